"""Image sharding across the GPUs of one node: one process per GPU, torch.distributed for the exchange
(backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

The reference parallelises over image rows inside one process (tbb::parallel_for,
/root/reference/src/RayTracer.cpp:237); pixels are independent, so here every rank keeps the whole
immutable scene and renders a share of the 64x64-pixel tiles.  There is no collective on the data
path of a frame.  One exchange step produces the image on rank 0:

  static  tile t belongs to rank t % world (interleaved, so expensive regions are spread);
          every rank packs the accumulators of its own tiles (one index_select over a cached pixel
          index) and rank 0 receives them with one gather -- each peer sends only its 1/world of the
          image over its own direct link -- and scatters them into the image with one index_copy.
  dynamic ranks pull batches of tiles from a shared counter (the rendezvous store's atomic add).  Rank 0
          renders straight into the accumulator; every other rank renders a frame into a buffer that is
          empty between frames and sends rank 0 the tiles it rendered -- their numbers and their packed
          pixels in one message, as much data as the static gather moves -- which rank 0 adds to the
          accumulator.  A zeroed buffer plus one frame's contribution is that contribution exactly, so
          accumulator + message is bit for bit what rendering the tile on rank 0 would have given.
"""
import math

import torch
import torch.distributed as dist

TILE = 64


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


class TileSharder:
    """Which tiles this rank renders, and how the pieces come together on rank 0."""

    def __init__(self, width, height, rank=0, world=1, group=None):
        self.width, self.height, self.rank, self.world, self.group = width, height, rank, world, group
        self.tiles_x, self.tiles_y = tile_grid(width, height)
        self.n_tiles = self.tiles_x * self.tiles_y
        self.per_rank = math.ceil(self.n_tiles / world)

    # ---- static interleave
    def static_share(self):
        """(first, stride, count) for rtx_scene_set_tiles."""
        return self.rank, self.world, -1

    def owned(self, rank=None):
        rank = self.rank if rank is None else rank
        return torch.arange(rank, self.n_tiles, self.world)

    def _pixel_index(self, rank):
        """Flat pixel indices of the per_rank tile slots of `rank`, TILE*TILE each; pixels beyond the image and
        slots beyond the rank's share get the index H*W (one past the image)."""
        n_pix = self.width * self.height
        t = self.owned(rank)
        ty, tx = t // self.tiles_x, t % self.tiles_x
        yy = (ty[:, None, None] * TILE + torch.arange(TILE)[None, :, None]).expand(len(t), TILE, TILE)
        xx = (tx[:, None, None] * TILE + torch.arange(TILE)[None, None, :]).expand(len(t), TILE, TILE)
        idx = torch.where((yy < self.height) & (xx < self.width), yy * self.width + xx, torch.full_like(yy, n_pix))
        full = torch.full((self.per_rank, TILE, TILE), n_pix, dtype=torch.int64)
        full[:len(t)] = idx
        return full.reshape(-1)

    def _plan(self, device):
        """Index tensors of the exchange, built once per device: what this rank packs, and where rank 0 puts what
        it receives (row H*W of its staging image swallows the padding)."""
        cache = self.__dict__.setdefault("_plans", {})
        key = str(device)
        if key not in cache:
            n_pix = self.width * self.height
            pack = self._pixel_index(self.rank).clamp(max=n_pix - 1).to(device)
            place = torch.cat([self._pixel_index(r) for r in range(self.world)]).to(device) if self.rank == 0 else None
            cache[key] = (pack, place)
        return cache[key]

    def gather_static(self, accum, force_collective=False):
        """accum: (H,W,4) tensor holding this rank's tiles.  Returns the whole image on rank 0, None elsewhere.
        One packing gather per rank, one collective, one scatter on rank 0.  force_collective: do all of it even in
        a world of one, where it moves the image onto itself (a one-rank RCCL group on a single GPU then runs the
        very calls a run on several GPUs makes)."""
        if self.world == 1 and not force_collective:
            return accum
        c = accum.shape[-1]
        n_pix = self.width * self.height
        pack, place = self._plan(accum.device)
        send = accum.reshape(n_pix, c).index_select(0, pack)  # padding slots carry some pixel; rank 0 drops them
        # gloo has no gather for device tensors: there (rehearsals on one GPU) the pieces travel through the host
        staged = send.is_cuda and dist.get_backend(self.group) == "gloo"
        if staged:
            send = send.cpu()
        if self.rank == 0:
            big = send.new_empty((self.world, send.shape[0], c))
            dist.gather(send, list(big.unbind(0)), dst=0, group=self.group)
            if staged:
                big = big.to(accum.device)
            out = accum.new_empty((n_pix + 1, c))
            out.index_copy_(0, place, big.reshape(-1, c))
            return out[:n_pix].reshape(self.height, self.width, c)
        dist.gather(send, None, dst=0, group=self.group)
        return None

    # ---- dynamic balancing
    def dynamic_batches(self, store, frame, batch):
        """Yields (first, count) tile ranges pulled from the shared counter of `frame` until the tiles run out."""
        key = "rtx_tiles_%d" % frame
        while True:
            start = store.add(key, batch) - batch
            if start >= self.n_tiles:
                return
            yield start, min(batch, self.n_tiles - start)

    def _tile_pixels(self, tiles, device):
        """Flat pixel indices of the TILE x TILE slots of `tiles` (int64 tensor on `device`) and which of them lie
        inside the image."""
        ar = torch.arange(TILE, device=device)
        ty, tx = tiles // self.tiles_x, tiles % self.tiles_x
        yy = (ty[:, None, None] * TILE + ar[None, :, None]).expand(len(tiles), TILE, TILE)
        xx = (tx[:, None, None] * TILE + ar[None, None, :]).expand(len(tiles), TILE, TILE)
        valid = ((yy < self.height) & (xx < self.width)).reshape(-1)
        return (yy * self.width + xx).reshape(-1), valid

    def exchange_dynamic(self, tiles, frame_buf, accum_total, force_collective=False):
        """One frame of the dynamic scheme.  tiles: the tile numbers this rank rendered this frame -- ranks > 0 into
        frame_buf (H,W,4), which holds nothing else, rank 0 straight into accum_total.  Ranks > 0 send (tile numbers,
        packed pixels) in one message each and leave frame_buf empty again; rank 0 adds what arrives to accum_total
        and returns it.  Messages have the ranks' own lengths: a one-int all_gather tells rank 0 what to expect.
        force_collective: run the collective calls even in a world of one (a one-rank RCCL group on a single GPU
        then exercises the call path of several)."""
        if self.world == 1 and not force_collective:
            return accum_total
        ref = accum_total if self.rank == 0 else frame_buf
        device = ref.device
        c = ref.shape[-1]
        n_pix = self.width * self.height
        tiles = torch.as_tensor(tiles, dtype=torch.int64).reshape(-1)
        backend = dist.get_backend(self.group)
        staged = device.type == "cuda" and backend == "gloo"   # gloo moves host tensors (rehearsals on one GPU)
        wire = torch.device("cpu") if staged else device
        counts = torch.zeros(self.world, dtype=torch.int64, device=wire)
        mine = torch.tensor([0 if self.rank == 0 else len(tiles)], dtype=torch.int64, device=wire)
        dist.all_gather_into_tensor(counts, mine, group=self.group)
        counts = counts.tolist()
        per_tile = TILE * TILE * c
        if self.rank != 0:
            if len(tiles):
                td = tiles.to(device)
                idx, valid = self._tile_pixels(td, device)
                flat = frame_buf.reshape(n_pix, c)
                data = flat.index_select(0, idx.clamp(max=n_pix - 1))
                flat[idx[valid]] = 0          # empty again for the next frame
                msg = torch.cat([data.reshape(-1), td.to(torch.int32).view(torch.float32)])
                if staged:
                    msg = msg.cpu()
                dist.send(msg, dst=0, group=self.group)
            return None
        bufs, ops = {}, []
        for r in range(1, self.world):
            if counts[r]:
                bufs[r] = torch.empty(counts[r] * (per_tile + 1), dtype=torch.float32, device=wire)
                ops.append(dist.P2POp(dist.irecv, bufs[r], r, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        flat = accum_total.reshape(n_pix, c)
        for r, buf in bufs.items():
            buf = buf.to(device) if staged else buf
            k = counts[r]
            ids = buf[k * per_tile:].view(torch.int32).to(torch.int64)
            if ids.numel() != k or int(ids.min()) < 0 or int(ids.max()) >= self.n_tiles:
                raise RuntimeError("rank %d sent tile numbers outside the image" % r)
            idx, valid = self._tile_pixels(ids, device)
            data = buf[:k * per_tile].reshape(-1, c)
            where = idx[valid]
            flat[where] = flat[where] + data[valid]      # every pixel once per frame: a plain x = x + contribution
        return accum_total


def default_store():
    """The key-value store of the default process group (its add() is an atomic fetch-and-add)."""
    return dist.distributed_c10d._get_default_store()
