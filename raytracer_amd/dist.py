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
  dynamic ranks pull batches of tiles from a shared counter (the rendezvous store's atomic add),
          render each frame into a zeroed frame buffer, and the frame buffers are sum-reduced to
          rank 0, which adds the frame to its accumulator.  Every pixel is non-zero on exactly one
          rank, and x + 0 is exact, so the result is bit-identical to a single-GPU run.
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

    def gather_static(self, accum):
        """accum: (H,W,4) tensor holding this rank's tiles.  Returns the whole image on rank 0, None elsewhere.
        One packing gather per rank, one collective, one scatter on rank 0."""
        if self.world == 1:
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

    def reduce_frame(self, frame_buf, accum_total):
        """Sum-reduces the per-frame buffers to rank 0 and adds the frame to rank 0's accumulator."""
        if self.world > 1:
            dist.reduce(frame_buf, dst=0, op=dist.ReduceOp.SUM, group=self.group)
        if self.rank == 0:
            accum_total += frame_buf
            return accum_total
        return None


def default_store():
    """The key-value store of the default process group (its add() is an atomic fetch-and-add)."""
    return dist.distributed_c10d._get_default_store()
