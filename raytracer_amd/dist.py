"""Image sharding across the GPUs of one node: one process per GPU, torch.distributed for the exchange
(backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

The reference parallelises over image rows inside one process (tbb::parallel_for,
/root/reference/src/RayTracer.cpp:237); pixels are independent, so here every rank keeps the whole
immutable scene and renders a share of the 64x64-pixel tiles.  There is no collective on the data
path of a frame.  One exchange step produces the image on rank 0:

  static  tile t belongs to rank t % world (interleaved, so expensive regions are spread);
          every rank packs the accumulators of its own tiles and rank 0 receives them with one
          gather -- each peer sends only its 1/world of the image over its own direct link.
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


def _as_tiles(img, tiles_x, tiles_y):
    """(H,W,4) -> (tiles, TILE, TILE, 4) with zero padding at the right/bottom edges."""
    h, w, c = img.shape
    pad = img.new_zeros((tiles_y * TILE, tiles_x * TILE, c))
    pad[:h, :w] = img
    return pad.view(tiles_y, TILE, tiles_x, TILE, c).permute(0, 2, 1, 3, 4).reshape(tiles_x * tiles_y, TILE, TILE, c)


def _from_tiles(tiles, tiles_x, tiles_y, width, height):
    c = tiles.shape[-1]
    pad = tiles.view(tiles_y, tiles_x, TILE, TILE, c).permute(0, 2, 1, 3, 4).reshape(tiles_y * TILE, tiles_x * TILE, c)
    return pad[:height, :width].contiguous()


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

    def gather_static(self, accum):
        """accum: (H,W,4) tensor holding this rank's tiles.  Returns the whole image on rank 0, None elsewhere."""
        if self.world == 1:
            return accum
        tiles = _as_tiles(accum, self.tiles_x, self.tiles_y)
        mine = self.owned().to(accum.device)
        send = accum.new_zeros((self.per_rank, TILE, TILE, accum.shape[-1]))
        send[:len(mine)] = tiles.index_select(0, mine)
        if self.rank == 0:
            recv = [torch.empty_like(send) for _ in range(self.world)]
            dist.gather(send, recv, dst=0, group=self.group)
            out = accum.new_zeros((self.n_tiles, TILE, TILE, accum.shape[-1]))
            for r in range(self.world):
                idx = self.owned(r).to(accum.device)
                out.index_copy_(0, idx, recv[r][:len(idx)])
            return _from_tiles(out, self.tiles_x, self.tiles_y, self.width, self.height)
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
