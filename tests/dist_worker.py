"""One rank of the two-process exchange test (tests/test_gpu_dist.py starts two of these on the one GPU of the box, with
the gloo backend: device tensors then travel through the host, everything else is the N > 1 flow of bench.py).
Renders its real tile share of a small scene with the HIP library and takes part in the exchange; rank 0 saves the
images.  argv: out-prefix width height frames"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import api, scenes  # noqa: E402
from raytracer_amd.dist import TileSharder, default_store  # noqa: E402


def main():
    out, w, h, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        sc = scenes.mixed_scene(seed=5, width=w, height=h, max_depth=3, n_diff=2, n_spec=1, with_sky=True)
        rt = api.RayTracer(0)
        rt.set_stream(torch.cuda.current_stream().cuda_stream)
        rt.load_scene(sc)
        sh = TileSharder(w, h, rank, world)
        # static interleave: this rank's tiles of every frame, one gather at the end
        accum = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
        rt.set_accum_device(accum.data_ptr())
        rt.set_tiles(*sh.static_share())
        rt.render(1, frames)
        torch.cuda.synchronize()
        img = sh.gather_static(accum)
        # dynamic balancing: tile batches from the shared counter; rank 0 renders into the accumulator, the other
        # rank into a frame buffer whose rendered tiles (the library's own log of them) travel to rank 0 every frame
        store = default_store()
        total = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
        buf = torch.zeros_like(total)
        rt.set_accum_device((total if rank == 0 else buf).data_ptr())
        rt.clear_tile_log()
        res = None
        for f in range(1, frames + 1):
            pulled = []
            for first, count in sh.dynamic_batches(store, f, 2):
                rt.set_tiles(first, 1, count)
                rt.render(f, 1)
                pulled += list(range(first, first + count))
            tiles = rt.tile_log()
            assert tiles.tolist() == sorted(pulled)
            rt.clear_tile_log()
            res = sh.exchange_dynamic(tiles, buf, total)
            torch.cuda.synchronize()
            assert rank == 0 or not bool(buf.any())
        if rank == 0:
            np.save(out + "_static.npy", img.cpu().numpy())
            np.save(out + "_dynamic.npy", res.cpu().numpy())
        else:
            assert img is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
