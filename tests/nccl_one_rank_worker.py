"""A world of one rank on the RCCL backend (tests/test_gpu_dist.py starts it): the exchange calls of bench.py --gpus N
-- TileSharder.gather_static over dist.gather with device tensors, and the count exchange of the dynamic scheme -- run
once on the one GPU of the box, forced through the collective path although there is nobody else to talk to.
argv: out-prefix width height"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import api, scenes  # noqa: E402
from raytracer_amd.dist import TileSharder  # noqa: E402


def main():
    out, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        sc = scenes.mixed_scene(seed=5, width=w, height=h, max_depth=3, n_diff=2, n_spec=1, with_sky=True)
        rt = api.RayTracer(0)
        rt.set_stream(torch.cuda.current_stream().cuda_stream)
        rt.load_scene(sc)
        accum = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
        rt.set_accum_device(accum.data_ptr())
        rt.render(1, 2)
        torch.cuda.synchronize()
        sh = TileSharder(w, h, 0, 1)
        before = accum.clone()
        img = sh.gather_static(accum, force_collective=True)      # pack -> dist.gather (RCCL) -> place
        assert img.data_ptr() != accum.data_ptr()
        same = sh.exchange_dynamic(rt.tile_log(), None, accum, force_collective=True)
        torch.cuda.synchronize()
        np.save(out + "_nccl.npy", np.stack([before.cpu().numpy(), img.cpu().numpy(), same.cpu().numpy()]))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
