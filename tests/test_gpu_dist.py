"""The N > 1 flow with real rendered tiles: two processes share the one GPU of the box (gloo backend, so the exchange
travels through the host; on a node with several GPUs the same code runs over RCCL), each renders its tile share of a
small scene with the HIP library, the shares are gathered (static interleave) or, tile by rendered tile, added on rank 0
(dynamic batches), and the image must equal the single-process image bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from raytracer_amd import api, scenes
from util import assert_bits_equal

pytestmark = pytest.mark.gpu

W, H, FRAMES = 200, 150, 2    # 4 x 3 tiles, the last column and row partial


def test_two_processes_render_real_tile_shares_and_exchange_them(tmp_path):
    sc = scenes.mixed_scene(seed=5, width=W, height=H, max_depth=3, n_diff=2, n_spec=1, with_sky=True)
    rt = api.RayTracer(0)
    rt.load_scene(sc)
    rt.reset()
    for f in range(1, FRAMES + 1):
        rt.render(f, 1)
    want = rt.accum()
    assert want[..., :3].any()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "img")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, worker, out, str(W), str(H), str(FRAMES)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=150)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    assert_bits_equal(want, np.load(out + "_static.npy"), "static interleave + gather, two ranks")
    assert_bits_equal(want, np.load(out + "_dynamic.npy"), "dynamic batches + exchange of the rendered tiles, two ranks")


def test_exchange_calls_over_a_one_rank_rccl_group(tmp_path):
    """The RCCL call path of bench.py --gpus N on one GPU: a process group of one rank on backend nccl, the static gather
    and the dynamic scheme forced through their collective calls (device tensors, dist.gather,
    all_gather_into_tensor).  The image must come back as it went in, and equal the image rendered without torch."""
    sc = scenes.mixed_scene(seed=5, width=W, height=H, max_depth=3, n_diff=2, n_spec=1, with_sky=True)
    rt = api.RayTracer(0)
    rt.load_scene(sc)
    rt.reset()
    rt.render(1, 2)
    want = rt.accum()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "img")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nccl_one_rank_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, worker, out, str(W), str(H)], env=env, capture_output=True, text=True, timeout=170)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    before, gathered, dynamic = np.load(out + "_nccl.npy")
    assert_bits_equal(want, before, "rendered into a torch tensor")
    assert_bits_equal(want, gathered, "gather_static over a one-rank RCCL group")
    assert_bits_equal(want, dynamic, "exchange_dynamic over a one-rank RCCL group")
