"""rtx_trace_rays_noalloc, the entry point that allocates nothing (its work memory is a fixed array inside the
library), against rtx_trace_rays on the same rays: triangle ids exactly, t/alpha/beta bit for bit, on two calls in
a row of which the second is smaller (the array keeps the first call's contents), in both traversal modes, plus
the size limit, and free device memory (hipMemGetInfo) the same before and after every call.  Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest

from raytracer_amd import api, scenes
from util import assert_bits_equal

pytestmark = pytest.mark.gpu

MAX_RAYS = 1 << 18   # RTX_NOALLOC_MAX_RAYS of include/rtx.h


def rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-35, 35, (n, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1])
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d[: n // 50, 0] = 0                  # axis-parallel rays
    d[n // 50: n // 25, 1] = 0
    d[n // 25: n // 25 + 5] = 0          # zero directions: must miss
    return o, d


def test_noalloc_equals_trace_rays():
    import torch
    sc = scenes.mixed_scene(seed=4, width=16, height=16, detail=0.8)
    rt = api.RayTracer(0)
    rt.load_scene(sc)
    dev = torch.device("cuda", 0)
    # one set of device arrays for all calls: the caller owns inputs and outputs
    ids_d = torch.empty(MAX_RAYS, dtype=torch.int32, device=dev)
    tab_d = torch.empty((MAX_RAYS, 3), dtype=torch.float32, device=dev)
    for n, seed, mode in ((200000, 1, 0), (777, 2, 0), (MAX_RAYS, 3, 0), (5000, 4, 1), (1, 5, 0)):
        o, d = rays(n, seed)
        want_ids, want_tab = rt.trace_rays(o, d, mode)
        o_d = torch.from_numpy(o).to(dev)
        d_d = torch.from_numpy(d).to(dev)
        ids_d.fill_(-7)
        tab_d.fill_(float("nan"))
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(dev)[0]      # hipMemGetInfo: what the driver has left on the device
        rt.trace_rays_noalloc(n, o_d.data_ptr(), d_d.data_ptr(), ids_d.data_ptr(), tab_d.data_ptr(), mode)
        assert torch.cuda.mem_get_info(dev)[0] == free_before, "the call changed the device's free memory"
        got_ids = ids_d[:n].cpu().numpy()
        got_tab = tab_d[:n].cpu().numpy()
        assert np.array_equal(got_ids, want_ids), "n=%d mode=%d: %d ids differ" % (n, mode, (got_ids != want_ids).sum())
        hit = want_ids >= 0
        assert_bits_equal(got_tab[hit], want_tab[hit], "t/alpha/beta n=%d" % n)
        assert 0.2 < hit.mean() < 0.999 or n == 1
        assert (ids_d[n:] == -7).all() if n < MAX_RAYS else True      # nothing written beyond n
        assert rt.stats()["rays"] == n
    # ids only
    o, d = rays(3000, 9)
    want_ids, _ = rt.trace_rays(o, d, 0)
    o_d, d_d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)     # kept alive over the call
    torch.cuda.synchronize()
    rt.trace_rays_noalloc(3000, o_d.data_ptr(), d_d.data_ptr(), ids_d.data_ptr(), None, 0)
    assert np.array_equal(ids_d[:3000].cpu().numpy(), want_ids)
    # beyond the fixed array: a clear error, nothing launched
    with pytest.raises(api.RtxError) as e:
        rt.trace_rays_noalloc(MAX_RAYS + 1, o_d.data_ptr(), d_d.data_ptr(), ids_d.data_ptr(), tab_d.data_ptr(), 0)
    assert e.value.code == -1 and "at most" in str(e.value)
    # a scene without a tree
    rt2 = api.RayTracer(0)
    with pytest.raises(api.RtxError) as e:
        rt2.trace_rays_noalloc(10, o_d.data_ptr(), d_d.data_ptr(), ids_d.data_ptr(), tab_d.data_ptr(), 0)
    assert e.value.code == -2
