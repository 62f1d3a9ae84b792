"""Parity of the HIP path (through the C ABI of include/rtx.h) with the CPU restatement and with the
golden vectors taken from the reference.  Needs an MI355X: run with -m gpu.

Bar: primary-hit triangle ids, t/alpha/beta, per-sample radiance and accumulators are all compared
bit for bit (integer ids exactly; float32 values by their bit patterns).  At the full benchmark sizes,
where the CPU restatement would take minutes, the checks are size-independent properties:
culled traversal == exhaustive traversal, tile shares compose to the whole image, small work arenas
give the same image as large ones, repeated runs are identical.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import binding
from raytracer_amd import api, scenes
from scene_io import load_fixture, oracle_from_fixture, scene_from_fixture
from util import assert_bits_equal, assert_pixels_equal, bits, load_into, scattered_blocks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def libs():
    from raytracer_amd import build
    for p in (build.HIP_SO, build.HOST_SO, binding.ORACLE_SO):
        assert os.path.exists(p), "%s must be built before the GPU tests run (no building on the GPU box)" % p
    assert api.hip_lib().rtx_device_count() >= 1
    return api


def gpu_scene(sc, cam=None):
    rt = api.RayTracer(0)
    rt.load_scene(sc)
    if cam is not None:
        rt.set_camera_raw(cam)
    return rt


# ---------------------------------------------------------------- against the golden vectors

@pytest.mark.parametrize("name", ["scene_mixed.npz", "scene_mixed_nosky.npz"])
def test_primary_hits_match_reference_goldens(libs, name):
    g = load_fixture(name)
    sc = scene_from_fixture(g)
    rt = gpu_scene(sc, sc.cam)
    for mode in (0, 1):
        ids, tab = rt.primary_hits(mode)
        assert np.array_equal(ids, g["hit_ids"]), "mode %d: %d ids differ" % (mode, (ids != g["hit_ids"]).sum())
        hit = ids >= 0
        assert_bits_equal(tab[hit], g["hit_tab"][hit], "t/alpha/beta mode %d" % mode)
    eids, etab = rt.trace_rays(g["edge_o"], g["edge_d"], 0)
    assert np.array_equal(eids, g["edge_ids"])
    hit = eids >= 0
    assert_bits_equal(etab[hit], g["edge_tab"][hit], "edge rays")
    eids1, _ = rt.trace_rays(g["edge_o"], g["edge_d"], 1)
    assert np.array_equal(eids1, g["edge_ids"])


# ---------------------------------------------------------------- against the CPU restatement

@pytest.mark.parametrize("seed,w,h,depth,nd,ns,sky", [
    (1, 96, 80, 3, 2, 1, True),
    (2, 70, 45, 2, 1, 2, False),
    (3, 33, 129, 4, 1, 1, True),      # ragged: neither a multiple of 4 nor of 64
    (4, 64, 64, 1, 3, 2, True),       # depth 1: children are pure occlusion queries
    (5, 16, 12, 15, 1, 1, True),      # the deepest recursion the ABI accepts
    (6, 24, 16, 1, 64, 64, False),    # the most rays per hit the ABI accepts
    (7, 20, 12, 2, 64, 1, True),
])
def test_radiance_and_accumulator_bit_exact(libs, seed, w, h, depth, nd, ns, sky):
    sc = scenes.mixed_scene(seed=seed, width=w, height=h, max_depth=depth, n_diff=nd, n_spec=ns, with_sky=sky)
    O = load_into(binding.Oracle(), sc)
    rt = gpu_scene(sc)
    ids_o, tab_o = O.primary_hits()
    ids_g, tab_g = rt.primary_hits(0)
    assert np.array_equal(ids_o, ids_g)
    hit = ids_o >= 0
    assert_bits_equal(tab_o[hit], tab_g[hit], "t/alpha/beta")
    rad_o = O.sample_radiance(2, 0, 0, w, h)
    rad_g = rt.sample_radiance(2)
    assert_bits_equal(rad_o[..., :3], rad_g[..., :3], "per-sample radiance")
    assert (rad_o[..., :3] > 0).mean() > 0.2
    acc_o = np.zeros((h, w, 4), np.float32)
    O.reset_counts()
    for f in (1, 2, 3):
        O.render_rows(f, acc_o, nthreads=8)
    rt.reset()
    rt.render(1, 3)
    assert_bits_equal(acc_o[..., :3], rt.accum()[..., :3], "accumulator after 3 frames")
    assert rt.stats()["rays"] == O.ray_count()
    # tone map on the host side of the product equals the restatement's
    assert np.array_equal(api.tonemap(rt.accum(), 3), O.tonemap(acc_o, 3))


@pytest.mark.parametrize("sky,black", [(True, False), (False, True), (False, False)])
def test_last_level_rays_whose_value_is_known_are_not_traced(libs, sky, black):
    """A ray of the last level returns zero on a hit (src/RayTracer.cpp:170); where a miss is zero too (no sky
    face for the direction, src/Skybox.cpp:32-84, or no sky over a black background) the product skips the
    traversal.  The image and the ray count must not change, and the skipped rays are reported."""
    w, h, depth = 80, 48, 2
    sc = scenes.mixed_scene(seed=9, width=w, height=h, max_depth=depth, n_diff=2, n_spec=1, with_sky=sky)
    if black:
        sc.bg = (0.0, 0.0, 0.0)
    O = load_into(binding.Oracle(), sc)
    rt = gpu_scene(sc)
    assert_bits_equal(O.sample_radiance(1, 0, 0, w, h)[..., :3], rt.sample_radiance(1)[..., :3], "per-sample radiance")
    acc_o = np.zeros((h, w, 4), np.float32)
    O.reset_counts()
    O.render_rows(1, acc_o, nthreads=8)
    rt.reset()
    rt.render(1, 1)
    assert_bits_equal(acc_o[..., :3], rt.accum()[..., :3], "accumulator")
    st = rt.stats()
    assert st["rays"] == O.ray_count()
    last = st["rays_per_level"][depth]
    if black:
        assert st["rays_untraced"] == last > 0
    elif sky:
        assert 0 < st["rays_untraced"] < last      # the directions around -y have no sky face
    else:
        assert st["rays_untraced"] == 0


def test_secondary_like_rays_and_degenerate_rays(libs):
    sc = scenes.mixed_scene(seed=4, width=16, height=16, detail=0.8)
    O = load_into(binding.Oracle(), sc)
    rt = gpu_scene(sc)
    rng = np.random.default_rng(11)
    n = 20000
    o = rng.uniform(-35, 35, (n, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1])
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d[:300, 0] = 0
    d[300:600, 1] = 0
    d[600:900, [0, 1]] = 0
    d[900:910] = 0                      # zero directions: must miss
    d[910:920, 2] = np.nan              # NaN directions: must miss
    d[920:930, 0] = np.inf
    o[930:940, 1] = np.nan
    o[1000:2000] = np.round(o[1000:2000])
    o[2000:2200, 1] = 0.0               # origins exactly on the floor plane
    d[2200:2400] *= np.float32(1e-7)
    d[2400:2600] *= np.float32(1e7)
    # axis-parallel rays from origins on round coordinates, where box planes lie: 0 * inf = NaN inside AABB::hit
    o[2600:3800] = np.round(o[2600:3800] / 5) * 5
    d[2600:3000, 1] = 0
    d[3000:3400, 2] = 0
    d[3400:3600, 0] = 0
    d[3600:3800, [1, 2]] = 0
    d[3800:3900, 1] = np.float32(1e-40)  # denormal component: its reciprocal overflows
    d[3900:4000, 2] = np.float32(-3e-39)
    ids_o, tab_o = O.closest_hits(o, d)
    for mode in (0, 1):
        ids_g, tab_g = rt.trace_rays(o, d, mode)
        assert np.array_equal(ids_o, ids_g), "mode %d: %d differ" % (mode, (ids_o != ids_g).sum())
        hit = ids_o >= 0
        assert_bits_equal(tab_o[hit], tab_g[hit], "t/alpha/beta mode %d" % mode)
    assert (ids_o[900:940] == -1).all()
    assert 0.3 < (ids_o >= 0).mean() < 0.99


def test_rays_along_an_x_plane_of_a_leaf_box(libs):
    """dx = 0 with the origin's x on an x plane of a triangle's box: (bound - o) * (1/d) is 0 * inf = NaN in lane 0 of
    the reference's slab test (src/BVH.cpp:14-41), which then accepts the box whatever y and z say, provided its
    traversal gets there.  A ray that meets the box's y and z slabs stays in the wide traversal (every ancestor passes
    as well); one that misses them by less than the traversal's slack is decided by the reference's own tree.  Both
    kinds against the CPU restatement, and both kinds must occur."""
    sc = scenes.mixed_scene(seed=6, width=16, height=16, detail=0.8)
    O = load_into(binding.Oracle(), sc)
    rt = gpu_scene(sc)
    v = sc.tris["v"][:, :, :3]
    lo, hi = v.min(1), v.max(1)
    rng = np.random.default_rng(21)
    n = 8000
    pick = rng.integers(0, len(v), n)
    side = rng.integers(0, 2, n)
    px = np.where(side == 0, lo[pick, 0], hi[pick, 0])
    py = lo[pick, 1] + rng.uniform(0, 1, n) * (hi[pick, 1] - lo[pick, 1])
    pz = lo[pick, 2] + rng.uniform(0, 1, n) * (hi[pick, 2] - lo[pick, 2])
    d = np.zeros((n, 3), np.float32)
    d[:, 1] = rng.normal(0, 1, n)
    d[:, 2] = rng.normal(0, 1, n)
    beside = np.arange(n) >= n // 2
    # the second half passes just outside the box in y, running along z (dy = 0) or nearly so
    off = rng.uniform(1e-5, 2e-3, n) * np.where(rng.integers(0, 2, n) == 0, -1.0, 1.0)
    py = np.where(beside, np.where(off > 0, hi[pick, 1] + off, lo[pick, 1] + off), py)
    d[beside, 1] = np.where(rng.integers(0, 2, beside.sum()) == 0, 0.0, rng.normal(0, 1e-4, beside.sum()))
    s = rng.uniform(0.5, 30.0, n)
    o = np.stack([px, py - s * d[:, 1], pz - s * d[:, 2]], 1).astype(np.float32)
    o[:, 0] = px   # exactly the plane
    ids_o, tab_o = O.closest_hits(o, d)
    for mode in (0, 1):
        ids_g, tab_g = rt.trace_rays(o, d, mode)
        assert np.array_equal(ids_o, ids_g), "mode %d: %d differ" % (mode, (ids_o != ids_g).sum())
        hit = ids_o >= 0
        assert_bits_equal(tab_o[hit], tab_g[hit], "t/alpha/beta mode %d" % mode)
        if mode == 0:
            aside = rt.stats()["rays_set_aside"]
            # some rays needed the reference's tree (also rays through their own box: the plane is shared by the boxes
            # of neighbouring triangles, which they pass by), others were settled by the wide traversal
            assert 0 < aside < n, aside
    assert 0.2 < (ids_o >= 0).mean() < 0.999


def test_ties_resolve_in_reference_candidate_order(libs):
    """Stacks of exactly coincident triangles: every ray sees equal t from several candidates; the
    winner must be the first in the reference's depth-first candidate order."""
    base = scenes.mixed_scene(seed=8, width=48, height=48, detail=0.5, with_sky=False).tris
    rng = np.random.default_rng(5)
    pick = rng.choice(len(base), 60, replace=False)
    dup = np.concatenate([base[pick]] * 3)
    dup["color"][:, 0] = np.linspace(0, 1, len(dup))
    tris = np.concatenate([base, dup])[rng.permutation(len(base) + len(dup))]
    sc = scenes.mixed_scene(seed=8, width=48, height=48, detail=0.5, with_sky=False)
    sc.tris = tris
    O = load_into(binding.Oracle(), sc)
    rt = gpu_scene(sc)
    ids_o, _ = O.primary_hits()
    for mode in (0, 1):
        ids_g, _ = rt.primary_hits(mode)
        assert np.array_equal(ids_o, ids_g)


def test_empty_and_single_triangle_scenes(libs):
    sc = scenes.mixed_scene(seed=1, width=20, height=12, with_sky=True)
    for n in (0, 1, 2):
        sc.tris = scenes.mixed_scene(seed=1).tris[:n]
        O = load_into(binding.Oracle(), sc)
        rt = gpu_scene(sc)
        ids_o, _ = O.primary_hits()
        ids_g, _ = rt.primary_hits(0)
        assert np.array_equal(ids_o, ids_g)
        acc_o = np.zeros((sc.height, sc.width, 4), np.float32)
        O.render_rows(1, acc_o, nthreads=2)
        rt.reset()
        rt.render(1, 1)
        assert_bits_equal(acc_o[..., :3], rt.accum()[..., :3], "accumulator n=%d" % n)


# ---------------------------------------------------------------- properties at the benchmark sizes

def test_bunny_class_1080p_culled_equals_exhaustive(libs):
    """Primary rays only (bunny-class mesh, ~69.5k triangles, 1080p, primary rays): the production
    traversal returns the same triangle for every one of the 8.3M rays as the traversal that
    visits exactly the reference's candidate set; a 64x36 window is also checked against the CPU."""
    sc = scenes.bunny_class_scene()
    rt = gpu_scene(sc)
    ids0, tab0 = rt.primary_hits(0)
    s0 = rt.stats()
    ids1, tab1 = rt.primary_hits(1)
    assert ids0.size == 1920 * 1080 * 4
    assert np.array_equal(ids0, ids1)
    assert_bits_equal(tab0[ids0 >= 0], tab1[ids0 >= 0], "t/alpha/beta")
    assert 0.2 < (ids0 >= 0).mean() < 0.9
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(api.build_bvh(sc.tris))
    x0, y0 = 928, 522
    ids_o, tab_o = O.primary_hits(x0, y0, x0 + 64, y0 + 36)
    win = ids0.reshape(1080, 1920, 4)[y0:y0 + 36, x0:x0 + 64].reshape(-1)
    assert np.array_equal(ids_o, win)
    assert s0["rays"] == 1920 * 1080 * 4
    # the same call into page-locked arrays (rtx_host_alloc): identical answers; its wall time, read-back included, is
    # printed (run pytest with -s) and loosely bounded -- the kernels take about 1 ms, 133 MB cross the link
    import time
    ids_p, tab_p = rt.primary_hits(0, pinned=True)
    assert np.array_equal(ids_p, ids0) and np.array_equal(bits(tab_p), bits(tab0))
    t_all, t_ids = [], []
    for _ in range(4):
        t0 = time.perf_counter()
        rt.primary_hits(0, pinned=True)
        t_all.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        rt.primary_hits(0, want_tab=False, pinned=True)
        t_ids.append(time.perf_counter() - t0)
    print("rtx_primary_hits 1920x1080, page-locked arrays: ids + t/alpha/beta %.2f ms, ids only %.2f ms (kernels %.2f ms)" % (
        min(t_all) * 1e3, min(t_ids) * 1e3, rt.stats()["trace_ms"] + rt.stats()["shade_ms"]))
    assert min(t_all) < 0.012 and min(t_ids) < 0.006


def test_atrium_tiles_arena_and_determinism(libs):
    """Sponza-class scene at reduced resolution: (a) two tile shares rendered separately compose to
    the single-share image, (b) a work arena 20x smaller (many pixel batches, cut anywhere inside the two frames)
    gives the same image, (c) a second run is identical, (d) a window of the image equals the CPU restatement,
    (e) one call over two frames (one merged batch) equals two calls of one frame."""
    sc = scenes.atrium_scene(width=320, height=200, max_depth=4)
    nodes = api.build_bvh(sc.tris)
    rt = api.RayTracer(0)
    rt.load_scene(sc, nodes)
    rt.reset()
    rt.render(1, 2)
    full = rt.accum()
    st = rt.stats()
    assert st["chunks"] == 1          # the two small frames share one pixel batch
    # (c)
    rt.reset()
    rt.render(1, 2)
    assert_bits_equal(full, rt.accum(), "repeat")
    # (e) frames merged into one batch accumulate in the order of frame-by-frame calls
    rt.reset()
    rt.render(1, 1)
    rt.render(2, 1)
    assert rt.stats()["chunks"] == 1
    assert_bits_equal(full, rt.accum(), "frame by frame")
    # (b)
    rt.set_arena_bytes(int(320 * 200 * 4 * 3000 / 20))
    rt.reset()
    rt.render(1, 2)
    assert rt.stats()["chunks"] > 10
    assert_bits_equal(full, rt.accum(), "small arena")
    rt.set_arena_bytes(0)
    # (a)
    parts = []
    for r in range(2):
        rt.set_tiles(r, 2, -1)
        rt.reset()
        rt.render(1, 2)
        parts.append(rt.accum())
    rt.set_tiles(0, 1, -1)
    tiles_x = (sc.width + 63) // 64
    merged = np.zeros_like(full)
    for y0 in range(0, sc.height, 64):
        for x0 in range(0, sc.width, 64):
            owner = ((y0 // 64) * tiles_x + x0 // 64) % 2
            merged[y0:y0 + 64, x0:x0 + 64] = parts[owner][y0:y0 + 64, x0:x0 + 64]
            other = parts[1 - owner][y0:y0 + 64, x0:x0 + 64]
            assert not other.any(), "a device wrote outside its tiles"
    assert_bits_equal(full, merged, "tile shares")
    # (d)
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(nodes)
    acc_o = np.zeros((sc.height, sc.width, 4), np.float32)
    for f in (1, 2):
        O.render_rows(f, acc_o, row0=96, row1=104, nthreads=8)
    assert_bits_equal(acc_o[96:104, :, :3], full[96:104, :, :3], "rows 96..103 vs CPU")


def test_atrium_full_size_rows_equal_restatement(libs):
    """The benchmark workload itself (atrium, 1920x1080, depth 4, two frames): 360 scattered 4x4-pixel blocks of the
    accumulator against the CPU restatement, the frame's ray count against the per-level counts, and the device tone map of
    the whole frame against the host's."""
    sc = scenes.atrium_scene(width=1920, height=1080, max_depth=4)
    nodes = api.build_bvh(sc.tris)
    rt = api.RayTracer(0)
    rt.load_scene(sc, nodes)
    rt.reset()
    rt.render(1, 2)
    st = rt.stats()
    assert st["rays"] == sum(st["rays_per_level"]) and st["rays_per_level"][0] == 2 * 1920 * 1080 * 4
    assert st["queue_retries"] == 0
    full = rt.accum()
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(nodes)
    # 360 blocks of 4x4 pixels scattered over the whole frame (sky at the top, floor at the bottom, tile seams, the
    # image's corners and edges), the CPU time of three rows
    xy = scattered_blocks(sc.width, sc.height, 360, seed=1080)
    assert_pixels_equal(O, (1, 2), xy, full, "360 scattered 4x4 blocks vs CPU")
    assert np.array_equal(rt.tonemap_read(2).reshape(-1, 3), api.tonemap(full, 2))


def test_atrium_4k_tile_shares_compose(libs):
    """The 4K multi-GPU configuration on one GPU: 3840x2160, depth 4, the image as one share, as two interleaved
    shares, and as a 1/8 share rendered for two frames in one call (merged batches) against the same share frame by
    frame; 480 scattered 4x4-pixel blocks against the CPU restatement."""
    sc = scenes.atrium_scene(width=3840, height=2160, max_depth=4)
    nodes = api.build_bvh(sc.tris)
    rt = api.RayTracer(0)
    rt.load_scene(sc, nodes)
    rt.reset()
    rt.render(1, 1)
    st = rt.stats()
    assert st["rays_per_level"][0] == 3840 * 2160 * 4 and st["rays"] == sum(st["rays_per_level"])
    full = rt.accum()
    tiles_x = (sc.width + 63) // 64
    ty, tx = np.meshgrid(np.arange(sc.height) // 64, np.arange(sc.width) // 64, indexing="ij")
    owner = (ty * tiles_x + tx) % 2
    merged = np.zeros_like(full)
    rays = 0
    for r in range(2):
        rt.set_tiles(r, 2, -1)
        rt.reset()
        rt.render(1, 1)
        rays += rt.stats()["rays"]
        part = rt.accum()
        assert not part[owner != r].any(), "a share wrote outside its tiles"
        merged[owner == r] = part[owner == r]
    assert_bits_equal(full, merged, "two shares")
    assert rays == st["rays"]
    rt.set_tiles(3, 8, -1)
    rt.reset()
    rt.render(1, 2)
    assert rt.stats()["chunks"] == 1
    both = rt.accum()
    rt.reset()
    rt.render(1, 1)
    rt.render(2, 1)
    assert_bits_equal(both, rt.accum(), "1/8 share, two frames merged")
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(nodes)
    xy = scattered_blocks(sc.width, sc.height, 480, seed=2160)
    assert_pixels_equal(O, (1,), xy, full, "480 scattered 4x4 blocks vs CPU")


def test_converged_image_matches_mt19937_sampling(libs):
    """The counter-based samples replace the reference's two std::mt19937 generators (src/Triangle.cpp:96,122); the
    distributions must be the same.  192 samples per pixel from the HIP path against 192 from the restatement
    drawing from mt19937 in the reference's order: the two converged images differ by no more than two HIP images
    with different seeds differ from one another (the Monte-Carlo noise floor, measured in the same test), and
    their mean brightness agrees within that noise."""
    import copy
    frames = 48
    sc = scenes.mixed_scene(seed=4, width=40, height=24, max_depth=3, n_diff=2, n_spec=1)
    nodes = api.build_bvh(sc.tris)

    def gpu_mean(seed):
        s2 = copy.copy(sc)
        s2.seed = seed
        rt = api.RayTracer(0)
        rt.load_scene(s2, nodes)
        rt.reset()
        rt.render(1, frames)
        return rt.accum()[..., :3].astype(np.float64) / frames

    a, b = gpu_mean(0x5EED), gpu_mean(0xBEEF)
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(nodes)
    rays = np.array([O.get_ray(x, y) for y in range(sc.height) for x in range(sc.width)]).reshape(-1, 2, 4)
    o = np.tile(rays[:, 0, :3], (frames, 1))
    d = np.tile(rays[:, 1, :3], (frames, 1))
    col, tir = O.color_mt(0, o, d, fresh_each=False)    # one generator pair running through all frames
    m = col[:, :3].astype(np.float64).reshape(frames, sc.height, sc.width, 4, 3).mean(axis=(0, 3))
    rmse = lambda x, y: float(np.sqrt(np.mean((x - y) ** 2)))
    floor = rmse(a, b)
    assert floor > 0
    assert rmse(a, m) < 1.35 * floor and rmse(b, m) < 1.35 * floor, (rmse(a, m), rmse(b, m), floor)
    assert abs(a.mean() - m.mean()) < 3.0 * max(abs(a.mean() - b.mean()), floor / np.sqrt(a.size) * 3), (a.mean(), b.mean(), m.mean())


def test_external_accumulator_and_resume(libs):
    import torch
    sc = scenes.mixed_scene(seed=2, width=64, height=40)
    rt = gpu_scene(sc)
    rt.reset()
    rt.render(1, 4)
    want = rt.accum()
    ext = torch.zeros((sc.height, sc.width, 4), dtype=torch.float32, device="cuda:0")
    rt.set_accum_device(ext.data_ptr())
    rt.render(1, 2)
    torch.cuda.synchronize()
    half = ext.cpu().numpy()
    rt.set_accum_device(None)
    rt.write_accum(half)      # resume from a saved accumulator
    rt.render(3, 2)
    assert_bits_equal(want, rt.accum(), "resume")


def test_trees_the_wide_kernel_cannot_use_and_short_stacks(libs, monkeypatch):
    """(a) A tree whose leaf boxes are not the triangles' own bounds breaks the premise of the wide traversal:
    every ray must then go through the binary-tree kernel and still match the restatement walking the same
    nodes.  (b) With a 16-entry stack some rays overflow in the wide kernel and start over in the binary-tree
    kernel; the image must not change."""
    sc = scenes.mixed_scene(seed=12, width=96, height=64, detail=0.7)
    nodes = api.build_bvh(sc.tris)
    fat = nodes.copy()
    fat["bmin"][:, :3] -= np.float32(0.25)
    fat["bmax"][:, :3] += np.float32(0.25)
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(fat)
    rt = api.RayTracer(0)
    rt.load_scene(sc, fat)
    ids_o, tab_o = O.primary_hits()
    for mode in (0, 1):
        ids_g, tab_g = rt.primary_hits(mode)
        assert np.array_equal(ids_o, ids_g)
        assert_bits_equal(tab_o[ids_o >= 0], tab_g[ids_o >= 0], "fat tree, mode %d" % mode)
    # (b)
    big = scenes.atrium_scene(width=320, height=200, max_depth=4)
    bn = api.build_bvh(big.tris)
    ref = api.RayTracer(0)
    ref.load_scene(big, bn)
    ref.reset()
    ref.render(1, 1)
    monkeypatch.setenv("RTX_WIDE_STACK", "16")
    short = api.RayTracer(0)
    monkeypatch.delenv("RTX_WIDE_STACK")
    short.load_scene(big, bn)
    short.reset()
    short.render(1, 1)
    assert_bits_equal(ref.accum(), short.accum(), "16-entry stack")
    assert short.stats()["rays_set_aside"] >= ref.stats()["rays_set_aside"]


@pytest.mark.parametrize("stack", ["24", "16"])
def test_nodes_smaller_than_the_culling_slack(libs, monkeypatch, stack):
    """The quantised child boxes are tested inflated by a slack of 2^-14 of the largest coordinate in play.  A scene a
    hundred times smaller than usual and 8,000 units from the origin has nodes narrower than that slack: the empty
    interval that stands for an absent child then opens up, absent children count as entered, a node can ask for more
    stack than the host's bound foresees, and the ray is set aside for the binary-tree kernel at run time (with the
    16-entry stack all the more).  Primary hits, per-sample radiance and the accumulator must still equal the CPU
    restatement bit for bit, in both traversal modes."""
    monkeypatch.setenv("RTX_WIDE_STACK", stack)
    sc = scenes.mixed_scene(seed=8, width=72, height=48, max_depth=3, n_diff=2, n_spec=1, with_sky=True)
    off = np.array([8000.0, 8000.0, 8000.0], np.float32)
    k = np.float32(0.01)
    sc.tris = sc.tris.copy()
    sc.tris["v"][:, :, :3] = sc.tris["v"][:, :, :3] * k + off
    sc.eye = tuple((np.asarray(sc.eye, np.float32) * k + off).tolist())
    sc.at = tuple((np.asarray(sc.at, np.float32) * k + off).tolist())
    O = load_into(binding.Oracle(), sc)
    rt = gpu_scene(sc)
    ids_o, tab_o = O.primary_hits()
    assert 0.3 < (ids_o >= 0).mean()
    for mode in (0, 1):
        ids_g, tab_g = rt.primary_hits(mode)
        assert np.array_equal(ids_o, ids_g), "mode %d: %d ids differ" % (mode, (ids_o != ids_g).sum())
        assert_bits_equal(tab_o[ids_o >= 0], tab_g[ids_o >= 0], "t/alpha/beta mode %d" % mode)
    assert_bits_equal(O.sample_radiance(1, 0, 0, sc.width, sc.height)[..., :3], rt.sample_radiance(1)[..., :3], "per-sample radiance")
    acc_o = np.zeros((sc.height, sc.width, 4), np.float32)
    O.reset_counts()
    for f in (1, 2):
        O.render_rows(f, acc_o, nthreads=8)
    rt.reset()
    rt.set_instrumented(True)
    rt.render(1, 2)
    st = rt.stats()
    assert_bits_equal(acc_o[..., :3], rt.accum()[..., :3], "accumulator")
    assert st["rays"] == O.ray_count()
    # the slack really is wider than the nodes: far more box tests pass than in the same scene at its usual place
    assert st["node_tests"] / st["rays"] > 100, st["node_tests"] / st["rays"]
    if stack == "16":
        assert st["rays_set_aside"] > 0


def test_leaves_of_two_triangles_change_nothing(libs, monkeypatch):
    """The traversal tree keeps the two halves of a quad (triangles that share an edge and, nearly, a box) in one
    leaf; each triangle is still tested on its own, so ids, t/alpha/beta, radiance and ray counts must equal those of
    the tree with one triangle per leaf (RTX_LEAF_PAIRS=0), which in turn equal the CPU restatement's -- while fewer
    nodes are visited.  Also with nearly every ray forced through the wave-per-ray kernel, which takes the leaves
    apart again."""
    sc = scenes.atrium_scene(width=160, height=90, max_depth=3, detail=0.25)
    nodes = api.build_bvh(sc.tris)
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(nodes)
    want = np.zeros((sc.height, sc.width, 4), np.float32)
    O.reset_counts()
    O.render_rows(1, want, nthreads=16)
    got = {}
    for pairs, visits in (("1", None), ("0", None), ("1", "3")):
        monkeypatch.setenv("RTX_LEAF_PAIRS", pairs)
        if visits:
            monkeypatch.setenv("RTX_LONG_VISITS", visits)
        rt = api.RayTracer(0)
        rt.load_scene(sc, nodes)
        ids, tab = rt.primary_hits(0)
        rt.set_instrumented(True)
        rt.reset()
        rt.render(1, 1)
        st = rt.stats()
        assert st["rays"] == O.ray_count()
        assert_bits_equal(want[..., :3], rt.accum()[..., :3], "accumulator, pairs=%s long=%s" % (pairs, visits))
        got[(pairs, visits)] = (ids, tab, st["wide_visits"], st["rays_wave_traced"])
    a, b, c = got[("1", None)], got[("0", None)], got[("1", "3")]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[0], c[0])
    assert_bits_equal(a[1], b[1], "t/alpha/beta, pairs against single leaves")
    assert_bits_equal(a[1], c[1], "t/alpha/beta, lane-per-ray against wave-per-ray")
    assert a[2] < b[2], "two-triangle leaves should visit fewer nodes (%d against %d)" % (a[2], b[2])
    assert c[3] > 0.2 * O.ray_count()


@pytest.mark.parametrize("visits", ["2", "9"])
def test_long_rays_taken_over_by_a_whole_wave(libs, monkeypatch, visits):
    """A ray that has visited its allowance of nodes in the one-ray-per-lane kernel starts over in the kernel that
    gives it a whole wave.  With an allowance of a few nodes nearly every ray takes that route: primary hits must
    equal the exhaustive traversal and the restatement, the accumulator the restatement's, bit for bit, in a scene
    with every material (closest-hit levels) and a last level (any-hit)."""
    sc = scenes.mixed_scene(seed=5, width=96, height=64, max_depth=3, n_diff=2, n_spec=1)
    nodes = api.build_bvh(sc.tris)
    monkeypatch.setenv("RTX_LONG_VISITS", visits)
    rt = api.RayTracer(0)
    monkeypatch.delenv("RTX_LONG_VISITS")
    rt.load_scene(sc, nodes)
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(nodes)
    ids_o, tab_o = O.primary_hits()
    ids0, tab0 = rt.primary_hits(0)
    ids1, tab1 = rt.primary_hits(1)
    assert np.array_equal(ids0, ids1) and np.array_equal(ids0, ids_o)
    assert_bits_equal(tab0[ids0 >= 0], tab_o[ids0 >= 0], "t/alpha/beta")
    rt.reset()
    rt.render(1, 2)
    st = rt.stats()
    assert st["rays_wave_traced"] > st["rays"] // 8, st   # the allowance, not the small-queue rule, sends them there
    want = np.zeros((sc.height, sc.width, 4), np.float32)
    for f in (1, 2):
        O.render_rows(f, want, nthreads=8)
    assert_bits_equal(want[..., :3], rt.accum()[..., :3], "accumulator")
    assert st["rays"] == O.ray_count()
    # degenerate and axis-parallel rays through the same route
    rng = np.random.default_rng(3)
    m = 3000
    o = rng.uniform(-3, 3, (m, 3)).astype(np.float32)
    d = rng.normal(0, 1, (m, 3)).astype(np.float32)
    d[:300, 0] = 0
    d[300:600, 1] = 0
    d[600:900, 2] = 0
    d[900:1000, :2] = 0
    i0, t0 = rt.trace_rays(o, d, 0)
    i1, t1 = rt.trace_rays(o, d, 1)
    assert np.array_equal(i0, i1)
    assert_bits_equal(t0[i0 >= 0], t1[i0 >= 0], "arbitrary rays")


def test_waves_that_leave_the_queue_early_lose_no_ray(libs, monkeypatch):
    """The waves of the one-ray-per-lane kernel leave one after the other as a level's queue runs low (a wave's chunk is
    its last one when few rays per remaining wave are left behind it).  Whatever the threshold -- never, the default, or
    so high that most waves leave after their first chunk and a few work through the rest -- every ray of the queue is
    traced once: same ray counts per level, same accumulator bit for bit, and that accumulator is the restatement's on
    blocks scattered over the frame.  960x540 gives every level but the first more rays than the grid's first chunks
    hold, so chunks are claimed from the shared cursor."""
    sc = scenes.atrium_scene(width=960, height=540, max_depth=4)
    nodes = api.build_bvh(sc.tris)
    got = {}
    for backlog in ("0", None, "1024"):
        if backlog is None:
            monkeypatch.delenv("RTX_RETIRE_BACKLOG", raising=False)
        else:
            monkeypatch.setenv("RTX_RETIRE_BACKLOG", backlog)
        rt = api.RayTracer(0)
        rt.load_scene(sc, nodes)
        rt.reset()
        rt.render(1, 2)
        st = rt.stats()
        got[backlog] = (rt.accum(), st["rays_per_level"], st["rays"])
        del rt
    monkeypatch.delenv("RTX_RETIRE_BACKLOG", raising=False)
    for backlog in ("0", "1024"):
        assert got[backlog][1] == got[None][1] and got[backlog][2] == got[None][2]
        assert_bits_equal(got[None][0], got[backlog][0], "accumulator, RTX_RETIRE_BACKLOG=%s against the default" % backlog)
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(nodes)
    xy = scattered_blocks(sc.width, sc.height, 60, seed=540)
    assert_pixels_equal(O, (1, 2), xy, got["1024"][0], "60 scattered 4x4 blocks vs CPU")


def test_multi_million_triangle_mesh_depth_8(libs):
    """The stress configuration in small: a 2-million-triangle displaced mesh at depth 8 (finely tessellated, so
    grazing rays visit hundreds of nodes and the wave-per-ray kernel gets real work).  Culled traversal ==
    exhaustive traversal on every primary sample, a band of rows of the accumulator == the restatement, and the
    image does not depend on where a ray was traced (allowance 16 instead of the default)."""
    sc = scenes.heightfield_scene(n_tris=2000000, width=480, height=270)
    sc.max_depth = 8
    sc.n_diff = 1
    sc.n_spec = 1
    nodes = api.build_bvh(sc.tris)
    rt = api.RayTracer(0)
    rt.load_scene(sc, nodes)
    ids0, tab0 = rt.primary_hits(0)
    ids1, tab1 = rt.primary_hits(1)
    assert np.array_equal(ids0, ids1)
    assert_bits_equal(tab0[ids0 >= 0], tab1[ids0 >= 0], "t/alpha/beta")
    rt.reset()
    rt.render(1, 1)
    st = rt.stats()
    full = rt.accum()
    assert st["rays_wave_traced"] > 0 and st["rays"] == sum(st["rays_per_level"])
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(nodes)
    acc_o = np.zeros((sc.height, sc.width, 4), np.float32)
    O.render_rows(1, acc_o, row0=150, row1=156, nthreads=16)
    assert_bits_equal(acc_o[150:156, :, :3], full[150:156, :, :3], "rows 150..155 vs CPU")
    os.environ["RTX_LONG_VISITS"] = "16"
    try:
        rt2 = api.RayTracer(0)
    finally:
        del os.environ["RTX_LONG_VISITS"]
    rt2.load_scene(sc, nodes)
    rt2.reset()
    rt2.render(1, 1)
    assert rt2.stats()["rays_wave_traced"] > st["rays_wave_traced"]
    assert_bits_equal(full, rt2.accum(), "allowance 16")


def test_ten_million_triangles_4k_depth_8(libs):
    """The stress configuration at its named size: a displaced mesh of more than ten million triangles,
    3840x2160, max_recursion_depth 8 (a gigabyte of nodes and triangle records: beyond the Infinity Cache).
    Culled traversal == exhaustive traversal on every one of the 33 million primary samples; the frame's ray
    count is the sum of its levels' and no pixel batch had to be rendered twice; 480 4x4-pixel blocks scattered over
    the accumulator == the CPU restatement of the reference."""
    sc = scenes.heightfield_scene(n_tris=10100000, width=3840, height=2160)
    assert len(sc.tris) >= 10000000
    sc.max_depth = 8
    sc.n_diff = 1
    sc.n_spec = 1
    nodes = api.build_bvh(sc.tris)
    rt = api.RayTracer(0)
    rt.load_scene(sc, nodes)
    ids0, tab0 = rt.primary_hits(0)
    ids1, tab1 = rt.primary_hits(1)
    assert ids0.size == 3840 * 2160 * 4
    assert np.array_equal(ids0, ids1)
    hit = ids0 >= 0
    assert 0.2 < hit.mean() < 0.999
    assert_bits_equal(tab0[hit], tab1[hit], "t/alpha/beta")
    del ids1, tab1, tab0
    rt.reset()
    rt.render(1, 1)
    st = rt.stats()
    assert st["rays_per_level"][0] == 3840 * 2160 * 4 and st["rays"] == sum(st["rays_per_level"])
    # an open scene: most paths leave after a bounce or two, a few go on (33.2 M, 11.5 M, 384 k, 27 k, 2.3 k, 202, ...)
    assert all(st["rays_per_level"][lv] > 0 for lv in range(6)), st["rays_per_level"]
    assert st["queue_retries"] == 0, st["queue_retries"]
    assert st["rays_wave_traced"] > 0
    full = rt.accum()
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(nodes)
    xy = scattered_blocks(sc.width, sc.height, 480, seed=10100000)
    assert full[xy[:, 1], xy[:, 0], :3].any()
    assert_pixels_equal(O, (1,), xy, full, "480 scattered 4x4 blocks vs CPU")
    # the second frame sizes its queues from the first one's populations and still needs no second try
    rt.render(2, 1)
    assert rt.stats()["queue_retries"] == 0


def test_queue_history_overflow_is_rendered_again(libs):
    """Queues are sized from the ray populations of earlier frames.  History taken from one tile of sky makes
    the queues of the full image far too small: the device notices, stores nothing of such a batch, and the
    host renders it again with worst-case queues.  The image must equal the one of a scene without history."""
    sc = scenes.mixed_scene(seed=9, width=256, height=192)
    nodes = api.build_bvh(sc.tris)
    fresh = api.RayTracer(0)
    fresh.load_scene(sc, nodes)
    fresh.reset()
    fresh.render(1, 2)
    want = fresh.accum()
    assert fresh.stats()["queue_retries"] == 0
    rt = api.RayTracer(0)
    rt.load_scene(sc, nodes)
    pops = []
    for t in range(12):                      # find the emptiest tile: its rays die at level 0
        rt.set_tiles(t, 1, 1)
        rt.reset()
        rt.render(1, 1)
        pops.append(rt.stats()["rays"])
    rt2 = api.RayTracer(0)
    rt2.load_scene(sc, nodes)
    rt2.set_tiles(int(np.argmin(pops)), 1, 1)
    rt2.render(1, 1)                         # the only history this scene has
    rt2.set_tiles(0, 1, -1)
    rt2.reset()
    rt2.render(1, 2)
    st = rt2.stats()
    assert st["queue_retries"] >= 1, (st["queue_retries"], pops)
    assert_bits_equal(want, rt2.accum(), "after overflow and retry")
    assert st["rays"] == fresh.stats()["rays"]
    # with honest history the next frames fit without retries
    rt2.render(3, 2)
    assert rt2.stats()["queue_retries"] == 0
    fresh.render(3, 2)
    assert_bits_equal(fresh.accum(), rt2.accum(), "frames 3-4")


@pytest.mark.parametrize("w,h,depth,nd,ns", [(8, 6, 8, 4, 4), (4, 2, 6, 8, 8)])
def test_deep_branching_recursion_grows_its_queues_within_the_arena_limit(libs, w, h, depth, nd, ns):
    """max_recursion_depth 8 with four diffuse and four specular rays per hit: the worst case is 8^8 rays per primary
    sample (1.7 GB of queue each), the scene's real population some ten thousand.  The first frame starts from a
    bounded guess, the device reports the level that outgrew it, the queues grow from the counts actually asked for
    and the batch is rendered again -- never beyond the arena limit.  Accumulator and ray count == the CPU
    restatement's recursion; the second frame needs no second try; a limit that one pixel block cannot meet is an
    error that names the limit."""
    sc = scenes.mixed_scene(seed=3, width=w, height=h, max_depth=depth, n_diff=nd, n_spec=ns, with_sky=True)
    nodes = api.build_bvh(sc.tris)
    O = binding.Oracle()
    load_into(O, sc, build=False)
    O.set_nodes(nodes)
    want = np.zeros((h, w, 4), np.float32)
    O.reset_counts()
    O.render_rows(1, want, nthreads=16)
    rays1 = O.ray_count()
    assert rays1 > 2000 * w * h * 4
    rt = api.RayTracer(0)
    rt.load_scene(sc, nodes)
    rt.set_arena_bytes(2 << 30)
    rt.reset()
    rt.render(1, 1)
    st = rt.stats()
    assert st["queue_retries"] >= 1
    assert st["rays"] == rays1 and st["rays"] == sum(st["rays_per_level"])
    assert_bits_equal(want[..., :3], rt.accum()[..., :3], "accumulator, frame 1")
    O.render_rows(2, want, nthreads=16)
    rt.render(2, 1)
    assert rt.stats()["queue_retries"] == 0
    assert_bits_equal(want[..., :3], rt.accum()[..., :3], "accumulator, frame 2")
    # a limit between what the populations need (a few hundred MiB for one pixel block) and what the first guesses would
    # ask for: the levels whose true counts are known must be planned with those counts, not with their guesses
    mid = api.RayTracer(0)
    mid.load_scene(sc, nodes)
    mid.set_arena_bytes(512 << 20)
    mid.reset()
    mid.render(1, 1)
    mid.render(2, 1)
    assert_bits_equal(want[..., :3], mid.accum()[..., :3], "accumulator under a 512 MiB limit")
    small = api.RayTracer(0)
    small.load_scene(sc, nodes)
    small.set_arena_bytes(24 << 20)
    small.reset()
    with pytest.raises(api.RtxError) as e:
        small.render(1, 1)
    assert e.value.code == -4 and "work-arena limit of 24 MiB" in str(e.value) and "rtx_scene_set_arena_bytes" in str(e.value)


def test_tile_exchange_over_rccl_on_one_gpu(libs, monkeypatch):
    """rtx_gather_tiles, the exchange step of the several-GPUs-in-one-process drop-in, as far as one GPU can run it:
    a single scene holds the share (0, 1), which is packed, sent to itself through RCCL (ncclCommInitAll over one
    device, ncclSend + ncclRecv in one group) and unpacked.  With RTX_GATHER_POISON the image is overwritten between
    pack and unpack, so it is whole again only if every step did its part.  Ragged image: the last tile column and
    row are partial.  Shares that do not match (r, n) are refused."""
    sc = scenes.mixed_scene(seed=4, width=200, height=150)
    rt = gpu_scene(sc)
    rt.reset()
    rt.render(1, 2)
    want = rt.accum()
    assert want[..., :3].any()
    monkeypatch.setenv("RTX_GATHER_POISON", "1")
    api.gather_tiles([rt])
    assert_bits_equal(want, rt.accum(), "image after pack -> wipe -> send/recv -> unpack")
    api.gather_tiles([rt])                                   # communicator and buffers are reused
    assert_bits_equal(want, rt.accum(), "second exchange")
    monkeypatch.delenv("RTX_GATHER_POISON")
    rt.set_tiles(1, 2, -1)
    with pytest.raises(api.RtxError) as e:
        api.gather_tiles([rt])
    assert e.value.code == -1 and "tile share" in str(e.value)


@pytest.mark.parametrize("n", [2, 3, 8])
def test_tile_exchange_between_n_scenes_on_one_gpu(libs, n):
    """The exchange of several GPUs in one process with n > 1, as far as one GPU can run it: n scenes on the one GPU,
    scene r rendering the share (r, n) of a ragged image (4 x 3 tiles, shares of unequal size for n = 8), then
    rtx_gather_tiles -- the plan of tile lists and receive offsets, k_pack_tiles on every sender, one message per
    sender (a copy inside the GPU where RCCL would cross a link), k_unpack_tiles from the offsets of the plan.  Scene
    0's image must equal the single-scene image.  Then the dynamic scheme on the same scenes: every frame's tiles
    dealt out in uneven runs, rtx_exchange_tiles after every frame; scene 0's accumulator again equals the
    single-scene one, and the senders' images are empty afterwards."""
    sc = scenes.mixed_scene(seed=4, width=200, height=150)
    one = gpu_scene(sc)
    one.reset()
    one.render(1, 2)
    want = one.accum()
    rts = [gpu_scene(sc) for _ in range(n)]
    for r, rt in enumerate(rts):
        rt.set_tiles(r, n, -1)
        rt.reset()
        rt.render(1, 2)
    assert not np.array_equal(bits(rts[0].accum()), bits(want))        # scene 0 holds its share only
    api.gather_tiles(rts)
    assert_bits_equal(want, rts[0].accum(), "static shares gathered to scene 0")
    api.gather_tiles(rts)
    assert_bits_equal(want, rts[0].accum(), "second gather")
    with pytest.raises(api.RtxError) as e:
        api.gather_tiles(rts[1:] + rts[:1])                              # scene r must hold share (r, n)
    assert e.value.code == -1
    # dynamic: runs of 1, 2, 3, ... tiles go to the scenes in a different rotation every frame
    n_tiles = 12
    for rt in rts:
        rt.reset()
    for frame in (1, 2, 3):
        first, k, run = 0, frame, 1
        while first < n_tiles:
            count = min(run, n_tiles - first)
            rts[k % n].set_tiles(first, 1, count)
            rts[k % n].render(frame, 1)
            first += count
            k += 1
            run = run % 3 + 1
        assert sum(len(rt.tile_log()) for rt in rts) == n_tiles
        api.exchange_tiles(rts)
        assert all(len(rt.tile_log()) == 0 for rt in rts)
    one.reset()
    one.render(1, 3)
    assert_bits_equal(one.accum(), rts[0].accum(), "dynamic frames added on scene 0")
    for rt in rts[1:]:
        assert not rt.accum().any(), "a sender's frame buffer is empty after the exchange"
    # a tile rendered by two scenes in one frame is refused
    rts[0].set_tiles(0, 1, 2)
    rts[0].render(4, 1)
    rts[1].set_tiles(1, 1, 2)
    rts[1].render(4, 1)
    with pytest.raises(api.RtxError) as e:
        api.exchange_tiles(rts)
    assert e.value.code == -1 and "tile 1" in str(e.value)


def test_device_tonemap_equals_host(libs):
    """rtx_tonemap_read (table search on the device) gives the bytes of the host tone map of
    src/RayTracer.cpp:248-257, on rendered accumulators and on synthetic ones with the awkward values."""
    sc = scenes.mixed_scene(seed=6, width=150, height=70)      # 10500 pixels: not a multiple of 1024
    rt = gpu_scene(sc)
    rt.reset()
    for frame in (1, 2, 3):
        rt.render(frame, 1)
        assert np.array_equal(rt.tonemap_read(frame).reshape(-1, 3), api.tonemap(rt.accum(), frame))
    rng = np.random.default_rng(2)
    acc = (10 ** rng.uniform(-12, 3, (sc.height, sc.width, 4))).astype(np.float32)
    acc[0, :20, 0] = [0.0, -0.0, -1.0, np.nan, np.inf, -np.inf, 1e-45, 1e-38, 1.0, 0.5, 2.0, 255.0, 1e10, 1e15, 3.0, 7.0, 0.25, 0.1, 0.01, 0.001]
    edges = api.tonemap_table()[0].view(np.float32)
    acc[1, :100, 1] = edges[:100]
    acc[2, :100, 2] = np.nextafter(edges[100:200], np.float32(0))
    rt.write_accum(acc)
    for frame in (1, 3, 7):
        want = api.tonemap(acc, frame).reshape(sc.height, sc.width, 3)
        got = rt.tonemap_read(frame)
        skip = np.zeros(got.shape, bool)
        skip[0, 4, 0] = True                                # +inf: the reference's long -> int narrowing; the device saturates
        assert np.array_equal(got[~skip], want[~skip]), "frame %d: %d bytes differ" % (frame, (got != want).sum())
        assert got[0, 4, 0] == 255


def test_frames_of_a_merged_batch_each_give_their_image(libs):
    """rtx_render_frames_tonemapped: five frames rendered in merged pixel batches hand back, frame by frame, the bytes
    that one frame per call followed by rtx_tonemap_read gives (the out_NNN.png sequence of src/RayTracer.cpp:263-269),
    and leave the same accumulator -- with every frame in one batch, and with an arena so small that the batches cut
    frames apart."""
    sc = scenes.mixed_scene(seed=8, width=200, height=136)
    rt = gpu_scene(sc)
    rt.reset()
    want = []
    for frame in range(1, 6):
        rt.render(frame, 1)
        want.append(rt.tonemap_read(frame))
    acc_want = rt.accum()
    assert any(not np.array_equal(want[0], w) for w in want[1:])         # the frames do differ
    for arena in (0, 24 << 20):
        rt.set_arena_bytes(arena)
        rt.reset()
        got = rt.render_images(1, 5)
        batches = rt.stats()["chunks"]
        assert batches == 1 if arena == 0 else batches > 5, batches
        for k in range(5):
            assert np.array_equal(got[k], want[k]), "frame %d, arena %d" % (k + 1, arena)
        assert np.array_equal(rt.accum().view(np.uint32), acc_want.view(np.uint32))
    rt.set_arena_bytes(0)
    # the images of a later group continue the sequence
    got = rt.render_images(6, 2)
    rt.render(8, 1)
    acc8 = rt.accum()
    rt.reset()
    for frame in range(1, 8):
        rt.render(frame, 1)
        if frame >= 6:
            assert np.array_equal(rt.tonemap_read(frame), got[frame - 6])
    rt.render(8, 1)
    assert np.array_equal(rt.accum().view(np.uint32), acc8.view(np.uint32))
    lib = api.hip_lib()
    assert lib.rtx_render_frames_tonemapped(rt.h, 0, 1, None) == -1
    ptrs = (C.c_void_p * 2)(got[0].ctypes.data, None)
    assert lib.rtx_render_frames_tonemapped(rt.h, 1, 2, C.cast(ptrs, C.c_void_p)) == -1 and b"null image" in lib.rtx_last_error()


def test_rays_outside_the_wide_traversal_go_where_they_belong(libs):
    """Directions with a component that is tiny (below 2^-99, denormal), huge (above 2^126), zero, all zero, infinite or
    NaN, and NaN origins, through rtx_trace_rays in both modes: the production path (the wide traversal, the rays it
    leaves to the binary-tree kernel, the rays it answers itself) gives what the exhaustive traversal of the reference's
    tree gives."""
    sc = scenes.mixed_scene(seed=4, width=32, height=32)
    rt = gpu_scene(sc)
    rng = np.random.default_rng(12)
    n = 4096
    o = rng.uniform(-30, 30, (n, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) + 1.0
    d = rng.normal(size=(n, 3)).astype(np.float32)
    special = np.array([0.0, -0.0, 1e-30, -1e-31, 1e-38, 1e-45, 2.0 ** -99, np.nextafter(np.float32(2.0 ** -99), np.float32(0)), 2.0 ** 126,
                        np.nextafter(np.float32(2.0 ** 126), np.float32(np.inf)), 3e38, -3.4e38, np.inf, -np.inf, np.nan], np.float32)
    for k in range(n):
        m = k % 7
        if m < 3:
            d[k, m] = special[(k // 7) % len(special)]
        elif m == 3:
            d[k, 0] = special[(k // 7) % len(special)]
            d[k, 2] = special[(k // 11) % len(special)]
        elif m == 4:
            d[k] = special[(k // 7) % 6]
        elif m == 5:
            o[k, k % 3] = np.nan
    with np.errstate(all="ignore"):
        ids0, tab0 = rt.trace_rays(o, d, mode=0)
        ids1, tab1 = rt.trace_rays(o, d, mode=1)
    assert (ids0 >= 0).sum() > n // 8
    assert np.array_equal(ids0, ids1)
    hit = ids0 >= 0
    assert_bits_equal(tab0[hit], tab1[hit], "t/alpha/beta of the special rays")


def test_error_paths(libs):
    lib = api.hip_lib()
    h = C.c_void_p()
    assert lib.rtx_scene_create(C.byref(h), 9999) == -1
    assert lib.rtx_scene_create(C.byref(h), 0) == 0
    assert lib.rtx_render_frames(h, 1, 1) == -2 and b"set_params" in lib.rtx_last_error()
    bg = (C.c_float * 3)(0, 0, 0)
    assert lib.rtx_scene_set_params(h, 0, 10, 1, 1, 1, bg, 0) == -1
    assert lib.rtx_scene_set_params(h, 16, 16, 16, 1, 1, bg, 0) == -1
    assert lib.rtx_scene_set_params(h, 16, 16, 2, 1, 1, bg, 0) == 0
    tris = scenes.mixed_scene(detail=0.4).tris.copy()
    nodes = api.build_bvh(tris)
    assert lib.rtx_scene_set_bvh(h, nodes.ctypes.data, len(nodes)) == -2      # triangles first
    bad = tris.copy()
    bad["v"][3, 1, 3] = 1.0
    assert lib.rtx_scene_set_triangles(h, bad.ctypes.data, len(bad)) == -1    # w lane not zero
    assert lib.rtx_scene_set_triangles(h, tris.ctypes.data, len(tris)) == 0
    broken = nodes.copy()
    broken["left"][0] = len(nodes) + 5
    assert lib.rtx_scene_set_bvh(h, broken.ctypes.data, len(broken)) == -1
    broken = nodes.copy()
    broken["tri"][broken["tri"] >= 0] = 0
    assert lib.rtx_scene_set_bvh(h, broken.ctypes.data, len(broken)) == -1    # one triangle in every leaf
    assert lib.rtx_scene_set_bvh(h, nodes.ctypes.data, len(nodes)) == 0
    assert lib.rtx_render_frames(h, 1, 1) == -2                                  # no camera, no tables
    lib.rtx_scene_destroy(h)
