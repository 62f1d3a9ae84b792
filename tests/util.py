"""Shared helpers for the tests: load a procedural Scene into the oracle / the compiled reference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import binding  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_into(obj, scene, build=True):
    """obj: binding.Oracle or binding.Reference."""
    obj.set_params(scene.width, scene.height, scene.max_depth, scene.n_diff, scene.n_spec, scene.bg, scene.seed)
    obj.set_camera(scene.eye, scene.at, scene.focal, scene.rotate)
    sky_ids = set(scene.sky[1]) if scene.sky else set()
    if isinstance(obj, binding.Reference):
        # the reference keeps sky faces apart from model textures
        remap, k = {}, 0
        for i, t in enumerate(scene.textures):
            if i in sky_ids:
                continue
            obj.add_texture(t)
            remap[i] = k
            k += 1
        tris = scene.tris.copy()
        if len(tris):
            tex = tris["tex"]
            for old, new in remap.items():
                tex[scene.tris["tex"] == old] = new
        obj.set_triangles(tris)
        if scene.sky:
            obj.load_skybox(scene.sky[0], [scene.textures[i] for i in scene.sky[1]])
    else:
        for t in scene.textures:
            obj.add_texture(t)
        obj.set_triangles(scene.tris)
        if scene.sky:
            obj.set_skybox(scene.sky[0], scene.sky[1])
    if build:
        obj.build_bvh()
    return obj


def bits(a):
    """View float32 data as uint32 for exact comparison (NaN payloads and signed zeros included)."""
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bits_equal(a, b, what=""):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    neq = bits(a) != bits(b)
    # any NaN equals any NaN (payload/sign of NaN is not meaningful across compilers)
    neq &= ~(np.isnan(a) & np.isnan(b))
    if neq.any():
        idx = np.argwhere(neq)[:8]
        raise AssertionError("%s: %d of %d values differ; first at %s: %s vs %s" % (
            what, int(neq.sum()), a.size, idx.tolist(), a[neq][:8], b[neq][:8]))


def scattered_blocks(width, height, n_blocks, seed, block=4):
    """(n, 2) int32 pixel list (col, row): n_blocks seeded block x block squares spread over the whole image -- its
    corners and edges always among them -- for comparing a large frame with the CPU at the price of a few rows."""
    rng = np.random.default_rng(seed)
    bx = rng.integers(0, max(1, width - block + 1), n_blocks)
    by = rng.integers(0, max(1, height - block + 1), n_blocks)
    fixed = [(0, 0), (width - block, 0), (0, height - block), (width - block, height - block),
             (width // 2, 0), (width // 2, height - block), (0, height // 2), (width - block, height // 2)]
    for k, (x, y) in enumerate(fixed[:n_blocks]):
        bx[k], by[k] = max(0, x), max(0, y)
    # no pixel twice: a block list is a set of block origins on distinct positions
    seen, keep = set(), []
    for x, y in zip(bx.tolist(), by.tolist()):
        key = (x // block, y // block)
        if key in seen:
            continue
        seen.add(key)
        keep.append((x - x % block, y - y % block))
    out = []
    for x0, y0 in keep:
        for dy in range(block):
            for dx in range(block):
                if x0 + dx < width and y0 + dy < height:
                    out.append((x0 + dx, y0 + dy))
    return np.asarray(out, np.int32)


def assert_pixels_equal(oracle, frames, xy, gpu_accum, what, nthreads=16):
    """Accumulates `frames` (a list of frame numbers) for the pixels xy with the CPU restatement and compares those
    pixels of gpu_accum (H, W, 4) bit for bit."""
    acc = np.zeros(gpu_accum.shape, np.float32)
    for f in frames:
        oracle.render_pixels(f, xy, acc, nthreads=nthreads)
    cols, rows = xy[:, 0], xy[:, 1]
    assert_bits_equal(acc[rows, cols, :3], gpu_accum[rows, cols, :3], what)
