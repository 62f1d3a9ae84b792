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
