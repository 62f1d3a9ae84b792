"""The CPU restatement (oracle/rt_oracle.cpp) against golden vectors produced by the reference's
own compiled objects (tests/golden/make_golden.py).  Runs without the reference and without a GPU.
Everything is compared bit for bit."""
import numpy as np
import pytest

from scene_io import load_fixture, nodes_fields, oracle_from_fixture
from util import assert_bits_equal, bits


def test_layout(oracle_lib):
    lay = load_fixture("layout.npz")["layout"]
    assert lay[0] == oracle_lib.TRI_DTYPE.itemsize == 176
    assert lay[1] == oracle_lib.NODE_DTYPE.itemsize == 48
    assert lay[2] == 32
    f = oracle_lib.TRI_DTYPE.fields
    names = ["v", "n", "pn", "color", "uv", "rough", "ior", "tex", "emit", "metal", "transp"]
    assert [f[n][1] for n in names] == list(lay[3:14])
    nf = oracle_lib.NODE_DTYPE.fields
    assert [nf[n][1] for n in ("left", "right", "tri")] == list(lay[14:17])


def test_camera_and_rays(oracle_lib):
    g = load_fixture("camera.npz")
    for i in range(len(g["rot"])):
        O = oracle_lib.Oracle()
        w, h = [int(x) for x in g["size"][i]]
        O.set_params(w, h, 1, 1, 1)
        O.set_camera(g["eye"][i], g["at"][i], float(g["focal"][i]), float(g["rot"][i]))
        assert_bits_equal(O.get_camera(), g["cam"][i], "camera %d" % i)
        for k, (x, y) in enumerate(g["pixels"][i]):
            assert_bits_equal(O.get_ray(int(x), int(y)), g["rays"][i][k], "ray %d/%d" % (i, k))


@pytest.mark.parametrize("name", ["scene_mixed.npz", "scene_mixed_nosky.npz"])
def test_bvh_build(oracle_lib, name):
    g = load_fixture(name)
    O, _ = oracle_from_fixture(oracle_lib, g)
    assert np.array_equal(nodes_fields(O.get_nodes()), g["nodes"])


def test_bvh_build_tiny(oracle_lib):
    g = load_fixture("bvh_tiny.npz")
    i = 0
    while "tris%d" % i in g:
        O = oracle_lib.Oracle()
        O.set_params(8, 8, 1, 1, 1)
        O.set_triangles(np.ascontiguousarray(g["tris%d" % i]).view(oracle_lib.TRI_DTYPE).reshape(-1))
        O.build_bvh()
        assert np.array_equal(nodes_fields(O.get_nodes()), g["nodes%d" % i]), i
        i += 1
    assert i == 9


@pytest.mark.parametrize("name", ["scene_mixed.npz", "scene_mixed_nosky.npz"])
def test_primary_and_edge_hits(oracle_lib, name):
    g = load_fixture(name)
    O, sc = oracle_from_fixture(oracle_lib, g)
    ids, tab = O.primary_hits()
    # fixture order is (y, x, sub) as well
    assert np.array_equal(ids, g["hit_ids"])
    assert_bits_equal(tab, g["hit_tab"], "primary t/alpha/beta")
    assert (ids >= 0).mean() > 0.5
    eids, etab = O.closest_hits(g["edge_o"], g["edge_d"])
    assert np.array_equal(eids, g["edge_ids"])
    assert_bits_equal(etab, g["edge_tab"], "edge t/alpha/beta")
    off = 0
    for k, n in enumerate(g["cand_len"]):
        c = O.bvh_hit(g["edge_o"][8 * k], g["edge_d"][8 * k])
        assert np.array_equal(c, g["cand"][off:off + n]), "candidate order, ray %d" % (8 * k)
        off += n


@pytest.mark.parametrize("name", ["scene_mixed.npz", "scene_mixed_nosky.npz"])
def test_color_recursion_mt(oracle_lib, name):
    """color() with default-seeded generators per ray equals the reference's radiance wherever
    the reference's own result is defined (no total internal reflection under the ray)."""
    g = load_fixture(name)
    O, sc = oracle_from_fixture(oracle_lib, g)
    rays = np.array([O.get_ray(x, y) for y in range(sc.height) for x in range(sc.width)]).reshape(-1, 2, 4)
    col, tir = O.color_mt(0, rays[:, 0, :3], rays[:, 1, :3], fresh_each=True)
    ok = tir == 0
    assert ok.mean() > 0.9
    assert_bits_equal(col[ok][:, :3], g["color"][ok], "radiance")
    assert (g["color"][ok] > 0).any()


def test_pointwise_box_and_triangle(oracle_lib):
    g = load_fixture("pointwise.npz")
    O = oracle_lib.Oracle()
    box = np.array([O.aabb_hit(g["box_min"][i], g["box_max"][i], g["box_o"][i], g["box_d"][i]) for i in range(len(g["box_hit"]))])
    assert np.array_equal(box, g["box_hit"])
    assert 0.05 < box.mean() < 0.95
    sg = load_fixture("scene_mixed.npz")
    O, _ = oracle_from_fixture(oracle_lib, sg)
    accepted = 0
    for k, t in enumerate(g["tri_ids"]):
        out = O.tri_hit(int(t), g["tri_o"], g["tri_d"][k])
        assert_bits_equal(out, g["tri_out"][k], "Triangle::hit tri %d" % t)
        accepted += int((out[:, 2] < 1e30).sum())
    assert accepted > 100


def test_scatter_refract_texture_sky(oracle_lib):
    g = load_fixture("shading.npz")
    sg = load_fixture("scene_mixed.npz")
    O, _ = oracle_from_fixture(oracle_lib, sg)
    mats = g["mats"]
    for i in range(len(g["normals"])):
        tri = int(mats[i % 3])
        assert_bits_equal(O.scatter_mt(tri, 0, g["normals"][i], g["o"][i], g["d"][i], 5), g["spec"][i], "specular %d" % i)
        assert_bits_equal(O.scatter_mt(tri, 1, g["normals"][i], g["o"][i], g["d"][i], 5), g["diff"][i], "diffuse %d" % i)
        r = O.refract(int(mats[1]), g["normals"][i], g["o"][i], g["d"][i])
        assert bits(r[0:1])[0] == bits(g["refr"][i][0:1])[0], "refract proportion %d" % i
        if r[5] == 0.0:  # the reference's direction is uninitialised under total internal reflection
            assert_bits_equal(r[1:5], g["refr"][i][1:5], "refract direction %d" % i)
        else:
            assert r[0] == 0.0
    for i in range(len(g["uv"])):
        assert_bits_equal(O.sample_texture(i % 2, float(g["uv"][i, 0]), float(g["uv"][i, 1])), g["tex"][i], "texel %d" % i)
    for i in range(len(g["sky_d"])):
        assert_bits_equal(O.sample_sky(g["sky_d"][i]), g["sky"][i], "sky %d" % i)
    assert (g["sky"][:, :3].sum(axis=1) > 0).mean() > 0.4


def test_add_triangle(oracle_lib):
    g = load_fixture("add_triangle.npz")
    O = oracle_lib.Oracle()
    for i in range(32):
        O.add_triangle_cfg(g["v"][i, 0], g["v"][i, 1], g["v"][i, 2], g["side"][i], (0.1 * i, 0.5, 1.0), i & 1, (i >> 1) & 1,
                           (i >> 2) & 1, 0.25 * i, 1.0 + 0.01 * i)
    t = O.get_triangles().view(np.uint8).reshape(32, -1)[:, :167]
    # bytes 128..151 are uvCoordinate, which RayTracer::addTriangle leaves uninitialised
    assert np.array_equal(t[:, :128], g["tris"][:, :128])
    assert np.array_equal(t[:, 152:], g["tris"][:, 152:])


def test_render_loop_image_mt(oracle_lib):
    """One frame of the render loop body on one thread: sequential generator consumption,
    4-sample sum, *0.25 accumulate, tone map."""
    g = load_fixture("image_mt.npz")
    sg = load_fixture("scene_mixed_nosky.npz")
    O, sc = oracle_from_fixture(oracle_lib, sg)
    rays = np.array([O.get_ray(x, y) for y in range(sc.height) for x in range(sc.width)]).reshape(-1, 2, 4)
    col, tir = O.color_mt(0, rays[:, 0, :3], rays[:, 1, :3], fresh_each=False)
    assert tir.sum() == 0
    c = col.reshape(sc.height, sc.width, 4, 4)
    temp = np.zeros((sc.height, sc.width, 4), np.float32)
    for k in range(4):
        temp = temp + c[:, :, k, :]
    acc = np.zeros((sc.height, sc.width, 4), np.float32) + temp * np.float32(0.25)
    assert_bits_equal(acc[..., :3], g["accum"], "accumulator")
    out8 = O.tonemap(acc, 1).reshape(sc.height, sc.width, 3)
    assert np.array_equal(out8, g["rgb8"])
