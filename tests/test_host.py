"""Host-side product code (librtx_host.so) against the CPU restatement and the golden vectors, and
the export tables of both product libraries against include/rtx.h.  No GPU needed."""
import ctypes
import os
import re

import numpy as np
import pytest

from raytracer_amd import api, scenes
from scene_io import load_fixture, nodes_fields, scene_from_fixture
from util import ROOT, assert_bits_equal


def _declared(prefix):
    text = open(os.path.join(ROOT, "include", "rtx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z_0-9]+)\s*\(" % prefix, text)))


def test_every_declared_symbol_is_exported(native_libs):
    hip = ctypes.CDLL(native_libs.HIP_SO)   # loading initialises nothing on the GPU
    host = ctypes.CDLL(native_libs.HOST_SO)
    dev = _declared("rtx_")
    hst = _declared("rtxh_")
    assert len(dev) >= 20 and len(hst) >= 6
    for name in dev:
        assert hasattr(hip, name), name
    for name in hst:
        assert hasattr(host, name), name
    # and the ctypes tables used by the Python host cover exactly the header
    assert sorted(api.HIP_SYMBOLS) == dev
    assert sorted(api.HOST_SYMBOLS) == hst


def test_camera_matches_golden(native_libs):
    g = load_fixture("camera.npz")
    for i in range(len(g["rot"])):
        w, h = [int(x) for x in g["size"][i]]
        cam = api.camera(g["eye"][i], g["at"][i], float(g["focal"][i]), float(g["rot"][i]), w, h)
        assert_bits_equal(cam, g["cam"][i], "camera %d" % i)


@pytest.mark.parametrize("name", ["scene_mixed.npz", "scene_mixed_nosky.npz"])
def test_bvh_builder_matches_golden(native_libs, name):
    g = load_fixture(name)
    sc = scene_from_fixture(g)
    assert np.array_equal(nodes_fields(api.build_bvh(sc.tris)), g["nodes"])


def test_bvh_builder_tiny_and_empty(native_libs):
    g = load_fixture("bvh_tiny.npz")
    for i in range(9):
        tris = np.ascontiguousarray(g["tris%d" % i]).view(scenes.TRI_DTYPE).reshape(-1)
        assert np.array_equal(nodes_fields(api.build_bvh(tris)), g["nodes%d" % i]), i
    nodes = api.build_bvh(np.zeros(0, scenes.TRI_DTYPE))
    assert len(nodes) == 1 and nodes["left"][0] == -1 and nodes["right"][0] == -1 and nodes["tri"][0] == -1


def test_bvh_builder_parallel_path_matches_oracle(native_libs, oracle_lib):
    """Above 16k triangles the builder hands subtrees to worker threads; the tree must not change."""
    sc = scenes.bunny_class_scene(width=8, height=8, rings=100, segs=101)
    tris = np.concatenate([sc.tris, sc.tris[:2000]])
    O = oracle_lib.Oracle()
    O.set_params(8, 8, 1, 1, 1)
    O.set_triangles(tris)
    O.build_bvh()
    assert len(tris) > 16384
    assert np.array_equal(nodes_fields(api.build_bvh(tris)), nodes_fields(O.get_nodes()))


def test_make_triangle_matches_golden(native_libs):
    g = load_fixture("add_triangle.npz")
    for i in range(32):
        t = api.make_triangle(g["v"][i, 0], g["v"][i, 1], g["v"][i, 2], g["side"][i], (0.1 * i, 0.5, 1.0), i & 1, (i >> 1) & 1,
                              (i >> 2) & 1, 0.25 * i, 1.0 + 0.01 * i)
        b = t.view(np.uint8).reshape(-1)[:167]
        assert np.array_equal(b[:128], g["tris"][i][:128]) and np.array_equal(b[152:], g["tris"][i][152:]), i


def test_tables_and_tonemap(native_libs, oracle_lib):
    O = oracle_lib.Oracle()
    for a, b in zip(api.tables(), O.tables()):
        assert_bits_equal(a, b, "table")
    g = load_fixture("image_mt.npz")
    acc = np.zeros(g["accum"].shape[:2] + (4,), np.float32)
    acc[..., :3] = g["accum"]
    assert np.array_equal(api.tonemap(acc, 1).reshape(g["rgb8"].shape), g["rgb8"])
    rng = np.random.default_rng(1)
    acc = rng.uniform(-0.1, 3.0, (500, 4)).astype(np.float32)
    for frame in (1, 3, 64):
        assert np.array_equal(api.tonemap(acc, frame), O.tonemap(acc, frame))


def test_host_errors(native_libs):
    with pytest.raises(api.RtxError):
        api.camera((0, 0, 0), (0, 0, 1), 30.0, 0.0, 0, 10)
    lib = api.host_lib()
    rc = lib.rtxh_build_bvh(None, 5, None, 0)
    assert rc == -1 and b"null" in lib.rtxh_last_error()
    tris = scenes.mixed_scene(detail=0.4).tris
    nodes = np.zeros(4, scenes.NODE_DTYPE)
    rc = lib.rtxh_build_bvh(tris.ctypes.data, len(tris), nodes.ctypes.data, 4)
    assert rc == -1 and b"too small" in lib.rtxh_last_error()


def test_tonemap_table_is_the_host_tonemap(native_libs):
    """rtxh_tonemap_table tabulates rtxh_tonemap (src/RayTracer.cpp:248-257) exactly on [0, 1e15], at and next to
    every breakpoint included; the device searches this table instead of calling powf."""
    bits, vals, neg, nan = api.tonemap_table()
    assert bits[0] == 0 and (np.diff(bits.astype(np.int64)) > 0).all() and vals[-1] == 255
    rng = np.random.default_rng(1)
    edges = bits.view(np.float32)
    x = np.concatenate([rng.uniform(0, 2, 100000), 10 ** rng.uniform(-44, 15, 100000), [0.0, 1.0, 1e15],
                        edges, np.nextafter(edges, np.float32(-1))[1:], np.nextafter(edges, np.float32(2))]).astype(np.float32)
    acc = np.zeros((len(x), 4), np.float32)
    for frame in (1, 3):
        acc[:, 0] = x * np.float32(frame)
        quotient = acc[:, 0] / np.float32(frame)          # what the tone map sees
        want = api.tonemap(acc, frame)[:, 0]
        got = vals[np.searchsorted(bits, quotient.view(np.uint32), side="right") - 1]
        assert np.array_equal(got, want)
    special = np.zeros((3, 4), np.float32)
    special[:, 0] = [-1.0, -0.0, np.nan]
    assert api.tonemap(special, 1)[:, 0].tolist() == [neg, neg, nan]


def test_texture_loaders_match_reference_goldens(native_libs, tmp_path):
    """tests/golden/images.npz holds image files and what the reference's stb_image decodes from them (three
    channels, bottom row first, src/Texture.cpp:4-8): JPEG baseline/progressive/subsampled/grey/CMYK, PNG in all
    bit depths incl. 16-bit, palette and Adam7, TGA (raw and RLE), PPM/PGM, BMP with colour tables and channel masks,
    GIF, PSD, PIC and HDR."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "images.npz"))
    names = [str(n) for n in g["names"]]
    assert len(names) >= 30 and any(n.endswith(".jpg") for n in names)
    for k, name in enumerate(names):
        path = str(tmp_path / name)
        with open(path, "wb") as f:
            f.write(g["file_%d" % k].tobytes())
        got = api.load_texture(path)
        want = g["pixels_%d" % k]
        assert got.shape == want.shape and np.array_equal(got, want), name
    with open(str(tmp_path / "broken.jpg"), "wb") as f:
        f.write(g["file_0"].tobytes()[:200])
    with pytest.raises(api.RtxError):
        api.load_texture(str(tmp_path / "broken.jpg"))


def test_bmp_version_5_header_decodes_like_version_4(native_libs, tmp_path):
    """The reference's loader stops on an assertion of its own 128-byte read buffer when a 16-, 24- or 32-bit BMP has
    the 124-byte header (include/stb_image.h:5227), so those files have no reference pixels to compare with; a build
    without assertions reads them like the 108-byte header, and that is what the product does."""
    import image_cases
    rng = np.random.default_rng(3)
    w, h = 13, 7
    v16 = rng.integers(0, 65536, (h, w)).astype("<u2")
    v32 = rng.integers(0, 1 << 32, (h, w), dtype=np.uint64).astype("<u4")
    rgb = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    for bpp, rows, compress, masks in [(16, v16, 3, (0x0F00, 0x00F0, 0x000F, 0xF000)), (32, v32, 3, (0xFF000000, 0x00FC0000, 0x0003E000, 0x7F)),
                                       (24, rgb, 0, (0, 0, 0, 0))]:
        out = []
        for hsz in (108, 124):
            path = str(tmp_path / ("v%d_%d.bmp" % (hsz, bpp)))
            with open(path, "wb") as f:
                f.write(image_cases.bmp_file(w, h, bpp, [r.tobytes() for r in rows], hsz=hsz, compress=compress, masks=masks))
            out.append(api.load_texture(path))
        assert out[0].shape == (h, w, 3) and np.array_equal(out[0], out[1])
    assert np.array_equal(out[1], rgb[::-1, :, ::-1])      # 24 bits: stored blue first; row 0 of a texture is the bottom row


def test_load_model_reads_obj_ply_stl_off_alike(native_libs, tmp_path):
    """rtxh_load_model = RayTracer::loadModel's triangle construction (src/RayTracer.cpp:78-105) over the built-in mesh
    readers: the same mesh written as OBJ (+ .mtl), binary PLY, binary and ASCII STL and OFF gives the records of the
    numpy restatement bit for bit; the texture index survives only with texture coordinates; a missing file reports
    the reference's message.  (The reference reads meshes through assimp, which has no library here: which vertices
    an importer would merge or which normals it would generate is not pinned by this test.)"""
    import struct
    from test_gpu_cli import grid_mesh, load_model_py, write_obj, write_ply_binary
    d = str(tmp_path)
    verts, uvs, nrm, faces = grid_mesh(7, 5)
    write_obj(os.path.join(d, "m.obj"), verts, uvs, nrm, faces)
    with open(os.path.join(d, "m.mtl"), "w") as f:
        f.write("newmtl a\nKd 0.25 0.5 0.75\n")
    write_ply_binary(os.path.join(d, "m.ply"), verts, faces)
    tri = verts[faces]                                            # (F, 3, 3)
    up = np.array([0, 1, 0], "<f4")
    with open(os.path.join(d, "m_bin.stl"), "wb") as f:
        f.write(b"solid but binary".ljust(80, b" ") + struct.pack("<I", len(faces)))
        for t in tri:
            f.write(up.tobytes() + t.astype("<f4").tobytes() + b"\0\0")
    with open(os.path.join(d, "m_txt.stl"), "w") as f:
        f.write("solid grid\n")
        for t in tri:
            f.write(" facet normal 0 0 0\n  outer loop\n")
            for v in t:
                f.write("   vertex %.9g %.9g %.9g\n" % tuple(v))
            f.write("  endloop\n endfacet\n")
        f.write("endsolid grid\n")
    with open(os.path.join(d, "m.off"), "w") as f:
        f.write("OFF\n# a comment\n%d %d 0\n" % (len(verts), len(faces) // 2))
        for v in verts:
            f.write("%.9g %.9g %.9g\n" % tuple(v))
        half = len(faces) // 2                                     # the grid's quads, fanned back into the same triangles
        for (a, dd, c), (_, _, b) in zip(faces[:half], faces[half:]):
            f.write("4 %d %d %d %d 255 0 0\n" % (a, dd, c, b))
    mat = dict(rough=0.3, ior=1.2, metal=1, emit=0, transp=1)
    kw = dict(origin=(0.5, -1.0, 2.0), scale=3.0, is_metal=1, is_light_emitting=0, is_transparent=1, specular_roughness=0.3, refractive_index=1.2)

    def same(got, want, what):
        for name in scenes.TRI_DTYPE.names:
            assert np.ascontiguousarray(got[name]).tobytes() == np.ascontiguousarray(want[name]).tobytes(), (what, name)

    got, textured = api.load_model(os.path.join(d, "m.obj"), texture_index=4, **kw)
    assert textured
    same(got, load_model_py(verts, uvs, nrm, faces, 3.0, (0.5, -1.0, 2.0), dict(mat, color=(0.25, 0.5, 0.75)), 4), "obj")
    got, textured = api.load_model(os.path.join(d, "m.obj"), override_color=(0.9, 0.8, 0.1), **kw)
    assert not textured
    same(got, load_model_py(verts, uvs, nrm, faces, 3.0, (0.5, -1.0, 2.0), dict(mat, color=(0.9, 0.8, 0.1)), -1), "obj, colour given")
    grey = dict(mat, color=(0.6, 0.6, 0.6))
    flat = load_model_py(verts, None, None, faces, 3.0, (0.5, -1.0, 2.0), grey, -1)
    got, textured = api.load_model(os.path.join(d, "m.ply"), texture_index=4, **kw)
    assert not textured                                            # no texture coordinates in the file
    same(got, flat, "ply")
    same(api.load_model(os.path.join(d, "m_txt.stl"), **kw)[0], flat, "ascii stl, zero normals")
    with_up = load_model_py(verts, None, np.tile(up, (len(verts), 1)), faces, 3.0, (0.5, -1.0, 2.0), grey, -1)
    same(api.load_model(os.path.join(d, "m_bin.stl"), **kw)[0], with_up, "binary stl, facet normals")
    quads = np.concatenate([faces[:len(faces) // 2], faces[len(faces) // 2:]], 1).reshape(-1, 3)   # a d c | a c b per quad
    same(api.load_model(os.path.join(d, "m.off"), **kw)[0], load_model_py(verts, None, None, quads, 3.0, (0.5, -1.0, 2.0), grey, -1), "off")
    with pytest.raises(api.RtxError, match="Can't load model in "):
        api.load_model(os.path.join(d, "nope.obj"))
    with open(os.path.join(d, "junk.obj"), "w") as f:
        f.write("not a mesh\n")
    with pytest.raises(api.RtxError, match="Can't load model in "):
        api.load_model(os.path.join(d, "junk.obj"))


def test_traversal_tree_pairs_and_collapse(tmp_path):
    """raytracer_amd/csrc/rtx_wide.h on the host (tests/wide_tree_check.cpp, compiled here with g++): triangles that share
    an edge and a box pair up (every quad of a displaced grid, given in the scene generator's order; nothing in a soup
    of loose triangles, except an exact duplicate), mates are mutual and their common box stays within the limit; the
    surface-area tree over the leaves, collapsed four wide and quantised, reaches every triangle record exactly once
    and every triangle lies inside the decoded box of its leaf slot -- with one thread and with several."""
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wide_tree_check.cpp")
    exe = str(tmp_path / "wide_tree_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", src, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
