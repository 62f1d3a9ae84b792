"""The drop-in command line (raytracer_amd/RayTracer <config>) end to end on the GPU: config grammar, OBJ and
PLY readers, texture and skybox files with the vertical flip, single triangles, frame loop, tone map, PNG
writer and the stdout contract of /root/reference/src/main.cpp and src/RayTracer.cpp:263-273.
The images it writes must equal, byte for byte, the tone-mapped accumulators of the same scene assembled in
Python and rendered through the C ABI."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from oracle import binding
from raytracer_amd import api, build, scenes

F = np.float32


def write_obj(path, verts, uvs, normals, faces):
    with open(path, "w") as f:
        f.write("# generated\nmtllib m.mtl\n")
        for v in verts:
            f.write("v %.9g %.9g %.9g\n" % tuple(v))
        for t in uvs:
            f.write("vt %.9g %.9g\n" % tuple(t))
        for n in normals:
            f.write("vn %.9g %.9g %.9g\n" % tuple(n))
        for a, b, c in faces:
            f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a + 1, a + 1, a + 1, b + 1, b + 1, b + 1, c + 1, c + 1, c + 1))


def write_ply_binary(path, verts, faces):
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                b"element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(verts), len(faces)))
        f.write(np.asarray(verts, "<f4").tobytes())
        for a, b, c in faces:
            f.write(struct.pack("<Biii", 3, a, b, c))


def grid_mesh(n, seed):
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 6.28, 2)
    s, t = np.meshgrid(np.linspace(0, 1, n + 1), np.linspace(0, 1, n + 1), indexing="ij")
    x = (s - 0.5) * 2
    z = (t - 0.5) * 2
    y = 0.15 * np.sin(5 * s + ph[0]) * np.cos(4 * t + ph[1])
    verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(F)
    uvs = np.stack([s, t], -1).reshape(-1, 2).astype(F)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a = (i * (n + 1) + j).reshape(-1)
    b, d = a + n + 1, a + 1
    c = b + 1
    faces = np.concatenate([np.stack([a, d, c], 1), np.stack([a, c, b], 1)])   # wound so that normals point +y
    nrm = np.zeros_like(verts)
    nrm[:, 1] = 1.0
    return verts, uvs, nrm, faces


def load_model_py(verts, uvs, normals, faces, scale, offset, mat, tex_id):
    """RayTracer::loadModel's triangle construction (src/RayTracer.cpp:78-105) in float32 numpy."""
    v = (verts.astype(F) * F(scale) + np.asarray(offset, F)).astype(F)
    t = np.zeros(len(faces), scenes.TRI_DTYPE)
    for k in range(3):
        t["v"][:, k, :3] = v[faces[:, k]]
        if normals is not None:
            t["n"][:, k, :3] = normals[faces[:, k]]
        if tex_id >= 0:
            t["uv"][:, k] = uvs[faces[:, k]]
    e1 = (t["v"][:, 1, :3] - t["v"][:, 0, :3]).astype(F)
    e2 = (t["v"][:, 2, :3] - t["v"][:, 0, :3]).astype(F)
    pn = scenes._normalize_rows(scenes._cross(e1, e2))
    t["pn"][:, :3] = pn
    if normals is None:
        for k in range(3):
            t["n"][:, k, :3] = pn
    t["color"][:, :3] = np.asarray(mat["color"], F)
    t["rough"], t["ior"], t["tex"] = F(mat["rough"]), F(mat["ior"]), tex_id
    t["metal"], t["emit"], t["transp"] = mat["metal"], mat["emit"], mat["transp"]
    return t


def read_png(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


@pytest.mark.gpu
def test_cli_matches_c_abi(tmp_path):
    assert os.path.exists(build.CLI_BIN), "raytracer_amd/RayTracer must be built before the GPU tests"
    d = str(tmp_path)
    W, H, frames = 96, 64, 3
    verts, uvs, nrm, faces = grid_mesh(12, 3)
    write_obj(os.path.join(d, "bumps.obj"), verts, uvs, nrm, faces)
    open(os.path.join(d, "m.mtl"), "w").write("newmtl a\nKd 0.25 0.5 0.75\n")
    pverts, _, _, pfaces = grid_mesh(6, 4)
    write_ply_binary(os.path.join(d, "lid.ply"), pverts, pfaces)
    tex = scenes.checker_texture(16, 1)
    binding.write_ppm(os.path.join(d, "tex.ppm"), tex)          # the loader's flip turns this back into `tex`
    faces6 = scenes.sky_faces(16)
    for k, nm in enumerate(["front", "back", "left", "right", "top", "bottom"]):
        binding.write_ppm(os.path.join(d, nm + ".ppm"), faces6[k])
    cfg = """frame %d %d
camera 6 5 7 0 0.5 0 30 10
background_color 0.1 0.2 0.3
max_recursion_depth 3
diffuse_ray_number 2
specular_ray_number 1
skybox 1.5 front.ppm back.ppm left.ppm right.ppm top.ppm bottom.ppm
model_start
model_path bumps.obj
texture_path tex.ppm
scale 3
position_offset 0 0 0
is_metal 0
is_light_emitting 0
is_transparent 0
specular_roughness 0.3
refractive_index 1.2
model_end
model_start
model_path lid.ply
texture_path no
scale 1.5
position_offset 0.5 1.5 0
is_metal 1
is_light_emitting 0
is_transparent 0
specular_roughness 0.1
refractive_index 1.2
override_color 0.9 0.8 0.3
model_end
triangle_start
vertex_0 -2 4 -2
vertex_1 2 4 -2
vertex_2 0 4 2
normal_side 0 -1 0
color 9 9 8
is_metal 0
is_light_emitting 1
is_transparent 0
specular_roughness 0
refractive_index 1
triangle_end
render_num %d
""" % (W, H, frames)
    open(os.path.join(d, "config.txt"), "w").write(cfg)
    res = subprocess.run([build.CLI_BIN, "config.txt"], cwd=d, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().splitlines()
    assert len(lines) == frames + 1 and lines[-1] == "Render finished", res.stdout
    for i in range(frames):
        assert re.fullmatch(r"Output frame %d, use [0-9.e+-]+s" % (i + 1), lines[i]), lines[i]

    # the same scene through the Python host + C ABI
    t1 = load_model_py(verts, uvs, nrm, faces, 3.0, (0, 0, 0), dict(color=(0.25, 0.5, 0.75), rough=0.3, ior=1.2, metal=0, emit=0, transp=0), 0)
    t2 = load_model_py(pverts, None, None, pfaces, 1.5, (0.5, 1.5, 0), dict(color=(0.9, 0.8, 0.3), rough=0.1, ior=1.2, metal=1, emit=0, transp=0), -1)
    t3 = api.make_triangle((-2, 4, -2), (2, 4, -2), (0, 4, 2), (0, -1, 0), (9, 9, 8), 0, 1, 0, 0.0, 1.0)
    rt = api.RayTracer(0)
    rt.set_frame(W, H, 3, 2, 1, (0.1, 0.2, 0.3), 0x5EED)
    rt.set_camera((6, 5, 7), (0, 0.5, 0), 30.0, 10.0)
    # texture order of the command line: skybox faces first (the skybox line precedes the models), then model textures
    ids = [rt.add_texture(f) for f in faces6]
    rt.set_skybox(1.5, ids)
    tid = rt.add_texture(tex)
    t1["tex"] = tid
    rt.set_triangles(np.concatenate([t1, t2, t3]))
    rt.reset()
    for i in range(1, frames + 1):
        rt.render(i, 1)
        want = api.tonemap(rt.accum(), i).reshape(H, W, 3)
        got = read_png(os.path.join(d, "out_%03d.png" % i))
        assert got.shape == want.shape
        assert np.array_equal(got, want), "frame %d: %d bytes differ" % (i, (got != want).sum())
    assert want.std() > 10
    # once more with the single tile share sent through the exchange step (rtx_gather_tiles: RCCL as the command line
    # finds it, no torch in that process), the image wiped between pack and unpack: the same files, byte for byte
    first = [open(os.path.join(d, "out_%03d.png" % i), "rb").read() for i in range(1, frames + 1)]
    env = dict(os.environ, RTX_GATHER_SELF="1", RTX_GATHER_POISON="1")
    res = subprocess.run([build.CLI_BIN, "config.txt"], cwd=d, capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    for i in range(1, frames + 1):
        assert open(os.path.join(d, "out_%03d.png" % i), "rb").read() == first[i - 1], "frame %d differs after the exchange" % i

    # Several workers in one process, as on a node with several GPUs, here all on the one GPU of the box (RTX_DEVICE_MAP:
    # their tiles then meet by a copy inside the GPU, every other step is the one several GPUs take): a 200 x 150 frame
    # has 4 x 3 tiles, the last column and row partial.  Dynamic balancing (an interleaved head, the rest of the tiles
    # pulled in batches from one atomic counter by whichever worker comes first; rtx_exchange_tiles adds every
    # frame's tiles to worker 0's accumulator), dynamic with every tile pulled, and the static interleave
    # (rtx_gather_tiles) must all write the files one worker writes.
    open(os.path.join(d, "config.txt"), "w").write(cfg.replace("frame %d %d" % (W, H), "frame 200 150"))
    res = subprocess.run([build.CLI_BIN, "config.txt"], cwd=d, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "Render finished" in res.stdout, res.stdout + res.stderr
    one = [open(os.path.join(d, "out_%03d.png" % i), "rb").read() for i in range(1, frames + 1)]
    assert read_png(os.path.join(d, "out_%03d.png" % frames)).shape == (150, 200, 3)
    for workers, extra in ((2, {}), (2, {"RTX_DYNAMIC_PERCENT": "100"}), (3, {"RTX_BALANCE": "static"}), (5, {"RTX_DYNAMIC_PERCENT": "60"})):
        for i in range(1, frames + 1):
            os.remove(os.path.join(d, "out_%03d.png" % i))
        env = dict(os.environ, RTX_DEVICES=str(workers), RTX_DEVICE_MAP=",".join(["0"] * workers), **extra)
        res = subprocess.run([build.CLI_BIN, "config.txt"], cwd=d, capture_output=True, text=True, timeout=180, env=env)
        assert res.returncode == 0 and "Render finished" in res.stdout, res.stdout + res.stderr
        for i in range(1, frames + 1):
            assert open(os.path.join(d, "out_%03d.png" % i), "rb").read() == one[i - 1], "%d workers %s: frame %d differs" % (workers, extra, i)


def ball_mesh(rings, segs, radius, seed):
    """Indexed lat-long ball with a little relief, open by a hair at the poles (no zero-area faces), radial normals."""
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 6.28, 2)
    th, pp = np.meshgrid(np.linspace(0.02, np.pi - 0.02, rings + 1), np.linspace(0, 2 * np.pi, segs + 1), indexing="ij")
    r = radius * (1.0 + 0.06 * np.sin(4 * th + ph[0]) * np.cos(5 * pp + ph[1]))
    d = np.stack([np.sin(th) * np.cos(pp), np.cos(th), np.sin(th) * np.sin(pp)], -1)
    verts = (r[..., None] * d).reshape(-1, 3).astype(F)
    nrm = d.reshape(-1, 3).astype(F)
    uvs = np.stack([pp / (2 * np.pi), th / np.pi], -1).reshape(-1, 2).astype(F)
    i, j = np.meshgrid(np.arange(rings), np.arange(segs), indexing="ij")
    a = (i * (segs + 1) + j).reshape(-1)
    b, dd = a + segs + 1, a + 1
    c = b + 1
    faces = np.concatenate([np.stack([a, c, b], 1), np.stack([a, dd, c], 1)])
    return verts, uvs, nrm, faces


@pytest.mark.gpu
def test_template_config_512_against_the_restatement(tmp_path):
    """The reference's own config.txt (/root/reference/config.txt:1-63) with its comments removed as its first
    line asks: frame 512 512, camera 50 50 0 -> origin at focal 400, black background, depth 2, two diffuse and
    two specular rays, the placeholder model replaced by a generated 2,048-triangle mesh (metal, roughness 1,
    override colour 0.8, as the template has it), the placeholder sky by six generated faces, the template's
    floor triangle, render_num 4 (16 samples per pixel).  The command line renders it on the GPU; all four
    out_NNN.png must equal, byte for byte, the tone-mapped accumulator of the CPU restatement of the reference."""
    from util import load_into
    assert os.path.exists(build.CLI_BIN), "raytracer_amd/RayTracer must be built before the GPU tests"
    d = str(tmp_path)
    W = H = 512
    frames = 4
    verts, uvs, nrm, faces = ball_mesh(32, 32, 1.5, 5)
    assert len(faces) == 2048
    write_obj(os.path.join(d, "model.obj"), verts, uvs, nrm, faces)
    open(os.path.join(d, "m.mtl"), "w").write("newmtl a\nKd 0.3 0.3 0.3\n")
    faces6 = scenes.sky_faces(64)
    names = ["front", "back", "left", "right", "top", "bottom"]
    for k, nm in enumerate(names):
        binding.write_ppm(os.path.join(d, nm + ".ppm"), faces6[k])
    cfg = """frame %d %d
camera 50 50 0 0 0 0 400 0
background_color 0 0 0
max_recursion_depth 2
diffuse_ray_number  2
specular_ray_number 2
skybox 1 front.ppm back.ppm left.ppm right.ppm top.ppm bottom.ppm
model_start
model_path model.obj
texture_path no
scale 1
position_offset 0 0 0
is_metal 1
is_light_emitting 0
is_transparent 0
specular_roughness 1
refractive_index 1.2
override_color 0.8 0.8 0.8
model_end
triangle_start
vertex_0 -1000 100 1000
vertex_1 1000 100 1000
vertex_2 0 100 -1000
normal_side 0 -1 0
color 1 1 1
is_metal 0
is_light_emitting 0
is_transparent 0
specular_roughness 0
refractive_index 1.2
triangle_end
render_num %d
""" % (W, H, frames)
    open(os.path.join(d, "config.txt"), "w").write(cfg)
    res = subprocess.run([build.CLI_BIN, "config.txt"], cwd=d, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().splitlines()
    assert len(lines) == frames + 1 and lines[-1] == "Render finished", res.stdout

    t1 = load_model_py(verts, None, nrm, faces, 1.0, (0, 0, 0), dict(color=(0.8, 0.8, 0.8), rough=1.0, ior=1.2, metal=1, emit=0, transp=0), -1)
    t2 = api.make_triangle((-1000, 100, 1000), (1000, 100, 1000), (0, 100, -1000), (0, -1, 0), (1, 1, 1), 0, 0, 0, 0.0, 1.2)
    sc = scenes.Scene("template", np.concatenate([t1, t2]), faces6, (1.0, list(range(6))), eye=(50.0, 50.0, 0.0), at=(0.0, 0.0, 0.0),
                      focal=400.0, rotate=0.0, width=W, height=H, max_depth=2, n_diff=2, n_spec=2, bg=(0.0, 0.0, 0.0), seed=0x5EED)
    O = load_into(binding.Oracle(), sc)
    acc = np.zeros((H, W, 4), np.float32)
    for i in range(1, frames + 1):
        O.render_rows(i, acc, nthreads=16)
        want = O.tonemap(acc, i).reshape(H, W, 3)
        got = read_png(os.path.join(d, "out_%03d.png" % i))
        assert got.shape == want.shape
        assert np.array_equal(got, want), "frame %d: %d bytes differ" % (i, (got != want).sum())
    assert want.std() > 10 and (want.reshape(-1, 3).max(axis=1) > 0).mean() > 0.5


def test_cli_contract_errors(tmp_path, native_libs):
    """Argument check and parse errors: printed, exit code as in src/main.cpp:4-16.  No GPU is touched before
    these errors are raised."""
    d = str(tmp_path)
    r = subprocess.run([build.CLI_BIN], capture_output=True, text=True)
    assert r.returncode == 255 and r.stdout == "Invalid arguments\n"
    open(os.path.join(d, "bad.txt"), "w").write("// comment first\nframe 10 10\n")
    r = subprocess.run([build.CLI_BIN, "bad.txt"], cwd=d, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == 'Can\'t parse "frame"'
    open(os.path.join(d, "bad2.txt"), "w").write("frame 10 10\ncamera 1 1 1 0 0 0 50 0\nbackground_color 0 0 0\nmax_recursion_depth 0\n")
    r = subprocess.run([build.CLI_BIN, "bad2.txt"], cwd=d, capture_output=True, text=True)
    assert r.stdout.strip() == 'Expect: "max_recursion_depth" > 0'
    open(os.path.join(d, "bad3.txt"), "w").write(
        "frame 10 10\ncamera 1 1 1 0 0 0 50 0\nbackground_color 0 0 0\nmax_recursion_depth 1\ndiffuse_ray_number 1\n"
        "specular_ray_number 1\nfoo\n")
    r = subprocess.run([build.CLI_BIN, "bad3.txt"], cwd=d, capture_output=True, text=True)
    assert r.stdout.strip() == 'Expect: "skybox" or "model_start" or "triangle_start" or "render_num"'
    open(os.path.join(d, "bad4.txt"), "w").write(
        "frame 10 10\ncamera 1 1 1 0 0 0 50 0\nbackground_color 0 0 0\nmax_recursion_depth 1\ndiffuse_ray_number 1\n"
        "specular_ray_number 1\nmodel_start\nmodel_path nope.obj\ntexture_path no\nscale 1\nposition_offset 0 0 0\nis_metal 0\n"
        "is_light_emitting 0\nis_transparent 0\nspecular_roughness 0\nrefractive_index 1\nmodel_end\nrender_num 1\n")
    r = subprocess.run([build.CLI_BIN, "bad4.txt"], cwd=d, capture_output=True, text=True)
    assert r.stdout.strip() == "Can't load model in nope.obj"
