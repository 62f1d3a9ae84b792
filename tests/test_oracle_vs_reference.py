"""Live pin of the CPU restatement against the reference's own compiled objects
(oracle/_ref/libref.so, built by oracle/build_ref.sh from /root/reference).  Skipped where that
library is absent; tests/test_oracle_golden.py covers the same ground from committed vectors.
Wider than the goldens: more seeds, more rays, bigger trees."""
import numpy as np
import pytest

from oracle import binding
from raytracer_amd import scenes
from scene_io import nodes_fields
from util import assert_bits_equal, bits, load_into

pytestmark = pytest.mark.skipif(not binding.have_ref(), reason="oracle/_ref/libref.so not built")


def _pair(sc):
    return load_into(binding.Oracle(), sc), load_into(binding.Reference(), sc)


def _rays(obj, sc):
    return np.array([obj.get_ray(x, y) for y in range(sc.height) for x in range(sc.width)]).reshape(-1, 2, 4)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_scene_pipeline(seed):
    sc = scenes.mixed_scene(seed=seed, width=40, height=32, detail=0.7, max_depth=3, n_diff=2, n_spec=2)
    O, R = _pair(sc)
    assert_bits_equal(O.get_camera(), R.get_camera(), "camera")
    assert np.array_equal(nodes_fields(O.get_nodes()), nodes_fields(R.get_nodes()))
    rays = _rays(R, sc)
    assert_bits_equal(_rays(O, sc), rays, "rays")
    ids_o, tab_o = O.primary_hits()
    ids_r, tab_r = R.closest_hits(rays[:, 0, :3], rays[:, 1, :3])
    assert np.array_equal(ids_o, ids_r)
    assert_bits_equal(tab_o, tab_r, "t/alpha/beta")
    col_o, tir = O.color_mt(0, rays[:, 0, :3], rays[:, 1, :3], fresh_each=True)
    col_r = R.color(0, rays[:, 0, :3], rays[:, 1, :3], fresh_each=True)
    ok = tir == 0
    assert ok.mean() > 0.9
    assert_bits_equal(col_o[ok][:, :3], col_r[ok][:, :3], "radiance")


def test_bvh_large_with_ties():
    sc = scenes.bunny_class_scene(width=8, height=8, rings=60, segs=61)
    tris = np.concatenate([sc.tris, sc.tris[:500]])  # exact duplicates: equal sort keys
    O, R = binding.Oracle(), binding.Reference()
    for obj in (O, R):
        obj.set_params(8, 8, 1, 1, 1)
        obj.set_triangles(tris)
        obj.build_bvh()
    assert np.array_equal(nodes_fields(O.get_nodes()), nodes_fields(R.get_nodes()))


def test_secondary_like_rays():
    sc = scenes.mixed_scene(seed=4, width=16, height=16, detail=0.6)
    O, R = _pair(sc)
    rng = np.random.default_rng(7)
    o = rng.uniform(-30, 30, (4000, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1])
    d = rng.normal(0, 1, (4000, 3)).astype(np.float32)
    d[:100, 0] = 0
    d[100:200, 1] = 0
    d[200:210] = 0
    ids_o, tab_o = O.closest_hits(o, d)
    ids_r, tab_r = R.closest_hits(o, d)
    assert np.array_equal(ids_o, ids_r)
    assert_bits_equal(tab_o, tab_r, "t/alpha/beta")
    for i in range(0, 4000, 97):
        assert np.array_equal(O.bvh_hit(o[i], d[i]), R.bvh_hit(o[i], d[i]))


def test_render_rows_sequential_generators():
    sc = scenes.mixed_scene(seed=6, width=24, height=20, detail=0.5, with_sky=True, ior=1.0, max_depth=3, n_diff=2, n_spec=1)
    O, R = _pair(sc)
    acc_r = np.zeros((sc.height, sc.width, 4), np.float32)
    out_r = np.zeros((sc.height, sc.width, 3), np.uint8)
    R.render_rows(1, acc_r, out8=out_r, nthreads=1)
    rays = _rays(O, sc)
    col, tir = O.color_mt(0, rays[:, 0, :3], rays[:, 1, :3], fresh_each=False)
    assert tir.sum() == 0
    c = col.reshape(sc.height, sc.width, 4, 4)
    temp = np.zeros((sc.height, sc.width, 4), np.float32)
    for k in range(4):
        temp = temp + c[:, :, k, :]
    acc_o = np.zeros_like(acc_r) + temp * np.float32(0.25)
    assert_bits_equal(acc_o[..., :3], acc_r[..., :3], "accumulator")
    assert np.array_equal(O.tonemap(acc_o, 1).reshape(sc.height, sc.width, 3), out_r)


def test_random_scenes_cameras_and_rays_equal_the_reference():
    """A fixed range of the seeds of tests/fuzz_parity.py: random scenes of four kinds, cameras, parameters and batches of
    awkward rays; camera, tree, rays, hits, candidate lists and recursive radiance of the restatement against the compiled
    reference, bit for bit."""
    import fuzz_parity
    failed = fuzz_parity.run(0, 16, scene=fuzz_parity.one_scene_cpu)
    assert not failed, "\n".join(failed)


def test_texture_files_decode_like_the_reference(tmp_path):
    """Texture::Texture(path) (src/Texture.cpp:4-8, the reference's vendored stb_image, three channels, flipped)
    against the loaders of rtx_image.cpp / rtx_jpeg.cpp: JPEG baseline and progressive in every subsampling, with
    restart markers, grey and CMYK; PNG in every bit depth, with palette, alpha, 16 bits and Adam7; TGA, PNM; BMP with
    colour tables, channel masks and every header version; GIF (first frame, interlaced, transparent, partial frames
    over a background colour); PSD raw and packed, 8 and 16 bits; Softimage PIC; Radiance HDR flat and run-length."""
    import image_cases
    from raytracer_amd import api
    R = binding.Reference()
    rng = np.random.default_rng(7)
    n = 0
    for name, data in list(image_cases.jpeg_cases(rng)) + list(image_cases.lossless_cases(rng)) + list(image_cases.more_cases(rng)):
        path = str(tmp_path / name)
        with open(path, "wb") as f:
            f.write(data)
        want = R.texture_pixels(path)
        assert want is not None, name
        got = api.load_texture(path)
        assert got.shape == want.shape and np.array_equal(got, want), name
        n += 1
    assert n > 500
    for name, data in image_cases.refused_cases(rng):
        path = str(tmp_path / name)
        with open(path, "wb") as f:
            f.write(data)
        assert R.texture_pixels(path) is None, name
        with pytest.raises(api.RtxError):
            api.load_texture(path)
