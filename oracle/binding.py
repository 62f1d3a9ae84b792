"""TEST INFRASTRUCTURE ONLY: ctypes bindings for the CPU restatement (oracle/liboracle.so)
and for the compiled reference harness (oracle/_ref/libref.so).

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py import this
module. Nothing under raytracer_amd/ does.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libref.so")

# In-memory layout of the reference's Triangle (include/RayTracer/Triangle.h:6-34) and
# LinearNode (include/RayTracer/BVH.h:35-43); checked against ref_layout() in the tests.
TRI_DTYPE = np.dtype([
    ("v", "<f4", (3, 4)), ("n", "<f4", (3, 4)), ("pn", "<f4", (4,)), ("color", "<f4", (4,)),
    ("uv", "<f4", (3, 2)), ("rough", "<f4"), ("ior", "<f4"), ("tex", "<i4"),
    ("emit", "u1"), ("metal", "u1"), ("transp", "u1"), ("pad", "u1", (9,)),
])
NODE_DTYPE = np.dtype([
    ("bmin", "<f4", (4,)), ("bmax", "<f4", (4,)), ("left", "<i4"), ("right", "<i4"), ("tri", "<i4"), ("pad", "<i4"),
])
assert TRI_DTYPE.itemsize == 176 and NODE_DTYPE.itemsize == 48

_f = C.c_float
_i = C.c_int
_vp = C.c_void_p
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def _fptr(a):
    return a.ctypes.data_as(_fp)


def _iptr(a):
    return a.ctypes.data_as(_ip)


def f3(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1))


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


def build_ref():
    subprocess.check_call(["bash", os.path.join(HERE, "build_ref.sh")])


def have_ref():
    return os.path.exists(REF_SO)


def write_ppm(path, rgb_mem):
    """Write an image file such that the reference's flipped load (src/Texture.cpp:6-7)
    reproduces `rgb_mem` (H,W,3 uint8, row 0 = first row in memory)."""
    h, w, _ = rgb_mem.shape
    with open(path, "wb") as fh:
        fh.write(b"P6\n%d %d\n255\n" % (w, h))
        fh.write(np.ascontiguousarray(rgb_mem[::-1]).tobytes())


class _Base:
    prefix = ""

    def _fn(self, name, restype, *argtypes):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = restype
        fn.argtypes = list(argtypes)
        return fn


class Oracle(_Base):
    """The CPU restatement. Method names mirror the C entry points of rt_oracle.cpp."""
    prefix = "orc_"

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        self.lib = C.CDLL(ORACLE_SO)
        L = self
        self._create = L._fn("create", _vp)
        self._destroy = L._fn("destroy", None, _vp)
        self._set_params = L._fn("set_params", None, _vp, _i, _i, _i, _i, _i, _fp, C.c_uint64)
        self._set_camera = L._fn("set_camera", None, _vp, _fp, _fp, _f, _f)
        self._get_camera = L._fn("get_camera", None, _vp, _fp)
        self._put_camera = L._fn("put_camera", None, _vp, _fp)
        self._get_ray = L._fn("get_ray", None, _vp, _i, _i, _fp)
        self._set_triangles = L._fn("set_triangles", None, _vp, _vp, _i)
        self._num_triangles = L._fn("num_triangles", _i, _vp)
        self._get_triangles = L._fn("get_triangles", None, _vp, _vp)
        self._add_triangle_cfg = L._fn("add_triangle_cfg", None, _vp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _f, _f)
        self._add_texture = L._fn("add_texture", _i, _vp, _vp, _i, _i)
        self._set_skybox = L._fn("set_skybox", None, _vp, _f, _ip)
        self._sample_texture = L._fn("sample_texture", None, _vp, _i, _f, _f, _fp)
        self._sample_sky = L._fn("sample_sky", None, _vp, _fp, _fp)
        self._build_bvh = L._fn("build_bvh", _i, _vp)
        self._num_nodes = L._fn("num_nodes", _i, _vp)
        self._get_nodes = L._fn("get_nodes", None, _vp, _vp)
        self._set_nodes = L._fn("set_nodes", None, _vp, _vp, _i)
        self._aabb_hit = L._fn("aabb_hit", _i, _fp, _fp, _fp, _fp)
        self._bvh_hit = L._fn("bvh_hit", _i, _vp, _fp, _fp, _ip, _i)
        self._tri_hit = L._fn("tri_hit", None, _vp, _i, _i, _fp, _fp, _fp)
        self._closest_hits = L._fn("closest_hits", None, _vp, _i, _fp, _fp, _ip, _fp)
        self._scatter_mt = L._fn("scatter_mt", None, _vp, _i, _i, _fp, _fp, _fp, _i, _fp)
        self._refract = L._fn("refract", None, _vp, _i, _fp, _fp, _fp, _fp)
        self._color_mt = L._fn("color_mt", None, _vp, _i, _i, _fp, _fp, _fp, _i, _ip)
        self._color_keys = L._fn("color_keys", None, _vp, _i, _i, _fp, _fp, C.POINTER(C.c_uint32), _fp)
        self.sample_key = L._fn("sample_key", C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32)
        self.child_key = L._fn("child_key", C.c_uint32, C.c_uint32, C.c_uint32)
        self.draw = L._fn("draw", C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32)
        self._sample_radiance = L._fn("sample_radiance", None, _vp, _i, _i, _i, _i, _i, _fp)
        self._render_rows = L._fn("render_rows", C.c_double, _vp, _i, _i, _i, _fp, _vp, _i)
        self._render_pixels = L._fn("render_pixels", C.c_double, _vp, _i, _i, _ip, _fp, _i)
        self._worst_ray = L._fn("worst_ray", None, _vp, _fp)
        self._ray_count = L._fn("ray_count", C.c_uint64, _vp)
        self._tir_count = L._fn("tir_count", C.c_uint64, _vp)
        self._reset_counts = L._fn("reset_counts", None, _vp)
        self._primary_hits = L._fn("primary_hits", None, _vp, _i, _i, _i, _i, _ip, _fp)
        self._tonemap = L._fn("tonemap", None, _fp, _i, _i, _vp)
        self._tables = L._fn("tables", None, _fp, _fp, _fp)
        self.h = self._create()
        self.width = self.height = 0

    def close(self):
        if self.h:
            self._destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene set-up
    def set_params(self, width, height, max_depth, n_diff, n_spec, bg=(0, 0, 0), seed=0):
        self.width, self.height = width, height
        self._set_params(self.h, width, height, max_depth, n_diff, n_spec, _fptr(f3(bg)), seed)

    def set_camera(self, eye, at, focal, rotate_deg=0.0):
        self._set_camera(self.h, _fptr(f3(eye)), _fptr(f3(at)), focal, rotate_deg)

    def get_camera(self):
        out = np.zeros(16, np.float32)
        self._get_camera(self.h, _fptr(out))
        return out.reshape(4, 4)

    def put_camera(self, cam):
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        self._put_camera(self.h, _fptr(cam))

    def get_ray(self, x, y):
        out = np.zeros(32, np.float32)
        self._get_ray(self.h, x, y, _fptr(out))
        return out.reshape(4, 2, 4)

    def set_triangles(self, tris):
        tris = np.ascontiguousarray(tris, TRI_DTYPE)
        self._set_triangles(self.h, tris.ctypes.data, len(tris))

    def get_triangles(self):
        out = np.zeros(self._num_triangles(self.h), TRI_DTYPE)
        self._get_triangles(self.h, out.ctypes.data)
        return out

    def add_triangle_cfg(self, v0, v1, v2, side, color, metal, emit, transp, rough, ior):
        self._add_triangle_cfg(self.h, _fptr(f3(v0)), _fptr(f3(v1)), _fptr(f3(v2)), _fptr(f3(side)),
                               _fptr(f3(color)), int(metal), int(emit), int(transp), rough, ior)

    def add_texture(self, rgb_mem):
        rgb_mem = np.ascontiguousarray(rgb_mem, np.uint8)
        h, w, _ = rgb_mem.shape
        return self._add_texture(self.h, rgb_mem.ctypes.data, w, h)

    def set_skybox(self, brightness, tex_ids):
        ids = np.ascontiguousarray(tex_ids, np.int32)
        self._set_skybox(self.h, brightness, _iptr(ids))

    def build_bvh(self):
        return self._build_bvh(self.h)

    def get_nodes(self):
        out = np.zeros(self._num_nodes(self.h), NODE_DTYPE)
        self._get_nodes(self.h, out.ctypes.data)
        return out

    def set_nodes(self, nodes):
        nodes = np.ascontiguousarray(nodes, NODE_DTYPE)
        self._set_nodes(self.h, nodes.ctypes.data, len(nodes))

    # -- point queries
    def sample_texture(self, index, u, v):
        out = np.zeros(4, np.float32)
        self._sample_texture(self.h, index, u, v, _fptr(out))
        return out

    def sample_sky(self, d):
        out = np.zeros(4, np.float32)
        self._sample_sky(self.h, _fptr(f3(d)), _fptr(out))
        return out

    def aabb_hit(self, bmin, bmax, o, d):
        return self._aabb_hit(_fptr(f3(bmin)), _fptr(f3(bmax)), _fptr(f3(o)), _fptr(f3(d)))

    def bvh_hit(self, o, d, cap=1 << 20):
        out = np.zeros(cap, np.int32)
        n = self._bvh_hit(self.h, _fptr(f3(o)), _fptr(f3(d)), _iptr(out), cap)
        return out[:min(n, cap)].copy()

    def tri_hit(self, tri, o, d):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = np.zeros((len(o), 4), np.float32)
        self._tri_hit(self.h, tri, len(o), _fptr(o), _fptr(d), _fptr(out))
        return out

    def closest_hits(self, o, d):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        ids = np.zeros(len(o), np.int32)
        tab = np.zeros((len(o), 3), np.float32)
        self._closest_hits(self.h, len(o), _fptr(o), _fptr(d), _iptr(ids), _fptr(tab))
        return ids, tab

    def scatter_mt(self, tri, lobe, normal, o, d, num):
        out = np.zeros((num, 4), np.float32)
        self._scatter_mt(self.h, tri, lobe, _fptr(f3(normal)), _fptr(f3(o)), _fptr(f3(d)), num, _fptr(out))
        return out

    def refract(self, tri, normal, o, d):
        out = np.zeros(6, np.float32)
        self._refract(self.h, tri, _fptr(f3(normal)), _fptr(f3(o)), _fptr(f3(d)), _fptr(out))
        return out

    def color_mt(self, depth, o, d, fresh_each=True):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = np.zeros((len(o), 4), np.float32)
        tir = np.zeros(len(o), np.int32)
        self._color_mt(self.h, depth, len(o), _fptr(o), _fptr(d), _fptr(out), int(fresh_each), _iptr(tir))
        return out, tir

    def color_keys(self, depth, o, d, keys):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        keys = np.ascontiguousarray(keys, np.uint32)
        out = np.zeros((len(o), 4), np.float32)
        self._color_keys(self.h, depth, len(o), _fptr(o), _fptr(d), keys.ctypes.data_as(C.POINTER(C.c_uint32)),
                         _fptr(out))
        return out

    # -- image-level
    def sample_radiance(self, frame, x0, y0, x1, y1):
        out = np.zeros(((y1 - y0) * (x1 - x0), 4, 4), np.float32)
        self._sample_radiance(self.h, frame, x0, y0, x1, y1, _fptr(out))
        return out

    def render_rows(self, frame, accum, row0=0, row1=None, out8=None, nthreads=1):
        if row1 is None:
            row1 = self.height
        assert accum.dtype == np.float32 and accum.flags.c_contiguous
        p8 = out8.ctypes.data if out8 is not None else None
        return self._render_rows(self.h, frame, row0, row1, _fptr(accum), p8, nthreads)

    def render_pixels(self, frame, xy, accum, nthreads=1):
        """The loop body for the listed pixels only: xy (n, 2) int32 as (col, row); accumulates into accum."""
        xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
        assert accum.dtype == np.float32 and accum.flags.c_contiguous
        return self._render_pixels(self.h, frame, len(xy), _iptr(xy), _fptr(accum), nthreads)

    def worst_ray(self):
        out = np.zeros(8, np.float32)
        self._worst_ray(self.h, _fptr(out))
        return out

    def ray_count(self):
        return self._ray_count(self.h)

    def tir_count(self):
        return self._tir_count(self.h)

    def reset_counts(self):
        self._reset_counts(self.h)

    def primary_hits(self, x0=0, y0=0, x1=None, y1=None):
        x1 = self.width if x1 is None else x1
        y1 = self.height if y1 is None else y1
        n = (x1 - x0) * (y1 - y0) * 4
        ids = np.zeros(n, np.int32)
        tab = np.zeros((n, 3), np.float32)
        self._primary_hits(self.h, x0, y0, x1, y1, _iptr(ids), _fptr(tab))
        return ids, tab

    def tonemap(self, accum, frame):
        accum = np.ascontiguousarray(accum, np.float32)
        npix = accum.size // 4
        out = np.zeros((npix, 3), np.uint8)
        self._tonemap(_fptr(accum), frame, npix, out.ctypes.data)
        return out

    def tables(self):
        c = np.zeros(512, np.float32)
        s = np.zeros(512, np.float32)
        g = np.zeros(256, np.float32)
        self._tables(_fptr(c), _fptr(s), _fptr(g))
        return c, s, g


class Reference(_Base):
    """The reference's own compiled objects (oracle/_ref/libref.so), where present."""
    prefix = "ref_"

    def __init__(self):
        self.lib = C.CDLL(REF_SO)
        L = self
        self.layout = L._fn("layout", _i, _i)
        self._create = L._fn("create", _vp)
        self._destroy = L._fn("destroy", None, _vp)
        self._set_params = L._fn("set_params", None, _vp, _i, _i, _i, _i, _i, _fp)
        self._set_camera = L._fn("set_camera", None, _vp, _fp, _fp, _f, _f)
        self._get_camera = L._fn("get_camera", None, _vp, _fp)
        self._put_camera = L._fn("put_camera", None, _vp, _fp)
        self._get_ray = L._fn("get_ray", None, _vp, _i, _i, _fp)
        self._add_triangles = L._fn("add_triangles", None, _vp, _vp, _i)
        self._num_triangles = L._fn("num_triangles", _i, _vp)
        self._get_triangles = L._fn("get_triangles", None, _vp, _vp)
        self._add_triangle_cfg = L._fn("add_triangle_cfg", None, _vp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _f, _f)
        self._add_texture_file = L._fn("add_texture_file", _i, _vp, C.c_char_p)
        self._load_skybox = L._fn("load_skybox", _i, _vp, _f, *([C.c_char_p] * 6))
        self._texture_pixels = L._fn("texture_pixels", C.c_longlong, C.c_char_p, _ip, _ip, _vp, C.c_size_t)
        self._sample_texture = L._fn("sample_texture", None, _vp, _i, _f, _f, _fp)
        self._sample_sky = L._fn("sample_sky", None, _vp, _fp, _fp)
        self._build_bvh = L._fn("build_bvh", _i, _vp)
        self._get_nodes = L._fn("get_nodes", None, _vp, _vp)
        self._aabb_hit = L._fn("aabb_hit", _i, _fp, _fp, _fp, _fp)
        self._bvh_hit = L._fn("bvh_hit", _i, _vp, _fp, _fp, _ip, _i)
        self._tri_hit = L._fn("tri_hit", None, _vp, _i, _i, _fp, _fp, _fp)
        self._closest_hits = L._fn("closest_hits", None, _vp, _i, _fp, _fp, _ip, _fp)
        self._scatter = L._fn("scatter", None, _vp, _i, _i, _fp, _fp, _fp, _i, _fp)
        self._refract = L._fn("refract", None, _vp, _i, _fp, _fp, _fp, _fp)
        self._color = L._fn("color", None, _vp, _i, _i, _fp, _fp, _fp, _i)
        self._render_rows = L._fn("render_rows", C.c_double, _vp, _i, _i, _i, _fp, _vp, _i)
        self.h = self._create()
        self.width = self.height = 0
        self._tmp = tempfile.TemporaryDirectory(prefix="rtref_tex_")
        self._ntex = 0

    def close(self):
        if self.h:
            self._destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def texture_pixels(self, path):
        """What Texture::Texture(path) holds after the reference's stb_image decoded the file: (H, W, 3) uint8,
        bottom row first; None when the reference cannot load it."""
        w, h = C.c_int(0), C.c_int(0)
        n = self._texture_pixels(os.fsencode(path), C.byref(w), C.byref(h), None, 0)
        if n < 0:
            return None
        out = np.zeros(n, np.uint8)
        self._texture_pixels(os.fsencode(path), C.byref(w), C.byref(h), out.ctypes.data, n)
        return out.reshape(h.value, w.value, 3)

    def set_params(self, width, height, max_depth, n_diff, n_spec, bg=(0, 0, 0), seed=0):
        self.width, self.height = width, height
        self._set_params(self.h, width, height, max_depth, n_diff, n_spec, _fptr(f3(bg)))

    def set_camera(self, eye, at, focal, rotate_deg=0.0):
        self._set_camera(self.h, _fptr(f3(eye)), _fptr(f3(at)), focal, rotate_deg)

    def get_camera(self):
        out = np.zeros(16, np.float32)
        self._get_camera(self.h, _fptr(out))
        return out.reshape(4, 4)

    def put_camera(self, cam):
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        self._put_camera(self.h, _fptr(cam))

    def get_ray(self, x, y):
        out = np.zeros(32, np.float32)
        self._get_ray(self.h, x, y, _fptr(out))
        return out.reshape(4, 2, 4)

    def set_triangles(self, tris):
        tris = np.ascontiguousarray(tris, TRI_DTYPE)
        assert self._num_triangles(self.h) == 0
        self._add_triangles(self.h, tris.ctypes.data, len(tris))

    def get_triangles(self):
        out = np.zeros(self._num_triangles(self.h), TRI_DTYPE)
        self._get_triangles(self.h, out.ctypes.data)
        return out

    def add_triangle_cfg(self, v0, v1, v2, side, color, metal, emit, transp, rough, ior):
        self._add_triangle_cfg(self.h, _fptr(f3(v0)), _fptr(f3(v1)), _fptr(f3(v2)), _fptr(f3(side)),
                               _fptr(f3(color)), int(metal), int(emit), int(transp), rough, ior)

    def _tex_file(self, rgb_mem):
        path = os.path.join(self._tmp.name, "t%d.ppm" % self._ntex)
        self._ntex += 1
        write_ppm(path, np.ascontiguousarray(rgb_mem, np.uint8))
        return path.encode()

    def add_texture(self, rgb_mem):
        r = self._add_texture_file(self.h, self._tex_file(rgb_mem))
        assert r >= 0
        return r

    def load_skybox(self, brightness, faces_mem):
        paths = [self._tex_file(f) for f in faces_mem]
        ok = self._load_skybox(self.h, brightness, *paths)
        assert ok == 1

    def build_bvh(self):
        self._nnodes = self._build_bvh(self.h)
        return self._nnodes

    def get_nodes(self):
        out = np.zeros(self._nnodes, NODE_DTYPE)
        self._get_nodes(self.h, out.ctypes.data)
        return out

    def sample_texture(self, index, u, v):
        out = np.zeros(4, np.float32)
        self._sample_texture(self.h, index, u, v, _fptr(out))
        return out

    def sample_sky(self, d):
        out = np.zeros(4, np.float32)
        self._sample_sky(self.h, _fptr(f3(d)), _fptr(out))
        return out

    def aabb_hit(self, bmin, bmax, o, d):
        return self._aabb_hit(_fptr(f3(bmin)), _fptr(f3(bmax)), _fptr(f3(o)), _fptr(f3(d)))

    def bvh_hit(self, o, d, cap=1 << 20):
        out = np.zeros(cap, np.int32)
        n = self._bvh_hit(self.h, _fptr(f3(o)), _fptr(f3(d)), _iptr(out), cap)
        return out[:min(n, cap)].copy()

    def tri_hit(self, tri, o, d):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = np.zeros((len(o), 4), np.float32)
        self._tri_hit(self.h, tri, len(o), _fptr(o), _fptr(d), _fptr(out))
        return out

    def closest_hits(self, o, d):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        ids = np.zeros(len(o), np.int32)
        tab = np.zeros((len(o), 3), np.float32)
        self._closest_hits(self.h, len(o), _fptr(o), _fptr(d), _iptr(ids), _fptr(tab))
        return ids, tab

    def scatter(self, tri, lobe, normal, o, d, num):
        out = np.zeros((num, 4), np.float32)
        self._scatter(self.h, tri, lobe, _fptr(f3(normal)), _fptr(f3(o)), _fptr(f3(d)), num, _fptr(out))
        return out

    def refract(self, tri, normal, o, d):
        out = np.zeros(5, np.float32)
        self._refract(self.h, tri, _fptr(f3(normal)), _fptr(f3(o)), _fptr(f3(d)), _fptr(out))
        return out

    def color(self, depth, o, d, fresh_each=True):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = np.zeros((len(o), 4), np.float32)
        self._color(self.h, depth, len(o), _fptr(o), _fptr(d), _fptr(out), int(fresh_each))
        return out

    def render_rows(self, frame, accum, row0=0, row1=None, out8=None, nthreads=1):
        if row1 is None:
            row1 = self.height
        assert accum.dtype == np.float32 and accum.flags.c_contiguous
        p8 = out8.ctypes.data if out8 is not None else None
        return self._render_rows(self.h, frame, row0, row1, _fptr(accum), p8, nthreads)
