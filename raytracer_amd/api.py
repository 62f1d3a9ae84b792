"""ctypes bindings over the C ABI of include/rtx.h plus a thin `RayTracer` host class.

The reference's only public method is RayTracer::parseConfigFile
(/root/reference/include/RayTracer/RayTracer.h:14); its private members are what the hot
path consumes.  `RayTracer` here mirrors those members (set_frame / set_camera /
add_triangles / add_texture / set_skybox / render) on top of the C ABI so that tests and the
benchmark read like the reference's own driver code.  All compute happens in
librtx_hip.so; there is no fallback: without the library (or without a GPU) calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build
from .scenes import NODE_DTYPE, TRI_DTYPE

_vp, _i, _f, _u32, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_uint32, C.c_uint64
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


class RtxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rtx error %d: %s" % (code, msg))
        self.code = code


class Stats(C.Structure):
    _fields_ = [("rays", _u64), ("rays_per_level", _u64 * 16), ("node_tests", _u64), ("tri_tests", _u64),
                ("shaded_hits", _u64), ("texture_fetches", _u64), ("trace_ms", C.c_double), ("shade_ms", C.c_double),
                ("total_ms", C.c_double), ("trace_launches", _u32), ("chunks", _u32),
                ("trace_ms_per_level", C.c_double * 16), ("wide_visits", C.c_uint64), ("leaf_fetches", C.c_uint64),
                ("queue_retries", C.c_uint64), ("rays_set_aside", C.c_uint64),
                ("rays_untraced", C.c_uint64), ("rays_wave_traced", C.c_uint64)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if not k.endswith("_per_level")}
        d["rays_per_level"] = [int(x) for x in self.rays_per_level]
        d["trace_ms_per_level"] = [float(x) for x in self.trace_ms_per_level]
        return d


HIP_SYMBOLS = {
    "rtx_last_error": (C.c_char_p, []),
    "rtx_device_count": (_i, []),
    "rtx_scene_create": (_i, [C.POINTER(_vp), _i]),
    "rtx_scene_destroy": (_i, [_vp]),
    "rtx_scene_set_stream": (_i, [_vp, _vp]),
    "rtx_scene_set_params": (_i, [_vp, _i, _i, _i, _i, _i, _fp, _u64]),
    "rtx_scene_set_camera": (_i, [_vp, _fp, _fp, _fp, _fp]),
    "rtx_scene_set_triangles": (_i, [_vp, _vp, _u32]),
    "rtx_scene_set_bvh": (_i, [_vp, _vp, _u32]),
    "rtx_scene_add_texture": (_i, [_vp, _vp, _i, _i]),
    "rtx_scene_set_skybox": (_i, [_vp, _f, _ip]),
    "rtx_scene_set_tables": (_i, [_vp, _fp, _fp, _fp]),
    "rtx_scene_set_tiles": (_i, [_vp, _i, _i, _i]),
    "rtx_scene_set_arena_bytes": (_i, [_vp, _u64]),
    "rtx_scene_set_accum_device": (_i, [_vp, _vp]),
    "rtx_accum_reset": (_i, [_vp]),
    "rtx_render_frames": (_i, [_vp, _u32, _u32]),
    "rtx_accum_read": (_i, [_vp, _fp]),
    "rtx_accum_write": (_i, [_vp, _fp]),
    "rtx_scene_set_tonemap_table": (_i, [_vp, _vp, _vp, _i, C.c_uint8, C.c_uint8]),
    "rtx_tonemap_read": (_i, [_vp, _i, _vp]),
    "rtx_render_frames_tonemapped": (_i, [_vp, _u32, _u32, _vp]),
    "rtx_primary_hits": (_i, [_vp, _i, _ip, _fp]),
    "rtx_host_alloc": (_vp, [C.c_size_t]),
    "rtx_host_free": (_i, [_vp]),
    "rtx_trace_rays": (_i, [_vp, _i, _u32, _fp, _fp, _ip, _fp]),
    "rtx_trace_rays_noalloc": (_i, [_vp, _i, _u32, _vp, _vp, _vp, _vp]),
    "rtx_sample_radiance": (_i, [_vp, _u32, _fp]),
    "rtx_gather_tiles": (_i, [C.POINTER(_vp), _i]),
    "rtx_exchange_tiles": (_i, [C.POINTER(_vp), _i]),
    "rtx_scene_tile_log": (_i, [_vp, _ip, _i]),
    "rtx_scene_tile_log_clear": (_i, [_vp]),
    "rtx_scene_set_instrumented": (_i, [_vp, _i]),
    "rtx_get_stats": (_i, [_vp, C.POINTER(Stats)]),
}

HOST_SYMBOLS = {
    "rtxh_last_error": (C.c_char_p, []),
    "rtxh_camera": (_i, [_fp, _fp, _f, _f, _i, _i, _fp]),
    "rtxh_build_bvh": (_i, [_vp, _u32, _vp, _u32]),
    "rtxh_make_triangle": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _f, _f, _vp]),
    "rtxh_tables": (_i, [_fp, _fp, _fp]),
    "rtxh_tonemap": (_i, [_fp, _i, C.c_size_t, _vp]),
    "rtxh_tonemap_table": (_i, [_vp, _vp, _i, _ip, _vp]),
    "rtxh_load_texture": (C.c_longlong, [C.c_char_p, _ip, _ip, _vp, C.c_size_t]),
    "rtxh_load_model": (C.c_longlong, [C.c_char_p, _fp, _f, _i, _i, _i, _f, _f, _i, _fp, _vp, C.c_size_t, _ip]),
    "rtxh_run_config": (_i, [C.c_char_p, C.c_char_p, _i]),
}


def _load(path, table):
    if not os.path.exists(path):
        raise RtxError(-3, "%s is missing: run `python -m raytracer_amd.build` (there is no CPU fallback)" % path)
    lib = C.CDLL(path)
    for name, (res, args) in table.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_hip = None
_host = None


def hip_lib():
    global _hip
    if _hip is None:
        # A process can hold only one HIP runtime on the GPU.  PyTorch ships its own copy of libamdhip64;
        # if torch is going to be used in this process (device tensors as accumulators, torch.distributed) it
        # has to be loaded first so that librtx_hip.so binds to that same copy instead of bringing in the
        # system one, after which torch would find "no HIP GPUs".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _hip = _load(_build.HIP_SO, HIP_SYMBOLS)
    return _hip


def host_lib():
    global _host
    if _host is None:
        _host = _load(_build.HOST_SO, HOST_SYMBOLS)
    return _host


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1))


def _p(a):
    return a.ctypes.data_as(_fp)


def _chk(rc):
    if rc < 0:
        raise RtxError(rc, hip_lib().rtx_last_error().decode())
    return rc


def _chkh(rc):
    if rc < 0:
        raise RtxError(rc, host_lib().rtxh_last_error().decode())
    return rc


# ---- host-side helpers (librtx_host.so) ----------------------------------------------------------

def camera(eye, at, focal, rotate_deg, width, height):
    """Camera::setCamera -> (4,4) array: origin, leftUpCorner, rightStep, downStep."""
    out = np.zeros(16, np.float32)
    _chkh(host_lib().rtxh_camera(_p(_f32(eye)), _p(_f32(at)), focal, rotate_deg, width, height, _p(out)))
    return out.reshape(4, 4)


def build_bvh(tris):
    """BVH::buildTree -> NODE_DTYPE array in the reference's breadth-first order."""
    tris = np.ascontiguousarray(tris, TRI_DTYPE)
    cap = 3 * len(tris) + 1
    nodes = np.zeros(cap, NODE_DTYPE)
    n = _chkh(host_lib().rtxh_build_bvh(tris.ctypes.data, len(tris), nodes.ctypes.data, cap))
    return nodes[:n].copy()


def make_triangle(v0, v1, v2, normal_side, color, metal, emit, transp, rough, ior):
    """RayTracer::addTriangle -> one TRI_DTYPE record."""
    out = np.zeros(1, TRI_DTYPE)
    _chkh(host_lib().rtxh_make_triangle(_p(_f32(v0)), _p(_f32(v1)), _p(_f32(v2)), _p(_f32(normal_side)),
                                        _p(_f32(color)), int(metal), int(emit), int(transp), rough, ior,
                                        out.ctypes.data))
    return out


def tables():
    c, s, g = np.zeros(512, np.float32), np.zeros(512, np.float32), np.zeros(256, np.float32)
    _chkh(host_lib().rtxh_tables(_p(c), _p(s), _p(g)))
    return c, s, g


def tonemap(accum, frame):
    accum = np.ascontiguousarray(accum, np.float32)
    npix = accum.size // 4
    out = np.zeros((npix, 3), np.uint8)
    _chkh(host_lib().rtxh_tonemap(_p(accum.reshape(-1)), frame, npix, out.ctypes.data))
    return out


def load_texture(path):
    """Texture::Texture(path): (H, W, 3) uint8, bottom row first."""
    w, h = C.c_int(0), C.c_int(0)
    need = _chkh(host_lib().rtxh_load_texture(os.fsencode(path), C.byref(w), C.byref(h), None, 0))
    out = np.zeros(need, np.uint8)
    _chkh(host_lib().rtxh_load_texture(os.fsencode(path), C.byref(w), C.byref(h), out.ctypes.data, need))
    return out.reshape(h.value, w.value, 3)


def load_model(path, origin=(0.0, 0.0, 0.0), scale=1.0, is_metal=0, is_light_emitting=0, is_transparent=0, specular_roughness=0.0,
               refractive_index=1.0, texture_index=-1, override_color=None):
    """RayTracer::loadModel without its texture file: (triangle records as scenes.TRI_DTYPE, whether texture_index was kept)."""
    from . import scenes
    o = (C.c_float * 3)(*[float(x) for x in origin])
    col = (C.c_float * 3)(*[float(x) for x in override_color]) if override_color is not None else None
    textured = C.c_int(0)
    args = (os.fsencode(path), o, float(scale), int(is_metal), int(is_light_emitting), int(is_transparent), float(specular_roughness),
            float(refractive_index), int(texture_index), col)
    n = _chkh(host_lib().rtxh_load_model(*args, None, 0, C.byref(textured)))
    out = np.zeros(n, scenes.TRI_DTYPE)
    _chkh(host_lib().rtxh_load_model(*args, out.ctypes.data, n, C.byref(textured)))
    return out, bool(textured.value)


def tonemap_table():
    """The tone map as a step function of accum/frame: (bits uint32[n], values uint8[n], negative, nan)."""
    bits, vals = np.zeros(1024, np.uint32), np.zeros(1024, np.uint8)
    n, special = C.c_int(0), np.zeros(2, np.uint8)
    _chkh(host_lib().rtxh_tonemap_table(bits.ctypes.data, vals.ctypes.data, 1024, C.byref(n), special.ctypes.data))
    return bits[:n.value].copy(), vals[:n.value].copy(), int(special[0]), int(special[1])


def run_config(config_path, out_dir=".", n_devices=1):
    _chkh(host_lib().rtxh_run_config(os.fsencode(config_path), os.fsencode(out_dir), n_devices))


def gather_tiles(tracers):
    """rtx_gather_tiles: the tile shares of tracers[1:] (one GPU each) travel to tracers[0]'s accumulator over RCCL."""
    arr = (_vp * len(tracers))(*[t.h for t in tracers])
    _chk(hip_lib().rtx_gather_tiles(arr, len(tracers)))


def exchange_tiles(tracers):
    """rtx_exchange_tiles: the tiles every tracer has rendered since the last exchange are added to tracers[0]'s
    accumulator and zeroed where they came from (dynamically balanced frames)."""
    arr = (_vp * len(tracers))(*[t.h for t in tracers])
    _chk(hip_lib().rtx_exchange_tiles(arr, len(tracers)))


# ---- the device-side scene -----------------------------------------------------------------------

class RayTracer:
    """Host driver of one scene on one GPU, mirroring the members of the reference's RayTracer."""

    def __init__(self, device=0):
        self.lib = hip_lib()
        h = _vp()
        _chk(self.lib.rtx_scene_create(C.byref(h), device))
        self.h = h
        self.width = self.height = 0
        self._tables_set = False

    def close(self):
        for p, _ in self.__dict__.pop("_pinned_pool", {}).values():
            self.lib.rtx_host_free(p)
        if getattr(self, "h", None):
            self.lib.rtx_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene description
    def set_frame(self, width, height, max_depth, n_diffuse, n_specular, background=(0, 0, 0), seed=0):
        self.width, self.height = width, height
        _chk(self.lib.rtx_scene_set_params(self.h, width, height, max_depth, n_diffuse, n_specular,
                                           _p(_f32(background)), seed))

    def set_camera(self, eye, at, focal, rotate_deg=0.0):
        self.set_camera_raw(camera(eye, at, focal, rotate_deg, self.width, self.height))

    def set_camera_raw(self, cam):
        cam = np.ascontiguousarray(cam, np.float32).reshape(4, 4)
        _chk(self.lib.rtx_scene_set_camera(self.h, _p(cam[0]), _p(cam[1]), _p(cam[2]), _p(cam[3])))

    def set_triangles(self, tris, nodes=None):
        tris = np.ascontiguousarray(tris, TRI_DTYPE)
        _chk(self.lib.rtx_scene_set_triangles(self.h, tris.ctypes.data, len(tris)))
        if nodes is None:
            nodes = build_bvh(tris)
        nodes = np.ascontiguousarray(nodes, NODE_DTYPE)
        _chk(self.lib.rtx_scene_set_bvh(self.h, nodes.ctypes.data, len(nodes)))
        if not self._tables_set:
            c, s, g = tables()
            _chk(self.lib.rtx_scene_set_tables(self.h, _p(c), _p(s), _p(g)))
            self._tables_set = True
        return nodes

    def add_texture(self, rgb_mem):
        rgb_mem = np.ascontiguousarray(rgb_mem, np.uint8)
        h, w, _ = rgb_mem.shape
        return _chk(self.lib.rtx_scene_add_texture(self.h, rgb_mem.ctypes.data, w, h))

    def set_skybox(self, brightness, tex_ids):
        ids = np.ascontiguousarray(tex_ids, np.int32)
        _chk(self.lib.rtx_scene_set_skybox(self.h, brightness, ids.ctypes.data_as(_ip)))

    def load_scene(self, scene, nodes=None):
        """Everything from a raytracer_amd.scenes.Scene."""
        self.set_frame(scene.width, scene.height, scene.max_depth, scene.n_diff, scene.n_spec, scene.bg, scene.seed)
        self.set_camera(scene.eye, scene.at, scene.focal, scene.rotate)
        for t in scene.textures:
            self.add_texture(t)
        if scene.sky:
            self.set_skybox(scene.sky[0], scene.sky[1])
        return self.set_triangles(scene.tris, nodes)

    # -- execution control
    def set_stream(self, stream_handle):
        _chk(self.lib.rtx_scene_set_stream(self.h, stream_handle))

    def set_tiles(self, first=0, stride=1, count=-1):
        _chk(self.lib.rtx_scene_set_tiles(self.h, first, stride, count))

    def tile_log(self):
        """Tile numbers rendered into the accumulator since the log was last cleared."""
        n = _chk(self.lib.rtx_scene_tile_log(self.h, None, 0))
        out = np.zeros(max(n, 1), np.int32)
        _chk(self.lib.rtx_scene_tile_log(self.h, out.ctypes.data_as(_ip), n))
        return out[:n]

    def clear_tile_log(self):
        _chk(self.lib.rtx_scene_tile_log_clear(self.h))

    def set_arena_bytes(self, nbytes):
        _chk(self.lib.rtx_scene_set_arena_bytes(self.h, nbytes))

    def set_accum_device(self, device_ptr):
        _chk(self.lib.rtx_scene_set_accum_device(self.h, device_ptr))

    def set_instrumented(self, on):
        _chk(self.lib.rtx_scene_set_instrumented(self.h, int(on)))

    # -- the hot path
    def reset(self):
        _chk(self.lib.rtx_accum_reset(self.h))

    def render(self, first_frame=1, n_frames=1):
        _chk(self.lib.rtx_render_frames(self.h, first_frame, n_frames))

    def accum(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        _chk(self.lib.rtx_accum_read(self.h, _p(out.reshape(-1))))
        return out

    def _set_tone_table(self):
        if not getattr(self, "_tone_set", False):
            bits, vals, neg, nan = tonemap_table()
            _chk(self.lib.rtx_scene_set_tonemap_table(self.h, bits.ctypes.data, vals.ctypes.data, len(bits), neg, nan))
            self._tone_set = True

    def render_images(self, first_frame=1, n_frames=1):
        """Frames first_frame .. first_frame + n_frames - 1 in merged pixel batches, and the 8-bit image after each of
        them: (n_frames, H, W, 3) uint8."""
        self._set_tone_table()
        out = np.zeros((n_frames, self.height, self.width, 3), np.uint8)
        ptrs = (C.c_void_p * n_frames)(*[out[k].ctypes.data for k in range(n_frames)])
        _chk(self.lib.rtx_render_frames_tonemapped(self.h, first_frame, n_frames, C.cast(ptrs, C.c_void_p)))
        return out

    def tonemap_read(self, frame):
        """The 8-bit image after `frame` frames, tone-mapped on the device: (H, W, 3) uint8."""
        if not getattr(self, "_tone_set", False):
            bits, vals, neg, nan = tonemap_table()
            _chk(self.lib.rtx_scene_set_tonemap_table(self.h, bits.ctypes.data, vals.ctypes.data, len(bits), neg, nan))
            self._tone_set = True
        out = np.zeros((self.height, self.width, 3), np.uint8)
        _chk(self.lib.rtx_tonemap_read(self.h, frame, out.ctypes.data))
        return out

    def write_accum(self, accum):
        accum = np.ascontiguousarray(accum, np.float32)
        assert accum.size == self.height * self.width * 4
        _chk(self.lib.rtx_accum_write(self.h, _p(accum.reshape(-1))))

    def _pinned(self, name, dtype, count):
        """A numpy array over page-locked memory from rtx_host_alloc, kept with the tracer and grown on demand."""
        pool = self.__dict__.setdefault("_pinned_pool", {})
        nbytes = int(count) * np.dtype(dtype).itemsize
        if name not in pool or pool[name][1] < nbytes:
            if name in pool:
                self.lib.rtx_host_free(pool[name][0])
            p = self.lib.rtx_host_alloc(nbytes)
            if not p:
                raise RtxError(-4, self.lib.rtx_last_error().decode())
            pool[name] = (p, nbytes)
        buf = (C.c_char * nbytes).from_address(pool[name][0])
        return np.frombuffer(buf, dtype=dtype, count=count)

    def primary_hits(self, mode=0, want_tab=True, pinned=False):
        """pinned: the arrays returned are views of page-locked memory that the next pinned call overwrites."""
        n = self.height * self.width * 4
        if pinned:
            ids = self._pinned("ids", np.int32, n)
            tab = self._pinned("tab", np.float32, n * 3).reshape(n, 3) if want_tab else None
        else:
            ids = np.zeros(n, np.int32)
            tab = np.zeros((n, 3), np.float32) if want_tab else None
        _chk(self.lib.rtx_primary_hits(self.h, mode, ids.ctypes.data_as(_ip), _p(tab.reshape(-1)) if want_tab else None))
        return ids, tab

    def trace_rays(self, origins, directions, mode=0):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        ids = np.zeros(len(o), np.int32)
        tab = np.zeros((len(o), 3), np.float32)
        _chk(self.lib.rtx_trace_rays(self.h, mode, len(o), _p(o.reshape(-1)), _p(d.reshape(-1)),
                                     ids.ctypes.data_as(_ip), _p(tab.reshape(-1))))
        return ids, tab

    def trace_rays_noalloc(self, n, d_origins, d_directions, d_tri_ids, d_t_alpha_beta=None, mode=0):
        """rtx_trace_rays_noalloc: the four arguments are device addresses (ints) on this scene's GPU."""
        _chk(self.lib.rtx_trace_rays_noalloc(self.h, mode, n, d_origins, d_directions, d_tri_ids, d_t_alpha_beta))

    def sample_radiance(self, frame):
        out = np.zeros((self.height * self.width, 4, 4), np.float32)
        _chk(self.lib.rtx_sample_radiance(self.h, frame, _p(out.reshape(-1))))
        return out

    def stats(self):
        st = Stats()
        _chk(self.lib.rtx_get_stats(self.h, C.byref(st)))
        return st.as_dict()
