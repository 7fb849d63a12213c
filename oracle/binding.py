"""ctypes binding of oracle/librt_oracle.so.  TEST INFRASTRUCTURE: imported only
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
the raytracer_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODE_KD = 0  # the reference's kd-tree and traversal order (CPU baseline)
MODE_CANONICAL = 1  # order-independent hit semantics (GPU bit-exact target)
MODE_BRUTE = 2  # canonical semantics, no accelerator


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "librt_oracle.so")
        if not os.path.exists(path):
            build()
        l = C.CDLL(path)
        vp, u32p, f64p, u8p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        l.orc_scene_create.argtypes = [vp, C.POINTER(vp)]
        l.orc_scene_create.restype = C.c_int
        l.orc_scene_destroy.argtypes = [vp]
        l.orc_scene_destroy.restype = None
        l.orc_scene_info.argtypes = [vp, C.POINTER(C.c_uint64)]
        l.orc_render.argtypes = [vp, C.c_int, vp, C.c_int, f64p, u8p, C.POINTER(C.c_uint64)]
        l.orc_render.restype = C.c_int
        l.orc_render_aov.argtypes = [vp, C.c_int, vp, C.c_int, f64p, f64p, f64p, u8p, C.POINTER(C.c_uint64)]
        l.orc_render_aov.restype = C.c_int
        l.orc_query_primary.argtypes = [vp, C.c_int, u32p, f64p, f64p]
        l.orc_query_primary.restype = C.c_int
        l.orc_trace_rays.argtypes = [vp, C.c_int, C.c_uint32, f64p, f64p, f64p, C.c_int, u32p, f64p]
        l.orc_trace_rays.restype = C.c_int
        l.orc_last_error.restype = C.c_char_p
        l.orc_quadratic.argtypes = [C.c_double] * 3 + [f64p, f64p]
        l.orc_quadratic.restype = C.c_int
        l.orc_remap_value.argtypes = [C.c_double] * 5
        l.orc_remap_value.restype = C.c_double
        l.orc_to_argb_u32.argtypes = [C.c_double] * 3
        l.orc_to_argb_u32.restype = C.c_uint32
        l.orc_philox4x32_10.argtypes = [u32p, u32p, u32p]
        l.orc_philox4x32_10.restype = None
        l.orc_path_mix.argtypes = [C.c_uint64, C.c_uint32]
        l.orc_path_mix.restype = C.c_uint64
        l.orc_uniforms2.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, f64p]
        l.orc_uniforms2.restype = None
        l.orc_cosine_sample_hemisphere.argtypes = [f64p, C.c_double, C.c_double, f64p]
        l.orc_cosine_sample_hemisphere.restype = None
        l.orc_uniform_sample_cone.argtypes = [f64p, C.c_double, C.c_double, C.c_double, f64p]
        l.orc_uniform_sample_cone.restype = None
        l.orc_reflect.argtypes = [f64p, f64p, f64p]
        l.orc_reflect.restype = None
        l.orc_refract.argtypes = [f64p, f64p, C.c_double, f64p]
        l.orc_refract.restype = C.c_int
        l.orc_tonemap.argtypes = [f64p, C.c_uint32, u8p]
        l.orc_tonemap.restype = None
        _LIB = l
    return _LIB


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise OracleError(f"[{rc}] {lib().orc_last_error().decode()}")


def _p(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


class OracleScene:
    def __init__(self, desc_ptr):
        """desc_ptr: ctypes pointer to rt_scene_desc (e.g. LoadedScene.desc)."""
        self._h = C.c_void_p()
        _check(lib().orc_scene_create(C.cast(desc_ptr, C.c_void_p), C.byref(self._h)))
        d = desc_ptr.contents
        self.width, self.height, self.spp = int(d.opts.width), int(d.opts.height), int(d.opts.samples_per_pixel)

    def info(self):
        out = (C.c_uint64 * 6)()
        lib().orc_scene_info(self._h, out)
        keys = ["kd_nodes", "kd_leaves", "kd_refs", "kd_depth", "bounded", "unbounded"]
        return dict(zip(keys, [int(v) for v in out]))

    def render(self, params=None, mode=MODE_KD, threads=None, want_rgba=True):
        """-> (rgb_sum[h,w,3] f64, rgba8[h,w,4] u8 or None, ray_count)."""
        from raytracer_amd._abi import RenderParams

        params = params or RenderParams()
        threads = threads or min(os.cpu_count() or 1, 16)
        acc = np.zeros((self.height, self.width, 3), np.float64)
        rgba = np.zeros((self.height, self.width, 4), np.uint8) if want_rgba else None
        count = C.c_uint64(0)
        _check(
            lib().orc_render(
                self._h, mode, C.cast(C.byref(params), C.c_void_p), threads, _p(acc, C.c_double),
                _p(rgba, C.c_uint8) if want_rgba else None, C.byref(count),
            )
        )
        return acc, rgba, int(count.value)

    def render_aov(self, params=None, mode=MODE_KD, threads=None):
        """-> (rgb_sum, albedo_sum, normal[h,w,3], ray_count): ColorData's three members per pixel (render/mod.rs:16-20)."""
        from raytracer_amd._abi import RenderParams

        params = params or RenderParams()
        threads = threads or min(os.cpu_count() or 1, 16)
        shape = (self.height, self.width, 3)
        acc, alb, nrm = np.zeros(shape, np.float64), np.zeros(shape, np.float64), np.zeros(shape, np.float64)
        count = C.c_uint64(0)
        _check(
            lib().orc_render_aov(
                self._h, mode, C.cast(C.byref(params), C.c_void_p), threads, _p(acc, C.c_double), _p(alb, C.c_double),
                _p(nrm, C.c_double), None, C.byref(count),
            )
        )
        return acc, alb, nrm, int(count.value)

    def query_primary(self, mode=MODE_CANONICAL):
        n = self.width * self.height
        ids = np.zeros(n, np.uint32)
        t = np.zeros(n, np.float64)
        nrm = np.zeros((n, 3), np.float64)
        _check(lib().orc_query_primary(self._h, mode, _p(ids, C.c_uint32), _p(t, C.c_double), _p(nrm, C.c_double)))
        shape = (self.height, self.width)
        return ids.reshape(shape), t.reshape(shape), nrm.reshape(shape + (3,))

    def trace_rays(self, origin, direction, tmax=None, shadow=False, mode=MODE_CANONICAL):
        origin = np.ascontiguousarray(origin, np.float64)
        direction = np.ascontiguousarray(direction, np.float64)
        n = origin.shape[0]
        tmax = np.ascontiguousarray(tmax if tmax is not None else np.zeros(n), np.float64)
        ids = np.zeros(n, np.uint32)
        t = np.zeros(n, np.float64)
        _check(
            lib().orc_trace_rays(
                self._h, mode, n, _p(origin, C.c_double), _p(direction, C.c_double), _p(tmax, C.c_double),
                1 if shadow else 0, _p(ids, C.c_uint32), _p(t, C.c_double),
            )
        )
        return ids, t

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
