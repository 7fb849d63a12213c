"""ctypes binding of libmi355rt.so (include/mi355rt.h) -- the HIP hot path.

There is no CPU fallback: if the library is missing or no GPU is usable every
entry point raises `DeviceError`."""
import ctypes as C
import os

import numpy as np

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EXPORTS = [
    "rt_scene_create", "rt_scene_destroy", "rt_scene_set_pool_bytes", "rt_render_device", "rt_tonemap_device",
    "rt_render", "rt_render_aov_device", "rt_render_aov", "rt_query_primary", "rt_trace_rays", "rt_scene_info", "rt_last_error", "rt_version",
    "rt_render_progressive", "rt_render_multi_device", "rt_render_multi", "rt_scene_bvh",
    "rt_render_device_noalloc",
]

PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64)


class DeviceError(RuntimeError):
    pass


def lib_path():
    # MI355RT_LIB: load another build of the same library (kernel tuning experiments)
    return os.environ.get("MI355RT_LIB") or os.path.join(_HERE, "libmi355rt.so")


def lib():
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise DeviceError(f"{path} is missing: build it with __graft_entry__.build() (hipcc, gfx950)")
        # PyTorch bundles its own HIP runtime.  If torch is going to be used in this
        # process it must initialise that runtime BEFORE this library pulls in
        # /opt/rocm's libamdhip64 (two runtimes in one process: torch then reports
        # "No HIP GPUs are available").  torch is plumbing here, never required.
        if os.environ.get("MI355RT_NO_TORCH_PRELOAD") != "1":
            try:
                import torch

                if torch.cuda.is_available():
                    torch.cuda.init()
            except Exception:
                pass
        l = C.CDLL(path)
        vp, u32p, f64p, u8p, u64p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
        l.rt_scene_create.argtypes = [vp, C.POINTER(vp)]
        l.rt_scene_create.restype = C.c_int
        l.rt_scene_destroy.argtypes = [vp]
        l.rt_scene_destroy.restype = None
        l.rt_scene_set_pool_bytes.argtypes = [vp, C.c_uint64]
        l.rt_scene_set_pool_bytes.restype = C.c_int
        l.rt_render_device.argtypes = [vp, C.POINTER(_abi.RenderParams), vp, u64p, C.POINTER(_abi.RenderStats), vp]
        l.rt_render_device.restype = C.c_int
        l.rt_render_aov_device.argtypes = [vp, C.POINTER(_abi.RenderParams), vp, vp, vp, u64p, C.POINTER(_abi.RenderStats), vp]
        l.rt_render_aov_device.restype = C.c_int
        l.rt_render_aov.argtypes = [vp, C.POINTER(_abi.RenderParams), f64p, f64p, f64p, u64p, C.POINTER(_abi.RenderStats)]
        l.rt_render_aov.restype = C.c_int
        l.rt_tonemap_device.argtypes = [vp, vp, vp, vp]
        l.rt_tonemap_device.restype = C.c_int
        l.rt_render.argtypes = [vp, C.POINTER(_abi.RenderParams), f64p, u8p, u64p, C.POINTER(_abi.RenderStats)]
        l.rt_render.restype = C.c_int
        l.rt_render_progressive.argtypes = [vp, C.POINTER(_abi.RenderParams), C.c_uint32, u32p, PROGRESS_FN, vp, f64p, u64p]
        l.rt_render_progressive.restype = C.c_int
        l.rt_render_device_noalloc.argtypes = [vp, C.POINTER(_abi.RenderParams), vp, u64p, vp]
        l.rt_render_device_noalloc.restype = C.c_int
        l.rt_render_multi_device.argtypes = [C.POINTER(vp), C.c_uint32, C.POINTER(_abi.RenderParams), vp, u64p, C.POINTER(_abi.RenderStats), vp]
        l.rt_render_multi_device.restype = C.c_int
        l.rt_render_multi.argtypes = [C.POINTER(vp), C.c_uint32, C.POINTER(_abi.RenderParams), f64p, u8p, u64p, C.POINTER(_abi.RenderStats)]
        l.rt_render_multi.restype = C.c_int
        l.rt_scene_bvh.argtypes = [vp, vp, C.c_uint64, u32p, C.c_uint64, u64p]
        l.rt_scene_bvh.restype = C.c_int
        l.rt_query_primary.argtypes = [vp, u32p, f64p, f64p]
        l.rt_query_primary.restype = C.c_int
        l.rt_trace_rays.argtypes = [vp, C.c_uint32, f64p, f64p, f64p, C.c_int, u32p, f64p]
        l.rt_trace_rays.restype = C.c_int
        l.rt_scene_info.argtypes = [vp, u64p]
        l.rt_scene_info.restype = C.c_int
        l.rt_last_error.restype = C.c_char_p
        l.rt_version.restype = C.c_char_p
        _LIB = l
    return _LIB


def _check(rc):
    if rc != 0:
        raise DeviceError(f"[{rc}] {lib().rt_last_error().decode()}")


def _p(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


def _handles(scenes):
    arr = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    return arr


def render_multi(scenes, params=None, want_rgba=True, want_stats=False):
    """rt_render_multi: one frame by several replicas of a scene (one per GPU) -> (rgb_sum, rgba8 or None, ray_count, [stats] or None)."""
    params = params or _abi.RenderParams()
    s0 = scenes[0]
    acc = np.zeros((s0.height, s0.width, 3), np.float64)
    rgba = np.zeros((s0.height, s0.width, 4), np.uint8) if want_rgba else None
    count = C.c_uint64(0)
    stats = (_abi.RenderStats * len(scenes))() if want_stats else None
    _check(lib().rt_render_multi(_handles(scenes), len(scenes), C.byref(params), _p(acc, C.c_double), _p(rgba, C.c_uint8) if want_rgba else None,
                                 C.byref(count), stats))
    return acc, rgba, int(count.value), (list(stats) if want_stats else None)


def render_multi_device(scenes, d_accum_ptr, params=None, stream=None, want_stats=False):
    """rt_render_multi_device: d_accum_ptr is a device address on scenes[0]'s device (+= semantics) -> (ray_count, [stats] or None)."""
    params = params or _abi.RenderParams()
    count = C.c_uint64(0)
    stats = (_abi.RenderStats * len(scenes))() if want_stats else None
    _check(lib().rt_render_multi_device(_handles(scenes), len(scenes), C.byref(params), C.c_void_p(d_accum_ptr), C.byref(count), stats,
                                        C.c_void_p(stream or 0)))
    return int(count.value), (list(stats) if want_stats else None)


class DeviceScene:
    """An `rt_scene*` resident on the current HIP device."""

    def __init__(self, desc_ptr, pool_bytes=None):
        self._h = C.c_void_p()
        _check(lib().rt_scene_create(C.cast(desc_ptr, C.c_void_p), C.byref(self._h)))
        d = desc_ptr.contents
        self.width, self.height, self.spp = int(d.opts.width), int(d.opts.height), int(d.opts.samples_per_pixel)
        if pool_bytes:
            _check(lib().rt_scene_set_pool_bytes(self._h, int(pool_bytes)))

    def info(self):
        out = (C.c_uint64 * 8)()
        _check(lib().rt_scene_info(self._h, out))
        keys = ["bvh_nodes", "bvh_leaves", "bvh_depth", "scene_bytes", "bounded", "planes", "node_capacity", "job_capacity"]
        return dict(zip(keys, [int(v) for v in out]))

    def bvh(self):
        """rt_scene_bvh -> (nodes structured array, leaf primitive ids, root ref, built_on_device)."""
        info = (C.c_uint64 * 4)()
        _check(lib().rt_scene_bvh(self._h, None, 0, None, 0, info))
        node_t = np.dtype([("lmin", "<u2", 3), ("lmax", "<u2", 3), ("rmin", "<u2", 3), ("rmax", "<u2", 3), ("left", "<u4"), ("right", "<u4")])
        nodes = np.zeros(int(info[0]), node_t)
        ids = np.zeros(int(info[1]), np.uint32)
        _check(lib().rt_scene_bvh(self._h, nodes.ctypes.data_as(C.c_void_p), len(nodes), _p(ids, C.c_uint32), len(ids), info))
        return nodes, ids, int(info[2]), bool(info[3])

    def render(self, params=None, want_rgba=True, want_stats=False):
        """Host-buffer render -> (rgb_sum[h,w,3] f64, rgba8[h,w,4] or None, ray_count, stats or None)."""
        params = params or _abi.RenderParams()
        acc = np.zeros((self.height, self.width, 3), np.float64)
        rgba = np.zeros((self.height, self.width, 4), np.uint8) if want_rgba else None
        count = C.c_uint64(0)
        stats = _abi.RenderStats() if want_stats else None
        _check(
            lib().rt_render(
                self._h, C.byref(params), _p(acc, C.c_double), _p(rgba, C.c_uint8) if want_rgba else None,
                C.byref(count), C.byref(stats) if want_stats else None,
            )
        )
        return acc, rgba, int(count.value), stats

    def render_aov(self, params=None):
        """ColorData's three members per pixel (src/render/mod.rs:16-20) ->
        (rgb_sum[h,w,3], albedo[h,w,3] final = clamp(sum/spp), normal[h,w,3] of sample 0, ray_count)."""
        params = params or _abi.RenderParams()
        shape = (self.height, self.width, 3)
        acc, alb, nrm = np.zeros(shape, np.float64), np.zeros(shape, np.float64), np.zeros(shape, np.float64)
        count = C.c_uint64(0)
        _check(lib().rt_render_aov(self._h, C.byref(params), _p(acc, C.c_double), _p(alb, C.c_double), _p(nrm, C.c_double),
                                   C.byref(count), None))
        return acc, alb, nrm, int(count.value)

    def render_aov_device(self, d_accum_ptr, d_albedo_ptr, d_normal_ptr, params=None, stream=None):
        """Device-buffer form: albedo is the SUM over samples (+= semantics), normal is sample 0's."""
        params = params or _abi.RenderParams()
        count = C.c_uint64(0)
        _check(lib().rt_render_aov_device(self._h, C.byref(params), C.c_void_p(d_accum_ptr), C.c_void_p(d_albedo_ptr or 0),
                                          C.c_void_p(d_normal_ptr or 0), C.byref(count), None, C.c_void_p(stream or 0)))
        return int(count.value)

    def render_device(self, d_accum_ptr, params=None, stream=None, want_stats=True):
        """Device-buffer render: d_accum_ptr is a device address (int) of w*h*3 doubles."""
        params = params or _abi.RenderParams()
        count = C.c_uint64(0)
        stats = _abi.RenderStats() if want_stats else None
        _check(
            lib().rt_render_device(
                self._h, C.byref(params), C.c_void_p(d_accum_ptr), C.byref(count),
                C.byref(stats) if want_stats else None, C.c_void_p(stream or 0),
            )
        )
        return int(count.value), stats

    def render_device_noalloc(self, d_accum_ptr, params=None, stream=None):
        """rt_render_device_noalloc: as render_device, but every pool lives in the library's static device scratch."""
        params = params or _abi.RenderParams()
        count = C.c_uint64(0)
        _check(lib().rt_render_device_noalloc(self._h, C.byref(params), C.c_void_p(d_accum_ptr), C.byref(count), C.c_void_p(stream or 0)))
        return int(count.value)

    def render_progressive(self, n_passes, params=None, on_pass=None, want_accum=False):
        """rt_render_progressive -> (argb[h,w] u32, rgb_sum or None, ray_count).  on_pass(argb_view, passes_done, n_passes, rays)
        may return True to stop."""
        params = params or _abi.RenderParams()
        argb = np.zeros((self.height, self.width), np.uint32)
        acc = np.zeros((self.height, self.width, 3), np.float64) if want_accum else None
        count = C.c_uint64(0)

        def trampoline(_user, done, total, rays):
            return 1 if (on_pass and on_pass(argb, int(done), int(total), int(rays))) else 0

        cb = PROGRESS_FN(trampoline)
        _check(lib().rt_render_progressive(self._h, C.byref(params), int(n_passes), _p(argb, C.c_uint32), cb, None,
                                           _p(acc, C.c_double) if want_accum else None, C.byref(count)))
        return argb, acc, int(count.value)

    def tonemap_device(self, d_accum_ptr, d_rgba_ptr, stream=None):
        _check(lib().rt_tonemap_device(self._h, C.c_void_p(d_accum_ptr), C.c_void_p(d_rgba_ptr), C.c_void_p(stream or 0)))

    def query_primary(self):
        n = self.width * self.height
        ids = np.zeros(n, np.uint32)
        t = np.zeros(n, np.float64)
        nrm = np.zeros((n, 3), np.float64)
        _check(lib().rt_query_primary(self._h, _p(ids, C.c_uint32), _p(t, C.c_double), _p(nrm, C.c_double)))
        shape = (self.height, self.width)
        return ids.reshape(shape), t.reshape(shape), nrm.reshape(shape + (3,))

    def trace_rays(self, origin, direction, tmax=None, shadow=False):
        origin = np.ascontiguousarray(origin, np.float64)
        direction = np.ascontiguousarray(direction, np.float64)
        n = origin.shape[0]
        tmax = np.ascontiguousarray(tmax if tmax is not None else np.zeros(n), np.float64)
        ids = np.zeros(n, np.uint32)
        t = np.zeros(n, np.float64)
        _check(
            lib().rt_trace_rays(
                self._h, n, _p(origin, C.c_double), _p(direction, C.c_double), _p(tmax, C.c_double),
                1 if shadow else 0, _p(ids, C.c_uint32), _p(t, C.c_double),
            )
        )
        return ids, t

    def close(self):
        if self._h:
            lib().rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
