"""ctypes binding of libmi355rt_host.so (include/mi355rt_host.h): the scene
front end that stands in for the reference's Rust loader (src/main.rs:34-49)."""
import ctypes as C
import os

import numpy as np

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class HostError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmi355rt_host.so")
        if not os.path.exists(path):
            raise HostError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        l = C.CDLL(path)
        l.rth_scene_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        l.rth_scene_load.restype = C.c_int
        l.rth_scene_load_string.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        l.rth_scene_load_string.restype = C.c_int
        l.rth_scene_desc.argtypes = [C.c_void_p]
        l.rth_scene_desc.restype = C.POINTER(_abi.SceneDesc)
        l.rth_scene_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        l.rth_scene_set_option.restype = C.c_int
        l.rth_scene_free.argtypes = [C.c_void_p]
        l.rth_scene_free.restype = None
        l.rth_transform_from_json.argtypes = [C.c_char_p] + [C.POINTER(C.c_double)] * 3
        l.rth_transform_from_json.restype = C.c_int
        l.rth_camera_to_world.argtypes = [C.POINTER(C.c_double)] * 4
        l.rth_camera_to_world.restype = C.c_int
        l.rth_write_png.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)]
        l.rth_write_png.restype = C.c_int
        l.rth_image_load.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.c_uint64]
        l.rth_image_load.restype = C.c_int
        l.rth_last_error.restype = C.c_char_p
        _LIB = l
    return _LIB


def _check(rc):
    if rc != 0:
        raise HostError(f"[{rc}] {lib().rth_last_error().decode()}")


class LoadedScene:
    """Owns an `rth_scene`; `.desc` is the `rt_scene_desc*` for rt_scene_create."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    @classmethod
    def from_file(cls, path):
        h = C.c_void_p()
        _check(lib().rth_scene_load(os.fsencode(path), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_string(cls, text, asset_base=""):
        h = C.c_void_p()
        _check(lib().rth_scene_load_string(text.encode(), os.fsencode(asset_base), C.byref(h)))
        return cls(h.value)

    @property
    def desc(self):
        p = lib().rth_scene_desc(self._h)
        p._owner = self  # the description points into this scene's arrays: keep it alive as long as the pointer
        return p

    def set_option(self, key, value):
        _check(lib().rth_scene_set_option(self._h, key.encode(), float(value)))

    # convenience views used by tests and reports
    @property
    def opts(self):
        return self.desc.contents.opts

    @property
    def n_prims(self):
        return int(self.desc.contents.n_prims)

    def prim_types(self):
        d = self.desc.contents
        return np.ctypeslib.as_array(d.prim_type, shape=(d.n_prims,)).copy() if d.n_prims else np.zeros(0, np.uint8)

    def world(self):
        d = self.desc.contents
        return np.ctypeslib.as_array(d.world, shape=(d.n_transforms, 3, 4)).copy()

    def close(self):
        if self._h:
            lib().rth_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def transform_from_json(text):
    m = (C.c_double * 16)()
    inv = (C.c_double * 16)()
    invt = (C.c_double * 16)()
    _check(lib().rth_transform_from_json(text.encode(), m, inv, invt))
    f = lambda a: np.array(a[:], dtype=np.float64).reshape(4, 4)
    return f(m), f(inv), f(invt)


def camera_to_world(pos, target, up):
    arr = lambda v: (C.c_double * 3)(*[float(x) for x in v])
    out = (C.c_double * 9)()
    _check(lib().rth_camera_to_world(arr(pos), arr(target), arr(up), out))
    return np.array(out[:], dtype=np.float64).reshape(3, 3)


def image_load(path):
    """Texture::load's `image::open(path).to_rgb8()` -> uint8 [h, w, 3]."""
    w, h = C.c_uint32(), C.c_uint32()
    _check(lib().rth_image_load(os.fsencode(path), C.byref(w), C.byref(h), None, 0))
    rgb = np.zeros((h.value, w.value, 3), np.uint8)
    _check(lib().rth_image_load(os.fsencode(path), C.byref(w), C.byref(h), rgb.ctypes.data_as(C.POINTER(C.c_uint8)), rgb.size))
    return rgb


def write_png(path, rgba8):
    rgba8 = np.ascontiguousarray(rgba8, dtype=np.uint8)
    h, w, c = rgba8.shape
    assert c == 4
    _check(lib().rth_write_png(os.fsencode(path), w, h, rgba8.ctypes.data_as(C.POINTER(C.c_uint8))))
