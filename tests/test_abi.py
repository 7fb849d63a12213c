"""The C-ABI libraries load and export every symbol include/*.h declares
(no compute call is made here: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

DECL = re.compile(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+((?:rt|rth)_[a-z0-9_]+)\s*\(", re.M)


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(DECL.findall(text)))


def test_device_library_exports_every_declared_symbol():
    names = declared("mi355rt.h")
    assert {"rt_scene_create", "rt_render_device", "rt_query_primary", "rt_trace_rays", "rt_shard_tiles"} <= set(names)
    lib = ctypes.CDLL(os.path.join(ROOT, "raytracer_amd", "libmi355rt.so"))
    for n in names:
        assert hasattr(lib, n), f"libmi355rt.so does not export {n}"
    lib.rt_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.rt_version()


def test_host_library_exports_every_declared_symbol(libs):
    names = declared("mi355rt_host.h")
    assert "rth_scene_load" in names
    lib = ctypes.CDLL(os.path.join(ROOT, "raytracer_amd", "libmi355rt_host.so"))
    for n in names:
        assert hasattr(lib, n), f"libmi355rt_host.so does not export {n}"


def test_python_binding_lists_the_same_symbols():
    from raytracer_amd import device

    assert sorted(device.EXPORTS + ["rt_shard_tiles"]) == declared("mi355rt.h")


def test_struct_mirrors_have_the_c_sizes():
    """ctypes mirrors vs sizeof() in the header, measured with the C compiler."""
    import subprocess
    import tempfile

    from raytracer_amd import _abi

    src = '#include <stdio.h>\n#include "mi355rt.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",' \
          "sizeof(rt_render_options),sizeof(rt_camera),sizeof(rt_material),sizeof(rt_light),sizeof(rt_texture)," \
          "sizeof(rt_scene_desc),sizeof(rt_render_params),sizeof(rt_render_stats));return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    mirrors = [_abi.RenderOptions, _abi.Camera, _abi.Material, _abi.Light, _abi.Texture, _abi.SceneDesc, _abi.RenderParams,
               _abi.RenderStats]
    assert sizes == [ctypes.sizeof(m) for m in mirrors]


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under raytracer_amd/ or include/ may reference it."""
    for base in ("raytracer_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle" not in text.lower() or f == "rt_internal.h" and "oracle/rt_oracle.cpp" in text, os.path.join(dirpath, f)


def test_scene_create_fails_loudly_without_a_gpu(libs):
    """No CPU fallback: on a box without a GPU rt_scene_create must return RT_ERR_DEVICE."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from conftest import load_scene
    from raytracer_amd import device

    s = load_scene("test", size=8)
    with pytest.raises(device.DeviceError, match="no HIP device"):
        device.DeviceScene(s.desc)


def test_null_arrays_are_invalid_not_faults(libs):
    """a NULL array with a non-zero count in rt_scene_desc is RT_ERR_INVALID (checked before any device call,
    so this runs without a GPU); a NULL descriptor likewise."""
    import ctypes as C

    from raytracer_amd import _abi, device, host

    s = host.LoadedScene.from_file(os.path.join(ROOT, "scenes", "test.json"))
    lib = device.lib()
    h = C.c_void_p()
    assert lib.rt_scene_create(None, C.byref(h)) == _abi.RT_ERR_INVALID
    for field, ctype in (("prim_type", C.c_uint8), ("world", C.c_double), ("materials", _abi.Material), ("lights", _abi.Light), ("textures", _abi.Texture)):
        d = _abi.SceneDesc.from_buffer_copy(s.desc.contents)
        setattr(d, field, C.POINTER(ctype)())
        assert lib.rt_scene_create(C.byref(d), C.byref(h)) == _abi.RT_ERR_INVALID, field
        assert b"null array" in lib.rt_last_error()
        assert not h.value


def test_render_multi_rejects_empty_scene_lists(libs):
    import ctypes as C

    from raytracer_amd import _abi, device

    lib = device.lib()
    n = C.c_uint64(0)
    assert lib.rt_render_multi_device(None, 0, None, None, C.byref(n), None, None) == _abi.RT_ERR_INVALID
    assert lib.rt_render_multi(None, 2, None, None, None, C.byref(n), None) == _abi.RT_ERR_INVALID


def test_shard_tiles_partition_the_frame():
    """rt_shard_tiles is pure host code: the shards of a frame are disjoint and cover it."""
    import numpy as np

    from raytracer_amd import _abi, distributed

    opts = _abi.RenderOptions(width=200, height=120, max_depth=3, samples_per_pixel=4)
    for world in (1, 2, 3, 8):
        cover = np.zeros((120, 200), np.int32)
        sizes = []
        for rank in range(world):
            p = distributed.shard_params(rank, world, 4, tile_size=32)
            xy, tw, th = distributed.shard_tiles(opts, p)
            sizes.append(len(xy))
            for x, y in xy:
                cover[y : y + th, x : x + tw] += 1
        assert (cover == 1).all()
        assert max(sizes) - min(sizes) <= 1
    # sample shards: same tiles, disjoint sample ranges
    a = distributed.shard_params(0, 4, 64, sample_shards=2)
    b = distributed.shard_params(1, 4, 64, sample_shards=2)
    assert (a.shard_index, a.shard_count, a.s0, a.s1) == (0, 2, 0, 32)
    assert (b.shard_index, b.shard_count, b.s0, b.s1) == (0, 2, 32, 64)
