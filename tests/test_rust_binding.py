"""bindings/rust/mi355rt-sys/src/lib.rs cannot be compiled here (no Rust toolchain), so its #[repr(C)] structs and
its extern "C" block are checked textually against the ctypes mirrors of include/mi355rt.h (whose sizes
tests/test_abi.py checks with the C compiler) and against the header's own declarations."""
import ctypes as C
import os
import re

from conftest import ROOT
from test_abi import declared

SRC = os.path.join(ROOT, "bindings", "rust", "mi355rt-sys", "src", "lib.rs")

SCALARS = {"u8": C.c_uint8, "u32": C.c_uint32, "i32": C.c_int32, "u64": C.c_uint64, "f32": C.c_float, "f64": C.c_double}


def rust_structs():
    text = re.sub(r"//[^\n]*", "", open(SRC).read())
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\][^{]*?pub struct (\w+)\s*\{(.*?)\n\}", text, re.S):
        fields = re.findall(r"pub\s+(\w+)\s*:\s*([^,\n]+),", m.group(2))
        out[m.group(1)] = [(n, t.strip()) for n, t in fields]
    return out


def to_ctype(rust_type, mirrors):
    if rust_type in SCALARS:
        return SCALARS[rust_type]
    m = re.fullmatch(r"\[(\w+);\s*(\d+)\]", rust_type)
    if m:
        return SCALARS[m.group(1)] * int(m.group(2))
    m = re.fullmatch(r"\*const\s+(\w+)", rust_type)
    if m:
        inner = SCALARS.get(m.group(1)) or mirrors[m.group(1)]
        return C.POINTER(inner)
    return mirrors[rust_type]


def test_rust_structs_mirror_the_header():
    from raytracer_amd import _abi

    mirrors = {"rt_render_options": _abi.RenderOptions, "rt_camera": _abi.Camera, "rt_material": _abi.Material,
               "rt_light": _abi.Light, "rt_texture": _abi.Texture, "rt_scene_desc": _abi.SceneDesc,
               "rt_render_params": _abi.RenderParams, "rt_render_stats": _abi.RenderStats}
    structs = rust_structs()
    for name, mirror in mirrors.items():
        assert name in structs, name
        rust = structs[name]
        assert [n for n, _ in rust] == [n for n, _ in mirror._fields_], name
        for (n, rt), (_, ct) in zip(rust, mirror._fields_):
            want = to_ctype(rt, mirrors)
            assert C.sizeof(want) == C.sizeof(ct) and C.alignment(want) == C.alignment(ct), (name, n, rt)
            assert issubclass(want, C._Pointer) == issubclass(ct, C._Pointer), (name, n, rt)


def test_rust_extern_block_lists_every_entry_point():
    text = open(SRC).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', text, re.S).group(1)
    names = sorted(re.findall(r"pub fn (rt_\w+)\s*\(", block))
    assert names == declared("mi355rt.h")
    # argument counts agree with the header
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mi355rt.h")).read(), flags=re.S)
    for n in names:
        h_args = re.search(r"\b" + n + r"\s*\(([^)]*)\)", header).group(1).strip()
        r_args = re.search(r"pub fn " + n + r"\s*\(([^)]*)\)", block).group(1).strip()
        count = lambda a: 0 if a in ("", "void") else a.count(",") + 1
        assert count(h_args) == count(r_args), n
