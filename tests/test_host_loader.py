"""Host front end (raytracer_amd/csrc/host) against the reference's own unit-test
vectors: src/core/transform.rs:146-371, src/core/material.rs:148-225,
src/render/mod.rs:161-249, and the primitive counts of renders/renders.md."""
import json
import math
import os

import numpy as np
import pytest

from conftest import ROOT, SCENES, load_scene


def rot_y(deg):
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1.0]])


def trans(t):
    m = np.eye(4)
    m[:3, 3] = t
    return m


def scale(s):
    return np.diag([s[0], s[1], s[2], 1.0])


def test_it_constructs_matrices(libs):
    host, _ = libs
    m, inv, invt = host.transform_from_json("[]")
    assert np.array_equal(m, np.eye(4)) and np.array_equal(inv, np.eye(4))
    m, inv, invt = host.transform_from_json('[{"translate": [1, 2, 3]}]')
    assert np.array_equal(m, trans([1, 2, 3]))
    assert np.array_equal(inv, trans([-1, -2, -3]))
    assert np.array_equal(invt, trans([-1, -2, -3]).T)
    m, inv, _ = host.transform_from_json('[{"rotate": [[0, 1, 0], 50]}]')
    assert np.allclose(m, rot_y(50), atol=1e-15)
    assert np.allclose(inv, rot_y(-50), atol=1e-15)
    m, inv, _ = host.transform_from_json('[{"scale": [1, 2, 3]}]')
    assert np.array_equal(m, scale([1, 2, 3]))
    assert np.allclose(inv, scale([1, 0.5, 1 / 3]), atol=1e-16)


def test_it_constructs_complex_matrices(libs):
    """transform.rs:232-295: each entry LEFT-multiplies (applied after the previous)."""
    host, _ = libs
    m, inv, invt = host.transform_from_json('[{"rotate": [[0,1,0],50]},{"scale":[3,2,1]},{"translate":[5,2,3]}]')
    expect = trans([5, 2, 3]) @ scale([3, 2, 1]) @ rot_y(50)
    assert np.allclose(m, expect, atol=1e-15)
    assert np.allclose(inv @ m, np.eye(4), atol=1e-14)
    assert np.array_equal(invt, inv.T)
    m, _, _ = host.transform_from_json('[{"translate":[1,2,3]},{"translate":[-1,-2,-3]}]')
    assert np.array_equal(m, np.eye(4))
    m, _, _ = host.transform_from_json(
        '[{"rotate":[[0,1,0],50]},{"scale":[1,2,4]},{"translate":[1,2,3]},{"translate":[-1,-2,-3]},'
        '{"scale":[1,0.5,0.25]},{"rotate":[[0,1,0],-50]}]'
    )
    assert np.array_equal(m, np.eye(4))  # the reference asserts exact equality (transform.rs:258)


def test_rotate_axis_is_not_normalised(libs):
    """Quirk Q1: serde leaves Unit<Vector3> as given; Rodrigues on the raw axis
    (verified against renders/test.png in SURVEY.md 8c)."""
    host, _ = libs
    m, _, _ = host.transform_from_json('[{"rotate": [[1, 0, -1], -30]}]')
    a = math.radians(-30)
    c, s = math.cos(a), math.sin(a)
    ux, uy, uz = 1.0, 0.0, -1.0
    expect = np.array(
        [
            [ux * ux + (1 - ux * ux) * c, ux * uy * (1 - c) - uz * s, ux * uz * (1 - c) + uy * s],
            [ux * uy * (1 - c) + uz * s, uy * uy + (1 - uy * uy) * c, uy * uz * (1 - c) - ux * s],
            [ux * uz * (1 - c) - uy * s, uy * uz * (1 - c) + ux * s, uz * uz + (1 - uz * uz) * c],
        ]
    )
    assert np.allclose(m[:3, :3], expect, atol=1e-15)
    assert abs(np.linalg.det(m[:3, :3]) - 1.0) > 0.1  # a shear, not a rotation


def test_unknown_transform_and_fields_are_rejected(libs):
    host, _ = libs
    with pytest.raises(host.HostError):
        host.transform_from_json('[{"shear": [1, 2, 3]}]')
    with pytest.raises(host.HostError):
        host.LoadedScene.from_string('{"widht": 10}')
    with pytest.raises(host.HostError):
        host.LoadedScene.from_string('{"objects": [{"type": "sphere", "radios": 2}]}')
    with pytest.raises(host.HostError):
        host.LoadedScene.from_string('{"objects": [{"type": "torus"}]}')
    with pytest.raises(host.HostError):
        host.LoadedScene.from_file(os.path.join(SCENES, "does_not_exist.json"))
    with pytest.raises(host.HostError):
        host.LoadedScene.from_string('{"objects": [{"type": "mesh", "file": "models/nope.obj"}]}', SCENES)


def test_defaults_of_an_empty_scene(libs):
    """render/mod.rs:113-159 defaults; mod.rs:166-173 builds an empty scene."""
    host, _ = libs
    s = host.LoadedScene.from_string("{}")
    o = s.opts
    assert (o.width, o.height, o.max_depth, o.samples_per_pixel) == (100, 100, 3, 4)
    assert (o.max_reflected_rays, o.max_illumination_rays, o.max_occlusion_distance) == (32, 16, 1.0)
    d = s.desc.contents
    assert d.n_prims == 0 and d.n_lights == 0
    assert d.camera.fov_deg == 65.0 and list(d.camera.position) == [0.0, 0.0, 1.0]
    assert np.array_equal(np.array(d.camera.cam_to_world[:]).reshape(3, 3), np.eye(3))


def test_a_description_keeps_its_scene_alive(libs):
    """`LoadedScene.desc` points into the scene's arrays: taking it from a temporary must not leave it dangling."""
    import gc

    host, _ = libs
    d = host.LoadedScene.from_file(os.path.join(SCENES, "test.json")).desc
    gc.collect()
    junk = [host.LoadedScene.from_file(os.path.join(SCENES, "refraction.json")) for _ in range(4)]  # reuse freed memory, if any
    assert (d.contents.opts.width, d.contents.opts.height, d.contents.n_prims) == (800, 800, 9)
    del junk


def test_material_defaults_and_parameters(libs):
    """material.rs:148-225."""
    host, _ = libs
    s = host.LoadedScene.from_string(
        json.dumps(
            {
                "objects": [
                    {"type": "sphere"},
                    {"type": "sphere", "material": {"type": "phong"}},
                    {"type": "sphere", "material": {"type": "physical"}},
                    {
                        "type": "cube",
                        "material": {
                            "type": "physical", "side": "Both", "color": [1, 0.5, 0.25], "opacity": 0.5,
                            "emissive": [0.1, 0.2, 0.3], "emissive_intensity": 3, "roughness": 0.25, "metalness": 0.75,
                            "refractive_index": 1.5,
                        },
                    },
                    {
                        "type": "cube",
                        "material": {"type": "phong", "side": "Back", "specular": [1, 1, 1], "reflectivity": 0.5, "shininess": 12},
                    },
                ]
            }
        )
    )
    d = s.desc.contents
    mats = [d.materials[d.prim_material[i]] for i in range(d.n_prims)]
    for m in mats[:2]:  # default material == default phong
        assert (m.kind, m.side) == (0, 1)
        assert list(m.color) == [0, 0, 0] and list(m.emissive) == [0, 0, 0] and list(m.specular) == [0, 0, 0]
        assert (m.reflectivity, m.shininess, m.texture_id) == (0.0, 30.0, -1)
    m = mats[2]
    assert (m.kind, m.side, m.opacity, m.roughness, m.metalness, m.refractive_index) == (1, 1, 1.0, 0.5, 0.0, 1.0)
    m = mats[3]
    assert (m.kind, m.side) == (1, 0) and list(m.color) == [1, 0.5, 0.25] and list(m.emissive) == [0.1, 0.2, 0.3]
    assert (m.opacity, m.roughness, m.metalness, m.refractive_index) == (0.5, 0.25, 0.75, 1.5)
    m = mats[4]
    assert (m.kind, m.side, m.reflectivity, m.shininess) == (0, 2, 0.5, 12.0) and list(m.specular) == [1, 1, 1]


def test_flatten_order_children_before_self_and_parent_times_child(libs):
    """e.g. sphere.rs:49-69 + transform.rs:35."""
    host, _ = libs
    s = host.LoadedScene.from_string(
        json.dumps(
            {
                "objects": [
                    {
                        "type": "sphere", "transform": [{"translate": [1, 0, 0]}],
                        "children": [
                            {"type": "cube", "transform": [{"scale": [2, 2, 2]}]},
                            {"type": "group", "transform": [{"translate": [0, 5, 0]}], "children": [{"type": "plane"}]},
                        ],
                    },
                    {"type": "triangle", "vertices": [[0, 0, 0], [1, 0, 0], [0, 1, 0]]},
                ]
            }
        )
    )
    assert list(s.prim_types()) == [1, 2, 0, 3]  # cube, plane, sphere, triangle
    d = s.desc.contents
    w = s.world()
    cube = w[d.prim_transform[0]]
    assert np.array_equal(cube, (trans([1, 0, 0]) @ scale([2, 2, 2]))[:3])
    plane = w[d.prim_transform[1]]
    assert np.array_equal(plane, trans([1, 5, 0])[:3])
    tri_n = np.array(d.tri_nrm[0:9]).reshape(3, 3)
    assert np.array_equal(tri_n, np.tile([0.0, 0.0, 1.0], (3, 1)))  # Triangle::compute_normal


@pytest.mark.parametrize(
    "name,prims",
    [("scene", 1524), ("mesh", 2269), ("refraction", 6), ("physical", 50), ("mesh_test", 41372), ("occlusion", 9),
     ("emissive", 15), ("test", 9), ("cornell_box", 32)],
)
def test_primitive_counts_match_published(libs, name, prims):
    """renders/renders.md:9-51 (cornell_box: 16 quads fan-triangulated)."""
    s = load_scene(name)
    assert s.n_prims == prims


def test_obj_values_are_parsed_as_f32(libs):
    """tobj parses floats as f32; mesh.rs:83-87 widens them."""
    s = load_scene("cornell_box")
    d = s.desc.contents
    pos = np.array(d.tri_pos[0 : 9 * d.n_tris])
    assert np.array_equal(pos, pos.astype(np.float32).astype(np.float64))
    assert np.any(pos == np.float64(np.float32(548.8)))
    assert not np.any(pos == 548.8)


def test_camera_look_at(libs):
    host, _ = libs
    m = host.camera_to_world([0, 0, 1], [0, 0, 0], [0, 1, 0])
    assert np.array_equal(m, np.eye(3))
    m = host.camera_to_world([2, 5, 15], [-1, 0, 0], [0, 1, 0])
    assert np.allclose(m.T @ m, np.eye(3), atol=1e-15)
    z = np.array([3, 5, 15.0])
    assert np.allclose(m[:, 2], z / np.linalg.norm(z), atol=1e-15)


def test_png_writer_roundtrip(libs, tmp_path):
    from PIL import Image

    host, _ = libs
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53, 4), dtype=np.uint8)
    path = tmp_path / "out.png"
    host.write_png(str(path), img)
    back = np.array(Image.open(path))
    assert back.shape == img.shape and np.array_equal(back, img)


def test_texture_decoders_png(libs, tmp_path):
    """rth_image_load = `image::open(path).to_rgb8()` (src/core/texture.rs:38), PNG side: the reference's checker.png
    (mode L with a gAMA chunk, which is NOT applied) texel for texel, every PNG flavour a scene might use, and our own
    writer's output."""
    import zlib

    from PIL import Image

    host, _ = libs
    chk = host.image_load(os.path.join(ROOT, "scenes", "textures", "checker.png"))
    assert chk.shape == (640, 640, 3) and zlib.crc32(chk.tobytes()) == 3581272205
    for (y, x), v in {(0, 0): 0, (10, 100): 255, (320, 320): 2, (639, 639): 0}.items():
        assert chk[y, x].tolist() == [v, v, v]
    assert sorted(np.unique(chk).tolist()) == [0, 2, 5, 31, 254, 255]  # grey replicated, no gamma applied
    assert np.array_equal(chk, np.array(Image.open(os.path.join(ROOT, "scenes", "textures", "checker.png")).convert("RGB")))
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    variants = {
        "rgb": Image.fromarray(rgb), "rgba": Image.fromarray(np.dstack([rgb, rng.integers(0, 256, size=(37, 53), dtype=np.uint8)])),
        "grey": Image.fromarray(rgb[..., 0]), "la": Image.fromarray(rgb[..., :2].copy(), "LA"),
        "palette": Image.fromarray(rgb).quantize(17), "bilevel": Image.fromarray(rgb[..., 0] > 127),
    }
    for name, im in variants.items():
        p = str(tmp_path / f"{name}.png")
        im.save(p, optimize=(name != "rgb"))
        assert np.array_equal(host.image_load(p), np.array(Image.open(p).convert("RGB"))), name
    wide = rng.integers(0, 65536, size=(9, 11), dtype=np.uint16)
    Image.fromarray(wide).save(str(tmp_path / "g16.png"))  # 16-bit grey: narrowed as image's u16 -> u8 conversion does
    assert np.array_equal(host.image_load(str(tmp_path / "g16.png"))[..., 0], ((wide.astype(np.uint32) + 128) // 257).astype(np.uint8))
    ours = str(tmp_path / "ours.png")
    host.write_png(ours, np.dstack([rgb, np.full((37, 53), 255, np.uint8)]))
    assert np.array_equal(host.image_load(ours), rgb)
    with pytest.raises(host.HostError):
        host.image_load(str(tmp_path / "missing.png"))
    open(tmp_path / "junk.png", "wb").write(b"\x89PNG\r\n\x1a\n" + b"\0" * 40)
    with pytest.raises(host.HostError):
        host.image_load(str(tmp_path / "junk.png"))


def test_texture_decoders_jpeg(libs, tmp_path):
    """JPEG side: the reference's test.jpg (baseline, 4:4:4, restart intervals, Adobe marker).  The reference decodes with
    jpeg-decoder 0.2.4 (not under /root/reference: PARITY UNPINNED at the texel level); our decoder follows its published
    arithmetic (stb_image's fixed-point IDCT and BT.601 conversion) and is held here to an independent decoder (libjpeg
    through PIL, islow IDCT): equal on > 99 % of the samples, never more than 3 levels apart, no bias.  The renders pin
    it end to end (tests/test_reference_renders.py: quiet pixels of the textured primitives within one level)."""
    import zlib

    from PIL import Image

    host, _ = libs
    path = os.path.join(ROOT, "scenes", "textures", "test.jpg")
    img = host.image_load(path)
    assert img.shape == (1024, 1024, 3) and zlib.crc32(img.tobytes()) == 2935947561
    for (y, x), v in {(0, 0): [255, 255, 255], (100, 200): [135, 28, 160], (700, 13): [207, 57, 69], (300, 900): [57, 65, 148]}.items():
        assert img[y, x].tolist() == v
    d = img.astype(np.int32) - np.array(Image.open(path).convert("RGB")).astype(np.int32)
    assert np.abs(d).max() <= 3 and (d == 0).mean() > 0.99 and abs(d.mean()) < 1e-3
    # flavours written here: greyscale, 4:2:2 and 4:2:0 chroma subsampling (smooth content: the upsampling filters differ
    # from libjpeg's at edges), restart intervals off; progressive is an error, not a wrong image
    yy, xx = np.mgrid[0:96, 0:120]
    smooth = np.dstack([(xx * 2) % 256, (yy * 2 + xx) % 256, 255 - (xx + yy) % 256]).astype(np.uint8)
    smooth = np.array(Image.fromarray(smooth).resize((120, 96), Image.BILINEAR))
    for name, kw in {"grey": dict(), "s444": dict(subsampling=0), "s422": dict(subsampling=1), "s420": dict(subsampling=2)}.items():
        p = str(tmp_path / f"{name}.jpg")
        (Image.fromarray(smooth[..., 0]) if name == "grey" else Image.fromarray(smooth)).save(p, quality=92, **kw)
        got = host.image_load(p).astype(np.int32)
        want = np.array(Image.open(p).convert("RGB")).astype(np.int32)
        assert got.shape == want.shape
        tol = 3 if name in ("grey", "s444") else 40
        assert np.abs(got - want).max() <= tol and np.abs(got - want).mean() < (0.1 if tol == 3 else 2.0), name
    Image.fromarray(smooth).save(str(tmp_path / "prog.jpg"), progressive=True)
    with pytest.raises(host.HostError, match="progressive"):
        host.image_load(str(tmp_path / "prog.jpg"))


def test_texture_decoders_refuse_hostile_files(libs, tmp_path):
    """Texture files are untrusted input: a PNG whose few KB of IDAT inflate far beyond the image IHDR describes, a
    JPEG cut off in its entropy data, and a JPEG header that asks for gigabytes on a few bytes are errors, promptly."""
    import struct
    import time
    import zlib

    from PIL import Image

    host, _ = libs

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data))

    bomb = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * (64 << 20), 9)) + chunk(b"IEND", b"")
    assert len(bomb) < 100000
    open(tmp_path / "bomb.png", "wb").write(bomb)
    t0 = time.perf_counter()
    with pytest.raises(host.HostError, match="inflates beyond"):
        host.image_load(str(tmp_path / "bomb.png"))
    assert time.perf_counter() - t0 < 1.0
    rng = np.random.default_rng(5)
    Image.fromarray(rng.integers(0, 256, size=(256, 256, 3), dtype=np.uint8)).save(str(tmp_path / "full.jpg"), quality=90)
    data = open(tmp_path / "full.jpg", "rb").read()
    open(tmp_path / "cut.jpg", "wb").write(data[: len(data) // 3] + b"\xff\xd9")
    with pytest.raises(host.HostError, match="truncated JPEG"):
        host.image_load(str(tmp_path / "cut.jpg"))
    sof = data.index(b"\xff\xc0")
    huge = data[: sof + 5] + b"\xff\xff\xff\xff" + data[sof + 9:]  # height = width = 65535
    open(tmp_path / "huge.jpg", "wb").write(huge)
    t0 = time.perf_counter()
    with pytest.raises(host.HostError, match="does not fit the file"):
        host.image_load(str(tmp_path / "huge.jpg"))
    assert time.perf_counter() - t0 < 1.0


def test_scene_textures_come_from_the_in_tree_decoder(libs):
    """The loader hands rt_scene_create the decoder's texels (no pre-decoded sidecar files any more)."""
    host, _ = libs
    s = load_scene("test")
    d = s.desc.contents
    assert d.n_textures == 2
    sizes = sorted((int(d.textures[i].width), int(d.textures[i].height)) for i in range(2))
    assert sizes == [(640, 640), (1024, 1024)]
    for i in range(2):
        t = d.textures[i]
        name = "checker.png" if t.width == 640 else "test.jpg"
        got = np.ctypeslib.as_array(t.rgb8, shape=(t.height, t.width, 3))
        assert np.array_equal(got, host.image_load(os.path.join(ROOT, "scenes", "textures", name)))
    assert not os.path.exists(os.path.join(ROOT, "scenes", "textures", "test.jpg.ppm"))
