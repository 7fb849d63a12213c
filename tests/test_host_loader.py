"""Host front end (raytracer_amd/csrc/host) against the reference's own unit-test
vectors: src/core/transform.rs:146-371, src/core/material.rs:148-225,
src/render/mod.rs:161-249, and the primitive counts of renders/renders.md."""
import json
import math
import os

import numpy as np
import pytest

from conftest import SCENES, load_scene


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
