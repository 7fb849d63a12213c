"""Parity of the HIP path (through the C ABI, ctypes -> libmi355rt.so) with the
CPU oracle.  Bars (BASELINE.json north_star): primary-hit primitive ids
bit-exact, hit distances / normals within 1e-5 relative, final radiance
RMSE <= 1e-3 at matched spp with the shared counter-based RNG."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCENES, load_scene

pytestmark = pytest.mark.gpu

RMSE_TOL = 1e-3  # north_star: "Final radiance must reach RMSE <= 1e-3"
REL_TOL = 1e-5   # north_star: "Hit distances and normals must agree within 1e-5 relative"
NO_HIT = 0xFFFFFFFF


@pytest.fixture(scope="module")
def rt():
    from oracle import binding as ob
    from raytracer_amd import _abi, device

    device.lib()  # fails loudly if the HIP library is missing
    return device, ob, _abi


def radiance_rmse(acc_a, acc_b, spp):
    return float(np.sqrt(np.mean(((acc_a - acc_b) / spp) ** 2)))


def check_primary(gpu, cpu, ob):
    ids_g, t_g, n_g = gpu.query_primary()
    ids_c, t_c, n_c = cpu.query_primary(ob.MODE_CANONICAL)
    assert np.array_equal(ids_g, ids_c), f"{int((ids_g != ids_c).sum())} primary ids differ"
    hit = ids_c != NO_HIT
    if hit.any():
        assert np.all(np.abs(t_g[hit] - t_c[hit]) <= REL_TOL * np.abs(t_c[hit]))
        assert np.all(np.abs(n_g[hit] - n_c[hit]) <= REL_TOL)
    return ids_c


SMALL = ["test", "scene", "mesh", "refraction", "physical", "cornell_box", "occlusion", "emissive"]


@pytest.mark.parametrize("name", SMALL)
def test_small_render_matches_oracle_and_fixture(rt, name):
    device, ob, _ = rt
    s = load_scene(name, size=64)
    gpu, cpu = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
    fix = np.load(os.path.join(GOLDEN, f"oracle_{name}.npz"))
    ids = check_primary(gpu, cpu, ob)
    assert np.array_equal(ids, fix["ids"])
    acc_g, rgba_g, n_g, _ = gpu.render()
    spp = int(fix["spp"])
    n_c = int(fix["ray_count"])
    assert abs(n_g - n_c) <= max(2, n_c // 100000), (n_g, n_c)
    assert radiance_rmse(acc_g, fix["accum"], spp) <= RMSE_TOL
    same = (rgba_g == fix["rgba"]).all(axis=2).mean()
    assert same >= (0.97 if name == "cornell_box" else 0.999), same  # cornell: coplanar light/ceiling (quirk Q17)


@pytest.mark.parametrize("name,size,spp", [("scene", 160, 4), ("physical", 128, 4), ("cornell_box", 96, 16), ("mesh_test", 96, 2)])
def test_medium_render_matches_live_oracle(rt, name, size, spp):
    """cornell_box at 16 spp is BASELINE config 2; mesh_test has 41k primitives."""
    device, ob, _ = rt
    s = load_scene(name, size=size, spp=spp)
    gpu, cpu = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
    check_primary(gpu, cpu, ob)
    acc_g, _, n_g, _ = gpu.render(want_rgba=False)
    acc_c, _, n_c = cpu.render(mode=ob.MODE_CANONICAL, threads=16, want_rgba=False)
    assert abs(n_g - n_c) <= max(2, n_c // 100000), (n_g, n_c)
    assert radiance_rmse(acc_g, acc_c, spp) <= RMSE_TOL


def _random_rays(rng, n, lo, hi):
    o = rng.uniform(lo, hi, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


@pytest.mark.parametrize("name,lo,hi", [("scene", -25, 25), ("test", -4, 6), ("cornell_box", 0, 555), ("mesh", -8, 12)])
def test_probe_rays_closest_and_shadow(rt, name, lo, hi):
    """rt_trace_rays vs the oracle on rays no camera produces: random interior
    origins, axis-parallel directions (zero components exercise the slab tests'
    inf/NaN paths, quirk Q14)."""
    device, ob, _ = rt
    s = load_scene(name, size=8)
    gpu, cpu = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
    rng = np.random.default_rng(7)
    o, d = _random_rays(rng, 20000, lo, hi)
    d[:600] = np.repeat(np.array([[1, 0, 0], [0, -1, 0], [0, 0, 1], [0, 1, 0], [-1, 0, 0], [0, 0, -1]], float), 100, axis=0)
    d[600:700, 1] = 0.0
    ids_g, t_g = gpu.trace_rays(o, d)
    ids_c, t_c = cpu.trace_rays(o, d, mode=ob.MODE_CANONICAL)
    assert np.array_equal(ids_g, ids_c)
    hit = ids_c != NO_HIT
    assert np.all(np.abs(t_g[hit] - t_c[hit]) <= REL_TOL * np.abs(t_c[hit]))
    ids_b, _ = cpu.trace_rays(o[:3000], d[:3000], mode=ob.MODE_BRUTE)
    assert np.array_equal(ids_g[:3000], ids_b)
    # shadow rays: limits straddling the closest hit by +-BIAS
    tmax = np.where(hit, t_c, 10.0) + rng.choice([-1e-10, 0.0, 1e-10, 5.0, -0.5], size=len(t_c))
    tmax = np.abs(tmax)
    occ_g, _ = gpu.trace_rays(o, d, tmax, shadow=True)
    occ_c, _ = cpu.trace_rays(o, d, tmax, shadow=True, mode=ob.MODE_CANONICAL)
    assert np.array_equal(occ_g, occ_c)


with open(os.path.join(GOLDEN, "ray_counts.json")) as _f:
    RAY_COUNTS = json.load(_f)


@pytest.fixture(scope="module")
def full_scene(rt):
    device, ob, _ = rt
    s = load_scene("scene")  # BASELINE config 3: 800x800, 4 spp, 1,524 primitives
    return s, device.DeviceScene(s.desc)


def test_full_size_ray_count(rt, full_scene):
    """800x800x4spp scene.json: +-0.1 % of the published 147,115,715 rays and
    (libm aside) equal to the oracle's committed full-size count."""
    _, gpu = full_scene
    _, _, n, st = gpu.render(want_rgba=False, want_stats=True)
    c = RAY_COUNTS["scene"]
    assert abs(n - c["published"]) / c["published"] < 1e-3
    assert abs(n - c["oracle_canonical"]) <= 200
    assert st.ray_count == n == st.closest_rays + st.shadow_rays and st.camera_rays == 800 * 800 * 4


PUBLISHED_RAYS = {k: v["published"] for k, v in RAY_COUNTS.items()}
PUBLISHED_RAYS["mesh_test"] = 41492072  # renders/renders.md:33


# The reference's committed renders of these scenes (all eight PNGs went through OIDN, renders/renders.md:3; measured in
# tests/refstats.py::filter_evidence -- NOT raw tracer output, whatever `skip_denoise_pass` says in the scene files today).
REFERENCE_RENDERS = ["test", "mesh", "refraction", "physical", "mesh_test"]


@pytest.mark.parametrize("name", sorted(PUBLISHED_RAYS))
def test_full_size_against_the_references_own_outputs(rt, name):
    """Every scene the reference publishes, at its full 800x800x4spp (BASELINE configs 3 and 4 included): ray count
    within 0.1 % of renders/renders.md (a 3-run average of a random process) and, for the five renders of scenes
    without area-wide denoising artefacts, the HIP path against the reference's PNG with bounds calibrated on 12 of
    its own seeds (tests/refstats.py, tests/test_reference_renders.py::check_against_reference): quiet pixels equal to
    the 8-bit level, block means inside [raw, converged] region by region."""
    from PIL import Image

    import refstats
    from test_reference_renders import check_against_reference

    device, _, abi = rt
    s = load_scene(name)
    gpu = device.DeviceScene(s.desc)
    _, _, n, _ = gpu.render(want_rgba=False)
    pub = PUBLISHED_RAYS[name]
    assert abs(n - pub) / pub < 1e-3, (n, pub)
    if name in RAY_COUNTS:
        assert abs(n - RAY_COUNTS[name]["oracle_canonical"]) <= max(200, pub // 1000000)
    if name in REFERENCE_RENDERS:
        ref = np.array(Image.open(os.path.join(GOLDEN, "renders", f"{name}.png")).convert("RGB"))
        ids = gpu.query_primary()[0]
        frames, conv = refstats.collect(lambda seed: gpu.render(abi.RenderParams(seed=seed))[:2], 12, int(s.opts.samples_per_pixel))
        regions = refstats.region_stats(ref, frames, conv, ids)
        quiet = refstats.quiet_stats(ref, frames, ids)
        ev = refstats.filter_evidence(ref, frames, ids)
        print(check_against_reference(name, regions, quiet, ev, few_seeds=False))
        if name == "test":
            assert ev["far_nonzero"] >= 100 and ev["far_nonzero_ours"] == 0, ev


def test_full_size_properties(rt, full_scene):
    """Size-independent properties at the BASELINE size: idempotence, independence
    of the pool budget (batch partition), tile shards and windows compose exactly,
    sample ranges compose up to f64 re-association."""
    device, _, abi = rt
    s, gpu = full_scene
    a1, img1, n1, _ = gpu.render()
    a2, img2, n2, _ = gpu.render()
    assert n1 == n2 and np.array_equal(a1, a2) and np.array_equal(img1, img2)
    small = device.DeviceScene(s.desc, pool_bytes=1 << 30)
    a3, _, n3, st3 = small.render(want_rgba=False, want_stats=True)
    assert st3.batches > 4 and n3 == n1 and np.array_equal(a3, a1)
    acc, total = np.zeros_like(a1), 0
    for rank in range(3):
        a, _, n, _ = gpu.render(abi.RenderParams(shard_index=rank, shard_count=3, tile_size=32), want_rgba=False)
        assert ((acc != 0) & (a != 0)).sum() == 0  # disjoint pixels
        acc += a
        total += n
    assert total == n1 and np.array_equal(acc, a1)
    w, _, _, _ = gpu.render(abi.RenderParams(x0=123, y0=77, x1=517, y1=391), want_rgba=False)
    assert np.array_equal(w[77:391, 123:517], a1[77:391, 123:517])
    w[77:391, 123:517] = 0
    assert not w.any()
    s0, _, c0, _ = gpu.render(abi.RenderParams(s0=0, s1=1), want_rgba=False)
    s1, _, c1, _ = gpu.render(abi.RenderParams(s0=1, s1=4), want_rgba=False)
    assert c0 + c1 == n1 and np.allclose(s0 + s1, a1, rtol=0, atol=1e-12)
    b, _, _, _ = gpu.render(abi.RenderParams(seed=1), want_rgba=False)
    assert not np.array_equal(b, a1)  # the seed reaches the sample stream


def _scene_json(**kw):
    base = {"width": 24, "height": 16, "camera": {"position": [0, 1, 6]},
            "lights": [{"type": "ambient", "color": [0.2, 0.2, 0.2]}, {"type": "point", "transform": [{"translate": [2, 5, 3]}]}],
            "objects": [{"type": "sphere", "material": {"type": "phong", "color": [1, 0.5, 0.2]}}]}
    base.update(kw)
    return json.dumps(base)


EDGE_CASES = {
    "empty_scene": _scene_json(objects=[]),
    "no_lights": _scene_json(lights=[]),
    "planes_only": _scene_json(objects=[{"type": "plane", "transform": [{"translate": [0, -1, 0]}],
                                         "material": {"type": "physical", "color": [1, 1, 1], "roughness": 0.3}}]),
    "one_pixel": _scene_json(width=1, height=1),
    "one_sample": _scene_json(samples_per_pixel=1),
    "depth_one": _scene_json(max_depth=1),
    "depth_zero": _scene_json(max_depth=0),
    "depth_six": _scene_json(max_depth=6, max_reflected_rays=4, max_illumination_rays=4,
                             objects=[{"type": "sphere", "material": {"type": "physical", "color": [1, 1, 1], "opacity": 0.4,
                                                                      "refractive_index": 1.3, "roughness": 0.2}},
                                      {"type": "plane", "transform": [{"translate": [0, -1, 0]}],
                                       "material": {"type": "phong", "color": [1, 1, 1], "reflectivity": 1.0}}]),
    "no_reflected_or_gi_rays": _scene_json(max_reflected_rays=0, max_illumination_rays=0,
                                           objects=[{"type": "cube", "material": {"type": "physical", "color": [1, 0, 0]}}]),
    "wide_and_tall": _scene_json(width=40, height=9),
    "tall": _scene_json(width=9, height=40),
    "both_and_back_sides": _scene_json(objects=[
        {"type": "cube", "size": 8, "material": {"type": "phong", "side": "Back", "color": [0.5, 0.5, 0.5]}},
        {"type": "triangle", "vertices": [[-1, 0, 0], [1, 0, 0], [0, 1.5, 0]], "material": {"type": "phong", "side": "Both", "color": [0, 1, 0]}},
        {"type": "sphere", "radius": 0.5, "transform": [{"translate": [0, 0, 2]}],
         "material": {"type": "physical", "side": "Both", "color": [1, 1, 1], "opacity": 0.3, "refractive_index": 1.5}}]),
    "degenerate_triangle": _scene_json(objects=[{"type": "triangle", "vertices": [[0, 0, 0], [0, 0, 0], [0, 0, 0]]},
                                                {"type": "sphere"}]),
}


_TEX = {"type": "phong", "color": [1, 1, 1], "texture": "textures/checker.png"}
_PTEX = {"type": "physical", "color": [0.9, 0.8, 0.7], "roughness": 0.4, "metalness": 0.3, "texture": "textures/test.jpg"}
EDGE_CASES.update({
    # uv of each primitive kind + nearest-texel lookup (texture.rs:46-69), both materials
    "textured_all_kinds": _scene_json(width=40, height=30, camera={"position": [0, 2, 7], "target": [0, 0.5, 0]}, objects=[
        {"type": "plane", "transform": [{"translate": [0, -1, 0]}], "material": _TEX},
        {"type": "sphere", "transform": [{"translate": [-2, 0, 0]}], "material": _PTEX},
        {"type": "cube", "size": 1.5, "transform": [{"rotate": [[0, 1, 0], 30]}, {"translate": [0.2, 0, 0]}], "material": _TEX},
        {"type": "triangle", "vertex_data": [{"position": [1.5, -1, 0], "normal": [0, 0, 1], "texcoords": [0, 0]},
                                             {"position": [3.5, -1, 0], "normal": [0, 0, 1], "texcoords": [3, 0]},
                                             {"position": [2.5, 1.5, -1], "normal": [0, 0.5, 1], "texcoords": [1.5, -2]}],
         "material": dict(_PTEX, side="Both")}]),
    # groups inside groups, un-normalised rotation axes (quirk Q1), non-uniform scale, a mesh under all of it
    "nested_sheared_groups": _scene_json(width=36, height=36, camera={"position": [0, 3, 9], "target": [0, 0, 0], "fov": 50}, objects=[
        {"type": "group", "transform": [{"rotate": [[1, 0, -1], -30]}, {"translate": [0, 0.5, 0]}], "children": [
            {"type": "group", "transform": [{"scale": [1.5, 0.5, 1]}, {"rotate": [[-0.25, 0.9, 0], 40]}], "children": [
                {"type": "cube", "size": 2, "material": {"type": "phong", "color": [0.2, 0.6, 1], "reflectivity": 0.3}},
                {"type": "sphere", "radius": 0.7, "transform": [{"translate": [0, 2, 0]}],
                 "material": {"type": "physical", "color": [1, 0.3, 0.3], "roughness": 0.1, "metalness": 1}}]},
            {"type": "mesh", "file": "models/cornell_box/short_block.obj", "transform": [{"scale": [0.01, 0.01, 0.01]}, {"translate": [-3, -1, 0]}],
             "material": {"type": "phong", "color": [0.8, 0.8, 0.2]}}]},
        {"type": "plane", "transform": [{"translate": [0, -2, 0]}], "material": {"type": "phong", "color": [0.5, 0.5, 0.5]}}]),
    # Cornell-box scale (coordinates ~550, BIAS 1e-10 far below f32 resolution) with a tiny emissive sphere
    "huge_and_tiny": _scene_json(width=32, height=32, max_occlusion_distance=100.0,
                                 camera={"position": [278, 273, -800], "target": [278, 273, 0], "fov": 38.2},
                                 lights=[{"type": "point", "transform": [{"translate": [278, 540, 279]}], "intensity": 100000}],
                                 objects=[{"type": "mesh", "file": "models/cornell_box/floor.obj", "material": {"type": "phong", "color": [0.7, 0.7, 0.7]}},
                                          {"type": "mesh", "file": "models/cornell_box/back_wall.obj", "material": {"type": "phong", "color": [0.7, 0.2, 0.2]}},
                                          {"type": "sphere", "radius": 0.001, "transform": [{"scale": [50000, 50000, 50000]}, {"translate": [278, 100, 279]}],
                                           "material": {"type": "physical", "color": [1, 1, 1], "emissive": [0.5, 0.5, 0.1], "roughness": 0.9}}]),
    # one reflected ray, one GI ray, two refractive shells inside each other (ior carried by the ray, quirk Q11 / Q20)
    "single_ray_budgets_nested_glass": _scene_json(width=30, height=30, max_depth=5, max_reflected_rays=1, max_illumination_rays=1, objects=[
        {"type": "sphere", "radius": 1.5, "material": {"type": "physical", "color": [1, 1, 1], "opacity": 0.1, "refractive_index": 1.5, "roughness": 0.0}},
        {"type": "sphere", "radius": 0.7, "material": {"type": "physical", "side": "Both", "color": [0.2, 1, 0.4], "opacity": 0.5, "refractive_index": 1.1}},
        {"type": "plane", "normal": [0, 0, 1], "transform": [{"translate": [0, 0, -4]}], "material": dict(_TEX, emissive=[0.1, 0.1, 0.1])}]),
    # light inside a Back-sided cube (inner faces shaded), ambient only outside contribution; no point light reaches the outer sphere
    "light_inside_back_cube": _scene_json(width=28, height=20, camera={"position": [0, 0, 0.5], "target": [0, 0, -1], "fov": 100},
                                          lights=[{"type": "ambient", "color": [0.05, 0.05, 0.05]}, {"type": "point", "transform": [{"translate": [0, 1, 0]}], "intensity": 3}],
                                          objects=[{"type": "cube", "size": 6, "material": {"type": "phong", "side": "Back", "color": [0.6, 0.6, 0.9], "specular": [1, 1, 1], "shininess": 8}},
                                                   {"type": "sphere", "radius": 0.5, "transform": [{"translate": [1, -1, -2]}],
                                                    "material": {"type": "phong", "color": [1, 0.4, 0.1], "reflectivity": 1.0}},
                                                   {"type": "sphere", "radius": 2, "transform": [{"translate": [0, 0, -12]}], "material": {"type": "phong", "color": [1, 1, 1]}}]),
})


@pytest.mark.parametrize("case", sorted(EDGE_CASES))
def test_edge_cases(rt, case):
    device, ob, _ = rt
    from raytracer_amd import host

    s = host.LoadedScene.from_string(EDGE_CASES[case], asset_base=SCENES)
    gpu, cpu = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
    check_primary(gpu, cpu, ob)
    acc_g, rgba_g, n_g, _ = gpu.render()
    acc_c, rgba_c, n_c = cpu.render(mode=ob.MODE_CANONICAL, threads=2)
    assert n_g == n_c
    assert radiance_rmse(acc_g, acc_c, s.opts.samples_per_pixel) <= 1e-9
    assert (rgba_g == rgba_c).all(axis=2).mean() >= 0.99
    if case == "empty_scene":
        assert n_g == 24 * 16 * 4 and not acc_g.any()
    if case == "depth_zero":
        assert n_g == 0 and not acc_g.any()


@pytest.mark.parametrize("seed", range(24))
def test_random_scenes_match_oracle(rt, seed):
    """Differential test on seeded random scenes (tests/random_scenes.py: every primitive kind, both materials, all
    sides, textures, nested groups, sheared rotations, odd ray budgets): primary ids and ray counts exact, radiance to
    1e-9.  tools/fuzz_parity.py runs the same check over any seed range (600 seeds clean in round 1)."""
    device, ob, _ = rt
    from raytracer_amd import host
    from random_scenes import random_scene

    s = host.LoadedScene.from_string(random_scene(seed), asset_base=SCENES)
    gpu, cpu = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
    check_primary(gpu, cpu, ob)
    acc_g, _, n_g, _ = gpu.render(want_rgba=False)
    acc_c, _, n_c = cpu.render(mode=ob.MODE_CANONICAL, threads=4, want_rgba=False)
    assert n_g == n_c
    assert radiance_rmse(acc_g, acc_c, s.opts.samples_per_pixel) <= 1e-9


def test_error_codes_instead_of_panics(rt):
    import ctypes as C

    device, _, abi = rt
    from raytracer_amd import host

    lib = device.lib()
    h = C.c_void_p()
    assert lib.rt_scene_create(None, C.byref(h)) == abi.RT_ERR_INVALID
    s = host.LoadedScene.from_string(_scene_json(max_depth=7))
    with pytest.raises(device.DeviceError, match="max_depth"):
        device.DeviceScene(s.desc)
    s = host.LoadedScene.from_string(_scene_json())
    gpu = device.DeviceScene(s.desc)
    for bad in (abi.RenderParams(x0=0, y0=0, x1=25, y1=16), abi.RenderParams(s0=2, s1=9), abi.RenderParams(shard_index=3, shard_count=2)):
        with pytest.raises(device.DeviceError):
            gpu.render(bad)
    assert b"" != lib.rt_last_error()
    # a 1 MiB pool forces many tiny batches and overflow retries; results must not change
    a, _, n, _ = gpu.render(want_rgba=False)
    tiny = device.DeviceScene(s.desc, pool_bytes=1 << 20)
    b, _, m, st = tiny.render(want_rgba=False, want_stats=True)
    assert n == m and np.array_equal(a, b) and st.batches > 1


def test_progressive_passes_end_in_the_one_shot_frame(rt):
    """rt_render_progressive (raytrace_to_buffer, raytracing_scene.rs:778-875): every pass adds pixels, none is touched
    twice, the callback sees the buffer grow, and the last pass equals rt_render's RGBA8 as to_argb_u32 words."""
    device, _, abi = rt
    s = load_scene("test", size=96)
    gpu = device.DeviceScene(s.desc)
    acc1, rgba, n1, _ = gpu.render(abi.RenderParams(seed=5))
    want = (np.uint32(255) << 24) | (rgba[..., 0].astype(np.uint32) << 16) | (rgba[..., 1].astype(np.uint32) << 8) | rgba[..., 2].astype(np.uint32)
    seen = []

    def on_pass(argb, done, total, rays):
        seen.append((done, total, rays, int((argb != 0).sum()), argb.copy()))
        return False

    argb, acc, n = gpu.render_progressive(5, abi.RenderParams(seed=5), on_pass, want_accum=True)
    assert n == n1 and np.array_equal(acc, acc1) and np.array_equal(argb, want)
    assert [x[0] for x in seen] == [1, 2, 3, 4, 5] and all(x[1] == 5 for x in seen)
    filled = [x[3] for x in seen]
    assert filled == sorted(filled) and filled[0] > 0 and filled[-1] == 96 * 96 and filled[0] < 96 * 96 // 3
    for a, b in zip(seen, seen[1:]):  # finished pixels never change
        m = a[4] != 0
        assert np.array_equal(a[4][m], b[4][m]) and a[2] < b[2]
    # one pass = the plain frame; a stop request ends the render early
    one, _, n_one = gpu.render_progressive(1, abi.RenderParams(seed=5))
    assert n_one == n1 and np.array_equal(one, want)
    part, _, n_part = gpu.render_progressive(5, abi.RenderParams(seed=5), lambda argb, done, total, rays: done == 2)
    assert 0 < n_part < n1 and (part == 0).any() and np.array_equal(part[part != 0], want[part != 0])


def test_raytrace_cli_and_render_scenes_report(rt, tmp_path):
    """The `raytrace` binary (src/main.rs:9-78) and tools/render_scenes.py (src/render_scenes.rs:43-77): the PNGs they write
    hold the pixels rt_render returns, and their report lines have the reference's form."""
    import re
    import subprocess
    import sys

    from PIL import Image

    from conftest import ROOT

    device, _, abi = rt
    s = load_scene("test")
    gpu = device.DeviceScene(s.desc)
    exe = os.path.join(ROOT, "raytracer_amd", "raytrace")
    _, want, n, _ = gpu.render(abi.RenderParams(seed=0))
    for extra in ([], ["--passes", "4"], ["--passes", "3", "--no-progress"]):
        out = str(tmp_path / "cli.png")
        r = subprocess.run([exe, os.path.join(ROOT, "scenes", "test.json"), "-o", out] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert np.array_equal(np.array(Image.open(out).convert("RGBA")), want)
        assert re.search(r"Took [0-9.]+s to load assets\.", r.stdout) and re.search(r"pre-process scene and construct bounding boxes for 9 primitives", r.stdout)
        assert re.search(r"Took [0-9.]+s to render the scene\.", r.stdout) and f"{n} rays" in r.stdout and "Output written to" in r.stdout
        assert ("pass 4/4: 640000/640000 pixels" in r.stdout) == (extra == ["--passes", "4"])
    r = subprocess.run([exe, os.path.join(ROOT, "scenes", "nope.json"), "-o", str(tmp_path / "x.png")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "failed to parse scene" in r.stderr
    assert subprocess.run([exe], capture_output=True, text=True, timeout=60).returncode == 2
    # render-scenes: report lines of src/render_scenes.rs:55-69 and the last iteration's frame (seed = iteration index)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "render_scenes.py"), str(tmp_path), "--scenes", "test", "--iterations", "2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert re.search(r"Raytracing .*scenes/test\.json \(9 primitives\)\.\.\.", r.stdout)
    assert len(re.findall(r"Iteration \d: rendered in [0-9.]+ms \(\d+ rays\)", r.stdout)) == 2
    assert re.search(r"Final render: 800 x 800 pixels, 4 spp, 9 primitives, \d+ rays, [0-9.]+ms", r.stdout)
    _, want1, _, _ = gpu.render(abi.RenderParams(seed=1))
    assert np.array_equal(np.array(Image.open(str(tmp_path / "test.png")).convert("RGBA")), want1)


def test_cornell_box_full_size_against_live_oracle(rt):
    """BASELINE config 2 at its full size: cornell_box.json 800x800 at 16 spp (10.24 M camera rays) on the HIP path against
    the live oracle.  The box's light and ceiling are coplanar (scenes/models/cornell_box/light.obj, ceiling.obj, quirk
    Q17): which of the two a ray through that rectangle hits is decided in the last place of t, so that region is
    reported on its own."""
    device, ob, abi = rt
    s = load_scene("cornell_box", spp=16)
    gpu, cpu = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
    ids = check_primary(gpu, cpu, ob)
    acc_g, _, n_g, _ = gpu.render(want_rgba=False)
    acc_c, _, n_c = cpu.render(mode=ob.MODE_CANONICAL, threads=16, want_rgba=False)
    err = (acc_g - acc_c) / 16.0
    rmse = float(np.sqrt(np.mean(err ** 2)))
    # the coplanar region: pixels whose centre ray hits the light or the ceiling inside the light's rectangle
    kd = cpu.query_primary(ob.MODE_KD)[0]
    tie = ids != kd  # the reference's kd-tree order and the canonical order disagree only on exact ties
    bad = np.abs(err).max(axis=2) > 1e-9
    print(f"cornell 800x800x16: rays {n_g} vs oracle {n_c}, radiance RMSE {rmse:.3e}; {int(bad.sum())} pixels differ by > 1e-9, "
          f"{int(tie.sum())} pixel centres on the coplanar light/ceiling tie, RMSE outside differing pixels {float(np.sqrt(np.mean(err[~bad] ** 2))):.3e}")
    assert abs(n_g - n_c) <= max(2, n_c // 100000), (n_g, n_c)
    assert rmse <= RMSE_TOL
    assert bad.mean() < 0.02  # differences are confined to a few pixels (ties flipped by last-place libm differences)


def test_device_buffers_and_tonemap_with_torch(rt):
    """rt_render_device / rt_tonemap_device on caller-owned HBM (torch tensors)."""
    import torch

    device, ob, abi = rt
    s = load_scene("test", size=64)
    gpu, cpu = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
    acc = torch.zeros((64, 64, 3), dtype=torch.float64, device="cuda:0")
    rgba = torch.zeros((64, 64, 4), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    n, st = gpu.render_device(acc.data_ptr())
    gpu.tonemap_device(acc.data_ptr(), rgba.data_ptr())
    torch.cuda.synchronize()
    acc_c, rgba_c, n_c = cpu.render(mode=ob.MODE_CANONICAL, threads=4)
    assert n == n_c and st.ms_total > 0
    assert radiance_rmse(acc.cpu().numpy(), acc_c, 4) <= RMSE_TOL
    assert (rgba.cpu().numpy() == rgba_c).all(axis=2).mean() >= 0.999
    n2, _ = gpu.render_device(acc.data_ptr())  # += semantics
    torch.cuda.synchronize()
    assert np.allclose(acc.cpu().numpy(), 2 * acc_c, atol=1e-9)


@pytest.mark.parametrize("name,size,spp", [("test", 64, 4), ("refraction", 96, 4), ("physical", 64, 2), ("scene", 96, 4),
                                           ("cornell_box", 48, 1), ("mesh", 64, 3)])
def test_aovs_match_oracle(rt, name, size, spp):
    """The other two members of screen_raycast's ColorData (src/render/mod.rs:16-20): albedo = clamp(mean over the
    samples of the clamped material colour of the first hit) and the normal of sample 0 -- which follows the refraction
    chain (:485; refraction.json) and is (0,0,1) on a miss.  The colour must not change when AOVs are asked for."""
    device, ob, abi = rt
    s = load_scene(name, size=size, spp=spp)
    gpu, cpu = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
    acc_g, alb_g, nrm_g, n_g = gpu.render_aov()
    acc_c, alb_sum_c, nrm_c, n_c = cpu.render_aov(mode=ob.MODE_CANONICAL, threads=16)
    acc_plain, _, n_plain, _ = gpu.render(want_rgba=False)
    assert n_g == n_plain and np.array_equal(acc_g, acc_plain)
    assert abs(n_g - n_c) <= max(2, n_c // 100000)
    alb_c = alb_sum_c if spp == 1 else np.clip(alb_sum_c * (1.0 / spp), 0.0, 1.0)
    ids_c, _, _ = cpu.query_primary(ob.MODE_CANONICAL)
    tie = 0.03 if name == "cornell_box" else 0.0  # coplanar light / ceiling (quirk Q17)
    assert (np.abs(alb_g - alb_c).max(axis=2) > 1e-12).mean() <= tie
    assert (np.abs(nrm_g - nrm_c).max(axis=2) > REL_TOL).mean() <= tie
    miss = ids_c == NO_HIT
    assert np.array_equal(nrm_g[miss], np.broadcast_to([0.0, 0.0, 1.0], nrm_g[miss].shape))  # sample 0 is the pixel centre
    assert np.allclose(np.linalg.norm(nrm_g, axis=2), 1.0, atol=1e-12)
    # sample ranges compose: albedo sums add up, the normal comes with sample 0 only
    if spp > 1:
        import torch

        bufs = torch.zeros((3, size, size, 3), dtype=torch.float64, device="cuda:0")
        ptr = lambda i: bufs[i].data_ptr()
        torch.cuda.synchronize()
        gpu.render_aov_device(ptr(0), ptr(1), ptr(2), abi.RenderParams(s0=1, s1=spp))
        torch.cuda.synchronize()
        assert not bufs[2].any()
        gpu.render_aov_device(ptr(0), ptr(1), ptr(2), abi.RenderParams(s0=0, s1=1))
        torch.cuda.synchronize()
        assert np.allclose(bufs[1].cpu().numpy(), alb_sum_c, atol=1e-12)
        assert np.allclose(bufs[2].cpu().numpy(), nrm_g, atol=0)


def test_work_counters(rt):
    """RT_FLAG_COUNT_WORK feeds bench.py's roofline: nodes visited / primitives tested per ray."""
    device, _, abi = rt
    s = load_scene("scene", size=96)
    gpu = device.DeviceScene(s.desc)
    a0, _, n0, st0 = gpu.render(want_rgba=False, want_stats=True)
    a1, _, n1, st1 = gpu.render(abi.RenderParams(flags=abi.RT_FLAG_COUNT_WORK), want_rgba=False, want_stats=True)
    assert n0 == n1 and np.array_equal(a0, a1)
    assert st0.nodes_visited_closest == 0 and st1.nodes_visited_closest > st1.closest_rays
    assert st1.prims_tested_closest > 0 and st1.nodes_visited_shadow > 0


TIE_SCENES = ["test", "scene", "mesh", "refraction", "physical", "mesh_test", "occlusion", "emissive", "cornell_box"]


@pytest.mark.parametrize("name", TIE_SCENES)
def test_primary_ids_against_the_references_traversal_order(rt, name):
    """The reference's closest hit is `min_by` = the FIRST minimum in kd traversal order (src/core/bounds.rs:303-307,
    :344-353, :359-362); the kernels take the minimum of (t, tie key) whatever the traversal order (DESIGN.md
    "canonical hit"; tie key: unbounded objects in scene order first -- the reference's own rule -- then the bounded
    primitive with the higher id).  The two can differ only where two primitives are hit at EXACTLY the same distance.
    At the full 800x800 the GPU's primary ids are held against the oracle's MODE_KD -- the restatement of the
    reference's own tree and order -- on every shipped scene: no difference anywhere, cornell_box included, whose light
    is coplanar with its ceiling (quirk Q17, scenes/models/cornell_box/light.obj / ceiling.obj, y = 548.8: 2,386 pixel
    centres of the frame hit both at the same distance, and the reference's order gives the light on all of them)."""
    device, ob, _ = rt
    s = load_scene(name)
    assert (s.opts.width, s.opts.height) == (800, 800)
    gpu, cpu = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
    ids_g, t_g, _ = gpu.query_primary()
    ids_k, t_k, _ = cpu.query_primary(ob.MODE_KD)
    n = int((ids_g != ids_k).sum())
    print(f"{name}: {n} of {ids_g.size} primary ids differ from the reference's traversal order (MODE_KD)")
    assert n == 0
    hit = ids_k != NO_HIT
    assert np.array_equal(t_g[hit], t_k[hit])


def test_cornell_box_ties_beyond_the_camera_rays(rt):
    """Rays other than camera rays meet cornell_box's coplanar light / ceiling too (a GI ray that ends on the light
    gathers its emission, one that ends on the ceiling nothing), and for them the reference's winner follows the kd
    traversal order of THAT ray.  The frame against the oracle's MODE_KD: inside the radiance bar; the pixels that
    differ at all are printed -- that remainder is PARITY UNPINNED (no reference render of cornell_box exists, and
    which of two coplanar primitives a tie goes to is decided by last-ulp rounding on the reference's own hardware)."""
    device, ob, _ = rt
    s = load_scene("cornell_box", size=200, spp=4)
    gpu, cpu = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
    acc_g, _, n_g, _ = gpu.render(want_rgba=False)
    acc_k, _, n_k = cpu.render(mode=ob.MODE_KD, threads=16, want_rgba=False)
    differ = int((np.abs(acc_g - acc_k).max(axis=2) > 1e-9).sum())
    rmse = radiance_rmse(acc_g, acc_k, 4)
    print(f"cornell_box 200x200x4 against MODE_KD: rays {n_g} / {n_k}, {differ} of 40000 pixels differ, radiance RMSE {rmse:.2e}")
    assert abs(n_g - n_k) <= max(2, n_k // 100000)
    assert rmse <= RMSE_TOL and differ <= 400  # 80 pixels, 4.2e-4 when this was written
