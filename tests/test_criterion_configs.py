"""The reference's criterion micro-configurations (benches/test_bench.rs:16-36, 52, 70): `screen_raycast(x, y)` at the
pixels (50,50) (50,150) (150,50) (150,150) of three 200x200, max_depth 5 scenes -- an empty scene with the camera at
(2,5,15) looking at (-1,0,0) (:23-36), scenes/benchmarks/simple.json (:52) and scenes/benchmarks/complex.json (:70).
The library offers no per-pixel call (the boundary is the frame, include/mi355rt.h), so a configuration is checked as
its whole frame, and the four pixels the bench times are printed and compared on their own (-s shows them)."""
import json
import os

import numpy as np
import pytest

from conftest import SCENES

PIXELS = [(50, 50), (50, 150), (150, 50), (150, 150)]  # test_bench.rs:16-21
EMPTY = json.dumps({"width": 200, "height": 200, "max_depth": 5, "camera": {"position": [2.0, 5.0, 15.0], "target": [-1.0, 0.0, 0.0]}})


def _load(name):
    from raytracer_amd import host

    if name == "empty":
        return host.LoadedScene.from_string(EMPTY, asset_base=SCENES)
    return host.LoadedScene.from_file(os.path.join(SCENES, "benchmarks", f"{name}.json"))


@pytest.mark.parametrize("name,prims", [("empty", 0), ("simple", 1), ("complex", 1523)])
def test_oracle_on_the_criterion_scenes(libs, name, prims):
    """CPU: the scenes load with the reference's options, and the reference-order kd-tree traversal (MODE_KD) and the
    traversal-order independent form the GPU implements (MODE_CANONICAL) agree on them."""
    _, ob = libs
    s = _load(name)
    assert (s.opts.width, s.opts.height, s.opts.max_depth, s.opts.samples_per_pixel) == (200, 200, 5, 4)
    assert s.n_prims == prims
    cpu = ob.OracleScene(s.desc)
    ids_kd, t_kd, _ = cpu.query_primary(ob.MODE_KD)
    ids_c, t_c, _ = cpu.query_primary(ob.MODE_CANONICAL)
    assert np.array_equal(ids_kd, ids_c) and np.array_equal(t_kd, t_c)
    if name == "empty":  # every get_color call is a counted miss (:530), nothing else is cast
        acc, _, n = cpu.render(mode=ob.MODE_KD, threads=4, want_rgba=False)
        assert n == 200 * 200 * 4 and not acc.any()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["empty", "simple", "complex"])
def test_criterion_scenes_match_the_oracle(name):
    from oracle import binding as ob
    from raytracer_amd import device

    s = _load(name)
    gpu, cpu = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
    ids_g, t_g, n_g = gpu.query_primary()
    ids_c, t_c, n_c = cpu.query_primary(ob.MODE_KD)  # the reference's own traversal order
    assert np.array_equal(ids_g, ids_c)
    hit = ids_c != 0xFFFFFFFF
    assert np.array_equal(t_g[hit], t_c[hit]) and np.abs(n_g[hit] - n_c[hit]).max(initial=0.0) <= 1e-5
    acc_g, rgba_g, rays_g, _ = gpu.render()
    acc_c, rgba_c, rays_c = cpu.render(mode=ob.MODE_CANONICAL, threads=16)
    assert rays_g == rays_c
    spp = int(s.opts.samples_per_pixel)
    assert np.abs(acc_g - acc_c).max() / spp <= 1e-9
    for x, y in PIXELS:
        print(f"{name}: screen_raycast({x},{y}) = {acc_g[y, x] / spp} rgba {rgba_g[y, x].tolist()} first hit {int(ids_g[y, x]) if ids_g[y, x] != 0xFFFFFFFF else None}")
        assert np.array_equal(rgba_g[y, x], rgba_c[y, x])
