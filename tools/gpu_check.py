#!/usr/bin/env python3
"""Quick GPU-vs-oracle check used while developing (the real tests live in tests/)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import binding as ob  # noqa: E402
from raytracer_amd import _abi, device, host  # noqa: E402


def check(name, size, spp=None, full=False):
    s = host.LoadedScene.from_file(f"scenes/{name}.json")
    if size:
        s.set_option("width", size)
        s.set_option("height", size)
    if spp:
        s.set_option("samples_per_pixel", spp)
    o = ob.OracleScene(s.desc)
    t0 = time.time()
    g = device.DeviceScene(s.desc)
    print(f"[{name}] {s.n_prims} prims, create {time.time() - t0:.3f}s, info {g.info()}", flush=True)
    ids_g, t_g, n_g = g.query_primary()
    ids_o, t_o, n_o = o.query_primary(ob.MODE_CANONICAL)
    ids_k, _, _ = o.query_primary(ob.MODE_KD)
    hit = ids_o != 0xFFFFFFFF
    print(f"  primary ids mismatch GPU/canonical: {(ids_g != ids_o).sum()}  canonical/kd: {(ids_o != ids_k).sum()}  hits {hit.sum()}")
    same = (ids_g == ids_o) & hit
    if same.any():
        rel_t = np.abs(t_g[same] - t_o[same]) / np.maximum(np.abs(t_o[same]), 1e-300)
        print(f"  max rel t err {rel_t.max():.3e}  max normal err {np.abs(n_g[same] - n_o[same]).max():.3e}  bit-equal t {(t_g[same] == t_o[same]).mean():.6f}")
    t0 = time.time()
    acc_g, rgba_g, cnt_g, st = g.render(want_stats=True)
    tg = time.time() - t0
    print(f"  GPU render {tg:.3f}s rays {cnt_g}  stats {st.as_dict()}", flush=True)
    if not full:
        t0 = time.time()
        acc_o, rgba_o, cnt_o = o.render(mode=ob.MODE_CANONICAL, threads=16)
        to = time.time() - t0
        spp_n = s.opts.samples_per_pixel
        rmse = np.sqrt(np.mean((acc_g / spp_n - acc_o / spp_n) ** 2))
        print(f"  oracle render {to:.2f}s rays {cnt_o}  ray diff {cnt_g - cnt_o}  radiance RMSE {rmse:.3e}  max abs {np.abs(acc_g - acc_o).max() / spp_n:.3e}  rgba mismatch px {(rgba_g != rgba_o).any(axis=2).sum()}")
    return g, st


if __name__ == "__main__":
    names = sys.argv[1:] or ["test", "refraction", "physical", "cornell_box", "scene", "mesh"]
    for n in names:
        check(n, 160)
    print("---- full size scene.json")
    g, st = check("scene", 0, full=True)
    for _ in range(2):
        t0 = time.time()
        acc, rgba, cnt, st = g.render(want_stats=True)
        dt = time.time() - t0
        print(f"  full render wall {dt:.3f}s  {cnt / dt / 1e6:.1f} Mrays/s (host wall incl. copies)  kernel ms_total {st.ms_total:.1f} -> {cnt / st.ms_total / 1e3:.1f} Mrays/s")
        print("  ", st.as_dict())
