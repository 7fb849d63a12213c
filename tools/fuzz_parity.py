"""GPU vs oracle on seeded random scenes (development aid / soak test): fuzz_parity.py FIRST LAST [aov]"""
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from oracle import binding as ob
from raytracer_amd import device, host
from random_scenes import random_scene
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    txt = random_scene(seed)
    try:
        s = host.LoadedScene.from_string(txt, asset_base="scenes")
    except Exception as e:
        print(seed, "load error", e); bad += 1; continue
    g, o = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
    ig, tg, ng = g.query_primary(); io, to, no = o.query_primary(ob.MODE_CANONICAL)
    ag, _, cg, _ = g.render(want_rgba=False); ao, _, co = o.render(mode=ob.MODE_CANONICAL, threads=4, want_rgba=False)
    spp = s.opts.samples_per_pixel
    rmse = float(np.sqrt(np.mean(((ag - ao) / spp) ** 2)))
    ok = np.array_equal(ig, io) and cg == co and rmse <= 1e-9
    if len(sys.argv) > 3 and sys.argv[3] == "aov":  # ColorData's albedo / normal members too
        a2, alb_g, nrm_g, c2 = g.render_aov()
        _, alb_sum, nrm_o, _ = o.render_aov(mode=ob.MODE_CANONICAL, threads=4)
        alb_o = alb_sum if spp == 1 else np.clip(alb_sum * (1.0 / spp), 0.0, 1.0)
        ok = ok and np.array_equal(a2, ag) and c2 == cg and np.abs(alb_g - alb_o).max() <= 1e-12 and np.abs(nrm_g - nrm_o).max() <= 1e-9
    if not ok:
        bad += 1
    print(seed, "prims", s.n_prims, "rays", cg, co, "ids", int((ig != io).sum()), "rmse %.2e" % rmse, "OK" if ok else "MISMATCH", flush=True)
print("bad", bad)
