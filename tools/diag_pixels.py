"""Pixels where GPU and oracle radiance differ on one random scene, per sample: diag_pixels.py SEED (development aid)"""
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from oracle import binding as ob
from raytracer_amd import _abi, device, host
from random_scenes import random_scene
seed = int(sys.argv[1])
txt = random_scene(seed)
s = host.LoadedScene.from_string(txt, asset_base="scenes")
g, o = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
w, h, spp = g.width, g.height, g.spp
print("size", w, h, "spp", spp, "prims", s.n_prims)
ag, _, cg, _ = g.render(want_rgba=False)
ao, _, co = o.render(mode=ob.MODE_CANONICAL, threads=8, want_rgba=False)
d = np.abs(ag - ao).max(axis=2)
ys, xs = np.nonzero(d > 1e-12)
print("rays", cg, co, "differing pixels", len(ys), "max diff", d.max())
for y, x in list(zip(ys, xs))[:6]:
    print("pixel", x, y, "gpu", ag[y, x], "oracle", ao[y, x])
    for k in range(spp):
        p = _abi.RenderParams(x0=int(x), y0=int(y), x1=int(x) + 1, y1=int(y) + 1, s0=k, s1=k + 1)
        a1, _, c1, _ = g.render(p, want_rgba=False)
        a2, _, c2 = o.render(p, mode=ob.MODE_CANONICAL, threads=1, want_rgba=False)
        if c1 != c2 or np.abs(a1[y, x] - a2[y, x]).max() > 1e-12:
            print("   sample", k, "rays", c1, c2, "gpu", a1[y, x], "oracle", a2[y, x])
if "--scene" in sys.argv:
    print(txt)
