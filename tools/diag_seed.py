"""Locate a ray-count difference between GPU and oracle on one random scene: diag_seed.py SEED (development aid)"""
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from oracle import binding as ob
from raytracer_amd import _abi, device, host
from random_scenes import random_scene
seed = int(sys.argv[1])
s = host.LoadedScene.from_string(random_scene(seed), asset_base="scenes")
g, o = device.DeviceScene(s.desc), ob.OracleScene(s.desc)
w, h, spp = g.width, g.height, g.spp
print("size", w, h, "spp", spp, "prims", s.n_prims)
for it in range(3):
    a, _, c, st = g.render(want_rgba=False, want_stats=True)
    print("gpu run", it, c, "closest", st.closest_rays, "shadow", st.shadow_rays)
g2 = device.DeviceScene(s.desc, pool_bytes=64 << 20)
print("gpu small pool", g2.render(want_rgba=False)[2])
print("oracle", o.render(mode=ob.MODE_CANONICAL, threads=8, want_rgba=False)[2], "kd", o.render(mode=ob.MODE_KD, threads=8, want_rgba=False)[2])
def counts(x0, y0, x1, y1, s0=0, s1=0):
    p = _abi.RenderParams(x0=x0, y0=y0, x1=x1, y1=y1, s0=s0, s1=s1)
    ag, _, cg, _ = g.render(p, want_rgba=False)
    ao, _, co = o.render(p, mode=ob.MODE_CANONICAL, threads=8, want_rgba=False)
    return cg, co, ag, ao
y0, y1 = 0, h
while y1 - y0 > 1:
    m = (y0 + y1) // 2
    cg, co, _, _ = counts(0, y0, w, m)
    if cg != co: y1 = m
    else: y0 = m
x0, x1 = 0, w
while x1 - x0 > 1:
    m = (x0 + x1) // 2
    cg, co, _, _ = counts(x0, y0, m, y1)
    if cg != co: x1 = m
    else: x0 = m
cg, co, ag, ao = counts(x0, y0, x1, y1)
print("pixel", x0, y0, "gpu", cg, "oracle", co, "rgb gpu", ag[y0, x0], "oracle", ao[y0, x0])
for k in range(spp):
    cg, co, ag, ao = counts(x0, y0, x1, y1, k, k + 1)
    print("  sample", k, cg, co, "diff rgb", (ag[y0, x0] - ao[y0, x0]))
