#!/usr/bin/env python3
"""What-if (development aid): how much faster would the closest-hit kernel trace the GI rays of scene.json if a wave's
64 rays were coherent?  Builds level-1-like rays (16 cosine-hemisphere directions about every primary hit, tile order
as the renderer emits them) and traces them through rt_trace_rays in native order and sorted by direction octant
inside windows of W rays.  Run under rocprofv3 --kernel-trace; the k_trace_closest<false, false> durations, in
order, are: native, then one per window size."""
import sys
sys.path.insert(0, ".")
import numpy as np
from raytracer_amd import device, host

s = host.LoadedScene.from_file("scenes/scene.json")
g = device.DeviceScene(s.desc)
ids, t, nrm = g.query_primary()
cam = s.desc.contents.camera
w = h = 800
pos = np.array(list(cam.position)); m = np.array(list(cam.cam_to_world)).reshape(3, 3)
ys, xs = np.mgrid[0:h, 0:w]
nx = (xs + 0.5) * 2.0 / w - 1.0; ny = 1.0 - (ys + 0.5) * 2.0 / h
f = np.tan(np.radians(cam.fov_deg) / 2)
d = np.stack([nx * f, ny * f, -np.ones_like(nx)], -1); d /= np.linalg.norm(d, axis=-1, keepdims=True)
d = d @ m.T
P = pos + d * t[..., None]
hit = ids != 0xFFFFFFFF
# tile order (8x8), as the renderer deals pixels
tile = (ys // 8) * (w // 8) + xs // 8
order = np.lexsort(((xs % 8).ravel(), (ys % 8).ravel(), tile.ravel()))
order = order[hit.ravel()[order]]
Pn, Nn = P.reshape(-1, 3)[order], nrm.reshape(-1, 3)[order]
rng = np.random.default_rng(1)
G = 16
u1, u2 = rng.random((len(Pn), G)), rng.random((len(Pn), G))
r, phi = np.sqrt(u1), 2 * np.pi * u2
a = np.where(np.abs(Nn[:, :1]) > 0.9, [[0.0, 1.0, 0.0]], [[1.0, 0.0, 0.0]])
tu = np.cross(Nn, a); tu /= np.linalg.norm(tu, axis=1, keepdims=True); tv = np.cross(Nn, tu)
D = (tu[:, None] * (r * np.cos(phi))[..., None] + tv[:, None] * (r * np.sin(phi))[..., None] + Nn[:, None] * np.sqrt(1 - u1)[..., None]).reshape(-1, 3)
O = (np.repeat(Pn, G, axis=0) + D * 1e-10)
print("rays", len(O), flush=True)
def run(o, dd, tag):
    i, tt = g.trace_rays(o, dd)
    print(tag, "hits", int((i != 0xFFFFFFFF).sum()), flush=True)
run(O, D, "native")
octant = (D[:, 0] > 0) * 4 + (D[:, 1] > 0) * 2 + (D[:, 2] > 0) * 1
for W in (1024, 4096, 65536, len(O)):
    win = np.arange(len(O)) // W
    perm = np.lexsort((octant, win))
    run(O[perm], D[perm], f"octant-sorted in windows of {W}")
# finer: 6 bits of direction (octant + dominant axis ...) via quantised direction
q = np.clip(((D + 1) * 2).astype(int), 0, 3); key = q[:, 0] * 16 + q[:, 1] * 4 + q[:, 2]
for W in (4096, 65536):
    win = np.arange(len(O)) // W
    perm = np.lexsort((key, win))
    run(O[perm], D[perm], f"64-bin direction sort in windows of {W}")
