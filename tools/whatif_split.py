#!/usr/bin/env python3
"""What-if (development aid): scene.json's GI rays of the primary hits, traced as one array and as two -- the rays
that miss the box of everything but the room (the root's other child) and the rays that enter it.  Run under
rocprofv3 --kernel-trace; the k_trace_closest<false,false> durations in order: primary, all, short, long, and the
same three again with the long rays' order shuffled (coherence lost)."""
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
# the two clusters of objects inside the room (tools/bvh_top.py: nodes 3 and 2 of the tree)
def enters(bmin, bmax):
    with np.errstate(divide="ignore", invalid="ignore"):
        t1, t2 = (np.array(bmin) - O) / D, (np.array(bmax) - O) / D
    tn, tf = np.minimum(t1, t2).max(axis=1), np.maximum(t1, t2).min(axis=1)
    return (tn <= tf) & (tf >= 0)
long_ = enters([-25.0, -11.0, -35.0], [13.8, 15.0, -11.4]) | enters([-13.3, -10.0, -14.0], [7.4, 9.7, 1.0])
print("rays", len(O), "long", int(long_.sum()), flush=True)
g.trace_rays(O, D)
g.trace_rays(O[~long_], D[~long_])
g.trace_rays(O[long_], D[long_])
