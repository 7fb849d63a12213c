#!/usr/bin/env python3
"""What-if (development aid): scene.json's GI rays of the primary hits traced against the scene, against the scene
without the room cube (object 0, whose box is the whole scene) and against the room alone -- what testing such a
primitive for every ray up front, outside the tree, could buy.  Run under rocprofv3 --kernel-trace."""
import json, os, sys, tempfile
sys.path.insert(0, ".")
import numpy as np
from raytracer_amd import device, host

src = json.load(open("scenes/scene.json"))
def variant(objs, name):
    d = dict(src); d["objects"] = objs
    p = os.path.join("scenes", name)
    json.dump(d, open(p, "w"))
    return p
paths = {"all": "scenes/scene.json", "no_room": variant(src["objects"][1:], "_whatif_noroom.json"), "room": variant(src["objects"][:1], "_whatif_room.json")}
s = host.LoadedScene.from_file(paths["all"])
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
for name, p in paths.items():
    gg = device.DeviceScene(host.LoadedScene.from_file(p).desc)
    i, _ = gg.trace_rays(O, D)
    print(name, "hits", int((i != 0xFFFFFFFF).sum()), flush=True)
    gg.close()
for p in ("scenes/_whatif_noroom.json", "scenes/_whatif_room.json"):
    os.remove(p)
