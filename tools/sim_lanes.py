#!/usr/bin/env python3
"""Development aid: per-ray traversal work of scene.json's GI rays (needs a probe build of the library that returns
node visits + 1e6 * primitive tests in place of t: MI355RT_LIB_COUNTS), and what wave-level schedules would make of it.
  python tools/sim_lanes.py            (normal library: writes /tmp/gi_rays.npz)
  MI355RT_LIB=build/lib_cnt.so python tools/sim_lanes.py sim"""
import sys
sys.path.insert(0, ".")
import numpy as np
from raytracer_amd import device, host

s = host.LoadedScene.from_file("scenes/scene.json")
g = device.DeviceScene(s.desc)
if len(sys.argv) < 2:
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
    n = 2_000_000 * 16 // 16
    np.savez("/tmp/gi_rays.npz", O=O[:n], D=D[:n])
    print("saved", n)
    sys.exit(0)
z = np.load("/tmp/gi_rays.npz")
O, D = z["O"], z["D"]
_, c = g.trace_rays(O, D)
prims = np.floor(c / 1e6); nodes = c - prims * 1e6
print("rays", len(c), "nodes/ray", nodes.mean(), "prims/ray", prims.mean())
cost = 75.0 * nodes + 220.0 * prims  # lane instructions
N = len(cost) // 4096 * 4096
cost = cost[:N]
base = cost.reshape(-1, 64).max(axis=1).sum()
print(f"lane use of one-ray-per-lane waves (cost-weighted): {cost.sum() / (64 * base):.3f}")
for K in (2, 4, 16):
    # lane-private queues: a wave owns 64*K consecutive rays; lane l takes rays l*K .. l*K+K-1 (a job's slots) one after the other
    q = cost.reshape(-1, 64, K).sum(axis=2).max(axis=1).sum()
    # and interleaved (lane l takes rays l, l+64, ...)
    qi = cost.reshape(-1, K, 64).sum(axis=1).max(axis=1).sum()
    print(f"K={K}: consecutive {base / q:.2f}x fewer wave-slots (lane use {cost.sum() / (64 * q):.3f}), interleaved {base / qi:.2f}x")
# node loop and leaf rounds separately (the while-while loop synchronises the two phases per outer iteration; this is an upper bound)
for nm, x in (("nodes", nodes[:N]), ("prims", prims[:N])):
    print(nm, "lane use", x.sum() / (64 * x.reshape(-1, 64).max(axis=1).sum()))
