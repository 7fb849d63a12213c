#!/usr/bin/env python3
"""What-if (development aid): the GI rays of a scene's primary hits traced as closest-hit queries (what k_trace_gi does)
and as any-hit queries limited to max_occlusion_distance (what most of them need: tools/../DESIGN.md "two-phase GI").
Run under rocprofv3 --kernel-trace; per scene the durations of k_trace_closest<false,false> and
k_trace_shadow<false,true>, in that order."""
import sys
sys.path.insert(0, ".")
import numpy as np
from raytracer_amd import device, host

for name in sys.argv[1:] or ["scene", "mesh", "mesh_test"]:
    s = host.LoadedScene.from_file(f"scenes/{name}.json")
    g = device.DeviceScene(s.desc)
    ids, t, nrm = g.query_primary()
    cam = s.desc.contents.camera
    o = s.desc.contents.opts
    w, h = int(o.width), int(o.height)
    pos = np.array(list(cam.position)); m = np.array(list(cam.cam_to_world)).reshape(3, 3)
    ys, xs = np.mgrid[0:h, 0:w]
    aspect = w / h
    nx = ((xs + 0.5) * 2.0 / w - 1.0) * aspect; ny = 1.0 - (ys + 0.5) * 2.0 / h
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
    i1, t1 = g.trace_rays(O, D)
    dmax = float(o.max_occlusion_distance)
    i2, _ = g.trace_rays(O, D, tmax=np.full(len(O), dmax), shadow=True)
    print(name, "rays", len(O), "closest hits", int((i1 != 0xFFFFFFFF).sum()), "within D", int(((i1 != 0xFFFFFFFF) & (t1 <= dmax)).sum()), "any-hit(D) hits", int((i2 != 0).sum()), "D", dmax, flush=True)
    g.close()
