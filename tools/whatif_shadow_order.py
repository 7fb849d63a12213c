#!/usr/bin/env python3
"""What-if (development aid): shadow rays of a scene's primary hits (one per point light facing the surface, cast from the
light, as k_expand emits them) traced in parent-major order and light-major inside blocks of 64 parents.  Run under
rocprofv3 --kernel-trace; per scene two k_trace_shadow<false, true> durations, in that order."""
import sys
sys.path.insert(0, ".")
import numpy as np
from raytracer_amd import device, host, _abi

for name in sys.argv[1:] or ["scene", "physical"]:
    s = host.LoadedScene.from_file(f"scenes/{name}.json")
    g = device.DeviceScene(s.desc)
    ids, t, nrm = g.query_primary()
    d_ = s.desc.contents
    cam = d_.camera
    w, h = int(d_.opts.width), int(d_.opts.height)
    pos = np.array(list(cam.position)); m = np.array(list(cam.cam_to_world)).reshape(3, 3)
    ys, xs = np.mgrid[0:h, 0:w]
    nx = ((xs + 0.5) * 2.0 / w - 1.0) * (w / h); ny = 1.0 - (ys + 0.5) * 2.0 / h
    f = np.tan(np.radians(cam.fov_deg) / 2)
    d = np.stack([nx * f, ny * f, -np.ones_like(nx)], -1); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = d @ m.T
    P = pos + d * t[..., None]
    hit = ids != 0xFFFFFFFF
    tile = (ys // 8) * (w // 8) + xs // 8
    order = np.lexsort(((xs % 8).ravel(), (ys % 8).ravel(), tile.ravel()))
    order = order[hit.ravel()[order]]
    Pn, Nn = P.reshape(-1, 3)[order], nrm.reshape(-1, 3)[order]
    Pn, Nn = np.repeat(Pn, 4, axis=0), np.repeat(Nn, 4, axis=0)  # 4 spp: four parents per pixel
    lights = np.array([list(d_.lights[i].position) for i in range(d_.n_lights) if d_.lights[i].kind == _abi.RT_LIGHT_POINT])
    nl = len(lights)
    ld = lights[None, :, :] - Pn[:, None, :]
    dist = np.linalg.norm(ld, axis=2)
    ldn = ld / dist[..., None]
    facing = (ldn * Nn[:, None, :]).sum(axis=2) > 0
    parent = np.repeat(np.arange(len(Pn)), nl).reshape(len(Pn), nl)
    light = np.tile(np.arange(nl), len(Pn)).reshape(len(Pn), nl)
    O = np.broadcast_to(lights[None], ld.shape)[facing]
    D = -ldn[facing]
    T = dist[facing] - 1e-10
    par, lig = parent[facing], light[facing]
    print(name, "point lights", nl, "shadow rays", len(T), flush=True)
    g.trace_rays(O, D, tmax=T, shadow=True)
    perm = np.lexsort((par, lig, par // 64))
    g.trace_rays(O[perm], D[perm], tmax=T[perm], shadow=True)
    g.close()
