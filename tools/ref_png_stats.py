#!/usr/bin/env python3
"""Full-size GPU renders against the reference's committed renders (development aid):
8x8 block-mean RMSE in 8-bit levels, mean difference, ray count vs published, frame time."""
import json, math, os, sys, time
sys.path.insert(0, ".")
import numpy as np
from PIL import Image
from raytracer_amd import device, host
counts = json.load(open("tests/golden/ray_counts.json"))
def block_mean(img, b=8):
    h, w, _ = img.shape
    return img.reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3))
for name in sys.argv[1:] or ["test", "mesh", "refraction", "physical", "mesh_test", "scene", "occlusion", "emissive"]:
    s = host.LoadedScene.from_file(f"scenes/{name}.json")
    g = device.DeviceScene(s.desc)
    g.render(want_rgba=False)
    t0 = time.perf_counter()
    acc, rgba, n, st = g.render(want_stats=True)
    dt = time.perf_counter() - t0
    ref = np.array(Image.open(f"tests/golden/renders/{name}.png").convert("RGB")).astype(np.float64)
    ours = rgba[..., :3].astype(np.float64)
    d = block_mean(ours) - block_mean(ref)
    pub = counts.get(name, {}).get("published")
    print(f"{name:11s} prims {s.n_prims:6d} rays {n:10d} pub-rel {((n - pub) / pub if pub else float('nan')):+.2e} block-RMSE {math.sqrt(float((d**2).mean())):6.3f} "
          f"mean diff {ours.mean() - ref.mean():+.3f}  kernels {st.ms_total:7.2f} ms -> {n / st.ms_total / 1e3:7.1f} Mrays/s", flush=True)
