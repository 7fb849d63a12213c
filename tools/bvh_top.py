#!/usr/bin/env python3
"""Development aid: the top three levels of scene.json's BVH as resident on the device (child boxes dequantised, leaf primitive ids)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from raytracer_amd import device, host
s = host.LoadedScene.from_file("scenes/scene.json")
g = device.DeviceScene(s.desc)
nodes, leaf_ids, root, _ = g.bvh()
origin, scale = np.array([-30.0, -11.0, -100.0]), 120.0 / 65535.0
def show(ref, depth, maxd):
    if ref & 0x80000000:
        first, cnt = ref & 0x0FFFFFFF, ((ref >> 28) & 7) + 1
        print("  " * depth + f"leaf prims {leaf_ids[first:first+cnt].tolist()}")
        return
    n = nodes[ref]
    for side in ("l", "r"):
        mn = origin + n[side + "min"].astype(float) * scale; mx = origin + n[side + "max"].astype(float) * scale
        child = int(n["left" if side == "l" else "right"])
        print("  " * depth + f"{side}: box {np.round(mn,1).tolist()} {np.round(mx,1).tolist()} ref {'leaf' if child & 0x80000000 else child}")
        if depth < maxd: show(child, depth + 1, maxd)
show(root, 0, 3)
