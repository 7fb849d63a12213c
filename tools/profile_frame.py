#!/usr/bin/env python3
"""Render a few frames of a scene (for rocprofv3 --kernel-trace --stats runs)."""
import sys

sys.path.insert(0, ".")
from raytracer_amd import device, host  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "scene"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pool = float(sys.argv[3]) if len(sys.argv) > 3 else 0
s = host.LoadedScene.from_file(f"scenes/{name}.json")
g = device.DeviceScene(s.desc, pool_bytes=int(pool * (1 << 30)) or None)
for _ in range(frames):
    _, _, n, st = g.render(want_rgba=False, want_stats=False)
print("rays", n)
