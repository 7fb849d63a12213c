#!/usr/bin/env python3
"""Render a few frames of a scene (for rocprofv3 runs): profile_frame.py SCENE [FRAMES] [POOL_GIB]
SCENE = a name under scenes/ (scene, mesh, ...) or synthetic:WIDTH:SPP (BASELINE config 5's 1,000,000-triangle scene, 16:9)."""
import os
import sys
import tempfile

sys.path.insert(0, ".")
from raytracer_amd import device, host, synthetic  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "scene"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pool = float(sys.argv[3]) if len(sys.argv) > 3 else 0
if name.startswith("synthetic"):
    _, w, spp = name.split(":")
    path = synthetic.write_scene(os.path.join(tempfile.gettempdir(), "mi355rt_synthetic"), width=int(w), height=int(w) * 9 // 16, spp=int(spp))
else:
    path = f"scenes/{name}.json"
s = host.LoadedScene.from_file(path)
g = device.DeviceScene(s.desc, pool_bytes=int(pool * (1 << 30)) or None)
for _ in range(frames):
    _, _, n, st = g.render(want_rgba=False, want_stats=True)
print("rays", n, "batches", st.batches, "switches", st.gi_two_phase, st.gi_waves)
