#!/usr/bin/env python3
"""Development aid: rt_scene_create on the 1M-triangle synthetic scene, three times (MI355RT_TIMING=1 prints the phases)."""
import os, sys, tempfile, time
sys.path.insert(0, ".")
from raytracer_amd import device, host, synthetic
p = synthetic.write_scene(os.path.join(tempfile.gettempdir(), "mi355rt_synthetic"), width=1920, height=1080, spp=4)
t0 = time.perf_counter(); s = host.LoadedScene.from_file(p); t1 = time.perf_counter()
print(f"host load {1e3*(t1-t0):.0f} ms", flush=True)
for i in range(3):
    t0 = time.perf_counter(); g = device.DeviceScene(s.desc); t1 = time.perf_counter()
    print(f"rt_scene_create {1e3*(t1-t0):.0f} ms", flush=True)
    g.close()
