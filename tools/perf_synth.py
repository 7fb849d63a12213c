#!/usr/bin/env python3
"""Per-class kernel times of the synthetic 1M-triangle scene (development aid). usage: perf_synth.py [width] [spp]"""
import os, sys, tempfile, time
sys.path.insert(0, ".")
from raytracer_amd import device, host, synthetic  # noqa: E402
w = int(sys.argv[1]) if len(sys.argv) > 1 else 960
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
p = synthetic.write_scene(os.path.join(tempfile.gettempdir(), "mi355rt_synthetic"), width=w, height=w * 9 // 16, spp=spp)
s = host.LoadedScene.from_file(p)
g = device.DeviceScene(s.desc)
for it in range(3):
    _, _, n, st = g.render(want_rgba=False, want_stats=True)
    d = st.as_dict()
    print(f"synthetic it{it}: total {d['ms_total']:7.1f} | closest {d['ms_closest']:6.2f} shadow {d['ms_shadow']:5.2f} expand {d['ms_expand']:6.2f} resolve {d['ms_resolve']:5.2f} | batches {d['batches']} rays {n} -> {n/d['ms_total']/1e3:7.1f} Mrays/s", flush=True)
print(g.info())
