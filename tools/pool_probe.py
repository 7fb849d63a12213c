import os, sys, tempfile
sys.path.insert(0, ".")
from raytracer_amd import device, host, synthetic
p = synthetic.write_scene(os.path.join(tempfile.gettempdir(), "mi355rt_synthetic"), width=3840, height=2160, spp=64)
s = host.LoadedScene.from_file(p)
for gib in (24, 64, 128):
    g = device.DeviceScene(s.desc, pool_bytes=gib << 30)
    for it in range(2):
        _, _, n, st = g.render(want_rgba=False, want_stats=True)
        d = st.as_dict()
        print(f"pool {gib} GiB it{it}: total {d['ms_total']:7.1f} | closest {d['ms_closest']:6.1f} shadow {d['ms_shadow']:5.1f} expand {d['ms_expand']:6.1f} resolve {d['ms_resolve']:5.1f} | batches {d['batches']} -> {n/d['ms_total']/1e3:7.1f} Mrays/s", flush=True)
    del g
