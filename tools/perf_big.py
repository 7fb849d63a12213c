import os, sys, tempfile, time
sys.path.insert(0, ".")
from raytracer_amd import device, host, synthetic
w, spp = int(sys.argv[1]), int(sys.argv[2])
p = synthetic.write_scene(os.path.join(tempfile.gettempdir(), "mi355rt_synthetic"), width=w, height=w * 9 // 16, spp=spp)
s = host.LoadedScene.from_file(p)
g = device.DeviceScene(s.desc)
for it in range(9):
    _, _, n, st = g.render(want_rgba=False, want_stats=True)
    d = st.as_dict()
    print(f"synthetic {w} x{spp} it{it}: total {d['ms_total']:8.1f} | closest {d['ms_closest']:7.2f} gi {d['ms_gi']:7.2f} ({''.join(c if d['gi_two_phase'] >> i & 1 else '-' for i, c in enumerate('ILN'))}/{d['gi_waves']}w) illum {d['ms_illum']:5.2f} shadow {d['ms_shadow']:6.2f} expand {d['ms_expand']:7.2f} resolve {d['ms_resolve']:6.2f} | batches {d['batches']} retries {d['retries']} rays {n} -> {n/d['ms_total']/1e3:7.1f} Mrays/s", flush=True)
