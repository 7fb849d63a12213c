#!/usr/bin/env python3
"""Per-class kernel times of one frame for a few pool budgets (development aid)."""
import sys
import time

sys.path.insert(0, ".")
from raytracer_amd import _abi, device, host  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "scene"
pools = [float(x) for x in sys.argv[2:]] or [8.0]
s = host.LoadedScene.from_file(f"scenes/{name}.json")
for gib in pools:
    g = device.DeviceScene(s.desc, pool_bytes=int(gib * (1 << 30)))
    for it in range(14):
        t0 = time.perf_counter()
        _, _, n, st = g.render(want_rgba=False, want_stats=True)
        dt = time.perf_counter() - t0
        d = st.as_dict()
        ksum = sum(d[k] for k in d if k.startswith("ms_") and k != "ms_total")
        print(f"pool {gib:5.2f} GiB it{it}: wall {dt*1e3:7.1f} ms total {d['ms_total']:7.1f} kernels {ksum:6.1f} | closest {d['ms_closest']:6.2f} gi {d['ms_gi']:6.2f} ({''.join(c if d['gi_two_phase'] >> i & 1 else '-' for i, c in enumerate('ILN'))}/{d['gi_waves']}w) illum {d['ms_illum']:5.2f} shadow {d['ms_shadow']:5.2f} "
              f"expand {d['ms_expand']:6.2f} resolve {d['ms_resolve']:5.2f} | batches {d['batches']} retries {d['retries']} rays {n} -> {n/d['ms_total']/1e3:7.1f} Mrays/s", flush=True)
    print(g.info())
    g.close()
