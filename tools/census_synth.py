#!/usr/bin/env python3
"""Census of the synthetic scene (development aid)."""
import os, sys, tempfile
os.environ["MI355RT_CENSUS"] = "1"
sys.path.insert(0, ".")
from raytracer_amd import _abi, device, host, synthetic  # noqa: E402
w = int(sys.argv[1]) if len(sys.argv) > 1 else 960
p = synthetic.write_scene(os.path.join(tempfile.gettempdir(), "mi355rt_synthetic"), width=w, height=w * 9 // 16, spp=4)
s = host.LoadedScene.from_file(p)
g = device.DeviceScene(s.desc)
g.render(want_rgba=False)
_, _, n, st = g.render(_abi.RenderParams(flags=_abi.RT_FLAG_COUNT_WORK), want_rgba=False, want_stats=True)
d = st.as_dict()
print({k: d[k] for k in d if "rays" in k or "visited" in k or "tested" in k})
