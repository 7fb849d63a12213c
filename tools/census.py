#!/usr/bin/env python3
"""Per-level node census of one frame (MI355RT_CENSUS=1 + RT_FLAG_COUNT_WORK; development aid)."""
import os, sys
os.environ["MI355RT_CENSUS"] = "1"
sys.path.insert(0, ".")
from raytracer_amd import _abi, device, host  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "scene"
s = host.LoadedScene.from_file(f"scenes/{name}.json")
g = device.DeviceScene(s.desc)
_, _, n, st = g.render(_abi.RenderParams(flags=_abi.RT_FLAG_COUNT_WORK), want_rgba=False, want_stats=True)
d = st.as_dict()
print({k: d[k] for k in d if "rays" in k or "visited" in k or "tested" in k})
