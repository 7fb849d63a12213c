#!/usr/bin/env python3
"""Development aid: one RT_FLAG_COUNT_WORK frame of a scene after the tuner has settled (MI355RT_CENSUS=1 prints node kinds per level): census.py [SCENE]"""
import sys
sys.path.insert(0, ".")
from raytracer_amd import _abi, device, host
s = host.LoadedScene.from_file(f"scenes/{sys.argv[1] if len(sys.argv) > 1 else 'scene'}.json")
g = device.DeviceScene(s.desc)
for _ in range(5): g.render(want_rgba=False)
_, _, n, st = g.render(_abi.RenderParams(flags=_abi.RT_FLAG_COUNT_WORK), want_rgba=False, want_stats=True)
print(st.as_dict())
