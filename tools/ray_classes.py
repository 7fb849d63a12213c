#!/usr/bin/env python3
"""Development aid: how a frame's rays split into classes (camera, GI in place, last-level illumination nodes, shadow): ray_classes.py SCENE..."""
import sys
sys.path.insert(0, ".")
from raytracer_amd import device, host
for name in sys.argv[1:]:
    s = host.LoadedScene.from_file(f"scenes/{name}.json")
    g = device.DeviceScene(s.desc)
    _, _, n, st = g.render(want_rgba=False, want_stats=True)
    d = st.as_dict()
    print(name, {k: d[k] for k in ("ray_count", "closest_rays", "shadow_rays", "camera_rays", "gi_rays", "illum_rays", "batches")})
