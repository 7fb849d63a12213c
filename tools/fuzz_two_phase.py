#!/usr/bin/env python3
"""Fuzz (development aid): forced two-phase queries against closest-hit queries on seeded random scenes, bit for bit.
  python tools/fuzz_two_phase.py FIRST LAST [size]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from random_scenes import random_scene
from raytracer_amd import device, host

def scene(desc, mode):
    os.environ["MI355RT_GI_TWO_PHASE"] = mode
    try:
        return device.DeviceScene(desc, pool_bytes=2 << 30)
    finally:
        del os.environ["MI355RT_GI_TWO_PHASE"]

first, last = int(sys.argv[1]), int(sys.argv[2])
size = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bad = illum = 0
for seed in range(first, last):
    s = host.LoadedScene.from_string(random_scene(seed), asset_base=os.path.join(ROOT, "scenes"))
    if size:
        s.set_option("width", size); s.set_option("height", size)
    a, b = scene(s.desc, "0"), scene(s.desc, "1")
    acc1, _, n1, _ = a.render(want_rgba=False)
    acc2, _, n2, st = b.render(want_rgba=False, want_stats=True)
    illum += st.illum_rays
    if n1 != n2 or not np.array_equal(acc1, acc2):
        bad += 1
        print("MISMATCH seed", seed, n1, n2, float(np.abs(acc1 - acc2).max()), flush=True)
    a.close(); b.close()
print(f"seeds {first}..{last}: {bad} mismatches; illumination nodes answered by k_trace_illum_last: {illum}")
