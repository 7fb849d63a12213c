"""Soak test (development aid): on seeded random scenes the frame must not depend on how it is cut up -- a 2 MiB pool
(many batches, overflow re-runs), three tile shards of odd tile size, two windows, two sample ranges.
usage: fuzz_partition.py FIRST LAST"""
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from raytracer_amd import _abi, device, host
from random_scenes import random_scene
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    s = host.LoadedScene.from_string(random_scene(seed), asset_base="scenes")
    w, h, spp = int(s.opts.width), int(s.opts.height), int(s.opts.samples_per_pixel)
    g = device.DeviceScene(s.desc)
    a, _, n, _ = g.render(want_rgba=False)
    small = device.DeviceScene(s.desc, pool_bytes=2 << 20)
    b, _, m, st = small.render(want_rgba=False, want_stats=True)
    ok = n == m and np.array_equal(a, b)
    acc, tot = np.zeros_like(a), 0
    for r in range(3):
        c, _, k, _ = g.render(_abi.RenderParams(shard_index=r, shard_count=3, tile_size=5), want_rgba=False)
        acc += c; tot += k
    ok = ok and tot == n and np.array_equal(acc, a)
    xm = max(1, w // 3)
    l, _, k1, _ = g.render(_abi.RenderParams(x0=0, y0=0, x1=xm, y1=h), want_rgba=False)
    r_, _, k2, _ = g.render(_abi.RenderParams(x0=xm, y0=0, x1=w, y1=h), want_rgba=False)
    ok = ok and k1 + k2 == n and np.array_equal(l + r_, a)
    if spp > 1:
        s0, _, c0, _ = g.render(_abi.RenderParams(s0=0, s1=1), want_rgba=False)
        s1, _, c1, _ = g.render(_abi.RenderParams(s0=1, s1=spp), want_rgba=False)
        ok = ok and c0 + c1 == n and np.allclose(s0 + s1, a, rtol=0, atol=1e-12)
    bad += 0 if ok else 1
    print(seed, w, h, spp, "rays", n, "batches(small pool)", st.batches, "retries", st.retries, "OK" if ok else "MISMATCH", flush=True)
print("bad", bad)
