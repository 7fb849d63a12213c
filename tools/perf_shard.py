#!/usr/bin/env python3
"""Wall vs kernel time of one rank's tile shard (development aid): perf_shard.py [shard_count]"""
import sys, time
sys.path.insert(0, ".")
import torch
from raytracer_amd import _abi, device, host  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 32
s = host.LoadedScene.from_file("scenes/scene.json")
g = device.DeviceScene(s.desc)
acc = torch.zeros((800, 800, 3), dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for stats in (True,):
    for it in range(2 * n):
        p = _abi.RenderParams(shard_index=(it // 2) % n, shard_count=n, tile_size=tile)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cnt, st = g.render_device(acc.data_ptr(), p, stream=stream, want_stats=stats)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        k = sum(getattr(st, "ms_" + x) for x in ("raygen", "closest", "gi", "illum", "shadow", "expand", "resolve", "accum")) if stats else float("nan")
        print(f"shards {n} tile {tile} stats {stats} it{it}: wall {dt:6.2f} ms  event total {st.ms_total if stats else float('nan'):6.2f}  kernels {k:6.2f}  rays {cnt}  batches {st.batches if stats else -1} | " + (" ".join(f"{x} {getattr(st, 'ms_' + x):.2f}" for x in ("raygen", "closest", "gi", "illum", "shadow", "expand", "resolve", "accum")) if stats else ""), flush=True)
