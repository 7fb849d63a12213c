#!/usr/bin/env python3
"""Wall time of rank 0's 1/N tile shard of scene.json, settled switches, best and median of 20 calls (development aid): shard_quick.py [N]"""
import statistics, sys, time
sys.path.insert(0, ".")
import torch
from raytracer_amd import _abi, device, distributed, host  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = host.LoadedScene.from_file("scenes/scene.json")
g = device.DeviceScene(s.desc)
acc = torch.zeros((800, 800, 3), dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for _ in range(14):
    g.render_device(acc.data_ptr(), _abi.RenderParams(), stream=stream, want_stats=False)
torch.cuda.synchronize()
whole = []
for _ in range(5):
    t0 = time.perf_counter(); g.render_device(acc.data_ptr(), _abi.RenderParams(), stream=stream, want_stats=False); torch.cuda.synchronize(); whole.append(1e3 * (time.perf_counter() - t0))
p = distributed.shard_params(0, n, 4, 0, 8, 1)
ts = []
for _ in range(22):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.render_device(acc.data_ptr(), p, stream=stream, want_stats=False); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
ts = ts[2:]
print(f"whole frame {min(whole):.3f} ms; 1/{n} shard best {min(ts):.3f} median {statistics.median(ts):.3f} ms; efficiency {min(whole) / (n * min(ts)):.3f}")
