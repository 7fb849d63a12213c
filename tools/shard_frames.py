#!/usr/bin/env python3
"""Render the 1/N tile shards of scene.json a few times without statistics (for rocprofv3 --kernel-trace): shard_frames.py [N] [rounds]"""
import sys
sys.path.insert(0, ".")
import torch
from raytracer_amd import _abi, device, host  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s = host.LoadedScene.from_file("scenes/scene.json")
g = device.DeviceScene(s.desc)
acc = torch.zeros((800, 800, 3), dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for r in range(rounds):
    for k in range(n):
        g.render_device(acc.data_ptr(), _abi.RenderParams(shard_index=k, shard_count=n, tile_size=8), stream=stream, want_stats=False)
torch.cuda.synchronize()
