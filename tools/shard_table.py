#!/usr/bin/env python3
"""What each of 8 ranks would do for one frame, measured on ONE GPU (the build has no 8-GPU node): the wall time of every
rank's shard (rt_render_device, no statistics, settled switches), max / mean -- the strong-scaling efficiency before the
exchange is mean-of-whole-frame / (8 x max shard) -- and the bytes the rank would send in the exchange step.
  scene.json       800x800x4: 8 tile shards (8x8 tiles, Morton-interleaved); gather of each rank's own tiles
  config 5         1,000,000 triangles 3840x2160x64: 4 tile shards x 2 sample ranges (--sample-shards 2); reduce(SUM)
                   of the frames (every pixel has two contributors)
usage: shard_table.py [out.json] [--quick]   (--quick: config 5 at 1920x1080x8)"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

from raytracer_amd import _abi, device, distributed, host, synthetic  # noqa: E402


def shards(scene, world, sample_shards, tile, rounds, settle):
    w, h, spp = int(scene.opts.width), int(scene.opts.height), int(scene.opts.samples_per_pixel)
    g = device.DeviceScene(scene.desc)
    acc = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(settle):  # whole frames: the library settles its switches on calls of a million GI rays and more
        g.render_device(acc.data_ptr(), _abi.RenderParams(), stream=stream, want_stats=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    whole_rays, _ = g.render_device(acc.data_ptr(), _abi.RenderParams(), stream=stream, want_stats=False)
    torch.cuda.synchronize()
    whole_ms = 1e3 * (time.perf_counter() - t0)
    rows = []
    for rank in range(world):
        p = distributed.shard_params(rank, world, spp, 0, tile, sample_shards)
        best, rays = 1e30, 0
        for _ in range(rounds):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rays, _ = g.render_device(acc.data_ptr(), p, stream=stream, want_stats=False)
            torch.cuda.synchronize()
            best = min(best, 1e3 * (time.perf_counter() - t0))
        rows.append({"rank": rank, "ms": round(best, 3), "rays": int(rays)})
    ms = [r["ms"] for r in rows]
    tile_shards = world // sample_shards
    if sample_shards == 1:  # gather: a rank sends its own tiles (f64 RGB), rank 0 keeps its own
        sent = w * h * 3 * 8 // tile_shards
        form = "gather of each rank's own tiles to rank 0"
    else:                   # reduce(SUM) of whole frames
        sent = w * h * 3 * 8
        form = "reduce(SUM) of the ranks' frames to rank 0"
    g.close()
    return {"frame": f"{w}x{h}x{spp}", "world": world, "tile_shards": tile_shards, "sample_shards": sample_shards, "tile": tile,
            "whole_frame_ms": round(whole_ms, 3), "whole_frame_rays": int(whole_rays), "shards": rows, "max_ms": max(ms), "mean_ms": round(sum(ms) / len(ms), 3),
            "max_over_mean": round(max(ms) / (sum(ms) / len(ms)), 3), "ideal_ms": round(whole_ms / world, 3),
            "efficiency_before_exchange": round(whole_ms / world / max(ms), 3), "exchange": form, "bytes_sent_per_rank": sent,
            "exchange_ms_at_one_xgmi_link_153GBs": round(sent / 153e9 * 1e3, 3)}


def main():
    quick = "--quick" in sys.argv
    out = {"_about": __doc__.split("usage:")[0].strip()}
    s = host.LoadedScene.from_file("scenes/scene.json")
    out["scene.json"] = shards(s, 8, 1, 8, 5, 12)
    print(json.dumps(out["scene.json"]), flush=True)
    kw = dict(width=1920, height=1080, spp=8) if quick else {}
    s5 = host.LoadedScene.from_file(synthetic.write_scene(os.path.join(tempfile.gettempdir(), "mi355rt_synthetic"), **kw))
    out["config5"] = shards(s5, 8, 2, 8, 1 if not quick else 2, 2 if not quick else 12)
    print(json.dumps(out["config5"]), flush=True)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if args:
        open(args[0], "w").write(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
