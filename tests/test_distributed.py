"""The N > 1 path on CPU: two `gloo` ranks render their tile / sample shards
(the CPU oracle stands in for the GPU renderer -- same rt_render_params
contract), one reduce(SUM) assembles the frame on rank 0.  On the GPU box
bench.py runs the same raytracer_amd.distributed.render_frame over RCCL."""
import os
import socket

import numpy as np
import pytest

from conftest import ROOT, load_scene


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, sample_shards, use_gather, out_path):
    import sys

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from oracle import binding as ob
    from raytracer_amd import distributed

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = load_scene("test", size=96)
        o = ob.OracleScene(s.desc)
        spp = s.opts.samples_per_pixel

        def render_local(params, accum):
            acc, _, count = o.render(params, mode=ob.MODE_CANONICAL, threads=2, want_rgba=False)
            # the oracle renders a window per call: emulate the tile shard by masking to this rank's tiles
            xy, tw, th = distributed.shard_tiles(s.opts, params)
            mask = np.zeros(acc.shape[:2], bool)
            for x, y in xy:
                mask[y : y + th, x : x + tw] = True
            accum += torch.from_numpy(acc * mask[..., None])
            return count  # (count is for the whole window; only the image is checked for tile shards)

        accum = torch.zeros((96, 96, 3), dtype=torch.float64)
        total, local = distributed.render_frame(render_local, accum, rank, world, spp, seed=0, tile_size=16,
                                                sample_shards=sample_shards, dist=dist, opts=s.opts if use_gather else None)
        if rank == 0:
            np.save(out_path, accum.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sample_shards,use_gather", [(1, True), (1, False), (2, False)])
def test_two_rank_gloo_frame_equals_single_rank(libs, tmp_path, sample_shards, use_gather):
    """use_gather: tile shards exchanged as one gather of each rank's own tiles instead of a full-frame reduce."""
    import torch.multiprocessing as mp

    _, ob = libs
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), sample_shards, use_gather, out), nprocs=2, join=True)
    frame = np.load(out)
    s = load_scene("test", size=96)
    full, _, _ = ob.OracleScene(s.desc).render(mode=ob.MODE_CANONICAL, threads=4, want_rgba=False)
    if sample_shards == 1:
        assert np.array_equal(frame, full)  # tile shards: one non-zero contributor per pixel
    else:
        assert np.allclose(frame, full, rtol=0, atol=1e-12)  # sample shards: f64 re-association only


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_on_one_gpu():
    """bench.py's N > 1 control flow (torch.distributed.run, shard plan, frame exchange, counter reduce, the JSON
    line) with both ranks on cuda:0 exchanging over gloo (MI355RT_BENCH_REHEARSE=1): RCCL itself needs one GPU per
    rank, which only the driver's multi-GPU node has."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, MI355RT_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--pool-gib", "6"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["config"]["parallelism"] == "tiles2"
    assert abs(line["config"]["rays_per_step"] - 147120917) <= 200 and line["value"] > 0
    assert "roofline" not in line and "cpu_baseline" not in line  # rank 0 at N = 1 only


@pytest.mark.gpu
def test_render_multi_one_process_two_and_three_replicas_on_one_gpu():
    """rt_render_multi / rt_render_multi_device (one host process, a host thread and a stream per scene)
    rehearsed with every replica on cuda:0 -- a one-GPU box has no second device, so the peer copy degenerates to a
    device-to-device copy; everything else (tile deal, threads, pack, inbox, unpack-add, counters) is the real path.
    The frame is bit-identical to rt_render's, whatever the number of replicas."""
    import torch

    from raytracer_amd import _abi, device

    s = load_scene("scene", size=200)
    scenes = [device.DeviceScene(s.desc, pool_bytes=2 << 30) for _ in range(3)]
    want_acc, want_rgba, want_n, _ = scenes[0].render(_abi.RenderParams(seed=9))
    for n in (1, 2, 3):
        acc, rgba, rays, stats = device.render_multi(scenes[:n], _abi.RenderParams(seed=9), want_stats=True)
        assert rays == want_n and np.array_equal(acc, want_acc) and np.array_equal(rgba, want_rgba)
        assert sum(st.ray_count for st in stats) == want_n and all(st.camera_rays > 0 for st in stats)
        if n > 1:  # an even deal: 8x8 tiles in Morton order
            cams = [st.camera_rays for st in stats]
            assert max(cams) - min(cams) <= 2 * 64 * 4
    # device form, += semantics, a window and a sample range; twice in a row (the cached tile lists are reused, then replaced)
    p = _abi.RenderParams(seed=9, x0=16, y0=40, x1=150, y1=180, s0=1, s1=3)
    ref, _, n_ref, _ = scenes[0].render(p, want_rgba=False)
    for _ in range(2):
        d_acc = torch.ones((200, 200, 3), dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        rays, _ = device.render_multi_device(scenes, d_acc.data_ptr(), p)
        torch.cuda.synchronize()
        assert rays == n_ref and np.array_equal(d_acc.cpu().numpy(), ref + 1.0)
    acc, _, rays, _ = device.render_multi(scenes[:2], _abi.RenderParams(seed=9), want_rgba=False)
    assert rays == want_n and np.array_equal(acc, want_acc)
    # misuse is an error code
    with pytest.raises(device.DeviceError, match="deals the tiles itself"):
        device.render_multi(scenes[:2], _abi.RenderParams(shard_index=0, shard_count=2))
    with pytest.raises(device.DeviceError, match="twice"):
        device.render_multi([scenes[0], scenes[0]])
    other = device.DeviceScene(load_scene("test", size=200).desc, pool_bytes=1 << 30)
    with pytest.raises(device.DeviceError, match="not replicas"):
        device.render_multi([scenes[0], other])


@pytest.mark.gpu
def test_bench_single_process_multi_rehearsal():
    """`python bench.py --gpus 2` without torch.distributed.run: one process, rt_render_multi_device (both replicas on
    cuda:0 here)."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, MI355RT_BENCH_REHEARSE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pool-gib", "6"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and "rt_render_multi_device" in line["config"]["parallelism"]
    assert abs(line["config"]["rays_per_step"] - 147120917) <= 200 and line["value"] > 0
