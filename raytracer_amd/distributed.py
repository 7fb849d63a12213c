"""Multi-GPU sharding of one frame: one process per GPU (torch.distributed; the
"nccl" backend is RCCL over xGMI on ROCm).

The path shards with no exchange during tracing (pixels and samples are
independent): every rank renders a disjoint, Morton-interleaved
set of tiles (and, for high sample counts, a disjoint sample range of them)
into a zero-initialised full-frame f64 accumulation buffer; ONE reduce(SUM) to
rank 0 assembles the frame, one more sums the ray counters.  The sum is exact
for tile shards (every pixel has one non-zero contributor) and equal up to f64
re-association for sample shards.
"""
import ctypes as C

import numpy as np

from . import _abi


def shard_params(rank, world, spp, seed=0, tile_size=32, sample_shards=1):
    """RenderParams of `rank`: world = tile shards x sample shards."""
    if world % sample_shards != 0:
        raise ValueError("world size must be a multiple of sample_shards")
    if sample_shards > spp:
        raise ValueError("more sample shards than samples")
    tile_shards = world // sample_shards
    k = rank % sample_shards
    p = _abi.RenderParams(seed=seed, tile_size=tile_size, shard_index=rank // sample_shards, shard_count=tile_shards)
    if sample_shards > 1:
        p.s0 = (k * spp) // sample_shards
        p.s1 = ((k + 1) * spp) // sample_shards
    return p


def shard_tiles(opts, params):
    """Tile origins [n,2], tile_w, tile_h of a RenderParams (rt_shard_tiles; host only)."""
    from . import device

    lib = device.lib()
    lib.rt_shard_tiles.argtypes = [C.POINTER(_abi.RenderOptions), C.POINTER(_abi.RenderParams), C.POINTER(C.c_uint32),
                                   C.c_uint32] + [C.POINTER(C.c_uint32)] * 3
    lib.rt_shard_tiles.restype = C.c_int
    n, tw, th = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = lib.rt_shard_tiles(C.byref(opts), C.byref(params), None, 0, C.byref(n), C.byref(tw), C.byref(th))
    if rc:
        raise device.DeviceError(f"[{rc}] {lib.rt_last_error().decode()}")
    xy = np.zeros((n.value, 2), np.uint32)
    if n.value:
        lib.rt_shard_tiles(C.byref(opts), C.byref(params), xy.ctypes.data_as(C.POINTER(C.c_uint32)), n.value, C.byref(n),
                           C.byref(tw), C.byref(th))
    return xy, int(tw.value), int(th.value)


def render_frame(render_local, accum, rank, world, spp, seed=0, tile_size=32, sample_shards=1, dist=None):
    """Render this rank's shard with `render_local(params, accum) -> ray_count`
    and reduce to rank 0.  `accum` is a zero-initialised torch tensor [h,w,3]
    float64 on the rank's device.  Returns (total_ray_count on rank 0, local count)."""
    import torch

    params = shard_params(rank, world, spp, seed, tile_size, sample_shards)
    local = int(render_local(params, accum))
    total = local
    if world > 1:
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        cnt = torch.tensor([local], dtype=torch.int64, device=accum.device)
        dist.reduce(cnt, dst=0, op=dist.ReduceOp.SUM)
        total = int(cnt.item())
    return total, local
