"""Multi-GPU sharding of one frame: one process per GPU (torch.distributed; the
"nccl" backend is RCCL over xGMI on ROCm).

The path shards with no exchange during tracing (pixels and samples are
independent): every rank renders a disjoint, Morton-interleaved
set of tiles (and, for high sample counts, a disjoint sample range of them)
into a zero-initialised full-frame f64 accumulation buffer; ONE collective
assembles the frame on rank 0 -- a gather of each rank's own tiles for pure tile
shards (every pixel has one contributor), a reduce(SUM) of the frames when the
sample range is split too (equal up to f64 re-association) -- and one reduce sums
the ray counters.
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


_INDEX_CACHE = {}


def _tile_pixels(opts, params, pad_tiles, device):
    """Flat pixel indices [pad_tiles * tw * th] of a shard's tiles (row-major inside a tile) and their
    validity (pixels of partial edge tiles / padding tiles are invalid and point at pixel 0)."""
    import torch

    xy, tw, th = shard_tiles(opts, params)
    w, h = int(opts.width), int(opts.height)
    idx = np.zeros((pad_tiles, th, tw), np.int64)
    ok = np.zeros((pad_tiles, th, tw), bool)
    for t, (x0, y0) in enumerate(xy):
        ys, xs = np.arange(y0, y0 + th)[:, None], np.arange(x0, x0 + tw)[None, :]
        v = (ys < h) & (xs < w)
        idx[t] = np.where(v, ys * w + xs, 0)
        ok[t] = v
    return torch.from_numpy(idx.reshape(-1)).to(device), torch.from_numpy(ok.reshape(-1)).to(device)


def _gather_plan(opts, world, spp, seed, tile_size, device):
    key = (int(opts.width), int(opts.height), world, tile_size, str(device))
    plan = _INDEX_CACHE.get(key)
    if plan is None:
        counts = [len(shard_tiles(opts, shard_params(r, world, spp, seed, tile_size))[0]) for r in range(world)]
        pad = max(1, max(counts))
        plan = [_tile_pixels(opts, shard_params(r, world, spp, seed, tile_size), pad, device) for r in range(world)]
        _INDEX_CACHE[key] = plan
    return plan


def exchange_form(sample_shards, opts, exchange=None):
    """Which collective assembles the frame -- decided from the call's arguments alone, BEFORE any collective is entered,
    so that every rank takes the same one (a rank that changed its mind inside a collective would desynchronise RCCL)."""
    if exchange is None:
        exchange = "gather" if (sample_shards == 1 and opts is not None) else "reduce"
    if exchange not in ("gather", "reduce"):
        raise ValueError("exchange must be 'gather' or 'reduce'")
    if exchange == "gather" and (sample_shards != 1 or opts is None):
        raise ValueError("the gather form needs pure tile shards and the scene's rt_render_options")
    return exchange


def render_frame(render_local, accum, rank, world, spp, seed=0, tile_size=32, sample_shards=1, dist=None, opts=None,
                 reduce_count=True, exchange=None):
    """Render this rank's shard with `render_local(params, accum) -> ray_count`
    and assemble the frame on rank 0.  `accum` is a zero-initialised torch tensor
    [h,w,3] float64 on the rank's device.  Returns (total_ray_count on rank 0, local count).

    Exchange step: with pure tile shards every pixel has exactly one contributor,
    so (when `opts` = the scene's rt_render_options is given) each rank sends only
    its own tiles -- 1/world of the frame -- to rank 0 in ONE gather, point to
    point over xGMI, and rank 0 scatters them into place: bit-identical to a
    reduce(SUM) of the full frames at 1/world of the bytes per link.  With sample
    shards pixels have several contributors and the full frames are reduce(SUM)ed.
    `reduce_count=False` leaves the ray counters to the caller (one reduce per
    run instead of one per frame).  `exchange` forces "gather" or "reduce" (same value on every rank)."""
    import torch

    form = exchange_form(sample_shards, opts, exchange)
    params = shard_params(rank, world, spp, seed, tile_size, sample_shards)
    local = int(render_local(params, accum))
    total = local
    if world > 1:
        if form == "gather":
            plan = _gather_plan(opts, world, spp, seed, tile_size, accum.device)
            flat = accum.view(-1, 3)
            mine = flat[plan[rank][0]]
            parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
            dist.gather(mine, parts, dst=0)
            if rank == 0:
                for r in range(1, world):
                    idx, ok = plan[r]
                    flat[idx[ok]] = parts[r][ok]
        else:
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        if reduce_count:
            cnt = torch.tensor([local], dtype=torch.int64, device=accum.device)
            dist.reduce(cnt, dst=0, op=dist.ReduceOp.SUM)
            total = int(cnt.item())
    return total, local
