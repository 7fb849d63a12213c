#!/usr/bin/env python3
"""bench.py -- Mrays/s of the MI355X hot path on BASELINE.json's headline config.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is ONE frame of scenes/scene.json (800x800, 4 spp, 1,524 primitives,
~147 M rays; BASELINE configs[2]): the scene is resident in HBM, every rank
renders its Morton-interleaved tile shard into a zeroed f64 accumulation buffer
through the C ABI (rt_render_device), one RCCL collective assembles the frame on
rank 0 (gather of each rank's own tiles; reduce(SUM) of the frames when samples
are sharded too), rank 0 tonemaps to RGBA8 on the device.  Total work is fixed as N
grows ("strong" scaling).  value = rays of all ranks / max-over-ranks time.

Rank 0 prints one JSON line; with N = 1 it also carries
  roofline     -- the dominant kernel class (HIP-event time inside the library,
                  on the stream the kernels run on) against the HBM peak, using
                  SURVEY.md 8(d)'s algorithmic bytes per ray;
  cpu_baseline -- the oracle's reference-order kd-tree renderer (a C++ port of
                  the Rust reference, which cannot be built here) on the host
                  cores, on a bounded sample of the same frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
PUBLISHED = {}         # BASELINE.json "published": {} -> vs_baseline is null


def _gi_form(bits):
    """rt_render_stats.gi_two_phase -> words (rt_api.hip GiTune)."""
    on = [w for i, w in enumerate(("GI rays of inner levels", "GI rays of the last level", "illumination nodes of the last level")) if bits >> i & 1]
    return "two-phase queries for " + " + ".join(on) if on else "closest-hit queries"


def gi_variant(bits, waves):
    """The switch set of a call in words: what `roofline.gi_query_form` says and what profiles/pmc_*.json carry as `_variant`."""
    return f"{_gi_form(bits)}, {waves} waves/SIMD"


def algorithmic_bytes(kind, nodes_per_ray, tri_per_ray, analytic_per_ray):
    """closest-hit ray 224 B, shadow ray 136 B of wavefront
    records + 64 B per BVH2 node visited + 72 B per triangle / 104 B per
    analytic primitive tested."""
    base = 224.0 if kind == "closest" else 136.0
    return base + 64.0 * nodes_per_ray + 72.0 * tri_per_ray + 104.0 * analytic_per_ray


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "scene.json"),
                    help="scene JSON, or 'synthetic' = BASELINE config 5 (1,000,000-triangle height field, generated on the fly; "
                         "3840x2160 64 spp unless --size/--spp say otherwise)")
    ap.add_argument("--spp", type=int, default=0, help="override samples_per_pixel (0 = scene file)")
    ap.add_argument("--size", type=int, default=0, help="override width=height (0 = scene file)")
    ap.add_argument("--tile", type=int, default=8, help="edge of the tiles dealt to ranks (8x8 = one 64-pixel granule: finest, best balanced)")
    ap.add_argument("--sample-shards", type=int, default=1)
    ap.add_argument("--pool-gib", type=float, default=0.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    from raytracer_amd import _abi, device, distributed, host

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Launched plainly with --gpus N > 1: ONE process drives N GPUs through the library (rt_render_multi_device: a host
    # thread and a stream per device, tiles sent to device 0 by peer copies).  Under torch.distributed.run (what the
    # driver does): one process per GPU, RCCL for the exchange.
    single_process_multi = world == 1 and args.gpus > 1
    if world != args.gpus and not single_process_multi:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # Rehearsal hook for a one-GPU box (tests only): MI355RT_BENCH_REHEARSE=1 puts every rank on cuda:0 and exchanges
    # over gloo, so the N > 1 control flow of this file can be run where RCCL cannot (one GPU cannot hold two ranks).
    rehearse = os.environ.get("MI355RT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    config_tag = "BASELINE configs[2]"
    if args.scene == "synthetic":
        import tempfile

        from raytracer_amd import synthetic

        out_dir = os.path.join(tempfile.gettempdir(), "mi355rt_synthetic")
        if rank == 0:
            args.scene = synthetic.write_scene(out_dir)
        if dist is not None:
            dist.barrier()
        args.scene = synthetic.write_scene(out_dir)  # the OBJ exists now; rewrites only the small JSON
        config_tag = "BASELINE configs[4], procedural"
    scene = host.LoadedScene.from_file(args.scene)
    if args.spp:
        scene.set_option("samples_per_pixel", args.spp)
    if args.size:  # square unless the scene is wider than tall (keeps the aspect of the 4K synthetic frame)
        aspect = scene.opts.height / scene.opts.width
        scene.set_option("width", args.size)
        scene.set_option("height", max(1, round(args.size * aspect)))
    w, h, spp = int(scene.opts.width), int(scene.opts.height), int(scene.opts.samples_per_pixel)
    gpu = device.DeviceScene(scene.desc, pool_bytes=int(args.pool_gib * (1 << 30)) or None)
    replicas = [gpu]
    if single_process_multi:
        if not rehearse and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} visible")
        for i in range(1, args.gpus):  # a replica of the scene per device (all on cuda:0 in a rehearsal)
            torch.cuda.set_device(0 if rehearse else i)
            replicas.append(device.DeviceScene(scene.desc, pool_bytes=int(args.pool_gib * (1 << 30)) or None))
        torch.cuda.set_device(local_rank)
    accum = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    stats_log = []

    def render_local(params, acc, flags=0):
        params.flags = flags
        # per-kernel HIP-event times feed the roofline, which only the N = 1 line carries; the events cost ~10 us per
        # launch (26 launches a frame), which a 1/8-frame shard of a few ms would notice
        want = world == 1
        n, st = gpu.render_device(acc.data_ptr(), params, stream=stream, want_stats=want)
        if want:
            stats_log.append(st)
        return n

    def step_multi():
        accum.zero_()
        torch.cuda.current_stream().synchronize()
        n, _ = device.render_multi_device(replicas, accum.data_ptr(), _abi.RenderParams(seed=0, tile_size=args.tile), stream=stream)
        gpu.tonemap_device(accum.data_ptr(), rgba.data_ptr(), stream=stream)
        return n

    def step(flags=0):
        if single_process_multi:
            return step_multi()
        accum.zero_()
        # frame exchange inside (gather of own tiles / reduce of frames); the ray counters are summed once per run
        _, local = distributed.render_frame(lambda p, a: render_local(p, a, flags), accum, rank, world, spp, seed=0,
                                            tile_size=args.tile, sample_shards=args.sample_shards, dist=dist, opts=scene.opts,
                                            reduce_count=False)
        if rank == 0:
            gpu.tonemap_device(accum.data_ptr(), rgba.data_ptr(), stream=stream)
        return local

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # scene set-up, not a step: the library settles which variant of the closest-hit kernels is fastest on this scene during
    # the scene's first render calls (an untimed one, two with its starting configuration, two per switch: eleven at most;
    # rt_tune.h GiTune; the frames are the same bit for bit)
    for _ in range(11):
        step()
    for _ in range(args.warmup):
        step()
    stats_log.clear()
    fence()
    t0 = time.perf_counter()
    rays = 0
    for _ in range(args.steps):
        rays += step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        cnt = torch.tensor([rays], dtype=torch.int64, device="cuda")
        dist.reduce(cnt, dst=0, op=dist.ReduceOp.SUM)
        rays = int(cnt.item())

    timed_stats = list(stats_log)
    out = None
    if rank == 0:
        value = rays / elapsed / 1e6
        out = {
            "metric": f"Mrays/sec (whole node) on {os.path.basename(args.scene)} {w}x{h} {spp}spp", "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": args.gpus if single_process_multi else world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / max(1, args.steps), 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic" if config_tag.endswith("procedural") else "scene file (the reference's own scenes/ input)",
            "config": {
                "workload": f"{os.path.basename(args.scene)} {w}x{h} {spp}spp, {scene.n_prims} primitives, "
                            f"{rays // max(1, args.steps)} rays/frame ({config_tag}); scene resident in HBM, frame reduced to rank 0",
                "parallelism": (f"tiles{args.gpus} (one process, rt_render_multi_device)" if single_process_multi else
                                f"tiles{world}" + (f"xsamples{args.sample_shards}" if args.sample_shards > 1 else "")),
                "tile": args.tile, "rays_per_step": rays // max(1, args.steps),
            },
        }

    if world == 1 and rank == 0 and not single_process_multi:
        # ---- roofline of the dominant kernel class (one extra untimed frame counts BVH work)
        stats_log.clear()
        step(_abi.RT_FLAG_COUNT_WORK)
        torch.cuda.synchronize()
        work = stats_log[-1]
        prim_types = scene.prim_types()
        tri_share = float((prim_types == _abi.RT_PRIM_TRIANGLE).mean()) if len(prim_types) else 0.0
        n_steps = max(1, len(timed_stats))
        ms = {k: sum(getattr(s, "ms_" + k) for s in timed_stats) / n_steps for k in ("raygen", "closest", "gi", "illum", "shadow", "expand", "resolve", "accum")}
        launches = {k: sum(getattr(s, "launches_" + k) for s in timed_stats) / n_steps for k in ("closest", "gi", "illum", "shadow")}
        nodes_c = work.nodes_visited_closest / max(1, work.closest_rays)
        prims_c = work.prims_tested_closest / max(1, work.closest_rays)
        nodes_s = work.nodes_visited_shadow / max(1, work.shadow_rays)
        prims_s = work.prims_tested_shadow / max(1, work.shadow_rays)
        bytes_c = algorithmic_bytes("closest", nodes_c, prims_c * tri_share, prims_c * (1 - tri_share))
        bytes_s = algorithmic_bytes("shadow", nodes_s, prims_s * tri_share, prims_s * (1 - tri_share))
        dominant = max(ms, key=ms.get)
        # the three traversal kernels: k_trace_closest (rays stored as nodes), k_trace_gi (GI rays drawn, traced and reduced
        # in place: closest-hit queries of SURVEY 8(d) all the same, counted in closest_rays) and k_trace_shadow
        kern = "gi" if dominant not in ("closest", "gi", "illum", "shadow") else dominant
        gi_rays, illum_rays = int(work.gi_rays), int(work.illum_rays)  # k_trace_illum_last: the last level's illumination nodes (two-phase queries)
        k_rays = {"closest": work.closest_rays - gi_rays - illum_rays, "gi": gi_rays, "illum": illum_rays, "shadow": work.shadow_rays}[kern]
        k_bytes = bytes_s if kern == "shadow" else bytes_c           # SURVEY 8(d) incl. the geometry term
        k_stream = 136.0 if kern == "shadow" else 224.0              # SURVEY 8(d) wavefront records only
        k_ms = ms[kern]
        n_launch = max(1.0, launches[kern])
        gbs = lambda nbytes, msec: nbytes / (msec * 1e-3) / 1e9 if msec > 0 else 0.0
        stream_gbs = gbs(k_rays * k_stream, k_ms)
        incl_cache_gbs = gbs(k_rays * k_bytes, k_ms)
        # HBM traffic (PMC passes cannot run inside this process: they are taken by tools/profile_all.sh on the same
        # workload and tree, committed under profiles/, and replayed here -- `source` says which file)
        default_workload = os.path.basename(args.scene) == "scene.json" and not args.spp and not args.size
        pmc_t, pmc_v = {}, {}
        if default_workload:
            try:
                pmc_t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                pmc_v = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu.json")))
            except Exception:
                pmc_t, pmc_v = {}, {}
        # ... and only if they were taken with the variant this run settled on: counters of another variant describe another kernel
        run_variant = gi_variant(timed_stats[-1].gi_two_phase, timed_stats[-1].gi_waves) if timed_stats else ""
        pmc_note = None
        for name, rec in (("pmc_traffic.json", pmc_t), ("pmc_valu.json", pmc_v)):
            if rec and rec.get("_variant") != run_variant:
                pmc_note = f"profiles/{name} was taken with [{rec.get('_variant')}], this run settled on [{run_variant}]: not replayed"
        if pmc_note:
            pmc_t, pmc_v = {}, {}
        src_t = f"profiles/pmc_traffic.json ({pmc_t.get('_tree', 'r01g')}, {pmc_t.get('_variant')}), replayed" if pmc_t else pmc_note
        src_v = f"profiles/pmc_valu.json ({pmc_v.get('_tree', 'r01g')}, {pmc_v.get('_variant')}), replayed" if pmc_v else pmc_note
        KERNEL_OF = {"closest": "k_trace_closest", "gi": "k_trace_gi", "illum": "k_trace_illum_last", "shadow": "k_trace_shadow"}
        traffic = pmc_t.get(KERNEL_OF[kern], {}).get("hbm_bytes_per_launch")
        measured_gbs = gbs(traffic * n_launch, k_ms) if traffic else None
        valu = None
        if pmc_v.get(KERNEL_OF[kern]):
            v = pmc_v[KERNEL_OF[kern]]
            valu = {k: v[k] for k in ("valu_busy", "lane_use", "fp64_tflops", "fp64_frac_of_peak", "wait_share") if k in v}
            valu["source"] = src_v
        elif pmc_note:
            valu = {"source": pmc_note}
        # whole frame: HBM bytes of every kernel of one step (PMC, replayed) against the step's wall time
        frame_bytes = None
        if pmc_t:
            per_step = {KERNEL_OF[k]: launches[k] for k in KERNEL_OF}
            tot = 0.0
            for k, rec in pmc_t.items():
                if not isinstance(rec, dict) or "hbm_bytes_per_launch" not in rec:
                    continue
                tot += rec["hbm_bytes_per_launch"] * per_step.get(k, rec.get("launches", 0) / max(1, pmc_t.get("_frames", 2)))
            frame_bytes = int(tot)
        step_ms = 1e3 * elapsed / max(1, args.steps)
        bound = "valu_issue" if (valu and valu.get("valu_busy", 0.0) >= 0.5 and (measured_gbs or 0.0) < 0.5 * HBM_PEAK_GBS) else "hbm"
        out["roofline"] = {
            # `achieved` / `frac`: the kernel's ALGORITHMIC HBM bytes -- SURVEY 8(d)'s wavefront records, 224 B per closest-hit
            # ray (136 B per shadow ray), which every design must move once -- over its HIP-event time.  The geometry term of
            # 8(d) (64 B per node visit, 72 / 104 B per primitive test) is cache traffic for scenes that fit L2 / LDS and is
            # reported apart (`algorithmic_incl_cache_gbs`: not a bound, it exceeds the HBM peak on L2-resident scenes).
            "bound": bound, "kernel": KERNEL_OF[kern], "achieved": round(stream_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(stream_gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src_t,
            "frac_hbm_stream": round(stream_gbs / HBM_PEAK_GBS, 4),
            "frac_hbm_measured": round(measured_gbs / HBM_PEAK_GBS, 4) if measured_gbs else None,
            "algorithmic_incl_cache_gbs": round(incl_cache_gbs, 2),
            "launches_per_step": launches[kern], "avg_launch_ms": round(k_ms / n_launch, 4),
            "stream_bytes_per_ray": k_stream, "algorithmic_bytes_per_ray_incl_cache": round(k_bytes, 1), "rays_per_step": int(k_rays),
            "nodes_per_ray": round(nodes_s if kern == "shadow" else nodes_c, 2),   # closest + gi rays together
            "prims_per_ray": round(prims_s if kern == "shadow" else prims_c, 2),
            "gi_query_form": run_variant,
            "kernel_ms_per_step": {k: round(v, 3) for k, v in ms.items()},
            # deep trees / fractions of a frame: the library runs a level's three traversal launches beside each other, so the
            # class times are spans that overlap (their sum exceeds the step) and a kernel's duration is its duration in company
            "launches_overlap": bool(timed_stats and timed_stats[-1].overlapped),
            "dominant_class": dominant,
            "valu": valu,  # co-limit: VALU issue occupancy / lane use / FP64 rate of the same kernel
            "frame": {
                "stream_bytes_per_step": int(work.closest_rays * 224.0 + work.shadow_rays * 136.0),
                "frac_hbm_stream": round(gbs(work.closest_rays * 224.0 + work.shadow_rays * 136.0, step_ms) / HBM_PEAK_GBS, 4),
                "hbm_bytes_per_step": frame_bytes, "hbm_bytes_source": src_t if frame_bytes else None,
                "frac_hbm_measured": round(gbs(frame_bytes, step_ms) / HBM_PEAK_GBS, 4) if frame_bytes else None,
            },
        }
        # ---- CPU baseline: the reference's algorithm (kd-tree order) on the host cores, bounded samples of the same frame
        if not args.no_cpu_baseline:
            from oracle import binding as ob

            # threads: the CPU share this process may use (a one-GPU box grants 16 of the machine's CPUs: OMP_NUM_THREADS)
            try:
                avail = len(os.sched_getaffinity(0))
            except Exception:
                avail = os.cpu_count() or 1
            share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
            cores = max(1, min(avail, share) if share > 0 else avail)
            model = "unknown"
            try:
                for line in open("/proc/cpuinfo"):
                    if line.startswith("model name"):
                        model = line.split(":", 1)[1].strip()
                        break
            except Exception:
                pass
            cpu = ob.OracleScene(scene.desc)
            rays_frame = rays // max(1, args.steps)

            def cpu_run(threads, budget_rays):  # a centred band of rows of the same frame, sized to ~budget_rays
                rows = max(1, min(h, int(h * min(1.0, budget_rays / max(1, rays_frame)))))
                y0 = (h - rows) // 2
                p = _abi.RenderParams(seed=0, x0=0, y0=y0, x1=w, y1=y0 + rows)
                t0 = time.perf_counter()
                _, _, n_cpu = cpu.render(p, mode=ob.MODE_KD, threads=threads, want_rgba=False)
                dt = time.perf_counter() - t0
                return n_cpu / dt / 1e6, f"rows {y0}..{y0 + rows} of the same frame ({n_cpu} rays, {dt:.1f} s)"

            v_all, s_all = cpu_run(cores, 10.0 * 3.0e6 * cores)  # ~10 s at ~3 Mrays/s per thread
            v_one, s_one = cpu_run(1, 10.0 * 3.0e6)
            out["cpu_baseline"] = {
                "value": round(v_all, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                "sample": s_all + ", oracle MODE_KD = C++ port of the reference (SAH kd-tree, f64, std::thread over rows)",
                "one_thread": {"value": round(v_one, 3), "cores": 1, "sample": s_one},
                "host": {"cpu_model": model, "nproc": os.cpu_count(), "usable_threads": cores},
            }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
