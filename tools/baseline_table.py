#!/usr/bin/env python3
"""one row per config on ONE MI355X -- GPU Mrays/s (kernel time of the frame, HIP events),
BVH work per ray, the HBM fractions of SURVEY 8(d)'s wavefront records (224 B per closest-hit ray, 136 B per shadow
ray: what must cross HBM; `frac_*`, each <= 1) for the closest-hit kernels and for the frame, 8(d)'s figure including
the geometry term as a rate (`incl_cache_gbs`: cache traffic on L2-resident scenes, not a bound), the variant of
k_trace_gi the library settled on, and the CPU port (oracle MODE_KD, std::thread) on a bounded sample of the same frame.
usage: baseline_table.py [out.json]   (run on the GPU box from the repo root)"""
import json, os, sys, tempfile, time
sys.path.insert(0, ".")
import numpy as np
from oracle import binding as ob
from raytracer_amd import _abi, device, host, synthetic

CORES = min(16, os.cpu_count() or 1)
HBM = 8000.0


def bytes_per_ray(kind, nodes, prims, tri_share):
    return (224.0 if kind == "closest" else 136.0) + 64.0 * nodes + 72.0 * prims * tri_share + 104.0 * prims * (1 - tri_share)


def row(label, scene, cpu_rows):
    g = device.DeviceScene(scene.desc)
    for _ in range(12):  # first frames: pool sizing, and the library settles its closest-hit switches (eleven calls at most)
        g.render(want_rgba=False)
    best = None
    for _ in range(3):
        _, _, n, st = g.render(want_rgba=False, want_stats=True)
        if best is None or st.ms_total < best.ms_total:
            best = st
    _, _, _, w = g.render(_abi.RenderParams(flags=_abi.RT_FLAG_COUNT_WORK), want_rgba=False, want_stats=True)
    types = scene.prim_types()
    tri = float((types == _abi.RT_PRIM_TRIANGLE).mean()) if len(types) else 0.0
    nc, pc = w.nodes_visited_closest / max(1, w.closest_rays), w.prims_tested_closest / max(1, w.closest_rays)
    bc = bytes_per_ray("closest", nc, pc, tri)
    ms_hit = best.ms_closest + best.ms_gi + best.ms_illum  # k_trace_closest + k_trace_gi + k_trace_illum_last: every closest-hit query of the frame
    gbs = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    h, wd = int(scene.opts.height), int(scene.opts.width)
    y0 = (h - cpu_rows) // 2
    o = ob.OracleScene(scene.desc)
    t0 = time.perf_counter()
    _, _, n_cpu = o.render(_abi.RenderParams(x0=0, y0=y0, x1=wd, y1=y0 + cpu_rows), mode=ob.MODE_KD, threads=CORES, want_rgba=False)
    dt = time.perf_counter() - t0
    r = {"config": label, "camera_rays": wd * h * int(scene.opts.samples_per_pixel), "primitives": int(scene.n_prims), "rays": int(n),
         "gpu_ms": round(best.ms_total, 2), "gpu_mrays": round(n / best.ms_total / 1e3, 1),
         "ms": {k: round(getattr(best, "ms_" + k), 2) for k in ("closest", "gi", "illum", "shadow", "expand", "resolve")}, "batches": int(best.batches),
         "gi_variant": "".join(c if best.gi_two_phase >> i & 1 else "-" for i, c in enumerate("ILN")) + f"/{best.gi_waves}w",
         "nodes_per_ray": round(nc, 2), "prims_per_ray": round(pc, 2),
         "launches_overlap": bool(best.overlapped),  # the class times are spans that overlap: no per-class fraction then
         "frac_hbm_stream_closest": None if best.overlapped else round(gbs(w.closest_rays * 224.0, ms_hit) / HBM, 3),
         "frac_hbm_stream_frame": round(gbs(w.closest_rays * 224.0 + w.shadow_rays * 136.0, best.ms_total) / HBM, 3),
         "bytes_per_ray_incl_cache": round(bc, 1), "incl_cache_gbs": None if best.overlapped else round(gbs(w.closest_rays * bc, ms_hit), 1),
         "cpu_mrays": round(n_cpu / dt / 1e6, 2), "cpu_cores": CORES, "cpu_sample": f"rows {y0}..{y0 + cpu_rows} ({n_cpu} rays, {dt:.1f} s)"}
    print(json.dumps(r), flush=True)
    return r


def load(name, **opts):
    s = host.LoadedScene.from_file(f"scenes/{name}.json")
    for k, v in opts.items():
        s.set_option(k, v)
    return s


rows = [row("test.json", load("test"), 800), row("cornell_box.json @16 spp", load("cornell_box", samples_per_pixel=16), 800),
        row("scene.json @4 spp", load("scene"), 800), row("mesh.json", load("mesh"), 800), row("mesh_test.json", load("mesh_test"), 200), row("physical.json", load("physical"), 200),
        row("refraction.json", load("refraction"), 400)]
p = synthetic.write_scene(os.path.join(tempfile.gettempdir(), "mi355rt_synthetic"), width=1920, height=1080, spp=4)
rows.append(row("synthetic 1M-tri 1920x1080 @4 spp", host.LoadedScene.from_file(p), 60))
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
