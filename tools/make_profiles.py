#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter_collection CSVs into the JSON summaries kept under profiles/.

  make_profiles.py traffic FETCH.csv WRITE.csv > profiles/pmc_traffic.json
  make_profiles.py valu SQ_A.csv SQ_B.csv      > profiles/pmc_valu.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section):
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at
64 bytes, so it is doubled.  VALU figures use rocprof's own derived-metric definitions:
VALUBusy = 4 * SQ_ACTIVE_INST_VALU / (SIMDs * GRBM_GUI_ACTIVE per XCD); lane use =
SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)."""
import csv
import json
import re
import sys
from collections import defaultdict

SIMDS = 256 * 4      # MI355X: 256 CUs x 4 SIMD16
XCDS = 8
FP64_PEAK_TFLOPS = 78.6  # vector FP64, MI355X_MICROARCH.md


FRAMES, SKIP = 3, 1  # frames rendered by the profiled command, and how many of the first ones are left out (cold caches, pool sizing)


def load(paths):
    """Sums per kernel over the dispatches of the frames that count: a pass renders FRAMES identical frames (one pinned
    variant, one batch each), so a kernel's dispatches split evenly over them and the first SKIP shares are dropped."""
    agg = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for path in paths:
        rows = defaultdict(list)
        for r in csv.DictReader(open(path)):
            m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
            if m:
                rows[m.group(1)].append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
        for k, rs in rows.items():
            ids = sorted({d for d, _, _ in rs})
            keep = set(ids[len(ids) * SKIP // FRAMES:]) if len(ids) % FRAMES == 0 else set(ids)
            for d, name, v in rs:
                if d in keep:
                    agg[k][name] += v
            n[k] = max(n[k], len(keep))
    return agg, n


def traffic(paths):
    agg, n = load(paths)
    out = {"_about": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over tools/profile_frame.py scene 3 24 (first frame left out) on one "
                     "MI355X; counters are in KiB. hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 / launches: FETCH_SIZE doubled "
                     "per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B). Far below the algorithmic bytes because BVH "
                     "nodes / primitives (counted at 64-128 B per visit in the algorithmic figure) are L1/L2 hits and freshly written "
                     "ray records are partly served by the 256 MiB Infinity Cache."}
    for k in sorted(agg):
        f, w = agg[k].get("FETCH_SIZE", 0.0), agg[k].get("WRITE_SIZE", 0.0)
        out[k] = {"launches": n[k], "fetch_kib_raw": f, "write_kib": w, "hbm_bytes_per_launch": int((2 * f + w) * 1024 / max(1, n[k]))}
    return out


def valu(paths):
    agg, n = load(paths)
    out = {"_about": "rocprofv3 --pmc SQ_* passes over tools/profile_frame.py scene 3 24 on one MI355X (sums over the frames after the first). "
                     "valu_busy = 4*SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE/8 XCDs) (rocprof's VALUBusy); lane_use = "
                     "SQ_THREAD_CYCLES_VALU / (64*SQ_ACTIVE_INST_VALU); fp64_tflops = 64*lane_use*(2*FMA_F64 + MUL_F64 + ADD_F64 + "
                     "TRANS_F64) wave instructions / kernel time, against the 78.6 TFLOP/s vector FP64 peak; wait_share = "
                     "SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (share of a wave's life spent waiting for an instruction's operands or memory)."}
    for k in sorted(agg):
        c = agg[k]
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 2.0  # collected in both passes
        if not gui or "SQ_ACTIVE_INST_VALU" not in c:
            continue
        cycles = gui / XCDS
        seconds = cycles / 2.4e9
        busy = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (SIMDS * cycles)
        lane = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        f64_ops = 2 * c.get("SQ_INSTS_VALU_FMA_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
        tflops = 64.0 * lane * f64_ops / seconds / 1e12
        out[k] = {"launches": n[k], "valu_busy": round(busy, 4), "lane_use": round(lane, 4), "fp64_tflops": round(tflops, 2),
                  "fp64_frac_of_peak": round(tflops / FP64_PEAK_TFLOPS, 4),
                  "wait_share": round(c.get("SQ_WAIT_INST_ANY", 0.0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1.0)), 4),
                  "insts_valu": c.get("SQ_INSTS_VALU", 0.0), "insts_vmem_rd": c.get("SQ_INSTS_VMEM_RD", 0.0),
                  "insts_vmem_wr": c.get("SQ_INSTS_VMEM_WR", 0.0), "insts_lds": c.get("SQ_INSTS_LDS", 0.0),
                  "insts_salu": c.get("SQ_INSTS_SALU", 0.0), "gpu_cycles": cycles}
        if "TA_TA_BUSY_sum" in c:  # busy cycles summed over the 256 texture-addresser units
            out[k]["ta_busy"] = round(c["TA_TA_BUSY_sum"] / 256.0 / max(1.0, cycles), 4)
    return out


if __name__ == "__main__":
    # usage: make_profiles.py traffic|valu [--tree TAG] [--variant TEXT | --cfg N] [--frames F] [--skip S] CSV...
    #   TAG = the profiles/ prefix of the tree the passes ran on; TEXT = the switch set the passes were pinned to, in
    #   bench.py's words (roofline.gi_query_form): bench.py replays the figures only for a run that settled on the same
    args = sys.argv[2:]
    tree = variant = None
    while args and args[0].startswith("--"):
        if args[0] == "--tree":
            tree = args[1]
        elif args[0] == "--variant":
            variant = args[1]
        elif args[0] == "--cfg":  # MI355RT_GI_CFG of the passes -> the words bench.py uses for it
            import os

            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from bench import gi_variant

            variant = gi_variant(int(args[1]) & 7, 3 if int(args[1]) & 8 else 4)
        elif args[0] == "--frames":
            FRAMES = int(args[1])
        elif args[0] == "--skip":
            SKIP = int(args[1])
        args = args[2:]
    out = {"traffic": traffic, "valu": valu}[sys.argv[1]](args)
    if tree:
        out["_tree"] = tree
    if variant:
        out["_variant"] = variant
    out["_frames"] = FRAMES - SKIP
    print(json.dumps(out, indent=1))
