#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_all.sh TAG [CFG]
# kernel stats of bench.py + PMC passes (each its own run) -> gpurun_out/prof_TAG/
# CFG = MI355RT_GI_CFG for the PMC passes (default 2 = two-phase GI queries on the last level, four waves: what bench.py
# settles on for scene.json): the counters then describe the variant that was timed; bench.py checks it (make_profiles.py --variant).
set -e
tag=$1
cfg=${2:-2}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o $tag -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_line.json 2> $out/bench.err
export MI355RT_GI_CFG=$cfg
pass() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 tools/profile_frame.py scene 3 24 > $out/$name.log 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq_a SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 GRBM_GUI_ACTIVE TA_TA_BUSY_sum
pass sq_b SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE
find $out -name "*.csv" | head -20
tail -1 $out/bench_line.json
