#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_all.sh TAG
# kernel stats of bench.py + PMC passes (each its own run) -> gpurun_out/prof_TAG/
set -e
tag=$1
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o $tag -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_line.json 2> $out/bench.err
# the PMC passes render two frames: pin the form of GI query the bench settles on for scene.json (closest-hit queries, k_trace_gi built for three waves per SIMD), or the
# first frames would be its trials of the other variants
export MI355RT_GI_TWO_PHASE=0 MI355RT_GI_WAVES=3
pass() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 tools/profile_frame.py scene 2 24 > $out/$name.log 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq_a SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 GRBM_GUI_ACTIVE TA_TA_BUSY_sum
pass sq_b SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE
find $out -name "*.csv" | head -20
tail -1 $out/bench_line.json
