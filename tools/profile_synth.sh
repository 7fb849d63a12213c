#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_synth.sh TAG [CFG]
# BASELINE config 5's scene (1,000,000 triangles) at 1920x1080x4: kernel stats of bench.py --scene synthetic + PMC passes
# (each its own run, pinned to CFG = MI355RT_GI_CFG, default 6 = what the scene settles on) -> gpurun_out/prof_TAG/
set -e
tag=$1
cfg=${2:-6}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o $tag -- python3 bench.py --scene synthetic --size 1920 --spp 4 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_line.json 2> $out/bench.err
export MI355RT_GI_CFG=$cfg
pass() { name=$1; shift; timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 tools/profile_frame.py synthetic:1920:4 3 24 > $out/$name.log 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq_a SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 GRBM_GUI_ACTIVE TA_TA_BUSY_sum
pass sq_b SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE
tail -n 2 $out/*.log
tail -c 1500 $out/bench_line.json
