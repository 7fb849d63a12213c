#!/bin/bash
# usage (on the GPU box, from the repo root): tools/shard_trace.sh OUTDIR [N]  -- per-launch timeline of the last 1/N tile-shard frames of scene.json
set -e
out=$PWD/$1; n=${2:-8}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --output-format rocpd -d $out/shardtrace -o t -- python3 tools/shard_frames.py $n 3 > $out/shardtrace.log 2>&1 || { tail -n 20 $out/shardtrace.log; exit 1; }
f=$(find $out/shardtrace -name "*.db" | head -n 1)
python3 tools/rocpd_summary.py $f --launches 75 > $out/shard_launches.txt
rm -rf $out/shardtrace
tail -n 80 $out/shard_launches.txt
