#!/bin/bash
# usage (on the GPU box): tools/probe_set.sh TAG [scenes...] -- per-class kernel times of the given scenes + the 1/8 shard timing -> gpurun_out/TAG/
tag=$1; shift
scenes=${@:-scene physical mesh}
out=gpurun_out/$tag
mkdir -p $out
for sc in $scenes; do
  MI355RT_TIMING=1 timeout -k 10 200 python tools/perf_probe.py $sc 24 > $out/probe_$sc.txt 2>&1 || exit 1
  grep "closest-hit kernels" $out/probe_$sc.txt
  tail -n 3 $out/probe_$sc.txt | head -n 2 | cut -c1-230
done
timeout -k 10 200 python tools/perf_shard.py 8 8 > $out/shard.txt 2>&1 || exit 1
tail -n 3 $out/shard.txt
