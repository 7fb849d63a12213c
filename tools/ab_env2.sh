#!/bin/bash
# usage: tools/ab_env2.sh VAR "V1 V2 .." SCENE...   kernel times of each scene with VAR set to each value
var=$1; vals=$2; shift 2
for v in $vals; do
  for s in "$@"; do
    echo -n "== $var=$v $s: "
    if [ "$s" = synthetic ]; then env $var=$v timeout -k 10 300 python3 tools/perf_synth.py 1920 4 2>&1 | grep it2 || exit 1
    else env $var=$v timeout -k 10 120 python3 tools/perf_probe.py $s 24 2>&1 | grep -E "it1[23]:" | cut -c30-180 || exit 1; fi
  done
done
