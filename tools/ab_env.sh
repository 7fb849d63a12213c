#!/bin/bash
# usage: tools/ab_env.sh SCENE VAR VALUE...   per-class kernel times of the in-tree library with VAR set to each VALUE
scene=$1; var=$2; shift 2
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v timeout -k 10 120 python3 tools/perf_probe.py $scene 24 2>&1 | grep -E "it1[23]:" || exit 1
done
