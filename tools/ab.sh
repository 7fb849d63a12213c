#!/bin/bash
# usage: tools/ab.sh SCENE VARIANT...   per-class kernel times (settled: the last of fourteen frames) of each build/lib_VARIANT.so ("main" = the in-tree library)
scene=$1; shift
for v in "$@"; do
  echo "== $v"
  if [ "$v" = main ]; then unset MI355RT_LIB; else export MI355RT_LIB=$PWD/build/lib_$v.so; fi
  timeout -k 10 120 python3 tools/perf_probe.py $scene 24 2>&1 | grep -E "it1[23]:" || exit 1
done
