#!/bin/bash
# usage: tools/fuzz_seeds.sh SEED...   GPU vs oracle on single seeds, in-tree library and (if present) build/lib_old.so
for s in "$@"; do
  echo "== seed $s (in-tree)"; timeout -k 10 120 python -u tools/fuzz_parity.py $s $((s+1)) 2>&1 | grep -v amdgpu.ids
  if [ -f build/lib_old.so ]; then echo "== seed $s (old)"; MI355RT_LIB=$PWD/build/lib_old.so timeout -k 10 120 python -u tools/fuzz_parity.py $s $((s+1)) 2>&1 | grep -v amdgpu.ids; fi
done
exit 0
