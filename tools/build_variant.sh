#!/bin/bash
# usage: tools/build_variant.sh NAME [-DFLAG ...]  -> build/lib_NAME.so (kernel tuning experiments; load with MI355RT_LIB)
set -e
name=$1; shift
mkdir -p build
cd raytracer_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -w "$@" -shared -o ../../build/lib_$name.so rt_api.hip rt_bvh.cpp
