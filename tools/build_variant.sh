#!/bin/bash
# tools/build_variant.sh <name> <source.hip> [-DMACRO ...]: link build_variants/lib_<name>.so = the in-tree objects with
# <source.hip> recompiled with the extra flags (A/B experiments; select with SATRENDER_LIB=...).
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build_variants/obj_$name
obj=$root/build_variants/obj_$name/$(basename ${src%.hip}).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment "$@" -c $root/satnerf_amd/csrc/$src -o $obj
others=$(ls $root/satnerf_amd/csrc/build/*.o | grep -v "/$(basename ${src%.hip}).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $obj -o $root/build_variants/lib_$name.so
echo built build_variants/lib_$name.so
