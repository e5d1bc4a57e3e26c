#!/bin/bash
# tools/ab_core3.sh <name> [-DSR_CORE_TIMING] [generator flags ...]: build build_variants/lib_<name>.so whose PARITY-mode forward core
# (AUXS = 1) is generated with the given csrc/gen/fwd_core3.py flags (--ablate nodma,nobarrier,noepi --PF n --FILL n); select it with
# SATRENDER_LIB=... python tools/ab_x3.py
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/build_variants/core3_$name
mkdir -p $d
defs=""
if [ "$1" = "-DSR_CORE_TIMING" ]; then defs="-DSR_CORE_TIMING"; shift; fi
python3 $root/satnerf_amd/csrc/gen/fwd_core3.py --out $d "$@" > $d/gen.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment $defs -DSR_CORE3_A1="\"$d/mlp_fwd3_core_a1.inc\"" -DSR_CORE3_A2="\"$d/mlp_fwd3_core_a2.inc\"" \
  -c $root/satnerf_amd/csrc/mlp_fwd_p3a1.hip -o $d/mlp_fwd_p3a1.o
others=$(ls $root/satnerf_amd/csrc/build/*.o | grep -v "/mlp_fwd_p3a1.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $d/mlp_fwd_p3a1.o -o $root/build_variants/lib_$name.so
echo built build_variants/lib_$name.so
