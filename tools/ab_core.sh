#!/bin/bash
# tools/ab_core.sh <name> [generator flags ...]: build build_variants/lib_<name>.so whose forward core (bf16, AUXS = 1 and 2) is
# generated with the given csrc/gen/fwd_core.py flags (--ablate nodma,... --PF n --GROUP n --FILL n); select with SATRENDER_LIB=...
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/build_variants/core_$name
mkdir -p $d
defs=""
if [ "$1" = "-DSR_CORE_TIMING" ]; then defs="-DSR_CORE_TIMING"; shift; fi
python3 $root/satnerf_amd/csrc/gen/fwd_core.py --out $d "$@" > $d/gen.log
for tu in mlp_fwd_p1a1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment $defs -DSR_CORE_A1="\"$d/mlp_fwd_core_a1.inc\"" -DSR_CORE_A2="\"$d/mlp_fwd_core_a2.inc\"" -DSR_CORE_A1S8="\"$d/mlp_fwd_core_a1s8.inc\"" -DSR_CORE_A2S8="\"$d/mlp_fwd_core_a2s8.inc\"" \
    -c $root/satnerf_amd/csrc/$tu.hip -o $d/$tu.o
done
others=$(ls $root/satnerf_amd/csrc/build/*.o | grep -v "/mlp_fwd_p1a1.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $d/mlp_fwd_p1a1.o -o $root/build_variants/lib_$name.so
echo built build_variants/lib_$name.so
