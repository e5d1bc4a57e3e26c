#!/bin/bash
# tools/ab_bwd_abl.sh: build build_variants/lib_bwdabl{1,2,3}.so = the dX kernel (width 256) with part of its workspace traffic removed
# (results wrong; timing only, tools/ab_step.py): 1 = no dpre stores, 2 = no phase loads, 3 = neither.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build_variants
build() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment $2 -c $root/satnerf_amd/csrc/mlp_bwd.hip -o $root/build_variants/mlp_bwd_abl$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $root/satnerf_amd/csrc/build/*.o | grep -v "/mlp_bwd.o") $root/build_variants/mlp_bwd_abl$1.o -o $root/build_variants/lib_bwdabl$1.so
  echo built lib_bwdabl$1.so
}
build 1 "-DSR_ABL_NO_WS_STORE" &
build 2 "-DSR_ABL_NO_PHASE_LOADS" &
build 3 "-DSR_ABL_NO_WS_STORE -DSR_ABL_NO_PHASE_LOADS" &
wait
