#!/bin/bash
# tools/ab_nt.sh: A/B builds of the cache policy (`nt` = streaming hint) on the training workspaces' traffic.  The saved state is written by one
# kernel and read by the next one or two (forward -> dX -> weight gradients), 408 MB in all against a 256-MB Infinity Cache:
#   lib_nt_bwdst   dX: dpre stores with the default policy (trunk stream + the C++ stages)
#   lib_nt_bwdall  dX: dpre stores AND the phase loads of the activation workspace with the default policy
#   lib_nt_w9      weight gradients: operand loads WITH nt (each block reads its units once)
#   lib_nt_fwd     training forward: activation stores with the default policy
# select with SATRENDER_LIB=build_variants/lib_<name>.so (tools/ab_step.py, bench.py)
set -e
root=$(cd "$(dirname "$0")/.." && pwd); c=$root/satnerf_amd/csrc; v=$root/build_variants/nt; mkdir -p $v
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment"
link() { name=$1; excl=$2; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $c/build/*.o | grep -v -e "/$excl.o") $v/$name.o -o $root/build_variants/lib_$name.so; echo built lib_$name.so; }
# dX
sed -e '/global_store_dwordx4/s/ nt\\n/\\n/' $c/mlp_bwd_trunk_a1.inc > $v/trunk_st_a1.inc
sed -e 's/ nt\\n/\\n/' $c/mlp_bwd_trunk_a1.inc > $v/trunk_all_a1.inc
$CC -DSR_WS_TEMPORAL -DSR_TRUNK_A1="\"$v/trunk_st_a1.inc\"" -c $c/mlp_bwd.hip -o $v/nt_bwdst.o & 
$CC -DSR_WS_TEMPORAL -DSR_TRUNK_A1="\"$v/trunk_all_a1.inc\"" -c $c/mlp_bwd.hip -o $v/nt_bwdall.o &
# weight gradients
for t in p m px mx; do sed -e '/global_load_dwordx4/s/\\n"$/ nt\\n"/' $c/wgrad9_loop_$t.inc > $v/w9_$t.inc; done
$CC -DSR_W9_P_INC="\"$v/w9_p.inc\"" -DSR_W9_M_INC="\"$v/w9_m.inc\"" -DSR_W9_PX_INC="\"$v/w9_px.inc\"" -DSR_W9_MX_INC="\"$v/w9_mx.inc\"" -c $c/wgrad9.hip -o $v/nt_w9.o &
# forward (saving core, AUXS = 1)
sed -e '/global_store_dwordx4/s/ nt\\n/\\n/' $c/mlp_fwd_core_a1s8.inc > $v/core_a1s8.inc
$CC -DSR_WS_TEMPORAL -DSR_CORE_A1="\"$c/mlp_fwd_core_a1.inc\"" -DSR_CORE_A2="\"$c/mlp_fwd_core_a2.inc\"" -DSR_CORE_A1S8="\"$v/core_a1s8.inc\"" -DSR_CORE_A2S8="\"$c/mlp_fwd_core_a2s8.inc\"" -c $c/mlp_fwd_p1a1.hip -o $v/nt_fwd.o &
wait
link nt_bwdst mlp_bwd; link nt_bwdall mlp_bwd; link nt_w9 wgrad9; link nt_fwd mlp_fwd_p1a1
grep -c " nt" $v/trunk_st_a1.inc $v/trunk_all_a1.inc $v/w9_p.inc $v/core_a1s8.inc $c/mlp_bwd_trunk_a1.inc $c/wgrad9_loop_p.inc $c/mlp_fwd_core_a1s8.inc
