#!/bin/bash
# tools/ab_wgrad_abl.sh: build build_variants/lib_w8abl{2,...,8}.so = the weight-gradient kernel with one resource removed (results are
# wrong; timing only): 2 = decode arithmetic without its LDS writes, 3 = no decode, 4 = no MFMAs, 5 = no LDS-DMA, 6 = no workgroup barrier, 7 = no transposed operand reads, 8 = LDS-DMA and rendezvous only.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build_variants
for n in 2 3 4 5 6 7 8; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -DSR_W8_ABL=$n -c $root/satnerf_amd/csrc/wgrad8.hip -o $root/build_variants/wgrad8_abl$n.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $root/satnerf_amd/csrc/build/*.o | grep -v "/wgrad8.o") $root/build_variants/wgrad8_abl$n.o -o $root/build_variants/lib_w8abl$n.so
    echo built lib_w8abl$n.so ) &
done
wait
