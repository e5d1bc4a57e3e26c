#!/bin/bash
# tools/ab_wgrad9_abl.sh: build build_variants/lib_w9<tag>.so = the 4-wave weight-gradient kernel (csrc/wgrad9.hip) with in-kernel stamps
# (w9time: -DSR_W9_TIMING, results correct) or with one resource removed from its generated slice loop (results are WRONG; timing only,
# tools/ab_wgrad8.py): nomfma, noload (no global loads / vmcnt waits), nodec (no decode arithmetic), noread (no transposed operand
# reads), nowrite (no LDS writes), nobar (no publish-counter check), and combinations given as extra arguments "a,b".
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
gen=$root/satnerf_amd/csrc/gen/wgrad9_loop.py
inc=$root/build_variants/inc9
mkdir -p $inc
$root/tools/build_variant.sh w9time wgrad9.hip -DSR_W9_TIMING
for abl in nomfma noload nodec noread nowrite nobar "$@"; do
  tag=$(echo $abl | tr -d ,)
  python $gen $inc _$tag $abl > /dev/null
  $root/tools/build_variant.sh w9$tag wgrad9.hip -I$inc -DSR_W9_TIMING "-DSR_W9_P_INC=\"wgrad9_loop_p_$tag.inc\"" "-DSR_W9_M_INC=\"wgrad9_loop_m_$tag.inc\"" \
    "-DSR_W9_PX_INC=\"wgrad9_loop_px_$tag.inc\"" "-DSR_W9_MX_INC=\"wgrad9_loop_mx_$tag.inc\""
done
