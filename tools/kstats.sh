#!/bin/bash
# tools/kstats.sh <outdir> <variant> ... : rocprofv3 --kernel-trace --stats of tools/ab_fwd.py per library variant (GPU box); prints the
# average duration of the fused forward kernels
root=$(pwd); out=$1; shift
mkdir -p $root/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then lib=""; else lib=$root/build_variants/lib_$v.so; fi
  SATRENDER_LIB=$lib rocprofv3 --kernel-trace --stats -d $root/gpurun_out/$out/$v -o p --output-format csv -- python $root/tools/ab_fwd.py > /dev/null 2>&1
  f=$(find $root/gpurun_out/$out/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -i "fwd" $f | awk -F, '{printf "%s calls=%s avg_ns=%s\n", substr($1,1,70), $2, $4}'
done
