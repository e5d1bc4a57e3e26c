#!/bin/bash
# tools/kstats_bench.sh <outdir> <variant> ...: rocprofv3 kernel stats of a 30-step training bench per library variant ("default" = in-tree)
root=$(pwd); out=$1; shift
mkdir -p $root/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then lib=""; else lib=$root/build_variants/lib_$v.so; fi
  SATRENDER_LIB=$lib rocprofv3 --kernel-trace --stats -d $root/gpurun_out/$out/$v -o p --output-format csv -- python $root/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras > /dev/null 2>&1
  f=$(find $root/gpurun_out/$out/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 - "$f" <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("sr::") or "sr::" in r["Name"][:12]:
        print("  %-60s %6s %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3)); 
PY
done
