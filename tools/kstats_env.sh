#!/bin/bash
# tools/kstats_env.sh <outdir> <ENV=a> <ENV=b> ...: rocprofv3 --kernel-trace --stats of a 30-step training bench per environment setting
# (A/B switches); prints the average duration of every sr:: kernel.  Each run is bounded by `timeout`.
root=$(pwd); out=$1; shift
mkdir -p $root/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  d=$root/gpurun_out/$out/$(echo $e | tr '= ' '__')
  env $e timeout 300 rocprofv3 --kernel-trace --stats -d $d -o p --output-format csv -- python $root/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras > /dev/null 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $e"; python3 - "$f" <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    if "sr::" in r["Name"][:12]:
        print("  %-70s %6s %8.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3)); 
PY
done
