root=$(pwd); out=$root/gpurun_out/profiles_r04w; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/w512 -o t --output-format csv -- python $root/tools/bench_width512.py > $out/w512.log 2>&1
cd $root
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/profiles_r04w/w512/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:10]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
