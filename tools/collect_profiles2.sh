#!/bin/bash
# Run on the GPU box (via gpurun): regenerates the round's profile artefacts under gpurun_out/profiles_${TAG}/.
# Counters are collected in their own passes with --kernel-trace only (never combined with other trace domains).
set -u
TAG=${1:-r04}; root=$(pwd); out=$root/gpurun_out/profiles_${TAG}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $out/train_stats -o t --output-format csv -- $B > $out/train_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/fwd_stats -o t --output-format csv -- $B --phase forward > $out/fwd_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$tag -o p --output-format csv -- $B > $out/pmc_$tag.log 2>&1
done
cd $root
if [ "${FULL:-0}" = "1" ]; then
python bench.py --steps 200 --warmup 50 > $out/bench_train.json 2> $out/bench_train.err
python bench.py --steps 200 --warmup 50 --phase forward > $out/bench_forward.json 2> $out/bench_forward.err
python bench.py --steps 100 --warmup 20 --phase forward --rays 4096 --no-cpu-baseline > $out/bench_forward_4096rays.json 2> /dev/null
python bench.py --steps 100 --warmup 20 --mode bf16x3 --no-cpu-baseline > $out/bench_train_bf16x3.json 2> /dev/null
python tools/bench_layer_path.py 512 > $out/layer_path_512.txt 2>&1
fi
TAG=$TAG python - <<'PY'
import csv, glob, collections
import os
out = "gpurun_out/profiles_" + os.environ.get("TAG", "r04")
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not k.startswith("sr::") and "sr::" not in k: continue
        a = acc[(k.split("(")[0][:60], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
disp = collections.Counter()
with open(out + "/train_pmc.csv", "w") as fo:
    fo.write("kernel,counter,per_launch_value,launches\n")
    for (k, c), (v, n) in sorted(acc.items()):
        fo.write(f"{k},{c},{v / n * (1 if not c.endswith('_SIZE') else 1):.1f},{n}\n")
print(open(out + "/train_pmc.csv").read()[:3000])
PY
