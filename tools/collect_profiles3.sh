#!/bin/bash
# Run on the GPU box (via gpurun): regenerates the round's profile artefacts under gpurun_out/profiles_${TAG}/ (copy what is judged into
# profiles/).  Counters are collected in their own passes with --kernel-trace only (never combined with other trace domains); every pass
# is bounded by `timeout`.  r05: every PMC pass also records the kernels' durations IN THAT PASS (DURATION_US_IN_PMC_PASS rows), and
# GRBM_GUI_ACTIVE rides with SQ_VALU_MFMA_BUSY_CYCLES so that the MFMA-busy fraction is a ratio of two counters of one pass.
set -u
TAG=${1:-r05}; root=$(pwd); out=$root/gpurun_out/profiles_${TAG}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/train_stats -o t --output-format csv -- $B > $out/train_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $out/fwd_stats -o t --output-format csv -- $B --phase forward > $out/fwd_stats.log 2>&1
if [ "${W512:-1}" = "1" ]; then
timeout 300 rocprofv3 --kernel-trace --stats -d $out/w512_stats -o t --output-format csv -- env AB_WIDTH=512 python $root/tools/ab_step.py > $out/w512_stats.log 2>&1
fi
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$tag -o p --output-format csv -- $B > $out/pmc_$tag.log 2>&1
done
cd $root
TAG=$TAG python - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/profiles_" + os.environ.get("TAG", "r05")
acc = collections.defaultdict(lambda: [0.0, 0])
short = lambda k: k.split("(")[0][:60]
for d in sorted(glob.glob(out + "/pmc_*/")):
    dur = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "sr::" not in k: continue
            x = dur[short(k)]
            x[0] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3; x[1] += 1
    seen = set()
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "sr::" not in k: continue
            a = acc[(short(k), r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
            if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": seen.add(short(k))
    for k in seen:   # the duration of the kernel in the pass that counted its MFMA-busy cycles
        if dur[k][1]: acc[(k, "DURATION_US_IN_PMC_PASS")] = [dur[k][0], dur[k][1]]
with open(out + "/train_pmc.csv", "w") as fo:
    fo.write("kernel,counter,per_launch_value,launches\n")
    for (k, c), (v, n) in sorted(acc.items()):
        fo.write(f"{k},{c},{v / n:.1f},{n}\n")
rows = {(k, c): v / n for (k, c), (v, n) in acc.items()}
for k in sorted({k for k, _ in rows}):
    f, w = rows.get((k, "FETCH_SIZE")), rows.get((k, "WRITE_SIZE"))
    b, g = rows.get((k, "SQ_VALU_MFMA_BUSY_CYCLES")), rows.get((k, "GRBM_GUI_ACTIVE"))
    print(f"{k[:58]:58s} traffic {((2 * f + w) / 1024 if f is not None and w is not None else float('nan')):8.1f} MB  mfma_busy {(b / 1024 / (g / 8) if b and g else float('nan')):.3f}  "
          f"dur_in_pmc {rows.get((k, 'DURATION_US_IN_PMC_PASS'), float('nan')):7.1f} us")
PY
for s in train fwd w512; do f=$(find $out/${s}_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${s}_kernel_stats.csv; done
ls $out | head -30
