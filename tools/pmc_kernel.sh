#!/bin/bash
# Usage: tools/pmc_kernel.sh <kernel-name-substring> <out-subdir> -- <command...>
# Runs the command under rocprofv3 once per PMC group (counters only, never with other trace domains) and prints per-kernel
# averages for the kernels whose name contains the substring.
set -u
pat=$1; out=$2; shift 3
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $root && rocprofv3 --kernel-trace --pmc $grp -d $root/gpurun_out/$out/g$i -o pmc --output-format csv -- "$@" > $root/gpurun_out/$out.g$i.log 2>&1)
done
cd $root
python - "$pat" gpurun_out/$out <<'PY'
import csv, glob, sys, collections
pat, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            a = acc[(r["Kernel_Name"][:40], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
disp = collections.defaultdict(int)
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:42s} {c:34s} total={v:.4g} rows={n}")
PY
