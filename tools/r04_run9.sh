mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_backward.py tests/test_hip_training.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
SATNERF_WGRAD_V1=1 python tools/ab_step.py 2>&1 | grep -v "amdgpu.ids\|Warning\|Trainer("
AB_TIMING9=1 SATRENDER_LIB=$PWD/build_variants/lib_w9time.so python tools/ab_step.py 2>&1 | grep -v "amdgpu.ids\|Warning\|Trainer("
done
for i in 1 2; do
SATNERF_WGRAD_V1=1 python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V1 step', d['ms_per_step'], d['roofline']['kernel_ms'])"
python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V9 step', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
root=$(pwd); cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d $root/gpurun_out/pmc9_$c -o p --output-format csv -- python $root/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras > /dev/null 2>&1; done
cd $root; python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc9_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad9" in r["Kernel_Name"] or "grad_tail" in r["Kernel_Name"]:
            a = acc[(r["Kernel_Name"][:24], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()): print(k, round(v / n, 1), n)
PY
