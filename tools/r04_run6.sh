mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_backward.py tests/test_hip_training.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r04_t6.log
out=gpurun_out/r04_ab6.log; : > $out
for i in 1 2; do
SATNERF_WGRAD_V1=1 python tools/ab_step.py 2>&1 | grep -v "amdgpu.ids\|Warning\|Trainer(" >> $out
AB_TIMING9=1 SATRENDER_LIB=$PWD/build_variants/lib_w9time.so python tools/ab_step.py 2>&1 | grep -v "amdgpu.ids\|Warning\|Trainer(" >> $out
done
for i in 1 2; do
SATNERF_WGRAD_V1=1 python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V1 step', d['ms_per_step'], d['roofline']['kernel_ms'])" >> $out
python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V9 step', d['ms_per_step'], d['roofline']['kernel_ms'])" >> $out
done
cat gpurun_out/r04_t6.log $out
