mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_backward.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_t1.log
for i in 1 2; do
SATNERF_WGRAD_V1=1 python tools/ab_wgrad8.py >> gpurun_out/r04_ab1.log 2>&1
python tools/ab_wgrad8.py >> gpurun_out/r04_ab1.log 2>&1
done
cat gpurun_out/r04_t1.log gpurun_out/r04_ab1.log
