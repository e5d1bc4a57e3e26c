mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_backward.py tests/test_hip_training.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r04_t4.log
out=gpurun_out/r04_ab4.log; : > $out
SATNERF_WGRAD_V1=1 python tools/ab_wgrad8.py 2>&1 | grep -v amdgpu.ids >> $out
python tools/ab_wgrad8.py 2>&1 | grep -v amdgpu.ids >> $out
AB_TIMING9=1 SATRENDER_LIB=$PWD/build_variants/lib_w9time.so python tools/ab_wgrad8.py 2>&1 | grep -v amdgpu.ids >> $out
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench4.json 2> gpurun_out/r04_bench4.err
cat gpurun_out/r04_t4.log $out; cat gpurun_out/r04_bench4.json | head -c 1500
