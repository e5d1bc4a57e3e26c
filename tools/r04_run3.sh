mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_backward.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r04_t3.log
out=gpurun_out/r04_ab3.log; : > $out
SATNERF_WGRAD_V1=1 python tools/ab_wgrad8.py 2>&1 | grep -v amdgpu.ids >> $out
for v in w9time w9nomfma w9noload w9nodec w9noread w9nowrite w9nobar; do
  AB_TIMING9=1 SATRENDER_LIB=$PWD/build_variants/lib_$v.so python tools/ab_wgrad8.py 2>&1 | grep -v amdgpu.ids >> $out
done
cat gpurun_out/r04_t3.log $out
