out=gpurun_out/r04_ab8.log; : > $out
for v in w9time w9nomfma w9noload w9nodec w9noread w9nowrite w9nobar; do
  AB_TIMING9=1 SATRENDER_LIB=$PWD/build_variants/lib_$v.so python tools/ab_wgrad8.py 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
