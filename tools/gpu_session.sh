#!/bin/bash
# one GPU-box session of round 6.  Outputs under gpurun_out/.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
{
echo "== cache policy of the workspace traffic (tools/ab_nt.sh), eager kernel timings"
for r in 1 2 3; do
  for v in "" nt_bwdst nt_bwdall nt_w9 nt_fwd; do
    if [ -z "$v" ]; then echo "default   :" $(python tools/ab_step.py 2>/dev/null | head -1); else echo "$v :" $(SATRENDER_LIB=$PWD/build_variants/lib_$v.so python tools/ab_step.py 2>/dev/null | head -1); fi
  done
done
echo "== the same, graph-replayed step (bench.py, 200 steps)"
for r in 1 2; do
  for v in "" nt_bwdst nt_bwdall nt_w9 nt_fwd; do
    lib=""; [ -n "$v" ] && lib="SATRENDER_LIB=$PWD/build_variants/lib_$v.so"
    echo "${v:-default}:" $(env $lib python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step_ms', round(b['ms_per_step'],4))")
  done
done
} > gpurun_out/ab_r06c.txt 2>&1
cat gpurun_out/ab_r06c.txt
{
python tools/convergence_ensemble.py run --arm ref --run 0 --steps 300 --every 100 2>&1 | grep -v amdgpu.ids | cut -c1-900
CONV_REF_GRAPH=0 python tools/convergence_ensemble.py run --arm ref --run 0 --steps 300 --every 100 2>&1 | grep -v amdgpu.ids | cut -c1-400
} > gpurun_out/ens_graph_check.txt
cat gpurun_out/ens_graph_check.txt
