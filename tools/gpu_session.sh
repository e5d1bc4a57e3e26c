#!/bin/bash
# one GPU-box session of round 6.  Outputs under gpurun_out/.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
{
echo "== stores and the counted vmcnt waits: fwd_storesame (ablation: every store of a wave on one unit), fwd_phasefirst / bwd_storelate (correct results: stores moved away from the waits)"
for r in 1 2 3; do
  for v in "" fwd_storesame fwd_phasefirst bwd_storelate; do
    if [ -z "$v" ]; then echo "default   :" $(python tools/ab_step.py 2>/dev/null | head -1); else echo "$v :" $(SATRENDER_LIB=$PWD/build_variants/lib_$v.so python tools/ab_step.py 2>/dev/null | head -1); fi
  done
done
echo "== graph-replayed step (bench.py, 200 steps)"
for r in 1 2 3; do
  for v in "" fwd_phasefirst bwd_storelate; do
    lib=""; [ -n "$v" ] && lib="SATRENDER_LIB=$PWD/build_variants/lib_$v.so"
    echo "${v:-default}:" $(env $lib python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step_ms', round(b['ms_per_step'],4), {k: round(v['ms']*1e3,1) for k,v in b['roofline']['all_kernels'].items()})")
  done
done
echo "== correctness of the two reordered streams: the gradient goldens and the one-launch-forward bit-identity test through each library"
for v in fwd_phasefirst bwd_storelate; do
  SATRENDER_LIB=$PWD/build_variants/lib_$v.so timeout 600 python -m pytest tests/test_hip_backward.py tests/test_hip_training.py -m gpu -q -x -k "golden or fused_training_forward or bit" 2>&1 | tail -1
done
} > gpurun_out/ab_r06d.txt 2>&1
cat gpurun_out/ab_r06d.txt
./build_variants/probe_war > gpurun_out/probe_war.txt 2>&1; cat gpurun_out/probe_war.txt
{
echo "== is the captured oracle step faithful?  graph replay vs the same patched step launched eagerly (same seed, 300 steps)"
python tools/convergence_ensemble.py run --arm ref --run 0 --steps 300 --every 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
CONV_REF_GRAPH_EAGER=1 python tools/convergence_ensemble.py run --arm ref --run 0 --steps 300 --every 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
} > gpurun_out/ens_graph_check.txt
cat gpurun_out/ens_graph_check.txt
