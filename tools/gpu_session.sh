#!/bin/bash
# one GPU-box session of round 6.  Outputs under gpurun_out/.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_hip_backward.py tests/test_hip_benched_shape.py tests/test_hip_wgrad9_sizes.py tests/test_hip_emax.py tests/test_hip_step_fusion.py -m gpu -q -x > gpurun_out/gputest_wgrad.log 2>&1; echo "pytest rc $?" >> gpurun_out/gputest_wgrad.log
tail -4 gpurun_out/gputest_wgrad.log
{
echo "== per-block cost of the 4-wave kernel: one block alone on 16 workgroups (128 tiles each), us per launch"
for b in 2 11 12 13; do AB_BLOCKS=$b python tools/ab_wgrad8.py 65536 16 2>&1 | tail -1; done
echo "== the same with full streams everywhere (SATNERF_WGRAD_THIN=0)"
for b in 12 13; do SATNERF_WGRAD_THIN=0 AB_BLOCKS=$b python tools/ab_wgrad8.py 65536 16 2>&1 | tail -1; done
echo "== whole kernel, stand-alone"
for r in 1 2; do
  SATNERF_WGRAD_THIN=0 python tools/ab_wgrad8.py 2>&1 | tail -1
  python tools/ab_wgrad8.py 2>&1 | tail -1
  SATNERF_WGRAD_THIN_COST=0.4,0.3 python tools/ab_wgrad8.py 2>&1 | tail -1
  SATNERF_WGRAD_THIN_COST=0.6,0.5 python tools/ab_wgrad8.py 2>&1 | tail -1
done
echo "== is the kernel's HBM traffic free?  lib_w9_l2: every tile re-reads the slice's first tile (operands from the L2)"
for r in 1 2 3; do
  python tools/ab_wgrad8.py 2>&1 | tail -1
  SATRENDER_LIB=$PWD/build_variants/lib_w9_l2.so python tools/ab_wgrad8.py 2>&1 | tail -1
done
echo "== in the step (eager kernel timings)"
for r in 1 2 3; do
  echo "thin off:" $(SATNERF_WGRAD_THIN=0 python tools/ab_step.py 2>&1 | head -1)
  echo "thin on :" $(python tools/ab_step.py 2>&1 | head -1)
done
echo "== graph-replayed step (bench.py, 200 steps): thin off / on / stream-K at 256"
for r in 1 2; do
  for e in "SATNERF_WGRAD_THIN=0" "SATNERF_WGRAD_THIN=1" "SATNERF_WGRAD_STREAMK=1"; do
    echo "$e:" $(env $e python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step_ms', round(b['ms_per_step'],4), {k: round(v['ms']*1e3,1) for k,v in b['roofline']['all_kernels'].items()})")
  done
done
echo "== training forward: save-stream ablations (mlp_fwd column; results of the ablated builds are wrong by construction)"
for r in 1 2; do
  for v in "" fwd_nostore fwd_noenc fwd_nostorenoenc; do
    if [ -z "$v" ]; then echo "default          :" $(python tools/ab_step.py 2>&1 | head -1); else echo "$v :" $(SATRENDER_LIB=$PWD/build_variants/lib_$v.so python tools/ab_step.py 2>&1 | head -1); fi
  done
done
} > gpurun_out/ab_r06.txt 2>&1
cat gpurun_out/ab_r06.txt
{
echo "== fp32 oracle arm: hipGraph-captured step vs eager, same seed, 300 steps"
python tools/convergence_ensemble.py run --arm ref --run 0 --steps 300 --every 100 2>&1 | tail -4
CONV_REF_GRAPH=0 python tools/convergence_ensemble.py run --arm ref --run 0 --steps 300 --every 100 2>&1 | tail -4
} > gpurun_out/ens_graph_check.txt 2>&1
cat gpurun_out/ens_graph_check.txt
