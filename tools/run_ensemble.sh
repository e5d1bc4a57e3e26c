#!/bin/bash
# tools/run_ensemble.sh [K] [steps]: the g1 ensemble study (tools/convergence_ensemble.py) on one GPU box, one training after the other (two
# processes on one GPU take turns at the granularity of whole kernels: side by side they ran 4-8 x slower each); results under gpurun_out/ens/,
# combined into gpurun_out/r06_convergence_ensemble.json.
K=${1:-8}; STEPS=${2:-20000}
mkdir -p gpurun_out/ens; rm -f gpurun_out/ens/*
export PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
for r in $(seq 0 $((K-1))); do
  python tools/convergence_ensemble.py run --arm hip --run $r --steps $STEPS > gpurun_out/ens/hip_$r.json 2> gpurun_out/ens/hip_$r.err
done
for r in $(seq 0 $((K-1))); do
  python tools/convergence_ensemble.py run --arm ref --run $r --steps $STEPS > gpurun_out/ens/ref_$r.json 2> gpurun_out/ens/ref_$r.err
  tail -1 gpurun_out/ens/ref_$r.err
done
python tools/convergence_ensemble.py combine gpurun_out/ens/ref_*.json gpurun_out/ens/hip_*.json > gpurun_out/r06_convergence_ensemble.json
python -c "import json; d=json.load(open('gpurun_out/r06_convergence_ensemble.json')); print(json.dumps(d['summary']))"
grep -h "capture failed" gpurun_out/ens/*.err | head -3
