#!/bin/bash
# tools/run_ensemble.sh [K] [steps]: the g1 ensemble study (tools/convergence_ensemble.py) on one GPU box -- the K fp32-oracle trainings in two
# lanes side by side (each step one hipGraph replay of ~600 small torch kernels), the K HIP trainings in a third lane; results under
# gpurun_out/ens/, combined into gpurun_out/r06_convergence_ensemble.json.
K=${1:-8}; STEPS=${2:-20000}
mkdir -p gpurun_out/ens; rm -f gpurun_out/ens/*
export PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
lane() { arm=$1; shift; for r in "$@"; do python tools/convergence_ensemble.py run --arm $arm --run $r --steps $STEPS > gpurun_out/ens/${arm}_$r.json 2> gpurun_out/ens/${arm}_$r.err; done; }
even=$(seq 0 2 $((K-1))); odd=$(seq 1 2 $((K-1)))
lane ref $even & p1=$!
lane ref $odd & p2=$!
lane hip $(seq 0 $((K-1))) & p3=$!
wait $p1 $p2 $p3
python tools/convergence_ensemble.py combine gpurun_out/ens/ref_*.json gpurun_out/ens/hip_*.json > gpurun_out/r06_convergence_ensemble.json
python -c "import json; d=json.load(open('gpurun_out/r06_convergence_ensemble.json')); print(json.dumps(d['summary']))"
grep -h "capture failed" gpurun_out/ens/*.err | head -3
