#!/bin/bash
mkdir -p gpurun_out/ens16
export PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
for r in 8 9 10 11 12 13 14; do
  python tools/convergence_ensemble.py run --arm ref --run $r --steps 20000 > gpurun_out/ens16/ref_$r.json 2> gpurun_out/ens16/ref_$r.err
  tail -1 gpurun_out/ens16/ref_$r.err
done
