#!/bin/bash
# one GPU-box session of round 6: the g1 ensemble study, sequential.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 3300 bash tools/run_ensemble.sh 8 20000 2>&1 | tail -12
