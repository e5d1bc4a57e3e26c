#!/bin/bash
# one GPU-box session of round 6.  Outputs under gpurun_out/.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/gputest.log
tail -5 gpurun_out/gputest.log
timeout 300 python tools/convergence_ensemble.py short --k 2 --steps 60 > gpurun_out/ens_short.json 2> gpurun_out/ens_short.err; tail -c 400 gpurun_out/ens_short.json; tail -3 gpurun_out/ens_short.err
timeout 1500 bash tools/run_ensemble.sh 8 20000 2>&1 | tail -3
