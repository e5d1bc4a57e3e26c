#!/bin/bash
# GPU-box session: (i) the layout probe, (ii) a finer trace of the runs' ends -- MAE every 250 steps over the last 5,000 -- for 8 HIP and 3 fp32 trainings of the ensemble
# study (same seeds as runs 0.. of the study: the same trajectories), to see whether the end-of-run excursions two of the 32 HIP runs show belong to the arithmetic.
mkdir -p gpurun_out/trace
export PYTHONDONTWRITEBYTECODE=1
./build_variants/probe_layout > gpurun_out/probe_layout.txt 2>&1; cat gpurun_out/probe_layout.txt
for r in 8 23 0 1 2 3 4 5; do python tools/convergence_ensemble.py run --arm hip --run $r --steps 20000 --every 250 --ncp 21 > gpurun_out/trace/hip_$r.json 2> gpurun_out/trace/hip_$r.err; done
for r in 0 1 2; do python tools/convergence_ensemble.py run --arm ref --run $r --steps 20000 --every 250 --ncp 21 > gpurun_out/trace/ref_$r.json 2> gpurun_out/trace/ref_$r.err; tail -1 gpurun_out/trace/ref_$r.err; done
