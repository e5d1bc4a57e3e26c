#!/bin/bash
# one GPU-box session of round 6.  Outputs under gpurun_out/.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
{
echo "== is the captured oracle step faithful?  graph replay vs the same patched step launched eagerly (same seed, 300 steps), then the unpatched eager step"
python tools/convergence_ensemble.py run --arm ref --run 0 --steps 300 --every 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
CONV_REF_GRAPH_EAGER=1 python tools/convergence_ensemble.py run --arm ref --run 0 --steps 300 --every 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
CONV_REF_GRAPH=0 python tools/convergence_ensemble.py run --arm ref --run 0 --steps 300 --every 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
} > gpurun_out/ens_graph_check.txt
cat gpurun_out/ens_graph_check.txt
timeout 2600 bash tools/run_ensemble.sh 8 20000 2>&1 | tail -4
tail -2 gpurun_out/ens/ref_0.err
