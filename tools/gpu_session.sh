#!/bin/bash
# The command list of a GPU-box session (gpurun runs it from the repository root; rewritten per session).  This is the round's FINAL one:
# full test suite, profiles, the default bench line, interleaved r05-vs-r06 A/B (tools/ab_trees.sh needs build_variants/tree_base = a built
# worktree of the previous round's commit).  Outputs under gpurun_out/; what is judged is copied into profiles/.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputest_final.log 2>&1; echo "pytest rc $?" >> gpurun_out/gputest_final.log
tail -3 gpurun_out/gputest_final.log
bash tools/collect_profiles3.sh r06 > gpurun_out/collect_r06.log 2>&1; tail -6 gpurun_out/collect_r06.log
cp gpurun_out/profiles_r06/train_pmc.csv profiles/r06_train_pmc.csv
python bench.py > gpurun_out/r06_bench_train.json 2> gpurun_out/bench_final.err; tail -c 300 gpurun_out/r06_bench_train.json
bash tools/ab_trees.sh 4 > gpurun_out/r06_ab_round.txt 2>&1; cat gpurun_out/r06_ab_round.txt
echo "width 512, eager kernel timings:" $(AB_WIDTH=512 python tools/ab_step.py 2>/dev/null | head -1) > gpurun_out/w512_step.txt; cat gpurun_out/w512_step.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
