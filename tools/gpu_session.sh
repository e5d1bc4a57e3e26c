#!/bin/bash
# one GPU-box session of round 6: tests, tail A/B, default bench line, interleaved r05-vs-r06 A/B.  Outputs under gpurun_out/.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "== tail A/B" > gpurun_out/ab_tail.txt
for r in 1 2; do
  ( cd build_variants/tree_base && echo "[r05 tree]" $(python tools/ab_tail.py 2>&1 | tail -2) ) >> gpurun_out/ab_tail.txt
  echo "[r06 release]" $(python tools/ab_tail.py 2>&1 | tail -2) >> gpurun_out/ab_tail.txt
  echo "[r06 relaxed]" $(SATRENDER_LIB=$PWD/build_variants/lib_tail_relaxed.so python tools/ab_tail.py 2>&1 | tail -2) >> gpurun_out/ab_tail.txt
done
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/gputest.log
tail -3 gpurun_out/gputest.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.json
bash tools/ab_trees.sh 2 > gpurun_out/ab_trees.txt 2>&1; cat gpurun_out/ab_trees.txt
cat gpurun_out/ab_tail.txt
