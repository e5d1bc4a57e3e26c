#!/bin/bash
# tools/ab_trees.sh [rounds] [extra env...]: interleaved A/B of two CHECK-OUTS of this repository on one GPU box -- the working tree against
# build_variants/tree_base (git worktree add build_variants/tree_base <commit>; build it there) -- per-kernel HIP-event timings of the
# eager training step (tools/ab_step.py) and the graph-replayed step (bench.py, 200 steps).  Boxes of the pool differ by up to 12 %:
# only numbers of one call compare.
rounds=${1:-3}
root=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq $rounds); do
  for t in build_variants/tree_base .; do
    ( cd $root/$t && echo "[$t]" $(python tools/ab_step.py 2>/dev/null | head -1) \
        $(python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step_ms', round(b['ms_per_step'],4))") )
  done
done
