#!/bin/bash
# tools/ab_env.sh <rounds> <ENV=a> <ENV=b> ...: interleaved bench.py runs (200 graph-replayed training steps) of ONE tree under different
# environment settings (A/B switches such as SATNERF_TRAIN_FUSED=0): step time and the eager per-kernel timings, one line per run.
rounds=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq $rounds); do
  for e in "$@"; do
    ( cd $root && echo "[$e]" $(env $e python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step_ms', round(b['ms_per_step'],4), {k: round(v['ms']*1e3,1) for k,v in b['roofline']['all_kernels'].items()})") )
  done
done
