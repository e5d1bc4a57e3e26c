#!/bin/bash
# Interleaved A/B of bench.py's training step under environment toggles, on ONE box (step times differ by up to 12 % between boxes of the
# pool): tools/ab_env.sh ROUNDS "A_ENV=.. B_ENV=.." "C_ENV=.." ...  -- every variant is an env string ("" = defaults); prints ms_per_step
# per run.  Example (r05: the four-launch step against the r04 launch sequence):
#   tools/ab_env.sh 3 "" "SATNERF_TAIL_PACK=0 SATNERF_GATHER_IN_FWD=0"
rounds=$1; shift
for r in $(seq $rounds); do
  for v in "$@"; do
    ms=$(env $v python bench.py --no-cpu-baseline --no-extras ${AB_ARGS:-} 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "[$r] {${v:-defaults}} $ms"
  done
done
