#!/bin/bash
# tools/ab_wgrad_blocks.sh: the weight-gradient kernel on single job blocks (AB_BLOCKS) with equal splits, shipped kernel against SATNERF_WGRAD_V2=1 (tools/ab_wgrad8.py per case).
for blocks in "0" "9" "8" "10" "11" "12,13" ""; do
  for v in 0 1; do
    echo -n "blocks=[$blocks] v2=$v: "
    if [ -n "$blocks" ]; then AB_BLOCKS=$blocks AB_SPLITS=18 SATNERF_WGRAD_V2=$v python tools/ab_wgrad8.py 65536 2>/dev/null | tail -1; else SATNERF_WGRAD_V2=$v python tools/ab_wgrad8.py 65536 2>/dev/null | tail -1; fi
  done
done
