#!/bin/bash
# tools/ab_core_run.sh <rounds> <variant> ...: interleaved rounds of tools/ab_fwd.py over build_variants/lib_<variant>.so ("default" = in-tree)
rounds=$1; shift
for r in $(seq $rounds); do
  for v in "$@"; do
    if [ "$v" = default ]; then lib=""; else lib=$PWD/build_variants/lib_$v.so; fi
    if [ "$v" = v1 ]; then SATNERF_FWD_V1=1 timeout 120 python tools/ab_fwd.py 2>/dev/null | sed "s/^default/v1/"; else SATRENDER_LIB=$lib timeout 120 python tools/ab_fwd.py 2>/dev/null; fi
  done
done
