#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests/test_hip_optin.py -m gpu -q -x > gpurun_out/gputest_optin.log 2>&1; echo "pytest rc $?" >> gpurun_out/gputest_optin.log
tail -4 gpurun_out/gputest_optin.log
/usr/bin/time -v python bench.py > gpurun_out/bench_pg.json 2> gpurun_out/bench_pg.err; grep -E "Elapsed" gpurun_out/bench_pg.err; python -c "
import json; b=json.load(open('gpurun_out/bench_pg.json')); print(b['value'], b['ms_per_step']); print({k: (round(b[k]['value']/1e6,3), round(b[k]['ms_per_step'],3)) for k in ('parity_mode','parity_grade','train_width512','train_width512_sc','train_width512_ds')})"
