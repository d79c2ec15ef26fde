#!/bin/bash
OUT=gpurun_out/${1:-r02k}
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
sum() { tail -1 $1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('ms_per_step', round(j['ms_per_step'],3), 'dfa in-loop ms', j['roofline']['avg_launch_ms'], 'frac', j['roofline']['frac'], 'serial dfa', j['kernels']['dfx_k_df_apply']['ms'])
"; }
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^$" | head -30
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_auto.log 2>&1; echo "auto:"; sum $OUT/bench_auto.log
rocm-smi --setperflevel high 2>&1 | tail -3
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^$" | head -30
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_high.log 2>&1; echo "high:"; sum $OUT/bench_high.log
timeout 600 python bench.py --steps 60 --warmup 3 --no-cpu-baseline > $OUT/bench_high60.log 2>&1; echo "high, 60 steps:"; sum $OUT/bench_high60.log
rocm-smi --setperflevel auto 2>&1 | tail -2
timeout 600 python bench.py --steps 60 --warmup 3 --no-cpu-baseline > $OUT/bench_auto60.log 2>&1; echo "auto, 60 steps:"; sum $OUT/bench_auto60.log
rocm-smi --showpower --showtemp 2>&1 | grep -v "^$" | head -20
