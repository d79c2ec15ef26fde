#!/bin/bash
OUT=gpurun_out/${1:-r02n}
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
sum() { tail -1 $1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('ms_per_step', round(j['ms_per_step'],3), 'dfa in-loop ms', j['roofline']['avg_launch_ms'], 'serial dfa', j['kernels']['dfx_k_df_apply']['ms'], 'serial syn', j['kernels']['dfx_k_synthesis']['ms'])
"; }
for sp in "" "1,5" "1,20" "1,45" "1,100" "256,45" "2048,45"; do
  DFX_DEV_SPIN=$sp timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/b.log 2>&1; echo "spin '$sp':"; sum $OUT/b.log
done
