#!/bin/bash
# quick GPU check: df_apply + enhance parity, bench (no CPU baseline).  Usage: tools/gpu_quick.sh <tag> [extra env assignments for the bench]
TAG=${1:-quick}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_df_apply.py tests/test_enhance.py tests/test_dsp_kernels.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
env "$@" timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('ms_per_step', round(j['ms_per_step'],3), 'roofline', j['roofline'])
print({k:v['ms'] for k,v in j['kernels'].items()})
"
