#!/bin/bash
# tests of the kernels touched in this step + a bench run.  Usage: tools/gpu_r4.sh <tag> [env assignments for the bench]
TAG=${1:-r4}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_dfnet_kernels.py tests/test_enhance.py tests/test_fp16_range.py tests/test_config_options.py tests/test_streaming.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
env "$@" timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --main-only > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench.log | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('ms_per_step', round(j['ms_per_step'],3), 'roofline', j['roofline']['frac'], j['roofline'].get('standalone',{}).get('frac'))
print({k:v['ms'] for k,v in j['kernels'].items()})
"
