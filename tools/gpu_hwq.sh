#!/bin/bash
OUT=gpurun_out/${1:-r02v}; shift
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for v in "$@"; do
  env $v timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/b.log 2>&1
  echo "$v: $(tail -1 $OUT/b.log | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('ms_per_step', round(j['ms_per_step'],3), 'dfa in-loop', j['roofline']['avg_launch_ms'], 'frac', j['roofline']['frac'])")"
done
