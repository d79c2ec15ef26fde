#!/bin/bash
# dev: bench (main loop only) under different environment settings.  Usage: tools/gpu_envs.sh <outtag> "A=1 B=2" "C=3" ...   ("-" = defaults)
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for v in "$@"; do
  e=""; [ "$v" != "-" ] && e="$v"
  for rep in 1 2; do
    env $e timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --main-only > $OUT/b.log 2>&1
    echo "[$v]: $(tail -1 $OUT/b.log | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3))' 2>/dev/null || tail -2 $OUT/b.log)"
  done
done
