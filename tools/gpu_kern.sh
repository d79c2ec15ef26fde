#!/bin/bash
# per-kernel times of the serialised step + ms per step, for a list of environment settings.  Usage: tools/gpu_kern.sh <tag> "ENV=VAL ..." ...
TAG=${1:-kern}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for cfg in "$@"; do
  e="$cfg"; [ "$cfg" = "-" ] && e="DFX_NOP=1"
  echo "== $cfg: $(env $e DFX_BENCH_SKIP_EXTRAS=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3), {k:v['ms'] for k,v in j.get('kernels',{}).items()})")" | tee -a $OUT/kern.log
done
