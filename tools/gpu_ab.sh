#!/bin/bash
# bench A/B over environment settings.  Usage: tools/gpu_ab.sh <tag> "ENV=VAL ..." "ENV=VAL ..." ...   ("-" = defaults)
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for cfg in "$@"; do
  e="$cfg"; [ "$cfg" = "-" ] && e="DFX_NOP=1"
  echo "== $cfg: $(env $e timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3), 'dfa', round(j.get('dfa_in_loop_ms',0),4), {k:v['ms'] for k,v in j.get('kernels',{}).items() if 'proj' in k or 'ggemm' in k})")" | tee -a $OUT/ab.log
done
