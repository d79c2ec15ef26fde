#!/bin/bash
# bench A/B over environment settings.  Usage: tools/gpu_ab.sh <tag> "ENV=VAL ..." "ENV=VAL ..." ...   ("-" = defaults)
# prints ms per step of the timed loop (--main-only: no extras) and the in-loop DF-apply time; appends to gpurun_out/<tag>/ab.log
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for cfg in "$@"; do
  e="$cfg"; [ "$cfg" = "-" ] && e="DFX_NOP=1"
  echo "== $cfg: $(env DFX_BENCH_PROF_ANALYSIS=1 $e timeout 300 python bench.py --steps ${STEPS:-20} --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3), 'dfa', round(j.get('dfa_in_loop_ms') or 0,4), 'finish', round(j.get('finish_in_loop_ms') or 0,4), 'analysis', round(j.get('analysis_in_loop_ms') or 0,4))")" | tee -a $OUT/ab.log
done
