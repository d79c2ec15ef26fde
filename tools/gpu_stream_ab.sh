#!/bin/bash
# streaming bench A/B over environment settings (BASELINE.json configs[3]: 4096 streams x 1 hop per call).
# Usage: tools/gpu_stream_ab.sh <tag> "ENV=VAL ..." ...   ("-" = defaults); appends ms per call to gpurun_out/<tag>/stream_ab.log
TAG=${1:-sab}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for cfg in "$@"; do
  e="$cfg"; [ "$cfg" = "-" ] && e="DFX_NOP=1"
  echo "== $cfg: $(env $e timeout 300 python tools/bench_stream.py --model df3_ll --streams 4096 --frames-per-call 1 --calls ${CALLS:-1000} 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_call', round(j['ms_per_call'],4))")" | tee -a $OUT/stream_ab.log
done
