#!/bin/bash
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4l; mkdir -p $OUT; rm -f $OUT/ab.log
V=$PWD/tools/dev/_build
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3))")" | tee -a $OUT/ab.log; }
run cur DFX_NOP=1
run prev DFX_LIBRARY=$V/libdfx_prev.so
run w0reg DFX_LIBRARY=$V/libdfx_w0reg.so
run cur2 DFX_NOP=1
run prev2 DFX_LIBRARY=$V/libdfx_prev.so
run w0reg2 DFX_LIBRARY=$V/libdfx_w0reg.so
