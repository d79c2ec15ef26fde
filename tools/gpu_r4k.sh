#!/bin/bash
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4k; mkdir -p $OUT; rm -f $OUT/ab.log
V=$PWD/tools/dev/_build
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3))")" | tee -a $OUT/ab.log; }
run cur DFX_NOP=1
run prev DFX_LIBRARY=$V/libdfx_prev.so
run cur2 DFX_NOP=1
run prev2 DFX_LIBRARY=$V/libdfx_prev.so
run skip_erb_tail DFX_DEV_SKIP=1
run skip_df_tail DFX_DEV_SKIP=2
run skip_both DFX_DEV_SKIP=3
run skip_convp DFX_DEV_SKIP=8
run cur3 DFX_NOP=1
