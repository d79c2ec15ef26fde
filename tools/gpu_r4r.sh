#!/bin/bash
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4r; mkdir -p $OUT; rm -f $OUT/ab.log
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3))")" | tee -a $OUT/ab.log; }
for rep in 1 2; do
run c10 DFX_SEQ_CHUNKS=10
run c12 DFX_SEQ_CHUNKS=12
run c14 DFX_SEQ_CHUNKS=14
run c16 DFX_SEQ_CHUNKS=16
run c12r16 DFX_SEQ_CHUNKS=12 DFX_SEQ_RAMP=16
run c14r48 DFX_SEQ_CHUNKS=14 DFX_SEQ_RAMP=48
done
