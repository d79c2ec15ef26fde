#!/bin/bash
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4t; mkdir -p $OUT; rm -f $OUT/ab.log
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3))")" | tee -a $OUT/ab.log; }
for rep in 1 2; do
run default DFX_NOP=1
run c14r0 DFX_SEQ_CHUNKS=14 DFX_SEQ_RAMP=0
run c12r0 DFX_SEQ_CHUNKS=12 DFX_SEQ_RAMP=0
run c16r0 DFX_SEQ_CHUNKS=16 DFX_SEQ_RAMP=0
run c13r0 DFX_SEQ_CHUNKS=13 DFX_SEQ_RAMP=0
run c12r48 DFX_SEQ_CHUNKS=12 DFX_SEQ_RAMP=48
done
