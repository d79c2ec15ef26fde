#!/bin/bash
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4q; mkdir -p $OUT; rm -f $OUT/ab.log
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3))")" | tee -a $OUT/ab.log; }
run base DFX_NOP=1
run tail64 DFX_TAIL_WGS=64
run tail96 DFX_TAIL_WGS=96
run tail128 DFX_TAIL_WGS=128
run dfo64 DFX_DFO_WGS=64
run dfo128 DFX_DFO_WGS=128
run dfo256 DFX_DFO_WGS=256
run base2 DFX_NOP=1
run tail96_dfo128 DFX_TAIL_WGS=96 DFX_DFO_WGS=128
run chunks12 DFX_SEQ_CHUNKS=12
run chunks8 DFX_SEQ_CHUNKS=8
run ramp16 DFX_SEQ_RAMP=16
