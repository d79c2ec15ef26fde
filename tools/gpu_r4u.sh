#!/bin/bash
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4u; mkdir -p $OUT; rm -f $OUT/ab.log
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3))")" | tee -a $OUT/ab.log; }
run default DFX_NOP=1
run grain8 DFX_FRONT_GRAIN=1,8
run grain32 DFX_FRONT_GRAIN=1,32
run grain64 DFX_FRONT_GRAIN=1,64
run phase_early DFX_PHASE_LATE=0
run ahead DFX_ENQUEUE_AHEAD=1
run projrt1 DFX_PROJ_RT=1
run default2 DFX_NOP=1
run gruseq0 DFX_GRU_SEQ=0
run syn2 DFX_SYN_SEGS=2
run syn4 DFX_SYN_SEGS=4
(timeout 200 python tools/dev/seq_trace.py 2>&1 | grep -v amdgpu | cut -c1-250)
