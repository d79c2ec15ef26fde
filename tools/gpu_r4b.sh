#!/bin/bash
# round 4, call B: df_convp inside the phase
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4b; mkdir -p $OUT; rm -f $OUT/ab.log
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3), 'finish', round(j.get('finish_in_loop_ms') or 0,4))")" | tee -a $OUT/ab.log; }
timeout 600 python -m pytest tests/test_enhance.py -x -q -k "variants_agree and (CONVP or GRU_SEQ)" 2>&1 | tail -3
run base DFX_NOP=1
run convp_phase DFX_CONVP_PHASE=1
run convp_phase_grain4 DFX_CONVP_PHASE=1 DFX_FRONT_GRAIN=1,4
run base2 DFX_NOP=1
run convp_phase2 DFX_CONVP_PHASE=1
(DFX_CONVP_PHASE=1 timeout 200 python tools/dev/seq_trace.py 2>&1 | grep -v amdgpu > $OUT/seq_trace_convp_phase.txt)
bash tools/gpu_trace.sh r4b_tl DFX_CONVP_PHASE=1
cut -c1-250 $OUT/seq_trace_convp_phase.txt
head -40 gpurun_out/r4b_tl/timeline.txt
