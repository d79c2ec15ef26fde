#!/bin/bash
# round 4, call F: df_convp on a capped number of CUs; exact mode with shorter chunks + df_convp behind df_conv1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4f; mkdir -p $OUT; rm -f $OUT/ab.log
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3), j.get('gru_phase_form'), j.get('exact_fp32'))")" | tee -a $OUT/ab.log; }
run base DFX_NOP=1
run convp_wgs64 DFX_CONVP_WGS=64
run convp_wgs96 DFX_CONVP_WGS=96
run convp_wgs128 DFX_CONVP_WGS=128
run convp_wgs192 DFX_CONVP_WGS=192
run base2 DFX_NOP=1
run exact DFX_EXACT_FP32=1
run exact_chunks10 DFX_EXACT_FP32=1 DFX_SEQ_CHUNKS=10 DFX_SEQ_RAMP=32
run exact_chunks14 DFX_EXACT_FP32=1 DFX_SEQ_CHUNKS=14 DFX_SEQ_RAMP=16
timeout 600 python -m pytest tests/test_enhance.py -m gpu -x -q -k "EXACT or golden" 2>&1 | tail -2
bash tools/gpu_trace.sh r4f_exact_tl DFX_EXACT_FP32=1 > /dev/null 2>&1; head -12 gpurun_out/r4f_exact_tl/timeline.txt | cut -c1-100; grep "gru_seq" gpurun_out/r4f_exact_tl/timeline.txt | cut -c1-100; tail -2 gpurun_out/r4f_exact_tl/timeline.txt
