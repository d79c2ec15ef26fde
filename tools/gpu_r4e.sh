#!/bin/bash
# round 4, call E: e0 recomputed in the decoder tail (tests, A/B), exact-mode phase trace
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4e; mkdir -p $OUT; rm -f $OUT/ab.log
timeout 1200 python -m pytest tests/test_enhance.py tests/test_fusions.py tests/test_streaming.py tests/test_streaming_gated.py tests/test_full_size.py tests/test_dfnet_kernels.py tests/test_capi.py -m gpu -x -q 2>&1 | tail -4
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3), 'finish', round(j.get('finish_in_loop_ms') or 0,4))")" | tee -a $OUT/ab.log; }
run e0_recompute DFX_NOP=1
run e0_stored DFX_E0_RECOMPUTE=0
run e0_recompute2 DFX_NOP=1
run e0_stored2 DFX_E0_RECOMPUTE=0
(DFX_BENCH_SKIP_EXTRAS=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('serialised kernels', {k:v['ms'] for k,v in j['kernels'].items()})")
(DFX_EXACT_FP32=1 timeout 200 python tools/dev/seq_trace.py 2>&1 | grep -v amdgpu > $OUT/seq_trace_exact.txt); cut -c1-260 $OUT/seq_trace_exact.txt
(timeout 200 python tools/dev/seq_trace.py 2>&1 | grep -v amdgpu > $OUT/seq_trace.txt); cut -c1-260 $OUT/seq_trace.txt
(timeout 200 python tools/bench_stream.py --model df3_ll --streams 4096 --frames-per-call 1 --calls 1000 2>&1 | tail -1 | cut -c1-300)
(timeout 200 python tools/bench_stream.py --model df3_ll --streams 4096 --frames-per-call 1 --calls 1000 --gating 2>&1 | tail -1 | cut -c1-300)
