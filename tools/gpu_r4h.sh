#!/bin/bash
# round 4, call H: fused DF encoder in the streaming runtime
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4h; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_streaming.py tests/test_streaming_gated.py tests/test_capi.py tests/test_lsnr_dropout.py tests/test_full_size.py tests/test_enhance.py -m gpu -x -q 2>&1 | tail -3
for v in "DFX_NOP=1" "DFX_FUSE_DFENC=0"; do
 for g in "" "--gating"; do echo "$v $g: $(env $v timeout 200 python tools/bench_stream.py --model df3_ll --streams 4096 --frames-per-call 1 --calls 1000 $g 2>&1 | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_call'],4))")"; done; done | tee $OUT/stream_ab.log
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | cut -c150-260; done
bash tools/gpu_trace_stream.sh r4h_stl > /dev/null 2>&1; cat gpurun_out/r4h_stl/*.txt 2>/dev/null | head -40 | cut -c1-150
