#!/bin/bash
# round 4, call A: baseline on this box, GRU variants under load (8-wave recurrence, side work ablations), exact-mode kernel times, streaming
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4a; mkdir -p $OUT; rm -f $OUT/ab.log
V=$PWD/tools/dev/_build
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | cut -c1-330)" | tee -a $OUT/ab.log; }
run base DFX_NOP=1
run nw8a DFX_LIBRARY=$V/libdfx_nw8a.so
run nw8b DFX_LIBRARY=$V/libdfx_nw8b.so
run base2 DFX_NOP=1
run skip_tails DFX_DEV_SKIP=3
run skip_tails_convp DFX_DEV_SKIP=11
run skip_all_side DFX_DEV_SKIP=15
run nw8a_skip_tails_convp DFX_LIBRARY=$V/libdfx_nw8a.so DFX_DEV_SKIP=11
(timeout 200 python tools/dev/seq_trace.py > $OUT/seq_trace_base.txt 2>&1)
(DFX_LIBRARY=$V/libdfx_nw8a.so timeout 200 python tools/dev/seq_trace.py > $OUT/seq_trace_nw8a.txt 2>&1)
(DFX_DEV_SKIP=11 timeout 200 python tools/dev/seq_trace.py > $OUT/seq_trace_skip11.txt 2>&1)
(DFX_EXACT_FP32=1 DFX_BENCH_SKIP_EXTRAS=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > $OUT/exact_kernels.json)
(timeout 200 python tools/bench_stream.py --model df3_ll --streams 4096 --frames-per-call 1 --calls 1000 2>&1 | tail -1 > $OUT/stream_ungated.json)
(timeout 200 python tools/bench_stream.py --model df3_ll --streams 4096 --frames-per-call 1 --calls 1000 --gating 2>&1 | tail -1 > $OUT/stream_gated.json)
cat $OUT/ab.log | cut -c1-200
tail -8 $OUT/seq_trace_base.txt | cut -c1-250
