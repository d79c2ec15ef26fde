#!/bin/bash
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4m; mkdir -p $OUT; rm -f $OUT/ab.log
timeout 900 python -m pytest tests/test_enhance.py tests/test_dfnet_kernels.py tests/test_config_options.py tests/test_streaming.py tests/test_capi.py -m gpu -x -q 2>&1 | tail -3
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3))")" | tee -a $OUT/ab.log; }
run lean DFX_NOP=1
run ggemm DFX_DFOUT_LEAN=0
run lean2 DFX_NOP=1
run ggemm2 DFX_DFOUT_LEAN=0
(DFX_BENCH_SKIP_EXTRAS=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('serialised kernels', {k:v['ms'] for k,v in j['kernels'].items()})")
for g in "" "--gating"; do timeout 200 python tools/bench_stream.py --model df3_ll --streams 4096 --frames-per-call 1 --calls 1000 $g 2>&1 | tail -1 | cut -c1-240; done
