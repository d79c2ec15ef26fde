#!/bin/bash
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4v; mkdir -p $OUT; rm -f $OUT/ab.log
V=$PWD/tools/dev/_build
timeout 600 python -m pytest tests/test_enhance.py tests/test_dfnet_kernels.py tests/test_config_options.py -m gpu -x -q 2>&1 | tail -2
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3))")" | tee -a $OUT/ab.log; }
run new DFX_NOP=1
run old DFX_LIBRARY=$V/libdfx_cphreg.so DFX_SEQ_CHUNKS=12 DFX_SEQ_RAMP=0
run new2 DFX_NOP=1
run old2 DFX_LIBRARY=$V/libdfx_cphreg.so DFX_SEQ_CHUNKS=12 DFX_SEQ_RAMP=0
(DFX_BENCH_SKIP_EXTRAS=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('serialised kernels', {k:v['ms'] for k,v in j['kernels'].items()})")
