#!/bin/bash
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4n; mkdir -p $OUT; rm -f $OUT/ab.log
V=$PWD/tools/dev/_build
timeout 600 python -m pytest tests/test_enhance.py tests/test_dfnet_kernels.py -m gpu -x -q 2>&1 | tail -2
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3))")" | tee -a $OUT/ab.log; }
run wlds_fasttanh DFX_NOP=1
run regs_fasttanh DFX_LIBRARY=$V/libdfx_cphreg.so
run wlds_fasttanh2 DFX_NOP=1
run regs_fasttanh2 DFX_LIBRARY=$V/libdfx_cphreg.so
for lib in "" "$V/libdfx_cphreg.so"; do (DFX_LIBRARY=$lib DFX_BENCH_SKIP_EXTRAS=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('serialised kernels', {k:v['ms'] for k,v in j['kernels'].items() if k in ('dfx_k_df_convp','dfx_k_ggemm','dfx_k_pwconv')})"); done
