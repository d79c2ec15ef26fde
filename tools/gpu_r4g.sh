#!/bin/bash
# round 4, call G: the DF branch of the encoder as one kernel (c1 never stored)
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4g; mkdir -p $OUT; rm -f $OUT/ab.log
timeout 900 python -m pytest tests/test_enhance.py tests/test_dfnet_kernels.py tests/test_config_options.py tests/test_onnx_targz.py -m gpu -x -q 2>&1 | tail -3
run() { tag=$1; shift; echo "== $tag: $(env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3))")" | tee -a $OUT/ab.log; }
run dfenc DFX_NOP=1
run two_kernels DFX_FUSE_DFENC=0
run dfenc2 DFX_NOP=1
run two_kernels2 DFX_FUSE_DFENC=0
(DFX_BENCH_SKIP_EXTRAS=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('serialised kernels', {k:v['ms'] for k,v in j['kernels'].items()})")
bash tools/gpu_trace.sh r4g_tl > /dev/null 2>&1; head -12 gpurun_out/r4g_tl/timeline.txt | cut -c1-100
