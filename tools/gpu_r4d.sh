#!/bin/bash
# round 4, call D: exact-fp32 pipelined recurrence (tests + timing), PMC traffic of the deep filter / finishing kernel, full bench line
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_enhance.py tests/test_faults.py tests/test_fp16_range.py -m gpu -x -q -k "EXACT or fault or range or golden" 2>&1 | tail -4
(DFX_EXACT_FP32=1 DFX_BENCH_SKIP_EXTRAS=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > $OUT/exact_kernels.json)
python -c "
import json; j=json.load(open('$OUT/exact_kernels.json')); print('exact ms_per_step', j['ms_per_step'], {k:v['ms'] for k,v in j['kernels'].items()})"
bash tools/gpu_trace.sh r4d_exact_tl DFX_EXACT_FP32=1 > /dev/null 2>&1; head -30 gpurun_out/r4d_exact_tl/timeline.txt | cut -c1-110
bash tools/gpu_pmc_dfa.sh r4d_pmc_dfa 2>&1 | tail -3
bash tools/gpu_pmc_finish.sh r4d_pmc_finish 2>&1 | tail -3
(timeout 1200 python bench.py --steps 20 --warmup 3 > $OUT/bench_full.json 2> $OUT/bench_full.err); tail -c 3000 $OUT/bench_full.json | cut -c1-3000
