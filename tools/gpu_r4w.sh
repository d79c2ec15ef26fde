#!/bin/bash
# cross-pass overlap A/B: enhance() per step vs EnhancePipeline (front of batch k+1 under the GRU phase of batch k)
mkdir -p gpurun_out/r4w
timeout 900 python -m pytest tests/test_enhance.py -m gpu -x -q -k "pipeline_equals" 2>&1 | tail -5 > gpurun_out/r4w/pytest.txt
: > gpurun_out/r4w/bench.txt
for cfg in "0 1 front_first" "1 1 front_first" "1 0 front_first" "1 1 rest_first" "0 1 front_first"; do
  set -- $cfg
  echo "== DFX_BENCH_PIPELINE=$1 DFX_FRONT_PRIO=$2 DFX_PIPE_ORDER=$3" >> gpurun_out/r4w/bench.txt
  DFX_BENCH_PIPELINE=$1 DFX_FRONT_PRIO=$2 DFX_PIPE_ORDER=$3 timeout 600 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> gpurun_out/r4w/bench.txt
done
cat gpurun_out/r4w/pytest.txt gpurun_out/r4w/bench.txt
