#!/bin/bash
mkdir -p gpurun_out/r4z
: > gpurun_out/r4z/bench.txt
for cfg in "12 0 16384" "16 0 4096" "20 0 4096" "24 0 4096" "12 0 4096" "16 16 4096" "20 32 4096" "12 0 16384"; do
  set -- $cfg
  echo "== CHUNKS=$1 RAMP=$2 FAN_SMALL=$3" >> gpurun_out/r4z/bench.txt
  DFX_SEQ_CHUNKS=$1 DFX_SEQ_RAMP=$2 DFX_FAN_SMALL_ROWS=$3 timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | grep -o '"ms_per_step": [0-9.]*\|Error.*\|error.*' >> gpurun_out/r4z/bench.txt
done
cat gpurun_out/r4z/bench.txt
