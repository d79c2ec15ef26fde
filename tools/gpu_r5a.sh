#!/bin/bash
# timing-only ablation: the chain's gi loads as L2 hits (every step re-reads the chunk's first rows) vs the real first-touch reads
mkdir -p gpurun_out/r5a
: > gpurun_out/r5a/bench.txt
for lib in "" tools/dev/_build/libdfx_gisame.so "" tools/dev/_build/libdfx_gisame.so; do
  echo "== DFX_LIBRARY=$lib" >> gpurun_out/r5a/bench.txt
  DFX_LIBRARY=$lib timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | grep -o '"ms_per_step": [0-9.]*\|Error.*\|error.*' >> gpurun_out/r5a/bench.txt
done
DFX_LIBRARY=tools/dev/_build/libdfx_gisame.so timeout 300 python tools/dev/seq_trace.py 2>&1 | tail -7 | cut -c1-330 > gpurun_out/r5a/seq_trace_gisame.txt
timeout 300 python tools/dev/seq_trace.py 2>&1 | tail -7 | cut -c1-330 > gpurun_out/r5a/seq_trace_base.txt
cat gpurun_out/r5a/bench.txt gpurun_out/r5a/seq_trace_base.txt gpurun_out/r5a/seq_trace_gisame.txt
