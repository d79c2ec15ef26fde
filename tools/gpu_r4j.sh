#!/bin/bash
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4j; mkdir -p $OUT
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | cut -c150-260; done | tee $OUT/bench_main.txt
for g in "" "--gating"; do timeout 200 python tools/bench_stream.py --model df3_ll --streams 4096 --frames-per-call 1 --calls 1000 $g 2>&1 | tail -1 | cut -c1-240; done
