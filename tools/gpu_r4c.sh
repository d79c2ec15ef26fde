#!/bin/bash
# round 4, call C: full GPU test suite + bench after the subtraction commit and the PCM16 boundary
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4c; mkdir -p $OUT
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $OUT/pytest_gpu.txt 2>&1
tail -6 $OUT/pytest_gpu.txt
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | cut -c1-300; done | tee $OUT/bench_main.txt
