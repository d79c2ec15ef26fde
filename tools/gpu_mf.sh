#!/bin/bash
TAG=${1:-r01m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_mf.py tests/test_io.py tests/test_streaming.py -m gpu -x -q > $OUT/pytest_mf.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_mf.log; tail -4 $OUT/pytest_mf.log
timeout 300 python tools/bench_mf.py > $OUT/bench_mf.jsonl 2> $OUT/bench_mf.err; echo "bench_mf rc=$?"; cat $OUT/bench_mf.jsonl; tail -3 $OUT/bench_mf.err
