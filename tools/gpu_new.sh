#!/bin/bash
# GPU session for the pieces added late in round 1: gated streaming, C API, config 1, IO kernels + their benches.
TAG=${1:-r01n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_streaming_gated.py tests/test_capi.py tests/test_config1.py tests/test_io.py -m gpu -x -q > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_new.log; tail -5 $OUT/pytest_new.log
timeout 300 python tools/bench_io.py > $OUT/bench_io.jsonl 2> $OUT/bench_io.err; echo "bench_io rc=$?"; cat $OUT/bench_io.jsonl
for args in "--model df3_ll" "--model df3_ll --gating" "--model df3 --gating"; do
  timeout 300 python tools/bench_stream.py $args --frames-per-call 1 --calls 100 >> $OUT/stream.jsonl 2>> $OUT/stream.err; echo "stream [$args] rc=$?"
done
cat $OUT/stream.jsonl
