#!/bin/bash
# Round 2, GPU session A: stream-copy / deep-filter microbench, GRU XCD-isolation experiment, quick parity + bench A/B.
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 300 tools/dev/_build/dfa_bench > $OUT/dfa_bench.log 2>&1; echo "dfa_bench rc=$?"; tail -32 $OUT/dfa_bench.log
export GPU_MAX_HW_QUEUES=16
G=tools/dev/_build/gru_h3_multi
{
  timeout 60 $G 5 167 -1 2048 3 p 0 5       # no background, product placement (per-layer 4-XCD windows)
  timeout 60 $G 5 167 -1 2048 3 07 0 5      # no background, all layers on XCDs 0-2
  timeout 60 $G 5 167 0 2048 3 p 0 5        # copy on every XCD, product placement
  timeout 60 $G 5 167 0 2048 3 07 0 5       # copy on every XCD, layers on XCDs 0-2
  timeout 60 $G 5 167 0 2048 3 07 f8 5      # copy on XCDs 3-7 only, layers on XCDs 0-2
  timeout 60 $G 5 167 0 2048 3 p f8 5       # copy on XCDs 3-7 only, product placement (control)
  timeout 60 $G 5 167 0 4096 3 07 f8 5      # the same with a 2x larger copy grid
  timeout 60 $G 5 167 2 2048 3 07 f8 5      # read-only stream on XCDs 3-7
  timeout 60 $G 5 167 0 2048 3 03 fc 5      # layers on XCDs 0-1 (80 workgroups on 64 CUs: two rounds?), copy on 2-7
} > $OUT/gru_xcd.log 2>&1
cat $OUT/gru_xcd.log
unset GPU_MAX_HW_QUEUES
timeout 600 python -m pytest tests/test_df_apply.py tests/test_enhance.py -m gpu -x -q > $OUT/pytest_dfa.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_dfa.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_pad.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench_pad.log | cut -c1-1500
DFX_SPEC_PAD=0 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_nopad.log 2>&1; echo "bench nopad rc=$?"; tail -1 $OUT/bench_nopad.log | cut -c1-700
du -sh $OUT
