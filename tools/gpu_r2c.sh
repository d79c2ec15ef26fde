#!/bin/bash
OUT=gpurun_out/${1:-r02c}
mkdir -p $OUT
timeout 60 tools/dev/_build/xcd_probe > $OUT/xcd_probe.log 2>&1; cat $OUT/xcd_probe.log
export GPU_MAX_HW_QUEUES=16
{
for a in _a0 _a19; do
  G=tools/dev/_build/gru_h3_multi$a
  echo "== ablation '$a'"
  timeout 60 $G 5 167 -1 2048 3 p 0 5 0      # alone
  timeout 60 $G 5 167 0 2048 3 p 0 5 0       # copy launched first
  timeout 60 $G 5 167 0 2048 3 p 0 5 1       # copy launched last
  timeout 60 $G 5 167 0 2048 3 07 f8 5 0     # isolated XCDs, copy first
  timeout 60 $G 5 167 0 2048 3 07 f8 5 1     # isolated XCDs, copy last
done
} > $OUT/gru_order.log 2>&1
cat $OUT/gru_order.log
