#!/bin/bash
OUT=gpurun_out/${1:-r02e}
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
{
for a in _a0 _a18; do
  G=tools/dev/_build/gru_h3_multi$a
  echo "== ablation '$a'; XCDs chosen by HW_REG_XCC_ID; copy launched after the recurrences"
  for cfg in "p 0" "07 0" "07 f8" "0f f0" "07 ff"; do
    set -- $cfg
    timeout 60 $G 5 167 -1 2048 3 $1 $2 5 1 | grep -v spans
    timeout 60 $G 5 167 0 2048 3 $1 $2 5 1 | grep concurrent
    timeout 60 $G 5 167 0 4096 3 $1 $2 5 1 | grep concurrent
  done
done
} > $OUT/gru_xcc.log 2>&1
cat $OUT/gru_xcc.log
