#!/bin/bash
OUT=gpurun_out/${1:-r02d}
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
{
for a in _a0 _a1 _a2 _a16 _a18 _a3 _a19; do
  G=tools/dev/_build/gru_h3_multi$a
  echo "== ablation '$a' (1 no W stream, 2 no gi loads, 16 no y stores); copy launched after the recurrences"
  timeout 60 $G 5 167 -1 2048 3 p 0 5 1 | grep concurrent
  timeout 60 $G 5 167 0 2048 3 p 0 5 1 | grep concurrent      # copy
  timeout 60 $G 5 167 1 2048 3 p 0 5 1 | grep concurrent      # nt copy
  timeout 60 $G 5 167 2 2048 3 p 0 5 1 | grep concurrent      # read only
done
} > $OUT/gru_ablate_last.log 2>&1
cat $OUT/gru_ablate_last.log
