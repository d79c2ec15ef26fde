#!/bin/bash
# Round 2, GPU session B: which memory stream of the GRU recurrence is slowed down by background HBM traffic (compile-time ablations)
OUT=gpurun_out/${1:-r02b}
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
for a in "" _a1 _a2 _a16 _a18 _a19; do
  G=tools/dev/_build/gru_h3_multi$a
  echo "== ablation '$a' (1 no W stream, 2 no gi loads, 16 no y stores)"
  timeout 60 $G 5 167 -1 2048 3 p 0 5
  timeout 60 $G 5 167 0 2048 3 p 0 5
  timeout 60 $G 5 167 2 2048 3 p 0 5
done > $OUT/gru_ablate_load.log 2>&1
cat $OUT/gru_ablate_load.log
