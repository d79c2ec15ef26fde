#!/bin/bash
mkdir -p gpurun_out/r06
P=tools/dev/_build/gru_2cu_probe
{
for mode in 0 1; do for nm in 0 48 96 144; do for sb in 0 768; do
  timeout 60 $P 80 1000 $nm $mode $sb
done; done; done
echo "--- 16 pairs only (one 16-clip group per layer would be 5 pairs; less L2 traffic)"; timeout 60 $P 16 1000 96 0 0; timeout 60 $P 16 1000 96 0 768
} 2>&1 | tee gpurun_out/r06/gru_2cu.log
