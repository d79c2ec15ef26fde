#!/bin/bash
mkdir -p gpurun_out/xkern
export LD_LIBRARY_PATH=$PWD/deepfilternet_amd/csrc:$LD_LIBRARY_PATH
P=xkern_probe
for va in "pk mfma" "pk m16x16x16" "pk m32x32x8" "pk m32x32x16" "pk bf16x32" "pk mfma32" "pkfma mfma" "pkmul mfma" "pkadd mfma" "valu mfma" "ana m16x16x16" "ana m32x32x8" "ana m32x32x16" "ana bf16x32"; do
  set -- $va
  echo "== $1 beside $2"; timeout 300 tools/dev/_build/$P $1 $2 10 > gpurun_out/xkern/${P}_$1_$2.log 2>&1; grep "SUMMARY\|rror" gpurun_out/xkern/${P}_$1_$2.log | tail -1; grep "trial" gpurun_out/xkern/${P}_$1_$2.log | head -2
done
