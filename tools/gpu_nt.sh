#!/bin/bash
# dev: does a non-temporal background stream disturb the recurrences less than a plain one?  (tools/dev/gru_h3_multi)
OUT=gpurun_out/${1:-nt}; mkdir -p $OUT
X=tools/dev/_build/gru_h3_multi
{
echo "== alone"; timeout 60 $X 5 334 -1 2048 2 p 0 5 1 | grep -v spans
for sb in 96 192 384 2048; do
  for mode in 0 1 2; do
    echo "== stream mode $mode blocks $sb"; timeout 60 $X 5 334 $mode $sb 8 p 0 5 1 | grep -v spans
  done
done
} > $OUT/nt.log 2>&1
cat $OUT/nt.log
