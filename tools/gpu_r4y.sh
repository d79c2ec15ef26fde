#!/bin/bash
# kernel-trace stats of the bench loop (per-kernel totals over 10 timed steps), for profiles/
OUT=gpurun_out/r4y
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
rm -rf $OUT/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --main-only > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
s=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp "$s" $OUT/kernel_stats.csv
t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py "$t" 2 > $OUT/timeline.txt
tail -1 $OUT/prof.log | cut -c1-300
head -40 $OUT/kernel_stats.csv
rm -rf $OUT/prof
