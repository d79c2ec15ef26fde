#!/bin/bash
# dev: per-launch durations of one serialised step (DFX_STREAMS=0: one stream, no time chunks)
OUT=gpurun_out/${1:-serial}; mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
(cd /tmp && DFX_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --main-only > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1); echo "rc=$?"
t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py "$t" 1 > $OUT/timeline.txt
rm -rf $OUT/prof
cut -c1-120 $OUT/timeline.txt
