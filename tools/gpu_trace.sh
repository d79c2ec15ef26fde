#!/bin/bash
# rocprofv3 kernel trace of a short bench, printed as the timeline of the last step.  Usage: tools/gpu_trace.sh <tag> [ENV=VAL ...]
TAG=${1:-trace}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
rm -rf $OUT/prof
(cd /tmp && env "$@" timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --main-only > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py "$t" 2 > $OUT/timeline.txt
tail -1 $OUT/prof.log | cut -c1-200
rm -rf $OUT/prof
