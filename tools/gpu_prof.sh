#!/bin/bash
# rocprofv3 kernel trace of a short bench run + the timeline of its last step.  Usage: tools/gpu_prof.sh <tag> [env...]
TAG=${1:-prof}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
(cd /tmp && env "$@" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --main-only > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1); echo "prof rc=$?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -30 $OUT/kernel_stats.csv
t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/timeline.py "$t" 2 > $OUT/timeline.txt && python tools/dev/timeline_util.py $OUT/timeline.txt
tail -1 $OUT/prof.log | cut -c1-400
find $OUT -name "*.csv" -size +8M -delete
rm -rf $OUT/prof
