#!/bin/bash
# rocprofv3 kernel trace of the streaming bench (BASELINE.json configs[3]), printed as the timeline of the last call.
# Usage: tools/gpu_trace_stream.sh <tag> [--gating]
TAG=${1:-trace_stream}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
rm -rf $OUT/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o st -- python $GRAFT_REPO_ROOT/tools/bench_stream.py --model df3_ll --streams 4096 --frames-per-call 1 --calls 30 "$@" > $GRAFT_REPO_ROOT/$OUT/prof_stream.log 2>&1)
t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py "$t" 2 > $OUT/timeline_stream.txt
python - "$t" $OUT/stream_trace_tail.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[-400:]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=list(keep[0].keys()))
w.writeheader()
w.writerows(keep)
PY
tail -1 $OUT/prof_stream.log | cut -c1-300
rm -rf $OUT/prof
