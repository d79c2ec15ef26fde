#!/bin/bash
# dev: serialised-step per-launch durations for library variants.  Usage: tools/gpu_serial_v.sh <outtag> <kernel-substring> <libtag>...
OUT=gpurun_out/$1; KEY=$2; shift; shift
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for t in "$@"; do
  L=""; [ "$t" != base ] && L=$GRAFT_REPO_ROOT/tools/dev/_build/libdfx_$t.so
  (cd /tmp && DFX_LIBRARY=$L DFX_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --main-only > $GRAFT_REPO_ROOT/$OUT/prof_$t.log 2>&1)
  f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
  python tools/timeline.py "$f" 1 > $OUT/timeline_$t.txt
  rm -rf $OUT/prof
  echo "== $t"; grep -E "$KEY" $OUT/timeline_$t.txt | cut -c1-100
done
