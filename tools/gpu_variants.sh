#!/bin/bash
# dev: bench (main loop only) with library variants built by tools/dev/build_variant.sh.  Usage: tools/gpu_variants.sh <outtag> <libtag>...
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for t in "$@"; do
  L=""; [ "$t" != base ] && L=$GRAFT_REPO_ROOT/tools/dev/_build/libdfx_$t.so
  for rep in 1 2; do
    DFX_LIBRARY=$L timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --main-only > $OUT/b_$t.log 2>&1
    echo "$t: $(tail -1 $OUT/b_$t.log | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3))' 2>/dev/null || tail -2 $OUT/b_$t.log)"
  done
done
