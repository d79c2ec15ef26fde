#!/bin/bash
# HBM traffic of enhance()'s finishing kernel from the PMC counters, separate passes per counter (MI355X_MICROARCH.md §HBM)
OUT=gpurun_out/${1:-pmc_finish}
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for c in FETCH_SIZE WRITE_SIZE; do
  sub=pmc_$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$sub -o fin -- python $GRAFT_REPO_ROOT/tools/dev/pmc_finish.py > $GRAFT_REPO_ROOT/$OUT/$sub.log 2>&1); echo "$c rc=$?"
done
python tools/pmc_finish_summary.py $OUT $OUT/finish_traffic.json
find $OUT -name "*.csv" -size +2M -delete
