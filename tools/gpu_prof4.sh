#!/bin/bash
OUT=gpurun_out/${1:-r02p}
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
run() {
  rm -rf $OUT/prof
  (cd /tmp && env "$@" timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
  t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
  python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for key in ("df_apply", "dfx_k_synthesis", "dfx_k_analysis"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])) if key in r["Kernel_Name"]]
    print(f"  {key:16s}", " ".join(f"{x:6.0f}" for x in d))
PY
  tail -1 $OUT/prof.log | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('  ms_per_step', round(j['ms_per_step'],3))"
}
for v in "$@"; do echo "$v"; run $v; done
rm -rf $OUT/prof
