#!/bin/bash
OUT=gpurun_out/${1:-r02l}
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1); echo "prof rc=$?"
t=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def nm(r): return r.get("Kernel_Name") or ""
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for i, r in enumerate(rows):
    if "df_apply" in nm(r):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        # what else overlaps [s, e]?
        ov = [nm(q)[:30] for q in rows if q is not r and int(q["Start_Timestamp"]) < e and int(q["End_Timestamp"]) > s]
        # previous kernel end on any stream
        pe = max(int(q["End_Timestamp"]) for q in rows[:i]) if i else s
        print(f"df_apply dur {(e-s)/1e3:8.1f} us  gap after last kernel end {(s-pe)/1e3:7.1f} us  overlaps: {ov}")
PY
tail -1 $OUT/prof.log | cut -c1-300
rm -rf $OUT/prof
