#!/bin/bash
# dev: SQ counters of the separable-conv kernels stand-alone (tools/dev/pw_bench), one counter group per pass
OUT=gpurun_out/${1:-pmc_pw}
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --list-avail > $GRAFT_REPO_ROOT/$OUT/avail.txt 2>&1)
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p$i -o pw -- $GRAFT_REPO_ROOT/tools/dev/_build/pw_bench 512 > $GRAFT_REPO_ROOT/$OUT/p$i.log 2>&1); echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "pwconv" not in k: continue
            acc[(k, r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
PY
find $OUT -name "*.csv" -size +2M -delete
