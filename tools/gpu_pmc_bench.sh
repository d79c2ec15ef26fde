#!/bin/bash
# SQ / TCP counters of selected kernels inside a short bench run, one counter group per pass (rocprofv3 --pmc with --kernel-trace only).
# Usage: tools/gpu_pmc_bench.sh <tag> <kernel name substrings, |-separated>
TAG=${1:-pmc}; PAT=${2:-conv01_h3|convp_h3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  (cd /tmp && DFX_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p$i -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --main-only > $GRAFT_REPO_ROOT/$OUT/p$i.log 2>&1); echo "pass $i rc=$?"
done
python - "$OUT" "$PAT" <<'PY'
import csv, glob, collections, sys, re
out, pat = sys.argv[1], re.compile(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if not pat.search(k): continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
find $OUT -name "*.csv" -size +1M -delete
