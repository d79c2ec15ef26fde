#!/bin/bash
# PMC counters of the persistent GRU launch in the two forms of the recurrence (pairs of CUs / one CU per 16 clips): matrix / vector / LDS instructions
# and the L2 requests of the launch.  Kernel-trace + --pmc only, one counter set per pass (guide: separate passes).  Usage: tools/gpu_pmc_gru.sh <tag>
TAG=${1:-pmc_gru}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1 DFX_HWQ_PROBE=0
run() {   # run <name> <pair 0|1> <counters...>
  local name=$1 pair=$2; shift 2
  (cd /tmp && DFX_GRU_PAIR=$pair timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --main-only > $GRAFT_REPO_ROOT/$OUT/$name.log 2>&1)
  echo "$name rc=$?"
}
for pair in 1 0; do
  run sq_p$pair $pair SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run lds_p$pair $pair SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY
  run tcc_p$pair $pair TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
done
python tools/pmc_summary.py $OUT "dfx_k_gru_seq" > $OUT/summary.txt 2>&1
python - <<'PY' $OUT
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*_p[01]")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "gru_seq" in r.get("Kernel_Name", ""):
                acc[(r["Kernel_Name"][:28], r["Counter_Name"])][0] += float(r["Counter_Value"]); acc[(r["Kernel_Name"][:28], r["Counter_Name"])][1] += 1
        for (k, c), (v, n) in sorted(acc.items()):
            print(f"{d.split('/')[-1]:10s} {k:30s} {c:34s} per launch {v / max(n, 1):.4g}  (launches {n})")
PY
find $OUT -name "*.csv" -size +4M -delete
du -sh $OUT
