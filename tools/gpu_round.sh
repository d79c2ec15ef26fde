#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace.  Usage: tools/gpu_round.sh <tag> [what...]
# Everything lands under gpurun_out/<tag>/ (merged back by gpurun).
TAG=${1:-r01}; shift
WHAT=${*:-"tests smoke bench prof"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for w in $WHAT; do
  case $w in
    tests) timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log ;;
    bench_small) timeout 600 python bench.py --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_small.log 2>&1; echo "bench_small rc=$?"; tail -2 $OUT/bench_small.log ;;
    bench) timeout 900 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -2 $OUT/bench.log ;;
    # prof: the timed loop only (14 passes); the handshake of the phase's streams, which spins for seconds under the tool, is skipped
    prof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- env DFX_HWQ_PROBE=0 python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 2 --main-only > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1); echo "prof rc=$?"; find $OUT/prof -name "*stats*" | head; f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" ;;
    pmc) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1); echo "pmc fetch rc=$?"
         (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_write -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1); echo "pmc write rc=$?"
         python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1; tail -30 $OUT/pmc_summary.txt ;;
    *) echo "unknown step $w" ;;
  esac
done
# the raw traces can be large: keep only CSV summaries small enough to travel back
find $OUT -name "*.csv" -size +8M -delete
du -sh $OUT
