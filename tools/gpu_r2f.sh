#!/bin/bash
# round 2, front-under-GRU overlap: parity of the variants, then bench A/B.  Usage: tools/gpu_r2f.sh <tag>
TAG=${1:-r2f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_enhance.py tests/test_onnx_targz.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
shift
for cfg in "DFX_FRONT_OVERLAP=0" "DFX_FRONT_OVERLAP=1" "DFX_FRONT_AHEAD=2" "DFX_FRONT_AHEAD=4" "$@"; do
  echo "== $cfg: $(env $cfg timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3), 'dfa', round(j.get('dfa_in_loop_ms',0),4))")" | tee -a $OUT/ab.log
done
