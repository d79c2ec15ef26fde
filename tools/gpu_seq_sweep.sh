#!/bin/bash
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for cfg in "$@"; do
  echo "== $cfg: $(env $cfg timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],3), 'dfa', round(j['dfa_in_loop_ms'],4))")"
done
