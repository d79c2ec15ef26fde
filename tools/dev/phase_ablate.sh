#!/bin/bash
# what the side work of the GRU phase costs the step with the pair form: dev build, DFX_DEV_SKIP bits (1 no ERB tail, 2 no DF tail, 8 no df_convp; results invalid)
L=$PWD/tools/dev/_build/libdfx_dev.so
run() { echo -n "$* : "; env DFX_LIBRARY=$L "$@" timeout 300 python bench.py --main-only --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
run DFX_DEV_SKIP=0
run DFX_DEV_SKIP=8
run DFX_DEV_SKIP=1
run DFX_DEV_SKIP=2
run DFX_DEV_SKIP=3
run DFX_DEV_SKIP=11
run DFX_DEV_SKIP=0
