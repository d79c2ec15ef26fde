#!/bin/bash
L=$PWD/tools/dev/_build/libdfx_dev.so
run() { echo -n "$* : "; env DFX_LIBRARY=$L "$@" timeout 300 python bench.py --main-only --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
run A=0
run DFX_CONVP_LATE=90 DFX_SEQ_TAIL_EVERY=2
run DFX_CONVP_LATE=95 DFX_SEQ_TAIL_EVERY=2
run DFX_CONVP_LATE=90 DFX_SEQ_TAIL_EVERY=3
run DFX_CONVP_LATE=90 DFX_SEQ_TAIL_EVERY=2 DFX_SEQ_DFTAIL_EVERY=6
run DFX_SEQ_TAIL_EVERY=3
run DFX_SEQ_TAIL_EVERY=4
run A=0
run DFX_CONVP_LATE=90 DFX_SEQ_TAIL_EVERY=2
