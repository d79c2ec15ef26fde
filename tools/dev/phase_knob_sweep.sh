#!/bin/bash
# knob sweep of the phase's side work with the pair form (dev build)
L=$PWD/tools/dev/_build/libdfx_dev.so
run() { echo -n "$* : "; env DFX_LIBRARY=$L "$@" timeout 300 python bench.py --main-only --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
run A=0
run DFX_CONVP_LATE=90
run DFX_CONVP_LATE=80
run DFX_CONVP_LATE=60
run DFX_SEQ_CHUNKS=10
run DFX_SEQ_CHUNKS=14
run DFX_SEQ_CHUNKS=16
run DFX_SEQ_P0_AHEAD=2
run DFX_SEQ_P0_AHEAD=4
run DFX_SEQ_DFTAIL_EVERY=2
run DFX_SEQ_DFTAIL_EVERY=6
run DFX_SEQ_TAIL_EVERY=2
run DFX_TAIL_SPLIT=0
run A=0
