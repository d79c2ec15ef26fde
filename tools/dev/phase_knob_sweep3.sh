#!/bin/bash
L=$PWD/tools/dev/_build/libdfx_dev.so
run() { echo -n "$* : "; env DFX_LIBRARY=$L "$@" timeout 300 python bench.py --main-only --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
run A=0
run DFX_FRONT_GRAIN_P=4
run DFX_FRONT_GRAIN_P=8
run DFX_FRONT_GRAIN_P=32
run DFX_FRONT_GRAIN_P=64
run DFX_FRONT_GRAIN_P=128
run A=0
