#!/bin/bash
mkdir -p gpurun_out/two_handles
D=gpurun_out/two_handles
run() { name=$1; shift; echo "== $name: $*" ; timeout 900 "$@" > $D/$name.log 2>&1; echo "rc $?"; grep "SUMMARY\|Error\|error" $D/$name.log | cut -c1-300 | tail -2; }
run ana_fwd_default  python tools/dev/two_analysis.py --iters 1500 --other forward
export DFX_LIBRARY=tools/dev/_build/libdfx_nopk.so
run ana_fwd_nopk     python tools/dev/two_analysis.py --iters 1500 --other forward
run ana_enh_nopk     python tools/dev/two_analysis.py --iters 1500 --other enhance
run two_own_nopk     python tools/dev/two_handles_diag.py --rounds 300 --max-dumps 2 --streams own
run two_shared_nopk  python tools/dev/two_handles_diag.py --rounds 300 --max-dumps 2
unset DFX_LIBRARY
python bench.py --steps 10 --warmup 3 > $D/bench_default.json 2> $D/bench_default.err; python - <<'PY'
import json
for n in ("default",):
    try:
        d=json.loads(open(f"gpurun_out/two_handles/bench_{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d.get("rooflines",{}).get("dfx_k_analysis"))
    except Exception as e: print(n, "failed", e)
PY
DFX_LIBRARY=tools/dev/_build/libdfx_nopk.so python bench.py --steps 10 --warmup 3 > $D/bench_nopk.json 2> $D/bench_nopk.err; python - <<'PY'
import json
for n in ("nopk",):
    try:
        d=json.loads(open(f"gpurun_out/two_handles/bench_{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d.get("rooflines",{}).get("dfx_k_analysis"))
    except Exception as e: print(n, "failed", e)
PY
