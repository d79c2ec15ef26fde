#!/bin/bash
# Dev: new concurrency tests, the two-handle rounds, the whole GPU suite and a bench line in one GPU call
mkdir -p gpurun_out/r06
python -m pytest tests/test_concurrency.py -m gpu -x -q -s > gpurun_out/r06/concurrency.txt 2>&1; tail -15 gpurun_out/r06/concurrency.txt
python tools/dev/two_handles_diag.py --rounds 100 --max-dumps 2 --streams own > gpurun_out/r06/two_handles_own.log 2>&1; grep "handle\|SUMMARY" gpurun_out/r06/two_handles_own.log | tail -4
python -m pytest tests -m gpu -x -q > gpurun_out/r06/gpu_tests.txt 2>&1; tail -3 gpurun_out/r06/gpu_tests.txt
python bench.py > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.err; tail -c 700 gpurun_out/r06/bench.json
