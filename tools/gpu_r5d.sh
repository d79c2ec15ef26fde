#!/bin/bash
# after the spill fixes: variant agreement, streaming, exact mode, io; bench main + exact + streaming
mkdir -p gpurun_out/r5d
timeout 1500 python -m pytest tests/test_enhance.py tests/test_streaming.py tests/test_streaming_gated.py tests/test_io.py tests/test_dsp_kernels.py tests/test_fusions.py -m gpu -x -q -n 4 2>&1 | tail -5 > gpurun_out/r5d/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 3 --main-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' > gpurun_out/r5d/bench.txt
DFX_EXACT_FP32=1 timeout 300 python bench.py --steps 10 --warmup 2 --main-only 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> gpurun_out/r5d/bench.txt
timeout 300 python tools/bench_stream.py --model df3_ll --streams 4096 --frames-per-call 1 --calls 1000 2>&1 | tail -1 | cut -c1-300 >> gpurun_out/r5d/bench.txt
timeout 300 python tools/bench_stream.py --model df3_ll --streams 4096 --frames-per-call 1 --calls 1000 --gating 2>&1 | tail -1 | cut -c1-300 >> gpurun_out/r5d/bench.txt
cat gpurun_out/r5d/pytest.txt gpurun_out/r5d/bench.txt
