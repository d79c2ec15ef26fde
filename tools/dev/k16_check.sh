#!/bin/bash
# Dev: the DFX_MFMA_K16 build (matrix kernels on v_mfma_f32_16x16x16_f16): parity, what it costs, and whether rocFFT beside it stays right
mkdir -p gpurun_out/r06
{
echo "== parity + faults subset on the k16 build"; DFX_LIBRARY=tools/dev/_build/libdfx_k16.so python -m pytest tests/test_enhance.py tests/test_dfnet_kernels.py -m gpu -x -q 2>&1 | tail -2
echo "== torch.fft.rfft beside forward passes: default build, then k16"
python tools/dev/two_analysis.py --iters 600 --other forward --victim torchfft 2>&1 | grep SUMMARY
DFX_LIBRARY=tools/dev/_build/libdfx_k16.so python tools/dev/two_analysis.py --iters 600 --other forward --victim torchfft 2>&1 | grep SUMMARY
echo "== step time: default, k16"
python bench.py --steps 10 --warmup 3 --main-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', round(d['ms_per_step'],3))"
DFX_LIBRARY=tools/dev/_build/libdfx_k16.so python bench.py --steps 10 --warmup 3 --main-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k16', round(d['ms_per_step'],3))"
} 2>&1 | tee gpurun_out/r06/k16.log
