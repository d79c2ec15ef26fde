"""PMC calibration run for DF-apply (run under `rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv`).
Dispatch order of dfx_k_df_apply_rows (the engine's layout: rows of 488 bins, tap-major coefficients): 4 x "calibration" (nb_df=2,
order=1, no gains: a pure stream through the same kernel with known byte counts: reads 241 float4 and writes 244 float4 per frame)
then 4 x the config-2 shape (O=5, nb_df=96, gains)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepfilternet_amd import _lib, libdf  # noqa: E402

B, T, F, Fs, E = 256, 1002, 481, 488, 32
dev = _lib.device()
df = libdf.DF(48000, 960, 480, 32, 2)
g = torch.Generator(device=dev).manual_seed(0)
spec = torch.randn((B, T, Fs, 2), device=dev, generator=g)
gains = torch.rand((B, T, E), device=dev, generator=g)
out = torch.empty_like(spec)
L = _lib.lib()


def run(nd, O, la, use_gains):
    coefs = torch.randn((B, O, T, nd, 2), device=dev, generator=g) * 0.3
    for _ in range(4):
        _lib.check(L.dfx_df_apply_strided(_lib.ptr(spec), Fs, _lib.ptr(coefs), 0, _lib.ptr(gains) if use_gains else None,
                                          df.bands_handle if use_gains else None, B, T, F, nd, O, la, 0.0, 0.0, _lib.ptr(out), Fs, _lib.stream()))
    torch.cuda.synchronize()


run(2, 1, 0, False)
run(96, 5, 2, True)
