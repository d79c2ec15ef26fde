#!/usr/bin/env python3
"""HBM roofline of the multi-frame Wiener / MVDR filter kernel (dfx_k_mf_filter) on one MI355X: B=256 clips x 1002 frames, nb=96
bins, N=5 (the deep-filter stress shape of BASELINE.json configs[4] with the MF model's filter stage).  One JSON line per variant."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main():
    from deepfilternet_amd import _lib
    from deepfilternet_amd import multiframe as MF

    dev = _lib.device()
    B, T, F, nb, N = 256, 1002, 481, 96, 5
    g = torch.Generator(device=dev).manual_seed(0)
    spec = torch.randn((B, 1, T, F, 2), device=dev, generator=g)
    ifc = torch.randn((B, T, nb, 2 * N), device=dev, generator=g)
    a = torch.randn((B, T, nb, N, N), device=dev, generator=g, dtype=torch.complex64)
    mat = torch.view_as_real(a @ a.mH / N + torch.eye(N, device=dev)).reshape(B, T, nb, 2 * N * N).contiguous()
    del a
    alg = B * T * (nb * (8 * N * N + 8 * N + 16) + (F - nb) * 16)
    for name, cls, kw in (("MfWf inverse", MF.MfWf, dict(inverse=True)), ("MfWf solve", MF.MfWf, dict(inverse=False)),
                          ("MfMvdr inverse", MF.MfMvdr, dict(inverse=True)), ("MfMvdr cholesky solve", MF.MfMvdr, dict(inverse=False, cholesky_decomp=True))):
        op = cls(nb, N, lookahead=2, **kw)
        for _ in range(2):
            op(spec, ifc, mat)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = op(spec, ifc, mat)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(json.dumps({"op": name, "N": N, "ms": ms, "algorithmic_GB": alg / 1e9, "GB/s": alg / ms / 1e6, "frac_of_8TBs": alg / ms / 1e6 / 8000,
                          "finite": bool(torch.isfinite(y).all())}), flush=True)


if __name__ == "__main__":
    main()
