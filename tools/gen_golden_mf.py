#!/usr/bin/env python3
"""Golden vectors for the multi-frame filter ops: the REFERENCE's own ``df.multiframe.MfWf`` / ``MfMvdr``
modules (multiframe.py:221-413) run on seeded inputs, every flag combination the MF model can select
(deepfilternetmf.py:335-352: cholesky_decomp x inverse) -> tests/golden/mf_ops.npz.  Build container only."""
from __future__ import annotations

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from tools.ref_import import install_shims  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden", "mf_ops.npz")
CASES = [  # name, op, N, lookahead, cholesky, inverse
    ("wf_inv", "wf", 5, 2, False, True), ("wf_chol_inv", "wf", 5, 0, True, True), ("wf_solve", "wf", 5, 2, False, False),
    ("wf_chol_solve", "wf", 3, 1, True, False), ("mvdr_inv", "mvdr", 5, 2, False, True), ("mvdr_chol_inv", "mvdr", 4, 0, True, True),
    ("mvdr_solve", "mvdr", 5, 1, False, False), ("mvdr_chol_solve", "mvdr", 2, 0, True, False), ("wf_n1", "wf", 1, 0, False, True),
]
B, T, F, NB = 2, 9, 20, 12


def make_inputs(rng, N, cholesky, inverse, mvdr):
    spec = rng.standard_normal((B, 1, T, F, 2)).astype(np.float32)
    ifc = rng.standard_normal((B, T, NB, N * 2)).astype(np.float32)
    a = rng.standard_normal((B, T, NB, N, N)) + 1j * rng.standard_normal((B, T, NB, N, N))
    if cholesky:
        m = np.tril(a) + 2.0 * np.eye(N)                      # a factor L; garbage above the diagonal must be ignored
        m = m + np.triu(rng.standard_normal((N, N)), 1) * 0.5
    elif not inverse:
        m = a @ a.conj().swapaxes(-1, -2) / N + np.eye(N)       # Hermitian positive definite ...
        m = m + 0.05 * np.triu(rng.standard_normal((N, N)), 1)  # ... with an upper triangle / diagonal imag part that gets overwritten
        m = m + 0.05j * np.eye(N)
    elif mvdr:
        m = a @ a.conj().swapaxes(-1, -2) / N + np.eye(N)       # positive definite: the MVDR denominator ifc^H M ifc stays away from 0
    else:
        m = a                                                   # an "inverse estimate": any matrix
    mat = np.stack([m.real, m.imag], -1).reshape(B, T, NB, N * N * 2).astype(np.float32)
    return spec, ifc, mat


def main():
    import torch

    install_shims()
    from df import multiframe as MF

    rng = np.random.default_rng(0)
    out = {}
    for name, op, N, la, chol, inv in CASES:
        spec, ifc, mat = make_inputs(rng, N, chol, inv, op == "mvdr")
        cls = MF.MfWf if op == "wf" else MF.MfMvdr
        mod = cls(NB, N, lookahead=la, cholesky_decomp=chol, inverse=inv).eval()
        with torch.no_grad():
            y = mod(torch.from_numpy(spec.copy()), torch.from_numpy(ifc.copy()), torch.from_numpy(mat.copy()))
        out.update({f"{name}.spec": spec, f"{name}.ifc": ifc, f"{name}.mat": mat, f"{name}.out": y.numpy(),
                    f"{name}.cfg": np.array([op == "mvdr", N, la, chol, inv, NB], dtype=np.int64)})
        print(name, y.shape, float(np.abs(y.numpy()).max()))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
