"""Multi-frame Wiener / MVDR filter ops (df/multiframe.py:221-413) on the HIP engine.
Golden: the reference's own MfWf / MfMvdr modules on seeded inputs (tools/gen_golden_mf.py -> tests/golden/mf_ops.npz), every
(cholesky_decomp, inverse) combination the MF model can select."""
import os

import numpy as np
import pytest
import torch

from oracle import mf_oracle as M


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "mf_ops.npz"))


def _cases(g):
    return sorted({k.split(".")[0] for k in g.files})


def _cfg(g, n):
    mv, N, la, ch, inv, nb = (int(v) for v in g[n + ".cfg"])
    return dict(mvdr=bool(mv), num_freqs=nb, frame_size=N, lookahead=la, cholesky_decomp=bool(ch), inverse=bool(inv))


def _tol(ref):
    return 2e-5 * max(1.0, float(np.abs(ref).max()))


def test_oracle_matches_reference_modules(g):
    for n in _cases(g):
        y = M.mf_filter(torch.from_numpy(g[n + ".spec"]), torch.from_numpy(g[n + ".ifc"]), torch.from_numpy(g[n + ".mat"]), **_cfg(g, n))
        assert np.abs(y.numpy() - g[n + ".out"]).max() <= 1e-6 * max(1.0, np.abs(g[n + ".out"]).max()), n


def test_mf_ops_match_reference_golden(backend, g):
    from deepfilternet_amd import multiframe as MF

    for n in _cases(g):
        c = _cfg(g, n)
        cls = MF.MfMvdr if c.pop("mvdr") else MF.MfWf
        op = cls(c.pop("num_freqs"), c.pop("frame_size"), **c)
        spec = torch.from_numpy(g[n + ".spec"].copy())
        y = op(spec, torch.from_numpy(g[n + ".ifc"]), torch.from_numpy(g[n + ".mat"]))
        ref = g[n + ".out"]
        assert y.shape == ref.shape and y.dtype == torch.float32
        assert np.abs(y.numpy() - ref).max() < _tol(ref), (n, np.abs(y.numpy() - ref).max())
        assert torch.equal(spec, torch.from_numpy(g[n + ".spec"]))           # the input is left alone (the reference overwrites it)


@pytest.mark.parametrize("mvdr,N,la,chol,inv", [(False, 5, 2, False, True), (True, 5, 0, False, False), (False, 8, 3, True, False),
                                                (True, 6, 1, True, True), (False, 7, 0, False, False)])
def test_mf_ops_match_oracle_ragged(backend, mvdr, N, la, chol, inv):
    """Shapes that do not fill a workgroup / wave, the largest supported frame size, T shorter than the filter."""
    from deepfilternet_amd import multiframe as MF

    rng = np.random.default_rng(N * 7 + la)
    for B, T, F, nb in ((1, 3, 33, 33), (3, 7, 481, 96) if backend == "hip" else (2, 5, 40, 17), (2, 1, 9, 4)):
        spec = rng.standard_normal((B, 1, T, F, 2)).astype(np.float32)
        ifc = rng.standard_normal((B, T, nb, 2 * N)).astype(np.float32)
        a = rng.standard_normal((B, T, nb, N, N)) + 1j * rng.standard_normal((B, T, nb, N, N))
        m = 0.3 * np.tril(a, -1) + 2 * np.eye(N) if chol else a @ a.conj().swapaxes(-1, -2) / N + np.eye(N)   # well conditioned
        mat = np.stack([m.real, m.imag], -1).reshape(B, T, nb, 2 * N * N).astype(np.float32)
        ref = M.mf_filter(torch.from_numpy(spec), torch.from_numpy(ifc), torch.from_numpy(mat), mvdr=mvdr, num_freqs=nb, frame_size=N,
                          lookahead=la, cholesky_decomp=chol, inverse=inv).numpy()
        op = (MF.MfMvdr if mvdr else MF.MfWf)(nb, N, lookahead=la, cholesky_decomp=chol, inverse=inv)
        y = op(torch.from_numpy(spec), torch.from_numpy(ifc), torch.from_numpy(mat)).numpy()
        assert np.abs(y - ref).max() < 5 * _tol(ref), (B, T, F, nb, np.abs(y - ref).max())
    with pytest.raises(ValueError):
        op(torch.zeros(1, 1, 2, 9, 2), torch.zeros(1, 2, 4, 2 * N), torch.zeros(1, 2, 4, 3))
    with pytest.raises(RuntimeError):
        MF.MfWf(4, 9)(torch.zeros(1, 1, 2, 9, 2), torch.zeros(1, 2, 4, 18), torch.zeros(1, 2, 4, 162))   # frame_size > 8
