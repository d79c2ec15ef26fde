"""Pins the C oracle (oracle/df_oracle.c) — the checker for the HIP DSP kernels.

The Rust reference cannot be built here, so the oracle is pinned by (a) the identities the survey verified against the
reference's semantics (torch.stft/numpy.fft recipe of DeepFilterNet/tests/test_dflib.py:39-81, STFT->ISTFT
round trip of libDF/src/transforms.rs:618-638, band-gain test of libDF/src/lib.rs:626-652), (b) the golden vectors
produced by the reference's own PyTorch twins (tests/golden/modules.npz: modules.py:929-967).
"""
import os

import numpy as np
import pytest

from oracle import libdf_oracle as L

W_MIN2 = [2] * 13 + [5, 5, 7, 7, 8, 10, 12, 13, 15, 18, 20, 24, 28, 31, 37, 42, 50, 56, 67]
W_MIN1 = [1, 1, 1, 1, 1, 1, 2, 2, 2, 3, 3, 4, 4, 5, 5, 7, 7, 8, 10, 12, 13, 15, 18, 20, 24, 28, 31, 37, 42, 50, 56, 67]


def test_erb_widths_goldens():
    assert L.DF(48000, 960, 480, 32, 2).erb_widths().tolist() == W_MIN2
    assert L.DF(48000, 960, 480, 32, 1).erb_widths().tolist() == W_MIN1
    for sr, n, nb, mn in [(48000, 960, 32, 2), (16000, 320, 24, 1), (48000, 192, 8, 1), (44100, 1024, 32, 3)]:
        w = L.erb_fb_widths(sr, n, nb, mn)
        assert int(w.sum()) == n // 2 + 1 and (w[:-1] >= mn).all()


def test_window_and_wnorm():
    d = L.DF(48000, 960, 480, 32, 2)
    i = np.arange(960)
    w = np.sin(0.5 * np.pi * np.sin(0.5 * np.pi * (i + 0.5) / 480) ** 2)
    assert np.array_equal(d.fft_window(), w.astype(np.float32))
    assert d.wnorm() == np.float32(1.0) / (np.float32(960 * 960) / np.float32(960))
    # power complementary: w[i]^2 + w[i+hop]^2 == 1
    assert np.allclose(w[:480] ** 2 + w[480:] ** 2, 1.0)


@pytest.mark.parametrize("N,H", [(960, 480), (96, 24), (192, 96), (960, 240), (512, 256)])
def test_stft_matches_numpy_fft(N, H):
    rng = np.random.default_rng(N + H)
    d = L.DF(48000, N, H, 8, 1)
    x = rng.standard_normal((3, H * 17 + 5)).astype(np.float32)   # trailing 5 samples are dropped (pyDF lib.rs:50)
    S = d.analysis(x)
    Tf = x.shape[1] // H
    assert S.shape == (3, Tf, N // 2 + 1) and S.dtype == np.complex64
    w = d.fft_window().astype(np.float64)
    xp = np.concatenate([np.zeros((3, N - H)), x.astype(np.float64)], 1)
    ref = np.stack([np.fft.rfft(xp[:, t * H:t * H + N] * w, axis=-1) for t in range(Tf)], 1) * d.wnorm()
    assert np.abs(S - ref).max() < 2e-7 * max(1.0, np.abs(ref).max() / 0.05)


@pytest.mark.parametrize("N,H", [(960, 480), (960, 240), (192, 96)])
def test_stft_istft_roundtrip_delay(N, H):
    rng = np.random.default_rng(3)
    d = L.DF(48000, N, H, 8, 1)
    x = rng.uniform(-1, 1, (2, H * 40)).astype(np.float32)
    y = d.synthesis(d.analysis(x))
    dl = N - H
    assert np.abs(y[:, dl:] - x[:, :-dl]).max() < 2e-6


def test_synthesis_ignores_dc_nyquist_imag_and_keeps_input():
    rng = np.random.default_rng(4)
    d = L.DF(48000, 960, 480, 32, 2)
    S = (rng.standard_normal((1, 6, 481)) + 1j * rng.standard_normal((1, 6, 481))).astype(np.complex64)
    S2 = S.copy()
    S2[..., 0] = S2[..., 0].real
    S2[..., -1] = S2[..., -1].real
    keep = S.copy()
    y1, y2 = d.synthesis(S), d.synthesis(S2)
    assert np.array_equal(y1, y2)
    assert np.array_equal(S, keep)          # the oracle (unlike pyDF, F7) never clobbers its input
    # against numpy irfft (unnormalised => * N) with window + overlap-add
    w = d.fft_window().astype(np.float64)
    fr = np.fft.irfft(S2[0].astype(np.complex128), n=960, axis=-1) * 960 * w
    ola = np.zeros(7 * 480)
    for t in range(6):
        ola[t * 480:t * 480 + 960] += fr[t]
    assert np.abs(y1[0] - ola[:6 * 480]).max() < 1e-3 * np.abs(ola).max()


def test_reset_false_continues_stream():
    rng = np.random.default_rng(6)
    d = L.DF(48000, 960, 480, 32, 2)
    x = rng.standard_normal((1, 480 * 10)).astype(np.float32)
    full = d.analysis(x)
    d.reset()
    a = d.analysis(x[:, :480 * 4].copy(), reset=False)
    b = d.analysis(x[:, 480 * 4:].copy(), reset=False)
    assert np.array_equal(np.concatenate([a, b], 1), full)


def test_erb_and_inverse_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "modules.npz"))
    w = g["widths"]
    lin = L.erb(g["erb_in"], w, db=False)
    assert np.allclose(lin, g["erb_lin"], rtol=1e-5, atol=1e-8)
    assert np.array_equal(L.erb_inv(lin, w), L.erb_inv(lin, w.astype(np.uint64)))
    assert np.allclose(L.erb_inv(g["erb_lin"].astype(np.float32), w), g["erb_inv"])
    db = L.erb(g["erb_in"], w, db=True)
    assert np.allclose(db, 10 * np.log10(g["erb_lin"].astype(np.float64) + 1e-10), atol=1e-4)
    # 2-D and 3-D inputs keep their rank (pyDF lib.rs:152-190)
    assert L.erb(g["erb_in"][0, 0], w).shape == (7, 32) and L.erb(g["erb_in"][0], w).shape == (3, 7, 32)
    with pytest.raises(ValueError, match="Dimension not supported for erb: 5"):
        L.erb(g["erb_in"][None], w)
    with pytest.raises(ValueError, match="Number of erb bands"):
        L.erb_inv(np.zeros((2, 31), np.float32), w)


def test_unit_norm_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "modules.npz"))
    x = g["unit_norm_in"]
    xc = np.ascontiguousarray((x[..., 0] + 1j * x[..., 1]).astype(np.complex64)[:, 0])
    keep = xc.copy()
    y = L.unit_norm(xc, float(g["alpha"]))
    assert np.array_equal(xc, keep)                                  # works on a copy (pyDF lib.rs:285)
    ref = g["unit_norm_out"][..., 0] + 1j * g["unit_norm_out"][..., 1]
    assert np.allclose(y, ref, rtol=2e-5, atol=1e-6)
    # explicit state continues the recursion
    st = L.unit_norm_init(96).repeat(2, 0)
    a = L.unit_norm(xc[:, :40].copy(), 0.99, st)
    assert np.allclose(a, y[:, :40])


def test_erb_norm_semantics():
    rng = np.random.default_rng(8)
    x = (rng.standard_normal((2, 50, 32)) * 10 - 50).astype(np.float32)
    xin = x.copy()
    y = L.erb_norm(xin, 0.99)
    assert np.array_equal(xin, y)                                     # in place + returned copy (F7)
    s = np.tile(np.linspace(-60, -90, 32, dtype=np.float32), (2, 1))
    ref = np.zeros_like(x)
    a = np.float32(0.99)
    for t in range(50):
        s = x[:, t] * (np.float32(1) - a) + s * a
        ref[:, t] = (x[:, t] - s) / np.float32(40)
    assert np.allclose(y, ref, rtol=1e-5, atol=2e-6)


def test_band_gain_equals_erb_inv_multiply(golden_dir):
    """libDF/src/lib.rs:626-652 test_erb_inout + reference Mask golden (modules.py:248-269)."""
    g = np.load(os.path.join(golden_dir, "modules.npz"))
    w = g["widths"]
    sp = g["mask_spec"][..., 0] + 1j * g["mask_spec"][..., 1]
    out = L.apply_band_gain(sp.astype(np.complex64), g["mask_m"], w)
    ref = g["mask_out"][..., 0] + 1j * g["mask_out"][..., 1]
    assert np.array_equal(out, ref.astype(np.complex64))
    assert np.array_equal(out, sp.astype(np.complex64) * L.erb_inv(g["mask_m"], w))


def test_post_filter_matches_torch_formula():
    import torch
    from oracle.dfnet_oracle import post_filter as pf_t

    rng = np.random.default_rng(9)
    n = (rng.standard_normal((3, 480)) + 1j * rng.standard_normal((3, 480))).astype(np.complex64)
    e = (n * rng.uniform(0.05, 1.0, (3, 480))).astype(np.complex64)
    a = L.post_filter(n, e, 0.02)
    b = pf_t(torch.from_numpy(n), torch.from_numpy(e), 0.02).numpy()
    assert np.allclose(a, b, rtol=1e-4, atol=1e-6)


def test_error_conventions():
    d = L.DF(48000, 960, 480, 32, 2)
    with pytest.raises(RuntimeError, match="empty or not contiguous"):
        d.analysis(np.zeros((2, 4800), np.float32)[:, ::2])
    with pytest.raises(TypeError):
        d.analysis(np.zeros((2, 4800), np.float64))
    with pytest.raises(RuntimeError):
        L.DF(48000, 960, 500, 32, 2)
    assert d.analysis(np.zeros((2, 100), np.float32)).shape == (2, 0, 481)
