"""Parity of the HIP DSP kernels (through the C ABI / the libdf-compatible module) against the C oracle.

Every test runs twice: backend 'emu' executes the same kernel sources on the CPU SIMT interpreter (CPU CI, no GPU),
backend 'hip' (-m gpu) executes libdfx.so on the MI355X.  Tolerances: index arithmetic (frame/band/bin placement) is
exact; float values within a few ulp of the oracle (different FFT factorisation / fma contraction)."""
import numpy as np
import pytest
import torch

from oracle import libdf_oracle as L
from tests.helpers import rms


def _libdf():
    from deepfilternet_amd import libdf

    return libdf


@pytest.mark.parametrize("N,H,nb", [(960, 480, 32), (192, 96, 8), (96, 24, 8), (960, 240, 32), (320, 160, 24)])
def test_analysis_matches_oracle(backend, N, H, nb):
    D = _libdf()
    rng = np.random.default_rng(N)
    x = (0.3 * rng.standard_normal((3, H * 11 + 7))).astype(np.float32)
    d, o = D.DF(48000, N, H, nb, 1), L.DF(48000, N, H, nb, 1)
    assert d.erb_widths().tolist() == o.erb_widths().tolist()
    assert np.array_equal(d.fft_window(), o.fft_window())
    S, R = d.analysis(x), o.analysis(x)
    assert S.shape == R.shape and S.dtype == np.complex64
    assert np.abs(S - R).max() < 3e-7 * max(1.0, np.abs(R).max() / 0.02)


def test_analysis_many_frames_gridstride(backend):
    D = _libdf()
    rng = np.random.default_rng(1)
    x = rng.standard_normal((5, 96 * 41)).astype(np.float32)
    S, R = D.DF(48000, 192, 96, 8, 1).analysis(x), L.DF(48000, 192, 96, 8, 1).analysis(x)
    assert np.abs(S - R).max() < 1e-6


@pytest.mark.parametrize("N,H", [(960, 480), (192, 96), (96, 24), (960, 240)])
def test_synthesis_matches_oracle(backend, N, H):
    D = _libdf()
    rng = np.random.default_rng(N + 1)
    F = N // 2 + 1
    Y = (rng.standard_normal((2, 19, F)) + 1j * rng.standard_normal((2, 19, F))).astype(np.complex64)
    keep = Y.copy()
    y, r = D.DF(48000, N, H, 8, 1).synthesis(Y), L.DF(48000, N, H, 8, 1).synthesis(Y)
    assert np.array_equal(Y, keep)
    assert y.shape == r.shape == (2, 19 * H)
    assert np.abs(y - r).max() < 2e-5 * np.abs(r).max()


def test_roundtrip_and_streaming_state(backend):
    D = _libdf()
    rng = np.random.default_rng(3)
    d = D.DF(48000, 960, 480, 32, 2)
    x = rng.uniform(-1, 1, (1, 480 * 12)).astype(np.float32)
    y = d.synthesis(d.analysis(x))
    assert np.abs(y[:, 480:] - x[:, :-480]).max() < 5e-6
    # reset=False continues the stream exactly like one long call (pyDF keeps one DFState)
    o = L.DF(48000, 960, 480, 32, 2)
    full = o.analysis(x)
    d.reset()
    a = d.analysis(x[:, :480 * 5].copy(), reset=False)
    b = d.analysis(x[:, 480 * 5:].copy(), reset=False)
    assert np.abs(np.concatenate([a, b], 1) - full).max() < 1e-6
    ys = o.synthesis(full.copy())
    d.reset()
    y1 = d.synthesis(full[:, :5].copy(), reset=False)
    y2 = d.synthesis(full[:, 5:].copy(), reset=False)
    assert np.abs(np.concatenate([y1, y2], 1) - ys).max() < 2e-6


def test_roundtrip_quarter_hop_streaming(backend):
    D = _libdf()
    rng = np.random.default_rng(4)
    d, o = D.DF(48000, 96, 24, 8, 1), L.DF(48000, 96, 24, 8, 1)
    x = rng.uniform(-1, 1, (2, 24 * 30)).astype(np.float32)
    S = o.analysis(x)
    ys = o.synthesis(S.copy())
    assert np.abs(d.synthesis(S.copy()) - ys).max() < 2e-6
    d2 = D.DF(48000, 96, 24, 8, 1)
    parts = [d2.synthesis(S[:1, a:b].copy(), reset=False) for a, b in ((0, 3), (3, 4), (4, 17), (17, 30))]
    assert np.abs(np.concatenate(parts, 1) - ys[:1]).max() < 2e-6


def test_erb_family_matches_oracle(backend):
    D = _libdf()
    rng = np.random.default_rng(5)
    w = L.DF(48000, 960, 480, 32, 2).erb_widths()
    X = (rng.standard_normal((2, 3, 9, 481)) + 1j * rng.standard_normal((2, 3, 9, 481))).astype(np.complex64) * 0.1
    for db in (True, False):
        a, b = D.erb(X, w, db), L.erb(X, w, db)
        assert a.shape == b.shape == (2, 3, 9, 32)
        assert np.allclose(a, b, rtol=2e-6, atol=2e-5 if db else 1e-9)
    assert D.erb(X[0, 0], w).shape == (9, 32)
    g = rng.uniform(0, 1, (2, 7, 32)).astype(np.float32)
    assert np.array_equal(D.erb_inv(g, w), L.erb_inv(g, w))
    with pytest.raises(ValueError, match="Dimension not supported for erb: 5"):
        D.erb(X[None], w)
    with pytest.raises(ValueError, match="Number of erb bands"):
        D.erb_inv(np.zeros((2, 31), np.float32), w)
    assert np.array_equal(D.unit_norm_init(96), L.unit_norm_init(96))


def test_norms_match_oracle(backend):
    D = _libdf()
    rng = np.random.default_rng(6)
    e = (rng.standard_normal((3, 37, 32)) * 8 - 50).astype(np.float32)
    e1, e2 = e.copy(), e.copy()
    a, b = D.erb_norm(e1, 0.99), L.erb_norm(e2, 0.99)
    assert np.allclose(a, b, rtol=1e-6, atol=1e-6)
    assert np.array_equal(e1, a)  # in place + returned (F7)
    st = (rng.standard_normal((3, 32)) - 70).astype(np.float32)
    assert np.allclose(D.erb_norm(e.copy(), 0.99, st), L.erb_norm(e.copy(), 0.99, st), rtol=1e-6, atol=1e-6)
    X = (rng.standard_normal((2, 29, 96)) + 1j * rng.standard_normal((2, 29, 96))).astype(np.complex64)
    keep = X.copy()
    a, b = D.unit_norm(X, 0.99), L.unit_norm(X, 0.99)
    assert np.array_equal(X, keep)
    assert np.abs(a - b).max() < 1e-5 * np.abs(b).max()
    stu = rng.uniform(1e-4, 1e-3, (2, 96)).astype(np.float32)
    assert np.abs(D.unit_norm(X, 0.99, stu) - L.unit_norm(X, 0.99, stu)).max() < 1e-5 * np.abs(b).max()


def test_features_fused_matches_oracle_pipeline(backend):
    from deepfilternet_amd.enhance import df_features

    D = _libdf()
    rng = np.random.default_rng(7)
    x = (0.1 * rng.standard_normal((3, 480 * 13))).astype(np.float32)
    d, o = D.DF(48000, 960, 480, 32, 2), L.DF(48000, 960, 480, 32, 2)
    spec, fe, fs = df_features(torch.from_numpy(x), d, 96)
    S = o.analysis(x)
    FE = L.erb_norm(L.erb(S, o.erb_widths()), 0.99)
    FS = L.unit_norm(np.ascontiguousarray(S[..., :96]), 0.99)
    assert spec.shape == (3, 1, 13, 481, 2) and fe.shape == (3, 1, 13, 32) and fs.shape == (3, 1, 13, 96, 2)
    sc = torch.view_as_complex(spec.squeeze(1).cpu()).numpy()
    assert np.abs(sc - S).max() < 1e-6
    assert np.abs(fe.squeeze(1).cpu().numpy() - FE).max() < 2e-5
    fsc = torch.view_as_complex(fs.squeeze(1).cpu()).numpy()
    assert np.abs(fsc - FS).max() < 2e-5 * np.abs(FS).max()


@pytest.mark.parametrize("N,H,nb,minf", [(960, 480, 32, 2), (960, 480, 80, 1), (320, 160, 24, 1), (192, 96, 8, 1), (960, 240, 64, 1)])
def test_fused_erb_feature_band_layouts(backend, N, H, nb, minf):
    """The ERB feature the STFT kernel writes (band energies in dB, before the norm) for several band layouts: summed on 64 segment lanes
    (dfx_bands_create cuts the bands into near-equal segments) and with more than 64 bands (no segment table: one lane per band, the kernel's
    second round of lanes).  Oracle: lib.rs:280-295 compute_band_corr, bins strictly in order."""
    from deepfilternet_amd.enhance import _norm_alpha, df_features

    D = _libdf()
    rng = np.random.default_rng(N + nb)
    x = (0.2 * rng.standard_normal((2, H * 9))).astype(np.float32)
    d, o = D.DF(48000, N, H, nb, minf), L.DF(48000, N, H, nb, minf)
    assert d.erb_widths().tolist() == o.erb_widths().tolist() and len(d.erb_widths()) == nb
    _, fe, _ = df_features(torch.from_numpy(x), d, min(96, N // 2))
    FE = L.erb_norm(L.erb(o.analysis(x), o.erb_widths()), _norm_alpha(d))   # (get_norm_alpha of this sr / hop, as df_features takes it)
    assert fe.shape == (2, 1, 9, nb)
    assert np.abs(fe.squeeze(1).cpu().numpy() - FE).max() < 2e-5


def test_device_tensor_path_no_host_roundtrip(backend):
    from deepfilternet_amd import _lib

    D = _libdf()
    rng = np.random.default_rng(8)
    x = rng.standard_normal((2, 96 * 6)).astype(np.float32)
    d = D.DF(48000, 192, 96, 8, 1)
    xt = torch.from_numpy(x).to(_lib.device())
    S = d.analysis(xt)
    assert isinstance(S, torch.Tensor) and S.dtype == torch.complex64 and S.device.type == _lib.device().type
    un = D.unit_norm(S[..., :40], 0.99)  # strided view read in place
    ref = L.unit_norm(np.ascontiguousarray(L.DF(48000, 192, 96, 8, 1).analysis(x)[..., :40]), 0.99)
    assert np.abs(un.cpu().numpy() - ref).max() < 1e-5 * np.abs(ref).max()


def test_error_conventions(backend):
    D = _libdf()
    d = D.DF(48000, 960, 480, 32, 2)
    with pytest.raises(RuntimeError, match="empty or not contiguous"):
        d.analysis(np.zeros((2, 4800), np.float32)[:, ::2])
    with pytest.raises(TypeError):
        d.analysis(np.zeros((2, 4800), np.float64))
    with pytest.raises(RuntimeError, match="hop_size"):
        D.DF(48000, 960, 500, 32, 2)
    with pytest.raises(RuntimeError, match="prime factor"):
        D.DF(48000, 2 * 7 * 11, 77, 8, 1)
    assert d.analysis(np.zeros((2, 100), np.float32)).shape == (2, 0, 481)
    assert (d.sr(), d.fft_size(), d.hop_size(), d.nb_erb()) == (48000, 960, 480, 32)
