"""Streaming (dfx_stream_*): the frame loop of the reference's DfTract::process (libDF/src/tract.rs:509-642).  The oracle is the
batch path, which is pinned against the reference's goldens: the concatenated streaming output equals enhance(pad=False) of the
whole signal delayed by the model's lookahead, however the signal is cut into calls; the first `lookahead` hops are silence."""
import numpy as np
import pytest
import torch

from oracle import dfnet_oracle as O
from tests.helpers import emu_subset, named_params, rms, torch_sd


def _run_stream(rt, x, cuts):
    hop = rt.frame_length
    out, pos = [], 0
    for n in cuts:
        out.append(rt.process(x[:, pos * hop:(pos + n) * hop]))
        pos += n
    assert pos * hop == x.shape[1]
    return torch.cat(out, dim=1)


@pytest.mark.parametrize("name", ["defaults", "df3", "pf32"])
def test_stream_equals_batch_delayed(backend, name):
    from deepfilternet_amd.enhance import enhance, init_df
    from deepfilternet_amd.streaming import DfStream

    if backend == "emu" and name != "pf32":
        pytest.skip("the interpreter is slow: it covers the conv_ch=32 model (kt=3, lookahead 1); the GPU run covers all three")
    p = named_params(name)
    model, df_state, _, _ = init_df(params=p, epoch="none", seed=9)
    hop, T = 480, (9 if backend == "emu" else 23)   # the interpreter is slow: fewer hops and cut patterns there
    rng = np.random.default_rng(2)
    x = torch.from_numpy((0.1 * rng.standard_normal((3, hop * T))).astype(np.float32))
    ref = enhance(model, df_state, x, pad=False)                      # batch path (itself checked against the oracle below)
    assert rms(ref.numpy() - O.enhance(p, torch_sd(p, 9), x.numpy(), pad=False)) < 2e-6
    tol = 1e-6
    if p.mask_pf:
        # with a post filter the two paths are different reference code: enhance() runs the PyTorch model's filter on every bin
        # (deepfilternet3.py:448-454), the runtime libDF's, which leaves the last (ch * F) % 4 bins of a frame alone (lib.rs:446-471)
        from oracle import stream_oracle as S

        refs = np.stack([S.process_stream(p, torch_sd(p, 9), xi, pf_beta=p.pf_beta, thresholds=(-1e9, 1e9, 1e9))[0] for xi in x.numpy()])
        tol = 1e-4
    rt = DfStream(model, df_state, streams=3, max_frames=7)
    d = rt.delay_frames
    assert d == p.df_lookahead and rt.frame_length == hop
    for cuts in (([1, 1, 1, 3, 1, 2],) if backend == "emu" else ([1] * T, [7, 7, 7, 2], [3, 1, 5, 2, 7, 1, 4])):
        rt.reset()
        y = _run_stream(rt, x, cuts)
        assert y.shape == x.shape
        if d:
            assert float(y[:, : d * hop].abs().max()) == 0.0  # warm-up hops are silence
        err = rms((y[:, d * hop:] - ref[:, : (T - d) * hop]).numpy())
        assert err < tol, (cuts, err)
        if p.mask_pf:
            assert rms(y.numpy() - refs) < 1e-6, (cuts, rms(y.numpy() - refs))
    model.check()


def test_stream_controls(backend):
    from deepfilternet_amd.enhance import enhance, init_df
    from deepfilternet_amd.streaming import DfStream

    if emu_subset(backend):
        pytest.skip("interpreter subset: the controls are exercised through test_streaming_gated.py there (DFX_EMU_ALL=1 runs this too)")
    p = named_params("pf32")
    model, df_state, _, _ = init_df(params=p, epoch="none", seed=3)
    hop, T = 480, (8 if backend == "emu" else 12)
    rng = np.random.default_rng(4)
    x = torch.from_numpy((0.1 * rng.standard_normal((2, hop * T))).astype(np.float32))
    rt = DfStream(model, df_state, streams=2, max_frames=4)
    d = rt.delay_frames
    # attenuation limit: same mix as enhance(atten_lim_db=...)
    rt.set_atten_lim(12.0)
    rt.set_post_filter_beta(0.0)          # (with the post filter the runtime and enhance() are different reference code, see above)
    model_nopf, _, _, _ = init_df(params=named_params("pf32_nopf"), epoch="none", seed=3)
    y = _run_stream(rt, x, [4] * (T // 4))
    ref = enhance(model_nopf, df_state, x, pad=False, atten_lim_db=12.0)
    assert rms((y[:, d * hop:] - ref[:, : (T - d) * hop]).numpy()) < 1e-6
    # |dB| < 0.01: the reference passes the input through untouched and undelayed (tract.rs:540-543)
    rt.reset()
    rt.set_atten_lim(0.0)
    assert torch.equal(rt.process(x[:, : 4 * hop]), x[:, : 4 * hop])
    # >= 100 dB switches the limit off again
    rt.reset()
    rt.set_atten_lim(100.0)
    y = _run_stream(rt, x, [4] * (T // 4))
    ref = enhance(model_nopf, df_state, x, pad=False)
    assert rms((y[:, d * hop:] - ref[:, : (T - d) * hop]).numpy()) < 1e-6
    lsnr = rt.process(x[:, : hop], return_lsnr=True)[1]
    assert lsnr.shape == (2, 1) and bool(torch.isfinite(lsnr).all())
    with pytest.raises(ValueError):
        rt.process(x[:, : 5 * hop])      # more than max_frames
    with pytest.raises(ValueError):
        rt.process(x[:, : hop + 1])      # not a whole number of hops


@pytest.mark.parametrize("channels", [1, 2])
def test_stream_post_filter_is_libdfs(backend, channels):
    """The real-time runtime filters with libDF's post_filter (lib.rs:446-471 via tract.rs:603-610): chunks_exact(4) over the frame's
    flattened [ch * F] bins — mono, F = 481: the Nyquist bin is never filtered; two channels: the last two bins of the second channel —
    not with the PyTorch model's formula on every bin (deepfilternet3.py:448-454).  The signal has energy at Nyquist, so the two differ."""
    from deepfilternet_amd.enhance import init_df
    from deepfilternet_amd.streaming import DfStream
    from oracle import stream_oracle as S

    if channels == 2 and emu_subset(backend):
        pytest.skip("interpreter subset: the mono case runs here, the two-channel case on the GPU (DFX_EMU_ALL=1 runs both)")
    p = named_params("defaults")
    sd = torch_sd(p, 4)
    model, df_state, _, _ = init_df(params=p, epoch="none", seed=4)
    T = 6 if backend == "emu" else 16
    hop = p.hop_size
    rng = np.random.default_rng(3)
    n = np.arange(T * hop)
    x = (0.05 * rng.standard_normal((channels, T * hop)) + 0.2 * np.cos(np.pi * n) + 0.1 * np.cos(np.pi * n * 479 / 480)).astype(np.float32)
    rt = DfStream(model, df_state, streams=channels, max_frames=2, channels=channels, gating=True, thresholds=(-1e9, 1e9, 1e9))
    rt.set_post_filter_beta(0.3)
    y = torch.cat([rt.process(torch.from_numpy(x[:, k * hop:(k + 2) * hop])) for k in range(0, T, 2)], dim=1).numpy()
    xin = x[0] if channels == 1 else x
    ref = S.process_stream(p, sd, xin, pf_beta=0.3, thresholds=(-1e9, 1e9, 1e9))[0].reshape(channels, -1)
    alt = S.process_stream(p, sd, xin, pf_beta=0.3, thresholds=(-1e9, 1e9, 1e9), pf_like_torch=True)[0].reshape(channels, -1)
    assert rms(y - ref) < 1e-6, rms(y - ref)
    assert rms(alt - ref) > 20 * rms(y - ref) and rms(alt - ref) > 1e-5     # the test can see which filter ran


def test_linear_rolling_spectra_equal_the_ring_form(backend, monkeypatch):
    """The rolling spectra of an ungated handle live in a linear buffer through which the deep filter's window slides (the last frames go
    back to the front when it reaches the end: DFX_STREAM_LINEAR=<slack> makes that happen every few hops here); the ring form
    (DFX_STREAM_LINEAR=0) rewrites the window every call.  Same bits — also across a switch to gating (the state changes form) and back,
    and through the pass-through setting of the attenuation limit, which keeps the spectra moving."""
    from deepfilternet_amd.enhance import init_df
    from deepfilternet_amd.streaming import DfStream

    p = named_params("pf32")
    model, df_state, _, _ = init_df(params=p, epoch="none", seed=9)
    hop, T = 480, (12 if backend == "emu" else 60)
    rng = np.random.default_rng(5)
    x = torch.from_numpy((0.1 * rng.standard_normal((2, hop * T))).astype(np.float32))
    cuts = [1, 2, 1, 1, 3, 1, 1, 2] + [1] * (T - 12)

    def run(env, toggle):
        monkeypatch.setenv("DFX_STREAM_LINEAR", env)
        rt = DfStream(model, df_state, streams=2, max_frames=3)
        out, pos = [], 0
        for i, n in enumerate(cuts):
            if toggle and i == 5:
                rt.set_gating(True)
                rt.set_thresholds(-1e9, 1e9, 1e9)   # decisions that never skip a stage: the gated runtime computes the same frames
            if toggle and i == 9:
                rt.set_gating(False)
            if i == 12:
                rt.set_atten_lim(0.0)               # pass-through: analysis + rolling spectra only
            if i == 14:
                rt.set_atten_lim(100.0)
            out.append(rt.process(x[:, pos * hop:(pos + n) * hop]))
            pos += n
        return torch.cat(out, dim=1)

    for toggle in ((False,) if backend == "emu" else (False, True)):   # (the interpreter is slow: the form switches run on the GPU)
        ref = run("0", toggle)   # (switching gating on mid-stream starts the DF decoder's delay line empty: like is compared with like)
        for env in (("6",) if (toggle or backend == "emu") else ("1", "6")):
            assert torch.equal(run(env, toggle), ref), (env, toggle)


@pytest.mark.gpu
def test_one_hop_kernels_agree_with_the_general_path(hip_backend):
    """A call of ONE hop takes kernels of its own (dfx_k_gru_step_h3, dfx_k_df_convp_step with its pending sums, window updates on the DF
    branch's streams, feature windows in linear buffers); calls of several hops take the general windowed forward pass (and rebuild the one-hop
    state behind them).  The same streams cut into single hops, into calls of three hops, and mixed agree to rounding (df_convp's sums are taken
    in a different order).  (Up to round 5 each one-hop form also had an environment switch; removed in round 6.)"""
    from deepfilternet_amd.enhance import init_df
    from deepfilternet_amd.streaming import DfStream

    p = named_params("df3")
    model, df_state, _, _ = init_df(params=p, epoch="none", seed=12)
    hop, T = 480, 48
    x = torch.from_numpy((0.1 * np.random.default_rng(3).standard_normal((3, hop * T))).astype(np.float32)).cuda()

    def run(cuts):
        rt = DfStream(model, df_state, streams=3, max_frames=3)
        out, pos = [], 0
        for n in cuts:
            out.append(rt.process(x[:, pos * hop:(pos + n) * hop]).cpu())
            pos += n
        assert pos == T
        return torch.cat(out, dim=1).numpy()

    ones = run([1] * T)
    threes = run([3] * (T // 3))
    mixed = run([1] * 9 + [2, 3, 1, 1, 1, 2] + [1] * (T - 19))
    scale = float(np.sqrt((ones ** 2).mean()))
    assert np.isfinite(ones).all() and float(np.abs(ones).max()) > 1e-4
    for tag, y in (("threes", threes), ("mixed", mixed)):
        assert rms(y - ones) < 2e-6 * max(scale, 1e-3) + 1e-7, (tag, rms(y - ones), scale)
