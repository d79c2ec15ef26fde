"""BASELINE.json configs[1] at full size (256 clips x 10 s of 48 kHz audio, DeepFilterNet3 shape) on the MI355X: the oracle cannot
run the whole batch in test time, so parity is shown through size-independent properties plus an oracle spot check of a few rows."""
import numpy as np
import pytest
import torch

from oracle import dfnet_oracle as O
from tests.helpers import rms

pytestmark = pytest.mark.gpu
SR, HOP, FFT = 48000, 480, 960


def _setup(hip_backend):
    from deepfilternet_amd.config import ModelParams
    from deepfilternet_amd.enhance import init_df
    from deepfilternet_amd.state_dict import random_state_dict

    p = ModelParams.deepfilternet3()
    sd = random_state_dict(p, 21)
    model, df_state, _, _ = init_df(params=p, state_dict=sd, epoch="none")
    return p, sd, model, df_state


def _audio(B, T, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    t = torch.arange(T, device="cuda", dtype=torch.float32) / SR
    f0 = 100.0 + 200.0 * torch.rand((B, 1), device="cuda", generator=g)
    x = 0.1 * torch.sin(2 * np.pi * f0 * t[None, :]) * (1 + 0.5 * torch.sin(2 * np.pi * 4.0 * t))[None, :]
    return (x + 0.05 * torch.randn((B, T), device="cuda", generator=g)).contiguous()


def test_full_batch_rows_match_oracle_and_single_clip_runs(hip_backend):
    from deepfilternet_amd.enhance import enhance

    p, sd, model, df_state = _setup(hip_backend)
    B, T = 256, 10 * SR
    x = _audio(B, T, 5)
    y = enhance(model, df_state, x)
    model.check()
    assert y.shape == (B, T) and bool(torch.isfinite(y).all())
    sdt = {k: torch.as_tensor(v) for k, v in sd.items()}
    for i in (0, 131, 255):
        # a clip gives the same bits alone as inside the batch (no cross-clip state, whatever the grid / chunking)
        yi = enhance(model, df_state, x[i:i + 1])
        assert torch.equal(yi, y[i:i + 1]), i
    # oracle (CPU restatement of the reference) on two of the rows
    rows = [7, 200]
    ref = O.enhance(p, sdt, x[rows].cpu().numpy())
    err = rms(y[rows].cpu().numpy() - ref)
    assert err < 2e-6, err


def test_full_batch_time_prefix_property(hip_backend):
    """The network is causal up to its lookahead: the enhanced first half of a clip does not depend on the second half, except for
    the hops within lookahead (+ the STFT overlap) of the cut."""
    from deepfilternet_amd.enhance import enhance

    p, sd, model, df_state = _setup(hip_backend)
    B, T = 256, 10 * SR
    x = _audio(B, T, 6)
    y = enhance(model, df_state, x)
    Th = 5 * SR
    yh = enhance(model, df_state, x[:, :Th].contiguous())
    keep = Th - (max(p.conv_lookahead, p.df_lookahead) + 2) * HOP - FFT
    assert torch.equal(yh[:, :keep], y[:, :keep])
    assert not torch.equal(yh[:, keep:], y[:, keep:Th])  # the tail of the short run saw zero padding instead of the future


def test_full_size_streaming_equals_batch(hip_backend):
    """4096 lockstep streams, hop by hop and in blocks: equal to the batch path delayed by the lookahead (dfx_stream_process)."""
    from deepfilternet_amd.enhance import enhance
    from deepfilternet_amd.streaming import DfStream

    p, sd, model, df_state = _setup(hip_backend)
    B, n_hops = 4096, 40
    x = _audio(B, n_hops * HOP, 7)
    ref = enhance(model, df_state, x, pad=False)
    rt = DfStream(model, df_state, streams=B, max_frames=8)
    d = rt.delay_frames
    for cuts in ([1] * n_hops, [8, 8, 8, 8, 8]):
        rt.reset()
        ys, pos = [], 0
        for n in cuts:
            ys.append(rt.process(x[:, pos * HOP:(pos + n) * HOP]))
            pos += n
        y = torch.cat(ys, dim=1)
        err = rms((y[:, d * HOP:] - ref[:, : (n_hops - d) * HOP]).cpu().numpy())
        assert err < 1e-6, (cuts[0], err)
        assert float(y[:, : d * HOP].abs().max()) == 0.0


def test_full_size_gated_streams(hip_backend):
    """BASELINE.json configs[3]: DeepFilterNet3 without lookahead, 4096 concurrent streams frame by frame, with the reference
    runtime's per-stream decisions (tract.rs:513-525,658-672) switched on.  Size-independent properties: a stream's output does not
    depend on its neighbours (row 0 == the same signal run alone), streams that take every stage equal the ungated runtime,
    digitally silent streams are frozen (zeros, lsnr -15) while their neighbours run; three rows against the streaming oracle."""
    from deepfilternet_amd.config import ModelParams
    from deepfilternet_amd.enhance import init_df
    from deepfilternet_amd.state_dict import random_state_dict
    from deepfilternet_amd.streaming import DfStream
    from oracle import stream_oracle as S

    p = ModelParams.deepfilternet3()
    p.conv_lookahead = p.df_lookahead = 0          # the "_ll" (low latency) model of ladspa/README.md:3
    sd = random_state_dict(p, 23)
    model, df_state, _, _ = init_df(params=p, state_dict=sd, epoch="none")
    B, n_hops = 4096, 24
    x = _audio(B, n_hops * HOP, 9)
    x[1::64] = 0.0                                  # every 64th stream is digital silence
    sdt = {k: torch.as_tensor(v) for k, v in sd.items()}
    # thresholds inside the lsnr distribution of this random-weight model (from three oracle rows), away from any observed value
    rows = [0, 2, 1]
    ungated = [S.process_stream(p, sdt, x[r].cpu().numpy(), thresholds=(-1e9, 1e9, 1e9)) for r in rows[:2]]
    vals = np.sort(np.concatenate([u[1] for u in ungated]))
    thr_df = float(vals[len(vals) // 2] + vals[len(vals) // 2 + 1]) / 2
    thr = (-1e9, 1e9, thr_df)                       # stage 1 always, stage 2 for the lower half of the lsnr values
    rt = DfStream(model, df_state, streams=B, max_frames=4, gating=True, thresholds=thr)
    ys, ls = [], []
    for c in range(0, n_hops, 4):
        y, l = rt.process(x[:, c * HOP:(c + 4) * HOP], return_lsnr=True)
        ys.append(y), ls.append(l)
    y, lsnr = torch.cat(ys, 1), torch.cat(ls, 1)
    assert bool(torch.isfinite(y).all())
    for r in rows:
        yr, lr, info = S.process_stream(p, sdt, x[r].cpu().numpy(), thresholds=thr)
        assert min(abs(v - thr_df) for v in info["lsnr_pass1"]) > 1e-4 if info["lsnr_pass1"] else True
        assert rms(y[r].cpu().numpy() - yr) < 1e-6, r
    # a stream alone == the same stream among 4095 others
    rt1 = DfStream(model, df_state, streams=1, max_frames=4, gating=True, thresholds=thr)
    y0 = torch.cat([rt1.process(x[:1, c * HOP:(c + 4) * HOP]) for c in range(0, n_hops, 4)], 1)
    assert rms((y0 - y[:1]).cpu().numpy()) < 1e-7
    # silent streams: with every lsnr above max_db_erb nothing resets the counter -> frozen after 3 hops (+1 silent, +1 no gains each)
    rt.reset()
    rt.set_thresholds(-1e9, -1e9, -1e9)
    y2, l2 = rt.process(x[:, : 4 * HOP], return_lsnr=True)
    y3, l3 = rt.process(x[:, 4 * HOP: 8 * HOP], return_lsnr=True)
    assert float(y3[1::64].abs().max()) == 0.0 and bool((l3[1::64] == -15.0).all())
    d = FFT - HOP                                   # nothing applied: the live streams come back delayed by the STFT only
    live = torch.cat([y2, y3], 1)[0::64]
    assert rms((live[:, d:] - x[0::64, : 8 * HOP - d]).cpu().numpy()) < 1e-6


def test_full_size_order10_deep_filter(hip_backend):
    """BASELINE.json configs[4]: the multi-frame stress configuration (df_order = 10, nb_df = 96) at batch 256 x 10 s — the whole
    enhance() with a DeepFilterNet3 whose deep filter has 10 taps (lookahead 3), and the deep-filter kernel alone on [256, 1002, 481]
    spectra.  Oracle rows + size-independent properties (a clip alone gives the same bits; linearity of the filter in its coefficients)."""
    from deepfilternet_amd import _lib
    from deepfilternet_amd.config import ModelParams
    from deepfilternet_amd.enhance import enhance, init_df
    from deepfilternet_amd.libdf import DF, _Bands
    from deepfilternet_amd.state_dict import random_state_dict

    p = ModelParams.deepfilternet3()
    p.df_order, p.df_lookahead, p.conv_lookahead = 10, 3, 3
    sd = random_state_dict(p, 25)
    model, df_state, _, _ = init_df(params=p, state_dict=sd, epoch="none")
    B, T = 256, 10 * SR
    x = _audio(B, T, 11)
    y = enhance(model, df_state, x)
    model.check()
    assert y.shape == (B, T) and bool(torch.isfinite(y).all())
    for i in (3, 255):
        assert torch.equal(enhance(model, df_state, x[i:i + 1]), y[i:i + 1]), i
    sdt = {k: torch.as_tensor(v) for k, v in sd.items()}
    rows = [17, 140]
    ref = O.enhance(p, sdt, x[rows].cpu().numpy())
    err = rms(y[rows].cpu().numpy() - ref)
    assert err < 2e-6, err
    # ---- the kernel alone at [256, 1002, 481], O = 10 (tap-major coefficients), dense rows and the engine's 64-byte aligned rows
    Tf, F, nd, O_, la, E = 1002, 481, 96, 10, 3, 32
    g = torch.Generator(device="cuda").manual_seed(3)
    spec = torch.randn((B, Tf, F, 2), device="cuda", generator=g)
    coefs = 0.3 * torch.randn((B, O_, Tf, nd, 2), device="cuda", generator=g)
    gains = torch.rand((B, Tf, E), device="cuda", generator=g)
    bands = _Bands.get(DF(48000, 960, 480, 32, 2).erb_widths())
    L = _lib.lib()

    def run(c, strided):
        Fs = 488 if strided else F
        s = torch.zeros((B, Tf, Fs, 2), device="cuda")
        s[:, :, :F] = spec
        out = torch.empty_like(s)
        if strided:
            _lib.check(L.dfx_df_apply_strided(_lib.ptr(s), Fs, _lib.ptr(c), 0, _lib.ptr(gains), bands.handle, B, Tf, F, nd, O_, la, 0.0, 0.0,
                                              _lib.ptr(out), Fs, _lib.stream()))
        else:
            _lib.check(L.dfx_df_apply(_lib.ptr(s), _lib.ptr(c), 0, _lib.ptr(gains), bands.handle, B, Tf, F, nd, O_, la, 0.0, 0.0, _lib.ptr(out),
                                      _lib.stream()))
        return out[:, :, :F]

    ya, yb = run(coefs, True), run(coefs, False)
    assert float((ya - yb).abs().max()) < 2e-5 * float(yb.abs().max())
    # oracle on one clip
    b = 77
    st = torch.view_as_complex(spec[b:b + 1].cpu().contiguous())
    ct = torch.view_as_complex(coefs[b:b + 1].cpu().contiguous())
    refk = st * O.band_gain(gains[b:b + 1].cpu(), DF(48000, 960, 480, 32, 2).erb_widths())
    refk[..., :nd] = O.df_apply(st, ct, O_, la, nd)
    assert float((torch.view_as_complex(ya[b:b + 1].cpu().contiguous()) - refk).abs().max()) < 2e-5 * float(refk.abs().max())
    # linearity in the coefficients on the deep-filter bins: DF(2c) == 2 DF(c) exactly (powers of two), DF(c1 + c2) == DF(c1) + DF(c2) up to rounding
    y2 = run(2.0 * coefs, True)
    assert torch.equal(y2[..., :nd, :], 2.0 * ya[..., :nd, :])
    c2 = 0.3 * torch.randn(coefs.shape, device="cuda", generator=g)
    ysum = run(coefs + c2, True)[..., :nd, :]
    ysep = ya[..., :nd, :] + run(c2, True)[..., :nd, :]
    assert float((ysum - ysep).abs().max()) < 1e-4 * float(ysep.abs().max())


def test_full_size_sub_batches_under_a_workspace_cap(hip_backend, monkeypatch):
    """384 clips x 10 s with the workspace capped at 24 GB (a 256-clip batch needs ~21 GB): enhance() runs the batch as
    256 + 128 clips in one workspace; every clip comes out as in a batch of its own size class (rows are independent)."""
    from deepfilternet_amd.enhance import enhance

    p, sd, model, df_state = _setup(hip_backend)
    B, T = 384, 10 * SR
    x = _audio(B, T, 9)
    monkeypatch.setenv("DFX_WORKSPACE_CAP_GB", "24")
    y = enhance(model, df_state, x)
    model.check()
    assert model._ws.numel() <= 24 * (1 << 30) and torch.isfinite(y).all()
    monkeypatch.delenv("DFX_WORKSPACE_CAP_GB")
    for rows in (slice(0, 16), slice(256, 272), slice(368, 384)):       # first sub-batch, the start and the end of the second
        yr = enhance(model, df_state, x[rows])
        assert torch.equal(y[rows], yr), rows


def test_full_size_fp16_split_vs_exact_fp32_every_row(hip_backend, monkeypatch):
    """The default arithmetic (fp16-split matrix ops: hi*hi + hi*lo + lo*hi, ~2^-21) against the exact fp32 kernels (DFX_EXACT_FP32=1) on
    ALL 256 clips of the config-2 batch, row by row: < 1e-6 RMS per row (signal RMS ~ 0.1; the north-star bar is 1e-4)."""
    from deepfilternet_amd.enhance import enhance, init_df

    p, sd, model, df_state = _setup(hip_backend)
    assert model.query(model.Q_EXACT_FP32) == 0
    monkeypatch.setenv("DFX_EXACT_FP32", "1")
    exact, _, _, _ = init_df(params=p, state_dict=sd, epoch="none")
    monkeypatch.delenv("DFX_EXACT_FP32")
    assert exact.query(exact.Q_EXACT_FP32) == 1
    B, T = 256, 10 * SR
    x = _audio(B, T, 9)
    y = enhance(model, df_state, x)
    model.check()
    ye = enhance(exact, df_state, x)
    exact.check()
    row_rms = (y - ye).pow(2).mean(dim=1).sqrt()
    worst = float(row_rms.max())
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(ye).all())
    assert worst < 1e-6, (worst, int(row_rms.argmax()))
