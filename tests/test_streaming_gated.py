"""Gated streaming (dfx_stream_set_gating): the per-frame decisions of the reference's real-time loop, DfTract::process
(libDF/src/tract.rs:509-616) — silent-input shortcut (:513-525), apply_stages (:658-672), skipped decoders keep their state — taken
independently by lockstep streams on the GPU, against oracle/stream_oracle.py (the reference's logic restated hop by hop over the
batch oracle, which is pinned against the reference's goldens; parity with tract itself is unpinned, see that file's header)."""
import numpy as np
import pytest
import torch

from oracle import stream_oracle as S
from tests.helpers import emu_subset, named_params, rms, torch_sd

HOP = 480


def _signals(T, seed):
    """stream 0: noise; stream 1: digital silence at the start (frozen before it ever ran a decoder), a near-silent stretch in the
    middle (mean square < 1e-7: frozen only while stage 1 is being skipped, tract.rs:562-567); stream 2: a level ramp."""
    rng = np.random.default_rng(seed)
    x = (0.1 * rng.standard_normal((3, HOP * T))).astype(np.float32)
    x[1, : HOP * 8] = 0
    x[1, HOP * (T // 2): HOP * (T - 3)] *= 1e-5
    x[2] *= np.linspace(0.01, 3, HOP * T).astype(np.float32)
    return x


def _thresholds(p, sd, x, quantiles):
    """Thresholds inside the lsnr distribution of these (random-weight) models so that every branch of apply_stages is taken; each
    one sits in the middle of the widest nearby gap between observed values, so the decisions do not hinge on the last bit."""
    vals = np.sort(np.concatenate([S.process_stream(p, sd, xi, thresholds=(-1e9, 1e9, 1e9))[1] for xi in x]))
    vals = vals[(vals != 0) & (vals > -14)]
    out = []
    for q in quantiles:
        i0 = int(q * (len(vals) - 1))
        cand = range(max(i0 - 3, 0), min(i0 + 3, len(vals) - 1))
        i = max(cand, key=lambda j: vals[j + 1] - vals[j])
        out.append(float(vals[i] + vals[i + 1]) / 2)
    return tuple(out)


def _run(rt, x, cuts):
    outs, ls, pos = [], [], 0
    for n in cuts:
        y, l = rt.process(torch.from_numpy(x[:, pos * HOP:(pos + n) * HOP]), return_lsnr=True)
        outs.append(y), ls.append(l)
        pos += n
    assert pos * HOP == x.shape[1]
    return torch.cat(outs, 1).numpy(), torch.cat(ls, 1).numpy()


CASES = [
    # model, hops, (min, max_erb, max_df) quantiles of the observed lsnr, cut pattern
    pytest.param("pf32", 24, (0.15, 0.85, 0.5), [4] * 6, id="pf32-all-branches"),           # kt = 3, lookahead 1, post filter
    pytest.param("pf32", 24, (0.0, 0.3, 0.15), [1] * 24, id="pf32-mostly-skipped"),           # long runs without stage 1 -> freezes mid-stream
    pytest.param("defaults", 24, (0.15, 0.85, 0.5), [3, 1, 4] * 3, id="defaults"),   # kt = 1, no lookahead
    pytest.param("df3", 30, (0.15, 0.85, 0.5), [5] * 6, id="df3-all-branches"),     # kt = 5, lookahead 2, conv_ch 64
    pytest.param("df3", 30, (0.0, 0.3, 0.15), [1] * 30, id="df3-mostly-skipped"),
]


@pytest.mark.parametrize("name,T,quantiles,cuts", CASES)
def test_gated_stream_matches_oracle(backend, name, T, quantiles, cuts):
    from deepfilternet_amd.enhance import init_df
    from deepfilternet_amd.streaming import DfStream

    if backend == "emu" and name != "pf32":
        pytest.skip("the interpreter covers the conv_ch=32 model; the others run on the GPU")
    if emu_subset(backend) and quantiles[0] == 0.0:
        pytest.skip("interpreter subset: the all-branches scenario runs here, this one on the GPU (DFX_EMU_ALL=1 runs both)")
    if backend == "emu":  # the interpreter is slow: fewer hops, one cut pattern
        T, cuts = 14, ([2] * 7 if len(cuts) < 10 else [1] * 14)
    p = named_params(name)
    sd = torch_sd(p, 9)
    model, df_state, _, _ = init_df(params=p, epoch="none", seed=9)
    x = _signals(T, 2)
    thr = _thresholds(p, sd, x, quantiles)
    ref = [S.process_stream(p, sd, xi, thresholds=thr) for xi in x]
    # the scenario really exercises the logic: every stage combination occurs, stream 1 is frozen at the start
    seen = {f for r in ref for f in r[2]["flags"]}
    assert {(True, False, True), (True, False, False), (False, False, False)} <= seen
    if quantiles[0] > 0:
        assert (False, True, False) in seen
    assert len(ref[0][2]["accepted"]) == T
    if name == "pf32":   # lookahead 1: hop 0 takes no decision, hops 1-2 skip stage 1 -> frozen from hop 3 until the silence ends
        acc1 = ref[1][2]["accepted"]
        assert acc1[:3] == [0, 1, 2] and acc1[3] >= 8
    if quantiles[1] < 0.5:
        assert len(ref[1][2]["accepted"]) < T - 5          # frozen again in the middle, with live state to preserve
    for r in ref:  # robust decisions only: no lsnr within 1e-4 dB of a threshold
        v = np.asarray(r[2]["lsnr_pass1"])
        assert min(np.abs(v - t).min() for t in thr) > 1e-4
    rt = DfStream(model, df_state, streams=3, max_frames=max(max(cuts), 2), gating=True, thresholds=thr)
    y, lsnr = _run(rt, x, cuts)
    d = p.df_lookahead
    for i, (yr, lr, info) in enumerate(ref):
        assert rms(y[i] - yr) < 1e-6, (i, rms(y[i] - yr))
        acc = info["accepted"]
        live = np.zeros(T, bool)
        live[acc[d:]] = True                                  # hops that emitted a net position
        assert np.abs(lsnr[i] - lr)[live].max() < 1e-3
        frozen = np.ones(T, bool)
        frozen[acc] = False
        assert np.all(lsnr[i][frozen] == -15.0)
        assert np.all(y[i].reshape(T, HOP)[frozen] == 0.0)
    if backend == "emu":
        return
    # same handle after reset, other cut pattern: identical bits (the gate buffers are part of the reset)
    rt.reset()
    y2, _ = _run(rt, x, [2] * (T // 2))
    assert np.array_equal(y, y2)
    # gating off again: back to the ungated runtime (== batch path delayed)
    rt.reset()
    rt.set_gating(False)
    y3, _ = _run(rt, x, cuts)
    ungated = np.stack([S.process_stream(p, sd, xi, thresholds=(-1e9, 1e9, 1e9))[0] for xi in x[[0, 2]]])
    assert rms(y3[[0, 2]] - ungated) < 1e-6
    model.check()


def test_gating_defaults_and_controls(backend):
    """Reference defaults -10 / 30 / 20 dB (tract.rs:177-189): with lsnr around 10 dB every frame runs both stages, so a gated
    runtime equals the ungated one (even on digital silence: a frame that ran stage 1 clears the skip counter, :562-564).  With
    every lsnr above max_db_erb_thresh nothing is applied: the output is the input delayed by fft-hop samples, and a silent stream
    is frozen once its counter (+1 per silent hop, +1 per hop without gains) has passed 5 - answered with zeros and -15 dB."""
    from deepfilternet_amd.enhance import init_df
    from deepfilternet_amd.streaming import DfStream

    p = named_params("defaults")
    model, df_state, _, _ = init_df(params=p, epoch="none", seed=3)
    T = 4 if backend == "emu" else 8
    rng = np.random.default_rng(4)
    x = (0.1 * rng.standard_normal((2, HOP * T))).astype(np.float32)
    x[1] = 0
    rt = DfStream(model, df_state, streams=2, max_frames=2)
    y0, _ = _run(rt, x, [2] * (T // 2))
    rt.reset()
    rt.set_gating(True)
    y1, l1 = _run(rt, x, [2] * (T // 2))
    # (to rounding, not to the bit: a gated handle walks hop by hop and its GRU layers then run as the one-step kernel, the ungated
    # two-hop calls above as projection + recurrence)
    assert rms(y0 - y1) < 1e-6 * max(rms(y0), 1e-3) and np.array_equal(y0[1], y1[1]) and np.all(l1 != -15.0)
    rt.reset()
    rt.set_thresholds(-1e9, -1e9, -1e9)
    y2, l2 = _run(rt, x, [2] * (T // 2))
    d = p.fft_size - p.hop_size
    assert rms(y2[0, d:] - x[0, :-d]) < 1e-6                               # nothing applied: STFT -> ISTFT
    assert np.all(y2[1] == 0) and np.all(l2[1, 3:] == -15.0) and np.all(l2[1, :3] != -15.0)
    sd = torch_sd(p, 3)
    for i in range(2):
        yr, lr, _ = S.process_stream(p, sd, x[i], thresholds=(-1e9, -1e9, -1e9))
        assert rms(y2[i] - yr) < 1e-6 and np.abs(l2[i] - lr).max() < 1e-3
    # attenuation limit ~0 dB: pass-through, lsnr = 35 (tract.rs:540-543) — behind the silent-input test (:513-525), so the stream that
    # has been silent for more than 5 hops keeps answering zeros / -15 dB
    rt.set_atten_lim(0.0)
    y3, l3 = rt.process(torch.from_numpy(x[:, :HOP]), return_lsnr=True)
    assert np.array_equal(y3.numpy(), x[:, :HOP]) and float(l3[0, 0]) == 35.0 and float(l3[1, 0]) == -15.0


@pytest.mark.parametrize("reduce_mask,gating", [("mean", True), ("max", False)])
def test_multichannel_streams(backend, reduce_mask, gating):
    """RuntimeParams::n_ch (tract.rs:119-176): rows [2k, 2k+1] are the two channels of stream k — own STFT / network state per channel,
    one reduced ERB mask per stream (ReduceMask, :96-118,868-902), one stage decision (channel 0's lsnr) and one silent-input counter
    per stream."""
    from deepfilternet_amd.enhance import init_df
    from deepfilternet_amd.streaming import DfStream

    if emu_subset(backend) and not gating:
        pytest.skip("interpreter subset: the gated mean-reduction case runs here (DFX_EMU_ALL=1 runs both)")
    p = named_params("pf32")
    sd = torch_sd(p, 9)
    model, df_state, _, _ = init_df(params=p, epoch="none", seed=9)
    T = 12 if backend == "emu" else 24
    rng = np.random.default_rng(5)
    x = (0.1 * rng.standard_normal((4, HOP * T))).astype(np.float32)
    x[1] *= 0.3                                        # the channels of a stream differ
    x[2:, : HOP * 8] = 0                               # stream 1 starts with silence on both channels
    x[3, HOP * 2: HOP * 3] = 0.05                      # ... except one hop on its second channel: the fold runs over all channels
    if gating:
        thr = _thresholds(p, sd, x[[0, 2]], (0.15, 0.85, 0.5))
    else:
        thr = (-1e9, 1e9, 1e9)
    ref = [S.process_stream(p, sd, x[2 * k: 2 * k + 2], thresholds=thr, reduce_mask=reduce_mask) for k in range(2)]
    if gating:
        for r in ref:
            v = np.asarray(r[2]["lsnr_pass1"])
            assert min(np.abs(v - t).min() for t in thr) > 1e-4
        assert len(ref[1][2]["accepted"]) < T          # stream 1 was frozen for a while, as a whole
    rt = DfStream(model, df_state, streams=4, max_frames=2, gating=gating, thresholds=thr if gating else None, channels=2,
                  reduce_mask=reduce_mask)
    y, lsnr = _run(rt, x, [2] * (T // 2))
    for k in range(2):
        assert rms(y[2 * k: 2 * k + 2] - ref[k][0]) < 1e-6, (k, rms(y[2 * k: 2 * k + 2] - ref[k][0]))
    # the reduction really couples the channels: a mono run of channel 0 differs
    mono = S.process_stream(p, sd, x[0], thresholds=thr)[0]
    assert rms(mono - ref[0][0][0]) > 1e-5
    with pytest.raises(RuntimeError):
        DfStream(model, df_state, streams=3, channels=2)
