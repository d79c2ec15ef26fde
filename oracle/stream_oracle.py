"""CPU restatement of the reference's real-time frame loop ``DfTract::process`` — TEST INFRASTRUCTURE ONLY.

Follows /root/reference/libDF/src/tract.rs:
  * :509-525  silent-input shortcut: ``rms = sum(x^2)/len`` (a mean square, accumulated sequentially in f32); below 1e-7 the
              skip counter goes up, otherwise it is cleared; above 5 the hop is answered with zeros and lsnr = -15 and *nothing*
              else runs (no STFT, no state moves);
  * :540-543  attenuation limit 1.0 (|dB| < 0.01): the hop is passed through, lsnr = 35;
  * :545-567  stage 1: gains present -> mask applied and the skip counter cleared; gains absent (lsnr above the ERB threshold)
              -> counter += 1;
  * :571-581  stage 2: DF only when its coefficients were produced;
  * :603-610  post filter only when stage 1 ran (``apply_erb``) and it is switched on — libDF's own ``post_filter`` (lib.rs:446-471): it
              walks the frame's flattened [ch * F] bins in chunks_exact(4), i.e. the last (ch * F) % 4 bins are NOT filtered (mono,
              F = 481: bin 480), and has no clamp on the sine;
  * :612-616  attenuation-limit mix;
  * :658-672  ``apply_stages``: lsnr < min_db_thresh -> zero mask, no DF; > max_db_erb_thresh -> nothing; > max_db_df_thresh ->
              mask only; else mask + DF.  Defaults -10 / 30 / 20 dB (:177-189).

Third-party boundary: the three sub-networks are *pulsed* tract 0.21.4 models (tract.rs:769-999, not available here), each a
stateful runner that only advances when it is run.  A decoder that is skipped for a frame therefore keeps its state (GRU hidden
state, the (kt-1)-frame delay line in front of ``df_convp``): each decoder sees the *compacted* sequence of the frames it ran on —
the same thing the reference's PyTorch model does under ``lsnr_dropout`` (deepfilternet3.py:413-441: ``emb[:, idcs]``,
``c0[:, :, idcs]``).  This restatement expresses exactly that with the batch oracle (dfnet_oracle, pinned against the reference's
goldens): encoder over the accepted hops, ERB decoder over the stage-1 frames, DF decoder over the stage-2 frames.
Pinning: the decoder compaction (:func:`decode_stages`) is checked against the reference's own ``DfNet.forward`` with
``lsnr_dropout=True`` (tests/golden/dfnet_lsnr_dropout_*.npz, tools/gen_golden_r2.py; tests/test_lsnr_dropout.py): same frames kept,
same mask / coefficients on them, zeros elsewhere.  With every threshold at +-inf this oracle reduces to the batch path delayed by the
lookahead, which is pinned too.  What stays **unpinned** is tract's own arithmetic (no Rust toolchain, no ONNX models in the
container) and the post filter's Rust formula order, restated in oracle/df_oracle.c from lib.rs:446-471.

Warm-up: like dfx_stream_process the first ``lookahead`` accepted hops of a stream produce silence and take no stage decision.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from deepfilternet_amd.config import ModelParams

from . import dfnet_oracle as O
from . import libdf_oracle as L

MIN_DB_THRESH, MAX_DB_ERB_THRESH, MAX_DB_DF_THRESH = -10.0, 30.0, 20.0   # RuntimeParams::default_with_ch, tract.rs:177-189


def apply_stages(lsnr: float, thr: Tuple[float, float, float]) -> Tuple[bool, bool, bool]:
    """tract.rs:658-672 -> (apply_gains, apply_gain_zeros, apply_df)."""
    if lsnr < thr[0]:
        return False, True, False
    if lsnr > thr[1]:
        return False, False, False
    if lsnr > thr[2]:
        return True, False, False
    return True, False, True


def hop_mean_square(x: np.ndarray) -> np.float32:
    """tract.rs:513-516: fold over the hop in f32, acc + x.powi(2), divided by the length.  x: [hop] or [ch, hop] (the reference
    iterates its [ch, hop] array in memory order: channel after channel)."""
    e = np.float32(0.0)
    flat = np.asarray(x, dtype=np.float32).reshape(-1)
    for v in flat:
        e = np.float32(e + np.float32(v * v))
    return np.float32(e / np.float32(flat.size))


def _features(p: ModelParams, hops: np.ndarray):
    """hops [K, ch, hop] -> (DF state, spec [ch, K, F], erb feat [ch, K, E], spec feat [ch, K, F']); channels are independent (each
    has its own DFState in the reference, tract.rs:424-436)."""
    st = L.DF(p.sr, p.fft_size, p.hop_size, p.nb_erb, p.min_nb_freqs)
    audio = np.ascontiguousarray(hops.transpose(1, 0, 2).reshape(hops.shape[1], -1), dtype=np.float32)
    spec, fe, fs = O.df_features(L, audio, st, p.nb_df, p.norm_alpha())
    return st, spec, fe, fs


def _encoder(p: ModelParams, sd, fe: np.ndarray, fs: np.ndarray):
    fe_t = O.pad_feat(torch.from_numpy(fe).unsqueeze(1), p.conv_lookahead)
    fs_t = torch.view_as_real(torch.from_numpy(fs)).permute(0, 3, 1, 2)
    fs_t = O.pad_feat(fs_t, p.conv_lookahead)
    return O.dfnet_encoder(p, sd, fe_t, fs_t)


@torch.no_grad()
def decode_stages(p: ModelParams, sd, enc: Dict[str, torch.Tensor], flags, ch: int = 1, reduce_mask: str = "mean"):
    """The two decoders on the frames their stage runs on — each decoder sees the COMPACTED sequence of its frames (a pulsed tract
    model only advances when it is run; deepfilternet3.py:413-441 states the same compaction with ``emb[:, idcs]`` / ``c0[:, :, idcs]``).
    enc: dfnet_encoder() output over all positions; flags[q] = apply_stages(lsnr[q]).
    -> gains [ch, P, E] (ones where no gains are applied, zeros for the zero mask), coefs complex [ch, O, nD, F'] for the stage-2
       positions idx_d (None if there are none), idx_g, idx_d."""
    P = len(flags)
    gains = torch.ones(ch, P, p.nb_erb)
    idx_g = [q for q in range(P) if flags[q][0]]
    idx_d = [q for q in range(P) if flags[q][2]]
    for q in range(P):
        if flags[q][1]:
            gains[:, q] = 0.0
    if idx_g:
        ig = torch.as_tensor(idx_g)
        m = O.dfnet_erb_decoder(p, sd, enc["emb"][:, ig], enc["e3"][:, :, ig], enc["e2"][:, :, ig], enc["e1"][:, :, ig],
                                enc["e0"][:, :, ig])["m"][:, 0]      # [ch, nG, E]
        if ch > 1 and reduce_mask == "mean":
            acc = m[0].clone()
            for c in range(1, ch):
                acc = acc + m[c]
            m = (acc * np.float32(1.0 / ch)).unsqueeze(0).expand(ch, -1, -1)
        elif ch > 1 and reduce_mask == "max":
            m = m.max(dim=0, keepdim=True).values.expand(ch, -1, -1)
        gains[:, ig] = m
    coefs = None
    if idx_d:
        idd = torch.as_tensor(idx_d)
        c = O.dfnet_df_decoder(p, sd, enc["emb"][:, idd], enc["c0"][:, :, idd])["df_coefs"]   # [ch,O,nD,F',2]
        coefs = torch.view_as_complex(c.contiguous())
    return gains, coefs, idx_g, idx_d


@torch.no_grad()
def process_stream(p: ModelParams, sd: Dict[str, torch.Tensor], x: np.ndarray, atten_lim_db: Optional[float] = None,
                   pf_beta: Optional[float] = None,
                   thresholds: Tuple[float, float, float] = (MIN_DB_THRESH, MAX_DB_ERB_THRESH, MAX_DB_DF_THRESH),
                   reduce_mask: str = "mean", pf_like_torch: bool = False):
    """One stream, hop by hop.  (pf_like_torch: the PyTorch model's post filter on every bin instead of libDF's — only there to show
    that tests can tell the two apart.)  x f32 [n_hops*hop] (mono) or [ch, n_hops*hop] -> (y like x, lsnr f32 [n_hops], info dict).

    Multi-channel (tract.rs:119-176 ``n_ch``, :96-118,868-902 ``ReduceMask``): per-channel STFT / features / network state; the
    silent-input fold runs over all channels of the hop; the stage decision is taken from channel 0's lsnr (:468 ``to_scalar``); the
    ERB decoder's masks are reduced over the channels ("mean": sum * (1/ch); "max"; "none") and the reduced mask is applied to every
    channel (:547-556).
    info["accepted"]: hop indices that were processed; info["flags"]: per net position (apply_gains, zeros, apply_df)."""
    assert p.conv_lookahead == p.df_lookahead, "like dfx_stream_create"
    mono = x.ndim == 1
    x2 = np.asarray(x, dtype=np.float32).reshape(1, -1) if mono else np.asarray(x, dtype=np.float32)
    ch = x2.shape[0]
    hop, Lk, O_ = p.hop_size, p.df_lookahead, p.df_order
    n_hops = x2.shape[1] // hop
    hops = np.ascontiguousarray(x2[:, : n_hops * hop].reshape(ch, n_hops, hop).transpose(1, 0, 2))   # [n_hops, ch, hop]
    y = np.zeros_like(hops)
    lsnr_out = np.zeros(n_hops, dtype=np.float32)

    def ret(yh, ls, info):
        out = np.ascontiguousarray(yh.transpose(1, 0, 2).reshape(ch, -1))
        return (out[0] if mono else out), ls, info

    lim = None
    if atten_lim_db is not None:
        a = abs(atten_lim_db)
        lim = None if a >= 100 else (1.0 if a < 0.01 else float(np.float32(10.0) ** np.float32(-a / 20.0)))
    if lim == 1.0:  # tract.rs:509-543: the silent-input test still runs (zeros / -15 dB), every other hop is passed through with lsnr = 35
        yh, ls, skip_counter = hops.copy(), np.full(n_hops, 35.0, np.float32), 0
        for a in range(n_hops):
            skip_counter = skip_counter + 1 if hop_mean_square(hops[a]) < np.float32(1e-7) else 0
            if skip_counter > 5:
                yh[a], ls[a] = 0.0, -15.0
        return ret(yh, ls, {"accepted": [], "flags": []})
    beta = pf_beta if pf_beta is not None else (p.pf_beta if p.mask_pf else 0.0)
    # ---- pass 1: which hops are processed.  The decision for hop a depends on the lsnr of earlier positions (counter += 1 when the
    # gains were skipped), which is causal: the encoder run on the accepted prefix gives it (prefix property of the batch path).
    accepted, skip_counter, lsnr_pos = [], 0, []
    for a in range(n_hops):
        if hop_mean_square(hops[a]) < np.float32(1e-7):
            skip_counter += 1
        else:
            skip_counter = 0
        if skip_counter > 5:
            lsnr_out[a] = -15.0
            continue
        accepted.append(a)
        k = len(accepted) - 1
        pos = k - Lk
        if pos < 0:
            continue
        _, _, fe, fs = _features(p, hops[accepted])
        v = float(_encoder(p, sd, fe, fs)["lsnr"][0, pos, 0])
        lsnr_pos.append(v)
        g, z, _ = apply_stages(v, thresholds)
        skip_counter = 0 if (g or z) else skip_counter + 1
    K = len(accepted)
    if K == 0:
        return ret(y, lsnr_out, {"accepted": [], "flags": []})
    # ---- pass 2: the accepted hops as one sequence; positions 0 .. K-1-Lk are emitted at steps Lk .. K-1
    st, spec, fe, fs = _features(p, hops[accepted])
    enc = _encoder(p, sd, fe, fs)
    P = max(K - Lk, 0)
    lsnr = enc["lsnr"][0, :, 0].numpy()
    flags = [apply_stages(float(lsnr[q]), thresholds) for q in range(P)]
    F = p.fft_size // 2 + 1
    widths = st.erb_widths()
    spec_t = torch.from_numpy(spec)                                    # [ch, K, F] complex
    gains, coefs, idx_g, idx_d = decode_stages(p, sd, enc, flags, ch, reduce_mask)
    spec_e = spec_t[:, :P] * O.band_gain(gains, widths)                # mask on frame q (rolling_spec_buf_y[df_order-1])
    if idx_d:
        xp = torch.view_as_real(spec_t[:, :, : p.nb_df])
        xp = torch.nn.functional.pad(xp, (0, 0, 0, 0, O_ - 1 - Lk, Lk))
        xp = torch.view_as_complex(xp.contiguous())                    # frame q + n - (O-1-la) at row q + n
        for j, q in enumerate(idx_d):
            acc = torch.zeros(ch, p.nb_df, dtype=spec_t.dtype)
            for n in range(O_):
                acc = acc + coefs[:, n, j] * xp[:, q + n]
            spec_e[:, q, : p.nb_df] = acc
    if beta > 0:
        for q in idx_g:                                                # tract.rs:603-610: only when stage 1 ran; lib.rs:446-471
            if pf_like_torch:
                spec_e[:, q] = O.post_filter(spec_t[:, q], spec_e[:, q], beta)
                continue
            flat = L.post_filter(spec_t[:, q].numpy().reshape(1, -1), spec_e[:, q].numpy().reshape(1, -1), beta)
            spec_e[:, q] = torch.from_numpy(flat.reshape(ch, F))
    if lim is not None:
        spec_e = spec_t[:, :P] * lim + spec_e * (1 - lim)
    out_spec = np.zeros((ch, K, F), dtype=np.complex64)
    out_spec[:, Lk:] = spec_e.numpy()
    ys = st.synthesis(out_spec).reshape(ch, K, hop)
    for k, a in enumerate(accepted):
        y[a] = ys[:, k]
        if k >= Lk:
            lsnr_out[a] = lsnr[k - Lk]
    return ret(y, lsnr_out, {"accepted": accepted, "flags": flags, "lsnr_pass1": lsnr_pos})


@torch.no_grad()
def process_raw_frames(p: ModelParams, sd: Dict[str, torch.Tensor], spec: np.ndarray,
                       thresholds: Tuple[float, float, float] = (MIN_DB_THRESH, MAX_DB_ERB_THRESH, MAX_DB_DF_THRESH)):
    """``DfTract::process_raw`` (tract.rs:441-507; df_process_frame_raw, capi.rs:172-210) frame after frame for one mono stream:
    spec complex64 [K, F] -> per call k a tuple (lsnr, gains [E] or None, coefs complex [O, F'] or None); the first ``lookahead`` calls
    yield (None, None, None) like the warm-up of dfx_stream_process_raw.  Features: lib.rs:206-217 with running means; stages and
    decoder-state semantics as in :func:`process_stream`."""
    K = spec.shape[0]
    Lk = p.df_lookahead
    spec1 = np.ascontiguousarray(spec[None].astype(np.complex64))
    widths = L.erb_fb_widths(p.sr, p.fft_size, p.nb_erb, p.min_nb_freqs)
    fe = L.erb_norm(L.erb(spec1, widths), p.norm_alpha())
    fs = L.unit_norm(np.ascontiguousarray(spec1[..., : p.nb_df]), p.norm_alpha())
    enc = _encoder(p, sd, fe, fs)
    P = max(K - Lk, 0)
    lsnr = enc["lsnr"][0, :, 0].numpy()
    flags = [apply_stages(float(lsnr[q]), thresholds) for q in range(P)]
    g_all, c_all, idx_g, idx_d = decode_stages(p, sd, enc, flags)
    gains = {q: g_all[0, q].numpy() for q in range(P) if flags[q][0] or flags[q][1]}     # the mask, or zeros below min_db_thresh
    coefs = {q: c_all[0, :, j].numpy() for j, q in enumerate(idx_d)}
    out = [(None, None, None)] * min(Lk, K)
    for q in range(P):
        out.append((float(lsnr[q]), gains.get(q), coefs.get(q)))
    return out
