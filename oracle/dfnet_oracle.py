"""torch-fp32 CPU restatement of DeepFilterNet3's forward pass and of ``enhance()`` — TEST INFRASTRUCTURE ONLY.

This is the checker for the HIP engine's DNN half (floating point, so a torch fp32 reference is the right oracle).
It is a *functional* restatement driven by a plain state-dict (reference key names) — it does not copy the reference's
``nn.Module`` classes.  Third-party arithmetic at the boundary is PyTorch's ATen CPU kernels (conv2d, conv_transpose2d,
GRU via ``torch.nn.GRU``), exactly what the reference itself calls.

Pinned against the reference's own modules by tools/gen_golden.py (run in the build container, where
/root/reference is importable): same seeded state-dict loaded with ``strict=True`` into ``df.deepfilternet3.DfNet``,
outputs compared and committed under tests/golden/.  For *pretrained* weights parity stays unpinned until the
checkpoints (missing blobs here) are available.

Each function cites the reference lines it follows (relative to /root/reference/DeepFilterNet/df).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

from deepfilternet_amd.config import ModelParams

BN_EPS = 1e-5  # torch.nn.BatchNorm2d default


def _t(sd, name) -> Tensor:
    v = sd[name]
    return v if isinstance(v, Tensor) else torch.as_tensor(np.asarray(v))


def _bn(x: Tensor, sd, prefix: str) -> Tensor:
    """nn.BatchNorm2d in eval mode."""
    return F.batch_norm(x, _t(sd, prefix + ".running_mean"), _t(sd, prefix + ".running_var"),
                        _t(sd, prefix + ".weight"), _t(sd, prefix + ".bias"), training=False, eps=BN_EPS)


def conv_norm_act(x: Tensor, sd, prefix: str, in_ch: int, out_ch: int, kernel: Tuple[int, int], fstride: int = 1,
                  act: str = "relu") -> Tensor:
    """modules.py:18-72 Conv2dNormAct(bias=False, separable=True): causal time pad, grouped conv, optional 1x1, BN, act.
    x: [B, C, T, F]."""
    kt, kf = kernel
    idx = 0
    if kt > 1:
        x = F.pad(x, (0, 0, kt - 1, 0))  # ConstantPad2d((0, 0, kt-1, 0)): modules.py:45-48
        idx = 1
    groups = math.gcd(in_ch, out_ch)
    sep = groups > 1 and max(kernel) > 1
    x = F.conv2d(x, _t(sd, f"{prefix}.{idx}.weight"), None, stride=(1, fstride), padding=(0, kf // 2), groups=groups)
    idx += 1
    if sep:
        x = F.conv2d(x, _t(sd, f"{prefix}.{idx}.weight"))
        idx += 1
    x = _bn(x, sd, f"{prefix}.{idx}")
    if act == "relu":
        return torch.relu(x)
    if act == "sigmoid":
        return torch.sigmoid(x)
    raise ValueError(act)


def convt_norm_act(x: Tensor, sd, prefix: str, ch: int, kernel: Tuple[int, int], fstride: int) -> Tensor:
    """modules.py:75-126 ConvTranspose2dNormAct(bias=False, separable=True) with in_ch == out_ch (depthwise)."""
    kt, kf = kernel
    assert kt == 1
    x = F.conv_transpose2d(x, _t(sd, f"{prefix}.0.weight"), None, stride=(1, fstride), padding=(kt - 1, kf // 2),
                           output_padding=(0, kf // 2), groups=ch)
    x = F.conv2d(x, _t(sd, f"{prefix}.1.weight"))
    return torch.relu(_bn(x, sd, f"{prefix}.2"))


def grouped_linear(x: Tensor, w: Tensor) -> Tensor:
    """modules.py:741-780 GroupedLinearEinsum: x [B,T,I], w [G, I/G, H/G] -> [B,T,H]."""
    b, t, _ = x.shape
    g = w.shape[0]
    return torch.einsum("btgi,gih->btgh", x.view(b, t, g, -1), w).flatten(2, 3)


def gru_manual(x: Tensor, sd, prefix: str, layers: int, h0: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Explicit GRU recurrence (PyTorch gate order r,z,n).  x: [B,T,H].  Slow: small cases only."""
    B, T, H = x.shape
    hs = []
    for l in range(layers):
        w_ih, w_hh = _t(sd, f"{prefix}.weight_ih_l{l}"), _t(sd, f"{prefix}.weight_hh_l{l}")
        b_ih, b_hh = _t(sd, f"{prefix}.bias_ih_l{l}"), _t(sd, f"{prefix}.bias_hh_l{l}")
        h = torch.zeros(B, H) if h0 is None else h0[l]
        ys = []
        for t in range(T):
            gi = x[:, t] @ w_ih.t() + b_ih
            gh = h @ w_hh.t() + b_hh
            r = torch.sigmoid(gi[:, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            h = (1 - z) * n + z * h
            ys.append(h)
        x = torch.stack(ys, 1)
        hs.append(h)
    return x, torch.stack(hs, 0)


def gru_aten(x: Tensor, sd, prefix: str, layers: int) -> Tensor:
    """torch.nn.GRU(batch_first=True), h0 = 0 — the ATen kernel the reference calls (modules.py:721)."""
    H = x.shape[-1]
    g = torch.nn.GRU(H, H, num_layers=layers, batch_first=True)
    with torch.no_grad():
        for l in range(layers):
            for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                getattr(g, f"{n}_l{l}").copy_(_t(sd, f"{prefix}.{n}_l{l}"))
    g.eval()
    with torch.no_grad():
        y, _ = g(x)
    return y


def squeezed_gru(x: Tensor, sd, prefix: str, layers: int, has_out: bool, manual: bool = False, skip: str = "none") -> Tensor:
    """modules.py:702-738 SqueezedGRU_S with linear_act_layer=ReLU; gru_skip_op none / identity / GroupedLinearEinsum: the skip takes
    the module's INPUT and joins after linear_out (:733-737)."""
    y = torch.relu(grouped_linear(x, _t(sd, f"{prefix}.linear_in.0.weight")))
    y = gru_manual(y, sd, f"{prefix}.gru", layers)[0] if manual else gru_aten(y, sd, f"{prefix}.gru", layers)
    if has_out:
        y = torch.relu(grouped_linear(y, _t(sd, f"{prefix}.linear_out.0.weight")))
    if skip == "identity":
        y = y + x
    elif skip == "groupedlinear":
        y = y + grouped_linear(x, _t(sd, f"{prefix}.gru_skip.weight"))
    return y


def pad_feat(x: Tensor, lookahead: int) -> Tensor:
    """deepfilternet3.py:357-361 ConstantPad2d((0, 0, -L, L)): drop the first L frames, append L zero frames."""
    if lookahead <= 0:
        return x
    return F.pad(x, (0, 0, -lookahead, lookahead))


def df_apply(spec: Tensor, coefs: Tensor, order: int, lookahead: int, nb_df: int) -> Tensor:
    """multiframe.py:85-95,126-136,169-180 MF.DF.forward.

    spec: complex [B, T, F]; coefs: complex [B, O, T, nb_df].  Returns complex [B, T, nb_df]:
    Y[b,t,f] = sum_n coefs[b,n,t,f] * spec[b, t + n - (O-1-lookahead), f]   (zero outside [0,T)).
    """
    B, T, _ = spec.shape
    x = spec[..., :nb_df]
    xp = F.pad(torch.view_as_real(x), (0, 0, 0, 0, order - 1 - lookahead, lookahead))
    xp = torch.view_as_complex(xp.contiguous())
    out = torch.zeros(B, T, nb_df, dtype=spec.dtype)
    for n in range(order):
        out = out + coefs[:, n] * xp[:, n:n + T]
    return out


def band_gain(mask: Tensor, widths: np.ndarray) -> Tensor:
    """modules.py:266-269 Mask: m.matmul(erb_inv_fb) with the 0/1 [E,F] matrix == per-band repeat (lib.rs:314-326)."""
    return torch.repeat_interleave(mask, torch.as_tensor(np.asarray(widths, dtype=np.int64)), dim=-1)


def post_filter(spec: Tensor, spec_e: Tensor, beta: float) -> Tensor:
    """deepfilternet3.py:448-454.  spec, spec_e complex [..., F]."""
    eps = 1e-12
    pi = 3.1415926535897932384626433
    mask = (spec_e.abs() / spec.abs().add(eps)).clamp(eps, 1)
    mask_sin = mask * torch.sin(pi * mask / 2).clamp_min(eps)
    pf = (1 + beta) / (1 + beta * mask.div(mask_sin).pow(2))
    return spec_e * pf


@torch.no_grad()
def dfnet_encoder(p: ModelParams, sd, fe: Tensor, fs: Tensor, manual_gru: bool = False) -> Dict[str, Tensor]:
    """deepfilternet3.py:166-185 Encoder.forward.  fe [B,1,T,E], fs [B,2,T,F'] (already shifted by pad_feat)."""
    C = p.conv_ch
    ck, cki = tuple(p.conv_kernel), tuple(p.conv_kernel_inp)
    e0 = conv_norm_act(fe, sd, "enc.erb_conv0", 1, C, cki)
    e1 = conv_norm_act(e0, sd, "enc.erb_conv1", C, C, ck, fstride=2)
    e2 = conv_norm_act(e1, sd, "enc.erb_conv2", C, C, ck, fstride=2)
    e3 = conv_norm_act(e2, sd, "enc.erb_conv3", C, C, ck, fstride=1)
    c0 = conv_norm_act(fs, sd, "enc.df_conv0", 2, C, cki)
    c1 = conv_norm_act(c0, sd, "enc.df_conv1", C, C, ck, fstride=2)
    cemb = c1.permute(0, 2, 3, 1).flatten(2)
    cemb = torch.relu(grouped_linear(cemb, _t(sd, "enc.df_fc_emb.0.weight")))
    e3f = e3.permute(0, 2, 3, 1).flatten(2)
    emb_in = torch.cat((e3f, cemb), dim=-1) if p.enc_concat else e3f + cemb    # :132-136,181 Concat / Add
    emb = squeezed_gru(emb_in, sd, "enc.emb_gru", 1, True, manual_gru, skip=p.emb_gru_skip_enc)
    lsnr = torch.sigmoid(F.linear(emb, _t(sd, "enc.lsnr_fc.0.weight"), _t(sd, "enc.lsnr_fc.0.bias")))
    lsnr = lsnr * (p.lsnr_max - p.lsnr_min) + p.lsnr_min
    return {"e0": e0, "e1": e1, "e2": e2, "e3": e3, "c0": c0, "c1": c1, "cemb": cemb, "emb_in": emb_in, "emb": emb, "lsnr": lsnr}


@torch.no_grad()
def dfnet_erb_decoder(p: ModelParams, sd, emb: Tensor, e3: Tensor, e2: Tensor, e1: Tensor, e0: Tensor,
                      manual_gru: bool = False) -> Dict[str, Tensor]:
    """deepfilternet3.py:245-254 ErbDecoder.forward -> m [B,1,T,E]."""
    C = p.conv_ch
    ck = tuple(p.conv_kernel)
    b, _, t, f8 = e3.shape
    d_emb = squeezed_gru(emb, sd, "erb_dec.emb_gru", p.emb_num_layers - 1, True, manual_gru, skip=p.emb_gru_skip)
    d_emb = d_emb.view(b, t, f8, -1).permute(0, 3, 1, 2)
    d3 = conv_norm_act(conv_norm_act(e3, sd, "erb_dec.conv3p", C, C, (1, 1)) + d_emb, sd, "erb_dec.convt3", C, C, ck)
    d2 = convt_norm_act(conv_norm_act(e2, sd, "erb_dec.conv2p", C, C, (1, 1)) + d3, sd, "erb_dec.convt2", C,
                        tuple(p.convt_kernel), 2)
    d1 = convt_norm_act(conv_norm_act(e1, sd, "erb_dec.conv1p", C, C, (1, 1)) + d2, sd, "erb_dec.convt1", C,
                        tuple(p.convt_kernel), 2)
    m = conv_norm_act(conv_norm_act(e0, sd, "erb_dec.conv0p", C, C, (1, 1)) + d1, sd, "erb_dec.conv0_out", C, 1, ck,
                      act="sigmoid")
    return {"m": m, "d3": d3, "d2": d2, "d1": d1}


@torch.no_grad()
def dfnet_df_decoder(p: ModelParams, sd, emb: Tensor, c0: Tensor, manual_gru: bool = False) -> Dict[str, Tensor]:
    """deepfilternet3.py:323-331 DfDecoder.forward + DfOutputReshapeMF :268-275 -> df_coefs [B,O,T,F',2]."""
    C, O = p.conv_ch, p.df_order
    b, t, _ = emb.shape
    c = squeezed_gru(emb, sd, "df_dec.df_gru", p.df_num_layers, False, manual_gru)
    if p.df_gru_skip == "groupedlinear":
        c = c + grouped_linear(emb, _t(sd, "df_dec.df_skip.weight"))
    elif p.df_gru_skip == "identity":
        c = c + emb
    c0p = conv_norm_act(c0, sd, "df_dec.df_convp", C, 2 * O, (p.df_pathway_kernel_size_t, 1)).permute(0, 2, 3, 1)
    c = torch.tanh(grouped_linear(c, _t(sd, "df_dec.df_out.0.weight")))
    c = c.view(b, t, p.nb_df, 2 * O) + c0p                      # [B,T,F',2O]
    df_coefs = c.view(b, t, p.nb_df, O, 2).permute(0, 3, 1, 2, 4)
    return {"df_coefs": df_coefs.contiguous(), "c0p": c0p, "coefs_raw": c}


@torch.no_grad()
def dfnet_forward(p: ModelParams, sd: Dict[str, Tensor], widths: np.ndarray, spec: Tensor, feat_erb: Tensor,
                  feat_spec: Tensor, manual_gru: bool = False, run_df: bool = True) -> Dict[str, Tensor]:
    """deepfilternet3.py:389-456 DfNet.forward (lsnr_dropout=False path); run_df=False: :433-446 (mask only, no coefficients).

    spec [B,1,T,F,2], feat_erb [B,1,T,E], feat_spec [B,1,T,F',2]  (all float32).
    Returns dict with spec_e [B,1,T,F,2], m [B,1,T,E], lsnr [B,T,1], df_coefs [B,O,T,F',2] plus intermediates.
    """
    O = p.df_order
    fs = feat_spec.squeeze(1).permute(0, 3, 1, 2)  # [B,2,T,F']   :407
    fe = pad_feat(feat_erb, p.conv_lookahead)      # :409
    fs = pad_feat(fs, p.conv_lookahead)            # :410
    enc = dfnet_encoder(p, sd, fe, fs, manual_gru)
    dec = dfnet_erb_decoder(p, sd, enc["emb"], enc["e3"], enc["e2"], enc["e1"], enc["e0"], manual_gru)
    m = dec["m"]
    # Mask :248-269 (no post filter / atten_lim inside the module for DF3)
    spec_c = torch.view_as_complex(spec.squeeze(1).contiguous())  # [B,T,F]
    spec_m = spec_c * band_gain(m.squeeze(1), widths)
    spec_e = spec_m.clone()
    dfd = {}
    if run_df:
        dfd = dfnet_df_decoder(p, sd, enc["emb"], enc["c0"], manual_gru)
        coefs_c = torch.view_as_complex(dfd["df_coefs"])
        # MF.DF on the *noisy* spec (:442), then high bins from the masked spec (:443)
        spec_e[..., : p.nb_df] = df_apply(spec_c, coefs_c, O, p.df_lookahead, p.nb_df)
    if p.mask_pf:
        spec_e = post_filter(spec_c, spec_e, p.pf_beta)
    out = {"spec_e": torch.view_as_real(spec_e).unsqueeze(1)}
    out.update(enc)
    out.update(dec)
    out.update(dfd)
    return out


def df_features(libdf, audio: np.ndarray, df_state, nb_df: int, alpha: float):
    """enhance.py:190-203 df_features() on numpy arrays with the oracle's libdf."""
    spec = df_state.analysis(audio)
    erb_feat = libdf.erb_norm(libdf.erb(spec, df_state.erb_widths()), alpha)
    spec_feat = libdf.unit_norm(np.ascontiguousarray(spec[..., :nb_df]), alpha)
    return spec, erb_feat, spec_feat


@torch.no_grad()
def enhance(p: ModelParams, sd: Dict[str, Tensor], audio: np.ndarray, pad: bool = True,
            atten_lim_db: Optional[float] = None, df_state=None) -> np.ndarray:
    """enhance.py:206-250 enhance() with the C oracle as libdf and dfnet_forward() as the model.  audio f32 [C,T]."""
    from . import libdf_oracle as libdf

    if df_state is None:
        df_state = libdf.DF(p.sr, p.fft_size, p.hop_size, p.nb_erb, p.min_nb_freqs)
    orig_len = audio.shape[-1]
    n_fft, hop = p.fft_size, p.hop_size
    if pad:
        audio = np.pad(audio, ((0, 0), (0, n_fft)))
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    spec, erb_feat, spec_feat = df_features(libdf, audio, df_state, p.nb_df, p.norm_alpha())
    spec_t = torch.view_as_real(torch.from_numpy(spec)).unsqueeze(1)
    out = dfnet_forward(p, sd, df_state.erb_widths(), spec_t, torch.from_numpy(erb_feat).unsqueeze(1),
                        torch.view_as_real(torch.from_numpy(spec_feat)).unsqueeze(1))
    enhanced = torch.view_as_complex(out["spec_e"].squeeze(1).contiguous())
    if atten_lim_db is not None and abs(atten_lim_db) > 0:
        lim = 10 ** (-abs(atten_lim_db) / 20)
        enhanced = torch.from_numpy(spec) * lim + enhanced * (1 - lim)
    y = df_state.synthesis(np.ascontiguousarray(enhanced.numpy()))
    if pad:
        d = n_fft - hop
        y = y[:, d: orig_len + d]
    return y
