"""DeepFilterNet3 state-dict manifest (names + shapes exactly as the reference's ``DfNet.state_dict()``) and a seeded
synthetic initialiser.

The pretrained checkpoints are missing blobs in this environment, so parity is established with the
*same seeded state-dict* loaded into the reference's PyTorch modules (tools/gen_golden.py, strict=True) and into the HIP
engine.  Names follow the reference's ``nn.Sequential`` indexing rule: a ``ConstantPad2d`` is
inserted at index 0 only when the time kernel is > 1 (modules.py:45-48), a 1x1 pointwise conv only when the block is
separable (groups > 1 and max(kernel) > 1, modules.py:49-67).

Reference structure: DeepFilterNet/df/deepfilternet3.py:100-185 (Encoder), :188-254 (ErbDecoder), :278-331 (DfDecoder),
DeepFilterNet/df/modules.py:18-126 (conv blocks), :702-738 (SqueezedGRU_S), :741-780 (GroupedLinearEinsum).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

from .config import ModelParams


def _conv_block(prefix: str, in_ch: int, out_ch: int, kernel: Tuple[int, int], transposed: bool = False,
                separable: bool = True) -> "OrderedDict[str, Tuple[int, ...]]":
    """Keys of one Conv2dNormAct / ConvTranspose2dNormAct (bias=False, BatchNorm2d, modules.py:18-126)."""
    kt, kf = kernel
    d: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    idx = 1 if kt > 1 else 0  # ConstantPad2d present only for a time kernel > 1
    groups = math.gcd(in_ch, out_ch) if separable else 1
    sep = separable and groups > 1
    if not transposed and max(kernel) == 1:
        sep = False  # modules.py:51-52 (Conv2dNormAct only)
    if transposed:
        d[f"{prefix}.{idx}.weight"] = (in_ch, out_ch // groups, kt, kf)
    else:
        d[f"{prefix}.{idx}.weight"] = (out_ch, in_ch // groups, kt, kf)
    idx += 1
    if sep:
        d[f"{prefix}.{idx}.weight"] = (out_ch, out_ch, 1, 1)
        idx += 1
    for n in ("weight", "bias", "running_mean", "running_var"):
        d[f"{prefix}.{idx}.{n}"] = (out_ch,)
    d[f"{prefix}.{idx}.num_batches_tracked"] = ()
    return d


def _gru(prefix: str, hidden: int, layers: int) -> "OrderedDict[str, Tuple[int, ...]]":
    d: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for l in range(layers):
        d[f"{prefix}.weight_ih_l{l}"] = (3 * hidden, hidden)
        d[f"{prefix}.weight_hh_l{l}"] = (3 * hidden, hidden)
        d[f"{prefix}.bias_ih_l{l}"] = (3 * hidden,)
        d[f"{prefix}.bias_hh_l{l}"] = (3 * hidden,)
    return d


def _glin(i: int, h: int, g: int) -> Tuple[int, int, int]:
    assert i % g == 0 and h % g == 0, (i, h, g)
    return (g, i // g, h // g)


def state_dict_manifest(p: ModelParams) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape in the reference's ``state_dict()`` order."""
    C, E, Fd, O = p.conv_ch, p.nb_erb, p.nb_df, p.df_order
    F = p.freq_bins
    emb = p.emb_dim
    H = p.emb_hidden_dim
    d: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    d["erb_fb"] = (F, E)
    d.update(_conv_block("enc.erb_conv0", 1, C, tuple(p.conv_kernel_inp)))
    for n in ("enc.erb_conv1", "enc.erb_conv2", "enc.erb_conv3"):
        d.update(_conv_block(n, C, C, tuple(p.conv_kernel)))
    d.update(_conv_block("enc.df_conv0", 2, C, tuple(p.conv_kernel_inp)))
    d.update(_conv_block("enc.df_conv1", C, C, tuple(p.conv_kernel)))
    d["enc.df_fc_emb.0.weight"] = _glin(C * Fd // 2, emb, p.enc_lin_groups)
    d["enc.emb_gru.linear_in.0.weight"] = _glin(2 * emb if p.enc_concat else emb, H, p.lin_groups)
    d.update(_gru("enc.emb_gru.gru", H, 1))
    if p.emb_gru_skip_enc == "groupedlinear":   # SqueezedGRU_S registers linear_in, gru, gru_skip, linear_out in this order
        d["enc.emb_gru.gru_skip.weight"] = _glin(emb, emb, p.lin_groups)
    d["enc.emb_gru.linear_out.0.weight"] = _glin(H, emb, p.lin_groups)
    d["enc.lsnr_fc.0.weight"] = (1, emb)
    d["enc.lsnr_fc.0.bias"] = (1,)
    d["erb_dec.emb_gru.linear_in.0.weight"] = _glin(emb, H, p.lin_groups)
    d.update(_gru("erb_dec.emb_gru.gru", H, p.emb_num_layers - 1))
    if p.emb_gru_skip == "groupedlinear":
        d["erb_dec.emb_gru.gru_skip.weight"] = _glin(emb, emb, p.lin_groups)
    d["erb_dec.emb_gru.linear_out.0.weight"] = _glin(H, emb, p.lin_groups)
    d.update(_conv_block("erb_dec.conv3p", C, C, (1, 1)))
    d.update(_conv_block("erb_dec.convt3", C, C, tuple(p.conv_kernel)))
    d.update(_conv_block("erb_dec.conv2p", C, C, (1, 1)))
    d.update(_conv_block("erb_dec.convt2", C, C, tuple(p.convt_kernel), transposed=True))
    d.update(_conv_block("erb_dec.conv1p", C, C, (1, 1)))
    d.update(_conv_block("erb_dec.convt1", C, C, tuple(p.convt_kernel), transposed=True))
    d.update(_conv_block("erb_dec.conv0p", C, C, (1, 1)))
    d.update(_conv_block("erb_dec.conv0_out", C, 1, tuple(p.conv_kernel)))
    d["mask.erb_inv_fb"] = (E, F)
    d.update(_conv_block("df_dec.df_convp", C, 2 * O, (p.df_pathway_kernel_size_t, 1)))
    d["df_dec.df_gru.linear_in.0.weight"] = _glin(emb, p.df_hidden_dim, 8)  # SqueezedGRU_S default linear_groups=8
    d.update(_gru("df_dec.df_gru.gru", p.df_hidden_dim, p.df_num_layers))
    if p.df_gru_skip == "groupedlinear":
        d["df_dec.df_skip.weight"] = _glin(emb, p.df_hidden_dim, p.lin_groups)
    d["df_dec.df_out.0.weight"] = _glin(p.df_hidden_dim, Fd * 2 * O, p.lin_groups)
    d["df_dec.df_fc_a.0.weight"] = (1, p.df_hidden_dim)  # constructed but unused in forward (deepfilternet3.py:321)
    d["df_dec.df_fc_a.0.bias"] = (1,)
    return d


def erb_fb_matrices(widths: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """modules.py:206-223 erb_fb(normalized=True): forward [F,E] with unit column sums, inverse [E,F] of 0/1."""
    widths = np.asarray(widths, dtype=np.int64)
    F, E = int(widths.sum()), len(widths)
    fb = np.zeros((F, E), dtype=np.float32)
    b = 0
    for i, w in enumerate(widths.tolist()):
        fb[b:b + w, i] = 1.0
        b += w
    inv = fb.T.copy()
    fwd = fb / fb.sum(axis=0, keepdims=True)
    return fwd.astype(np.float32), inv.astype(np.float32)


def random_state_dict(p: ModelParams, seed: int = 0, widths: np.ndarray | None = None) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights (numpy PCG64: stable across library versions), float32, reference key names.

    BatchNorm statistics are randomised (running_mean ~ N(0, 0.1), running_var ~ U(0.5, 1.5), gamma ~ U(0.5, 1.5),
    beta ~ N(0, 0.1)) so that BN folding is actually exercised.  Conv / linear / GRU weights use the
    fan-in uniform ranges of PyTorch's default initialisers so activations stay O(1).
    """
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = OrderedDict()
    man = state_dict_manifest(p)
    for name, shape in man.items():
        leaf = name.rsplit(".", 1)[-1]
        if name in ("erb_fb", "mask.erb_inv_fb"):
            continue
        if leaf == "num_batches_tracked":
            out[name] = np.array(0, dtype=np.int64)
        elif leaf == "running_mean":
            out[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "running_var":
            out[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif len(shape) == 1 and leaf == "weight":  # BN gamma
            out[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif len(shape) == 1 and leaf == "bias" and "gru" not in name and "fc" not in name:  # BN beta
            out[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif ".gru." in name:
            k = 1.0 / math.sqrt(p.emb_hidden_dim)
            out[name] = rng.uniform(-k, k, shape).astype(np.float32)
        elif len(shape) == 4:  # conv: fan_in = in/groups * kh * kw
            fan_in = shape[1] * shape[2] * shape[3]
            k = math.sqrt(3.0 / max(fan_in, 1))
            out[name] = rng.uniform(-k, k, shape).astype(np.float32)
        elif len(shape) == 3:  # grouped linear [G, I/G, H/G]
            k = math.sqrt(3.0 / shape[1])
            out[name] = rng.uniform(-k, k, shape).astype(np.float32)
        elif len(shape) == 2:  # nn.Linear weight [out, in]
            k = 1.0 / math.sqrt(shape[1])
            out[name] = rng.uniform(-k, k, shape).astype(np.float32)
        else:  # nn.Linear bias
            out[name] = rng.uniform(-0.05, 0.05, shape).astype(np.float32)
    if widths is not None:
        fwd, inv = erb_fb_matrices(widths)
        out["erb_fb"] = fwd
        out["mask.erb_inv_fb"] = inv
    # keep the reference's key order
    return OrderedDict((k, out[k]) for k in man if k in out)


def check_state_dict(p: ModelParams, sd: Dict[str, "np.ndarray"]) -> List[str]:
    """Returns a list of problems (missing keys / shape mismatches); buffers erb_fb / erb_inv_fb are optional because
    checkpoints may carry stale copies that the reference drops (checkpoint.py:85-103)."""
    problems = []
    for name, shape in state_dict_manifest(p).items():
        if name in ("erb_fb", "mask.erb_inv_fb") or name.endswith("num_batches_tracked"):
            continue
        if name.startswith("df_dec.df_fc_a"):
            continue
        if name not in sd:
            problems.append(f"missing {name}")
        elif tuple(sd[name].shape) != tuple(shape):
            problems.append(f"shape {name}: got {tuple(sd[name].shape)}, want {tuple(shape)}")
    return problems
