"""Fused deep-filter / ERB-gain / post-filter / attenuation-limit kernel (dfx_df_apply) vs the reference's MF.DF, Mask
and post filter (goldens from the reference's own modules) and vs the torch oracle on random shapes."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import dfnet_oracle as O
from oracle import libdf_oracle as L


def run_df_apply(spec, coefs, layout, gains, widths, nb_df, order, la, pf_beta=0.0, lim=0.0):
    """spec [B,T,F] c64, coefs c64 in `layout`, gains [B,T,E] f32 or None -> out [B,T,F] c64 (numpy in/out)."""
    from deepfilternet_amd import _lib
    from deepfilternet_amd.libdf import _Bands

    dev = _lib.device()
    B, T, F = spec.shape
    s = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(spec))).to(dev).contiguous()
    c = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(coefs))).to(dev).contiguous()
    g = torch.from_numpy(np.ascontiguousarray(gains)).to(dev) if gains is not None else None
    bands = _Bands.get(widths) if gains is not None else None
    out = torch.empty_like(s)
    _lib.check(_lib.lib().dfx_df_apply(_lib.ptr(s), _lib.ptr(c), layout, _lib.ptr(g), bands.handle if bands else None, B, T,
                                       F, nb_df, order, la, float(pf_beta), float(lim), _lib.ptr(out), _lib.stream()))
    return torch.view_as_complex(out.cpu()).numpy()


def run_df_apply_strided(spec, coefs, layout, gains, widths, nb_df, order, la, pf_beta=0.0, lim=0.0, pad=1):
    """The same operator on rows padded to an even stride (the engine's own spec layout; dfx_k_df_apply_rows)."""
    from deepfilternet_amd import _lib
    from deepfilternet_amd.libdf import _Bands

    dev = _lib.device()
    B, T, F = spec.shape
    Fs = ((F + 1) // 2) * 2 + 2 * (pad - 1)
    sp = np.full((B, T, Fs), np.nan + 1j * np.nan, np.complex64)   # pad bins of the input must be ignored
    sp[..., :F] = spec
    s = torch.view_as_real(torch.from_numpy(sp)).to(dev).contiguous()
    c = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(coefs))).to(dev).contiguous()
    g = torch.from_numpy(np.ascontiguousarray(gains)).to(dev) if gains is not None else None
    bands = _Bands.get(widths) if gains is not None else None
    out = torch.full_like(s, 7.0)
    _lib.check(_lib.lib().dfx_df_apply_strided(_lib.ptr(s), Fs, _lib.ptr(c), layout, _lib.ptr(g), bands.handle if bands else None, B,
                                               T, F, nb_df, order, la, float(pf_beta), float(lim), _lib.ptr(out), Fs, _lib.stream()))
    o = torch.view_as_complex(out.cpu()).numpy()
    if F % 2:
        assert np.all(o[..., F] == 0)                               # the pad bin that shares a float4 with bin F-1 is written as zero
    return o[..., :F]


@pytest.mark.parametrize("B,T,F,nd,O_,la,pf,lim,layout", [
    (2, 19, 481, 96, 5, 2, 0.0, 0.0, 0),    # DF3 shape: 4 compile-time passes, two waves (16 + 3 frames)
    (9, 70, 481, 96, 5, 0, 0.02, 0.0, 2),   # 5 chunks = two workgroups per clip, more clips than one XCD group, post filter
    (1, 1, 481, 96, 5, 2, 0.0, 0.25, 0),    # single frame + attenuation limit
    (2, 37, 481, 96, 10, 3, 0.02, 0.5, 2),  # order 10 (BASELINE.json configs[4])
    (2, 9, 97, 32, 16, 7, 0.0, 0.0, 0),     # run-time pass count, largest order
    (3, 21, 33, 8, 1, 0, 0.0, 0.0, 2),      # one tap, one pass
    (2, 5, 64, 64, 3, 1, 0.0, 0.0, 0),      # even F (dense rows are already aligned), nb_df == F
    (2, 33, 257, 128, 2, 1, 0.0, 0.0, 2),   # nb_df fills pass 0 completely
])
def test_rows_kernel_matches_oracle_and_flat_kernel(backend, B, T, F, nd, O_, la, pf, lim, layout):
    rng = np.random.default_rng(B * 1000 + T + O_)
    spec = (rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))).astype(np.complex64)
    cbotf = (rng.standard_normal((B, O_, T, nd)) + 1j * rng.standard_normal((B, O_, T, nd))).astype(np.complex64) * 0.3
    coefs = cbotf if layout == 0 else np.ascontiguousarray(cbotf.transpose(0, 2, 1, 3))
    nb = 8
    widths = np.full(nb, F // nb, np.uint64)
    widths[-1] += F - int(widths.sum())
    gains = rng.uniform(0, 1, (B, T, nb)).astype(np.float32)
    for gg in (gains, None):
        out = run_df_apply_strided(spec, coefs, layout, gg, widths, nd, O_, la, pf, lim)
        st = torch.from_numpy(spec)
        ref = st * O.band_gain(torch.from_numpy(gains), widths) if gg is not None else st.clone()
        ref[..., :nd] = O.df_apply(st, torch.from_numpy(cbotf), O_, la, nd)
        if pf > 0:
            ref = O.post_filter(st, ref, pf)
        if lim > 0:
            ref = st * lim + ref * (1 - lim)
        ref = ref.numpy()
        assert np.abs(out - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
        if F % 2:   # the flat-stream kernel on the dense rows gives the same numbers up to the order of the tap sum
            flat = run_df_apply(spec, coefs, layout, gg, widths, nd, O_, la, pf, lim)
            assert np.abs(out - flat).max() < 2e-5 * max(1.0, np.abs(ref).max())
            if pf == 0 and lim == 0:
                assert np.array_equal(out[..., nd:], flat[..., nd:])    # gain bins: one multiply per component
    # a wider pad (stride F + 3 rounded to even) only moves the rows
    out2 = run_df_apply_strided(spec, coefs, layout, gains, widths, nd, O_, la, pf, lim, pad=2)
    assert np.array_equal(out2, run_df_apply_strided(spec, coefs, layout, gains, widths, nd, O_, la, pf, lim))


@pytest.mark.parametrize("O_,la", [(5, 0), (5, 2), (10, 0), (10, 3), (1, 0)])
def test_matches_reference_mf_df_golden(backend, O_, la, golden_dir):
    g = np.load(os.path.join(golden_dir, "modules.npz"))
    s, c, y = g[f"df_{O_}_{la}_spec"], g[f"df_{O_}_{la}_coefs"], g[f"df_{O_}_{la}_out"]
    sc = (s[:, 0, ..., 0] + 1j * s[:, 0, ..., 1]).astype(np.complex64)
    cc = (c[..., 0] + 1j * c[..., 1]).astype(np.complex64)          # [B,O,T,F']  == DFX_COEF_BOTF
    ref = (y[:, 0, ..., 0] + 1j * y[:, 0, ..., 1]).astype(np.complex64)
    out = run_df_apply(sc, cc, 0, None, None, 96, O_, la)
    assert np.abs(out - ref).max() < 5e-6 * np.abs(ref).max()
    assert np.array_equal(out[..., 96:], sc[..., 96:])              # MF.DF leaves the high bins untouched
    # same coefficients in the DfDecoder layout [B,T,F',O]
    out2 = run_df_apply(sc, np.ascontiguousarray(cc.transpose(0, 2, 3, 1)), 1, None, None, 96, O_, la)
    assert np.array_equal(out, out2)
    # and in the engine's own tap-major layout [B,T,O,F'] (DFX_COEF_BTOF)
    out3 = run_df_apply(sc, np.ascontiguousarray(cc.transpose(0, 2, 1, 3)), 2, None, None, 96, O_, la)
    assert np.array_equal(out, out3)
    # the row-streaming kernel on rows padded to 482 bins (the engine's own layout)
    out4 = run_df_apply_strided(sc, cc, 0, None, None, 96, O_, la)
    assert np.abs(out4 - ref).max() < 5e-6 * np.abs(ref).max()
    assert np.array_equal(out4[..., 96:], sc[..., 96:])


def test_mask_matches_reference_golden(backend, golden_dir):
    g = np.load(os.path.join(golden_dir, "modules.npz"))
    w = g["widths"]
    sp = (g["mask_spec"][:, 0, ..., 0] + 1j * g["mask_spec"][:, 0, ..., 1]).astype(np.complex64)
    ref = (g["mask_out"][:, 0, ..., 0] + 1j * g["mask_out"][:, 0, ..., 1]).astype(np.complex64)
    B, T, F = sp.shape
    coefs = np.zeros((B, T, 2, 1), np.complex64)
    coefs[..., 0] = 1.0                                              # identity filter on 2 bins
    out = run_df_apply(sp, coefs, 1, g["mask_m"][:, 0], w, 2, 1, 0)
    assert np.array_equal(out[..., 2:], ref[..., 2:])                # exact: one multiply per component
    assert np.array_equal(out[..., :2], sp[..., :2])


@pytest.mark.parametrize("B,T,F,nd,O_,la,pf,lim", [
    (2, 19, 481, 96, 5, 2, 0.0, 0.0),      # DF3 shape, ragged last chunk (19 = 2*8+3), odd/even row alignment
    (3, 8, 481, 96, 5, 0, 0.02, 0.0),      # post filter
    (1, 1, 481, 96, 5, 2, 0.0, 0.25),      # single frame + attenuation limit
    (2, 9, 97, 32, 10, 3, 0.02, 0.5),      # other sizes, order 10
    (2, 5, 64, 64, 3, 1, 0.0, 0.0),        # even F, nb_df == F
    (3, 37, 481, 95, 5, 2, 0.0, 0.0),      # odd T*F (odd clips start 8-byte aligned only), odd nb_df, 3 row chunks
    (9, 17, 33, 7, 2, 0, 0.0, 0.0),        # more clips than one XCD group, tiny rows
])
def test_fused_matches_oracle(backend, B, T, F, nd, O_, la, pf, lim):
    rng = np.random.default_rng(B * 100 + T)
    spec = (rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))).astype(np.complex64)
    coefs = (rng.standard_normal((B, T, nd, O_)) + 1j * rng.standard_normal((B, T, nd, O_))).astype(np.complex64) * 0.3
    nb = 8
    widths = np.full(nb, F // nb, np.uint64)
    widths[-1] += F - int(widths.sum())
    gains = rng.uniform(0, 1, (B, T, nb)).astype(np.float32)
    out = run_df_apply(spec, coefs, 1, gains, widths, nd, O_, la, pf, lim)
    st, ct = torch.from_numpy(spec), torch.from_numpy(np.ascontiguousarray(coefs.transpose(0, 3, 1, 2)))
    ref = st * O.band_gain(torch.from_numpy(gains), widths)
    ref[..., :nd] = O.df_apply(st, ct, O_, la, nd)
    if pf > 0:
        ref = O.post_filter(st, ref, pf)
    if lim > 0:
        ref = st * lim + ref * (1 - lim)
    ref = ref.numpy()
    assert np.abs(out - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_argument_errors(backend):
    from deepfilternet_amd import _lib

    spec = np.zeros((1, 4, 481), np.complex64)
    coefs = np.zeros((1, 4, 96, 5), np.complex64)
    with pytest.raises(_lib.DfxError, match="lookahead"):
        run_df_apply(spec, coefs, 1, None, None, 96, 5, 5)
    with pytest.raises(_lib.DfxError, match="coef_layout"):
        run_df_apply(spec, coefs, 3, None, None, 96, 5, 0)
    with pytest.raises(_lib.DfxError, match="nb_df"):
        run_df_apply(spec, np.zeros((1, 4, 482, 5), np.complex64), 1, None, None, 482, 5, 0)
