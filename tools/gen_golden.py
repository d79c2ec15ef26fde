#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by running the REFERENCE's own Python code (imported from /root/reference).

Run in the build container only (``python tools/gen_golden.py``); the GPU box has no /root/reference and only reads the
committed fixtures.  What is pinned:

  manifest_<cfg>.json   ``DfNet.state_dict()`` key -> shape of the reference model (strict load check of our manifest)
  dfnet_<cfg>.npz       reference ``DfNet.forward`` outputs for a seeded state-dict (deepfilternet_amd.state_dict) and
                        seeded inputs  -> pins oracle/dfnet_oracle.py and, through it, the HIP engine
  modules.npz           the reference's own inline identities (modules.py:929-1009): erb == |X|^2 @ erb_fb, erb_inv,
                        ExponentialUnitNorm == unit_norm, Mask, MF.DF for several (order, lookahead)
  enhance_<cfg>.npz     reference ``df.enhance.enhance()`` (pad / no pad / atten_lim) with the reference model, with
                        ``libdf`` provided by the C oracle (the Rust original cannot be built here)

The Rust half (STFT/ISTFT/ERB/norms) has no buildable reference in this image; it is pinned by numpy identities in
tests/test_oracle_dsp.py instead.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from tools.ref_import import install_shims, load_reference_config, reference_available  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")


def ref_overrides(p):
    from deepfilternet_amd.config import _OPTIONS

    ov = {}
    for attr, opt, section, cast, default in _OPTIONS:
        v = getattr(p, attr)
        if isinstance(v, tuple):
            v = ",".join(str(x) for x in v)
        ov[(section, opt)] = v
    return ov


def build_reference_model(p, seed):
    import torch
    from deepfilternet_amd.state_dict import random_state_dict

    load_reference_config(ref_overrides(p))
    import libdf  # the oracle front end (installed by install_shims)
    from df.deepfilternet3 import init_model

    df_state = libdf.DF(sr=p.sr, fft_size=p.fft_size, hop_size=p.hop_size, nb_bands=p.nb_erb,
                        min_nb_erb_freqs=p.min_nb_freqs)
    model = init_model(df_state)
    sd = random_state_dict(p, seed, widths=df_state.erb_widths())
    sd_t = {k: torch.as_tensor(v) for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(sd_t, strict=False)
    # our generator does not emit df_fc_a's (unused) parameters with special care: they exist, so nothing may be missing
    assert not missing and not unexpected, (missing, unexpected)
    model.eval()
    return model, df_state, sd


def seeded_inputs(p, seed, B, T):
    rng = np.random.default_rng(seed + 1000)
    spec = (0.05 * rng.standard_normal((B, 1, T, p.freq_bins, 2))).astype(np.float32)
    feat_erb = (0.5 * rng.standard_normal((B, 1, T, p.nb_erb))).astype(np.float32)
    feat_spec = rng.standard_normal((B, 1, T, p.nb_df, 2)).astype(np.float32)
    return spec, feat_erb, feat_spec


def synth_audio(seed, C, T, sr=48000):
    """SURVEY.md §8d recipe: harmonic 'speech-like' tone with 4 Hz AM + white noise at 0 dB SNR, clipped to [-1,1]."""
    rng = np.random.default_rng(seed)
    t = np.arange(T) / sr
    out = np.zeros((C, T), dtype=np.float64)
    for c in range(C):
        f0 = rng.uniform(100, 300)
        s = sum(np.sin(2 * np.pi * f0 * (h + 1) * t + rng.uniform(0, 2 * np.pi)) / (h + 1) for h in range(5))
        s *= 0.5 * (1 + np.sin(2 * np.pi * 4 * t))
        s *= 0.1 / (np.sqrt(np.mean(s ** 2)) + 1e-12)
        n = rng.standard_normal(T)
        n *= np.sqrt(np.mean(s ** 2)) / np.sqrt(np.mean(n ** 2))
        out[c] = s + n
    return np.clip(out, -1, 1).astype(np.float32)


def gen_model_goldens(name, p, seed=0, B=2, T=12):
    import torch

    model, df_state, sd = build_reference_model(p, seed)
    man = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(GOLDEN, f"manifest_{name}.json"), "w") as f:
        json.dump({"params": sum(p_.numel() for p_ in model.parameters()), "state_dict": man}, f, indent=0)
    spec, fe, fs = seeded_inputs(p, seed, B, T)
    with torch.no_grad():
        spec_e, m, lsnr, coefs = model(torch.from_numpy(spec).clone(), torch.from_numpy(fe), torch.from_numpy(fs))
    np.savez_compressed(os.path.join(GOLDEN, f"dfnet_{name}.npz"), seed=seed, B=B, T=T, spec=spec, feat_erb=fe,
                        feat_spec=fs, spec_e=spec_e.numpy(), m=m.numpy(), lsnr=lsnr.numpy(),
                        df_coefs=coefs.contiguous().numpy())
    # enhance(): reference orchestration + reference model + oracle libdf
    from df.enhance import enhance

    audio = synth_audio(seed + 7, 2, 4800 * 3 + 123)
    outs = {}
    for tag, kw in (("pad", dict(pad=True)), ("nopad", dict(pad=False)), ("lim12", dict(pad=True, atten_lim_db=12.0))):
        y = enhance(model, df_state, torch.from_numpy(audio.copy()), **kw)
        outs["y_" + tag] = y.numpy()
    np.savez_compressed(os.path.join(GOLDEN, f"enhance_{name}.npz"), seed=seed, audio_seed=seed + 7, audio=audio, **outs)
    print(f"[{name}] params={sum(p_.numel() for p_ in model.parameters())} dfnet+enhance goldens written")


def gen_module_goldens():
    """The reference's inline tests (modules.py:929-1009) evaluated with the oracle libdf + extra DF-apply cases."""
    import torch

    load_reference_config({})
    import libdf
    from df.modules import ExponentialUnitNorm, Mask, erb_fb
    from df.multiframe import DF as MFDF
    from df.utils import get_norm_alpha

    rng = np.random.default_rng(5)
    out = {}
    df_state = libdf.DF(sr=48000, fft_size=960, hop_size=480, nb_bands=32, min_nb_erb_freqs=2)
    widths = df_state.erb_widths()
    fb = erb_fb(widths, 48000)
    fb_inv = erb_fb(widths, 48000, inverse=True)
    x = (rng.standard_normal((2, 3, 7, 481)) + 1j * rng.standard_normal((2, 3, 7, 481))).astype(np.complex64)
    py_erb = torch.matmul(torch.from_numpy(x).abs().square(), fb)
    lib_erb = libdf.erb(x, widths, False)
    assert np.allclose(lib_erb, py_erb.numpy(), rtol=1e-5, atol=1e-8)          # test_erb :929-947
    py_inv = torch.matmul(py_erb, fb_inv)
    assert np.allclose(libdf.erb_inv(lib_erb, widths), py_inv.numpy())
    out.update(erb_in=x, erb_lin=py_erb.numpy(), erb_inv=py_inv.numpy(), widths=widths.astype(np.int64))
    alpha = get_norm_alpha(log=False)
    spec = rng.standard_normal((2, 1, 100, 96, 2)).astype(np.float32)
    un_t = ExponentialUnitNorm(alpha, 96)(torch.from_numpy(spec)).squeeze(1).contiguous()
    un_l = libdf.unit_norm(np.ascontiguousarray(torch.view_as_complex(torch.from_numpy(spec)).squeeze(1).numpy()), alpha)
    assert np.allclose(un_l.real, un_t[..., 0].numpy(), rtol=1e-5, atol=1e-6)  # test_unit_norm :950-967
    assert np.allclose(un_l.imag, un_t[..., 1].numpy(), rtol=1e-5, atol=1e-6)
    out.update(unit_norm_in=spec, unit_norm_out=un_t.numpy(), alpha=np.float64(alpha))
    # Mask (modules.py:248-269) on real-view spec
    sp = rng.standard_normal((2, 1, 5, 481, 2)).astype(np.float32)
    mk = rng.uniform(0, 1, (2, 1, 5, 32)).astype(np.float32)
    out.update(mask_spec=sp, mask_m=mk, mask_out=Mask(fb_inv)(torch.from_numpy(sp), torch.from_numpy(mk)).numpy())
    # MF.DF (multiframe.py:160-180)
    for (O, la) in ((5, 0), (5, 2), (10, 0), (10, 3), (1, 0)):
        T = 9
        s = rng.standard_normal((2, 1, T, 481, 2)).astype(np.float32)
        c = rng.standard_normal((2, O, T, 96, 2)).astype(np.float32)
        y = MFDF(num_freqs=96, frame_size=O, lookahead=la).eval()(torch.from_numpy(s).clone(), torch.from_numpy(c))
        out[f"df_{O}_{la}_spec"], out[f"df_{O}_{la}_coefs"], out[f"df_{O}_{la}_out"] = s, c, y.numpy()
    np.savez_compressed(os.path.join(GOLDEN, "modules.npz"), **out)
    print("modules goldens written; alpha =", alpha)


def main():
    if not reference_available():
        raise SystemExit("/root/reference not available: goldens can only be regenerated in the build container")
    install_shims()
    os.makedirs(GOLDEN, exist_ok=True)
    from deepfilternet_amd.config import ModelParams

    gen_module_goldens()
    gen_model_goldens("defaults", ModelParams.defaults(), seed=0)
    gen_model_goldens("df3", ModelParams.deepfilternet3(), seed=1)
    pf = ModelParams.defaults()
    pf.mask_pf = True
    pf.df_lookahead = 1
    pf.conv_lookahead = 1
    pf.df_gru_skip = "identity"
    pf.df_pathway_kernel_size_t = 3
    pf.conv_ch = 32
    gen_model_goldens("pf32", pf, seed=2)


if __name__ == "__main__":
    main()
