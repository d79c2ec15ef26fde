#!/usr/bin/env python3
"""Round-2 goldens, produced by the REFERENCE's own Python (``/root/reference/DeepFilterNet/df``; build container only):

  dfnet_lsnr_dropout_<cfg>.npz  ``DfNet.forward`` with ``lsnr_dropout=True`` (deepfilternet3.py:413-441): the reference's own statement of
                                the decoder compaction that its real-time runtime performs through pulsed tract models — frames whose
                                local SNR is <= -10 dB are left out of BOTH decoders' input sequences (a decoder only sees, and only
                                advances on, the frames it runs on), their mask / coefficients are zero.  Pins oracle/stream_oracle.py
                                (thresholds (-10, +inf, +inf)) and, through it, the engine's stage gating.
  dfnet_opts_<name>.npz         ``DfNet.forward`` for the options round 1 refused: emb_gru_skip_enc / emb_gru_skip (identity,
                                groupedlinear), enc_concat, run_df=False.
"""
from __future__ import annotations

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from tools.gen_golden import GOLDEN, ref_overrides, seeded_inputs  # noqa: E402
from tools.ref_import import install_shims, load_reference_config, reference_available  # noqa: E402


def build(p, seed, run_df=True, sd_edit=None):
    import torch
    from deepfilternet_amd.state_dict import random_state_dict

    load_reference_config(ref_overrides(p))
    import libdf
    from df.deepfilternet3 import init_model

    df_state = libdf.DF(sr=p.sr, fft_size=p.fft_size, hop_size=p.hop_size, nb_bands=p.nb_erb, min_nb_erb_freqs=p.min_nb_freqs)
    model = init_model(df_state, run_df=run_df)
    sd = random_state_dict(p, seed, widths=df_state.erb_widths())
    if sd_edit:
        sd_edit(sd)
    missing, unexpected = model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    model.eval()
    return model, df_state, sd


def gen_lsnr_dropout(name, p, seed, T=48):
    """B = 1 (the reference's boolean-mask indexing squeezes the batch axis).  Inputs: the reference's df_features() of synthetic audio
    (with the oracle as libdf, like the enhance goldens), so that the streaming runtime can be driven with the same signal.  The lsnr
    head's bias is moved so that the local SNR of this signal straddles the -10 dB test (a good share of the frames on either side,
    none within 0.05 dB of it)."""
    import torch
    from oracle import dfnet_oracle as O
    from tools.gen_golden import synth_audio

    import libdf
    from deepfilternet_amd.state_dict import random_state_dict
    from df.enhance import df_features

    load_reference_config(ref_overrides(p))
    st = libdf.DF(sr=p.sr, fft_size=p.fft_size, hop_size=p.hop_size, nb_bands=p.nb_erb, min_nb_erb_freqs=p.min_nb_freqs)
    audio = synth_audio(seed + 3, 1, T * p.hop_size)
    audio[0, 10 * p.hop_size: 14 * p.hop_size] *= 0.02          # level changes make the local-SNR estimate move
    audio[0, 30 * p.hop_size: 36 * p.hop_size] *= 8.0
    audio = np.clip(audio, -1, 1).astype(np.float32)
    spec_t, fe_t, fs_t = df_features(torch.from_numpy(audio), st, p.nb_df, device="cpu")   # enhance.py:190-203
    spec, fe, fs = spec_t.numpy(), fe_t.numpy(), fs_t.numpy()
    # where does fc(emb) sit for these inputs?  (the oracle's encoder, pinned by round 1's goldens, with the unmodified bias)
    widths = st.erb_widths()
    sd0 = {k: torch.as_tensor(v) for k, v in random_state_dict(p, seed, widths=widths).items()}
    fsp = O.pad_feat(torch.from_numpy(fs).squeeze(1).permute(0, 3, 1, 2), p.conv_lookahead)
    enc = O.dfnet_encoder(p, sd0, O.pad_feat(torch.from_numpy(fe), p.conv_lookahead), fsp)
    # the seeded lsnr head barely moves (its pre-activation spans 0.03 over this signal): widen it so that the estimate covers tens of dB
    GAIN = 150.0
    pre = GAIN * torch.nn.functional.linear(enc["emb"], sd0["enc.lsnr_fc.0.weight"])[0, :, 0].numpy()
    srt = np.sort(pre)
    # centre of the widest gap that leaves at least 10 frames on either side -> logit(0.1) = -10 dB exactly between two observed values
    j = max(range(9, len(srt) - 10), key=lambda i: srt[i + 1] - srt[i])
    shift = float(np.log(0.1 / 0.9) - 0.5 * (srt[j] + srt[j + 1]))

    def edit(sd):
        sd["enc.lsnr_fc.0.weight"] = (np.asarray(sd["enc.lsnr_fc.0.weight"]) * np.float32(GAIN)).astype(np.float32)
        sd["enc.lsnr_fc.0.bias"] = np.full((1,), shift, np.float32)

    p.lsnr_dropout = True
    model, df_state, sd = build(p, seed, sd_edit=edit)
    p.lsnr_dropout = False
    with torch.no_grad():
        spec_e, m, lsnr, coefs = model(torch.from_numpy(spec).clone(), torch.from_numpy(fe), torch.from_numpy(fs))
    l = lsnr.numpy()[0, :, 0]
    kept = l > -10.0
    assert 8 <= kept.sum() <= T - 8 and np.abs(l + 10.0).min() > 0.05, (kept.sum(), np.abs(l + 10.0).min())
    np.savez_compressed(os.path.join(GOLDEN, f"dfnet_lsnr_dropout_{name}.npz"), seed=seed, T=T, lsnr_fc_gain=np.float32(GAIN), lsnr_fc_bias=np.float32(shift),
                        audio=audio, spec=spec, feat_erb=fe, feat_spec=fs, spec_e=spec_e.numpy(), m=m.numpy(), lsnr=lsnr.numpy(),
                        df_coefs=coefs.contiguous().numpy(), kept=kept)
    print(f"[lsnr_dropout {name}] {int(kept.sum())} of {T} frames kept, min |lsnr + 10| = {np.abs(l + 10).min():.3f} dB")


def gen_opts(name, p, seed, run_df=True, B=2, T=11):
    import torch

    model, df_state, sd = build(p, seed, run_df=run_df)
    spec, fe, fs = seeded_inputs(p, seed, B, T)
    with torch.no_grad():
        spec_e, m, lsnr, coefs = model(torch.from_numpy(spec).clone(), torch.from_numpy(fe), torch.from_numpy(fs))
    np.savez_compressed(os.path.join(GOLDEN, f"dfnet_opts_{name}.npz"), seed=seed, B=B, T=T, run_df=run_df, ini=p.to_ini(), spec=spec,
                        feat_erb=fe, feat_spec=fs, spec_e=spec_e.numpy(), m=m.numpy(), lsnr=lsnr.numpy(),
                        df_coefs=coefs.contiguous().numpy())
    print(f"[opts {name}] written ({sum(q.numel() for q in model.parameters())} parameters)")


def opt_cases():
    """name -> (ModelParams, run_df): shared with tests/test_config_options.py."""
    from deepfilternet_amd.config import ModelParams

    cases = {}
    a = ModelParams.deepfilternet3()
    a.emb_gru_skip_enc, a.emb_gru_skip = "identity", "groupedlinear"
    cases["skip_id_gl"] = (a, True)
    b = ModelParams.defaults()
    b.emb_gru_skip_enc, b.emb_gru_skip, b.conv_ch, b.lin_groups = "groupedlinear", "identity", 32, 8
    cases["skip_gl_id"] = (b, True)
    c = ModelParams.defaults()
    c.enc_concat, c.conv_ch, c.emb_num_layers, c.pad_mode = True, 32, 3, "output"
    cases["concat"] = (c, True)
    d = ModelParams.deepfilternet3()
    d.mask_pf = True
    cases["mask_only"] = (d, False)
    return cases


def main():
    if not reference_available():
        raise SystemExit("/root/reference not available: goldens can only be regenerated in the build container")
    install_shims()
    from deepfilternet_amd.config import ModelParams

    gen_lsnr_dropout("df3", ModelParams.deepfilternet3(), seed=11)
    gen_lsnr_dropout("df3_ll", ModelParams.deepfilternet3_ll(), seed=12)
    for i, (name, (p, run_df)) in enumerate(opt_cases().items()):
        gen_opts(name, p, seed=20 + i, run_df=run_df)


if __name__ == "__main__":
    main()
