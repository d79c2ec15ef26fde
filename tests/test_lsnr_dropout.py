"""Stage gating pinned against the reference's own statement of it: ``DfNet.forward`` with ``lsnr_dropout=True``
(deepfilternet3.py:413-441; goldens tests/golden/dfnet_lsnr_dropout_*.npz from tools/gen_golden_r2.py).  Frames whose local SNR is not
above -10 dB are left out of both decoders' input sequences (each decoder sees the compacted sequence of the frames it runs on — what a
pulsed tract model does in the real-time runtime), their mask and coefficients are zero.

  * the streaming oracle's ``decode_stages`` reproduces the reference's mask / coefficients (CPU);
  * the engine's gated runtime (dfx_stream_process_raw / dfx_stream_process with thresholds (-10, +inf, +inf)) reproduces them frame by
    frame, and the enhanced audio of the reference's ``spec_e`` ('emu': kernel sources on the CPU interpreter; 'hip': MI355X).
"""
import os

import numpy as np
import pytest
import torch

from deepfilternet_amd.config import ModelParams
from deepfilternet_amd.state_dict import random_state_dict
from oracle import dfnet_oracle as O
from oracle import libdf_oracle as L
from oracle import stream_oracle as S
from tests.helpers import rms, widths_for

INF = 1e30
CASES = {"df3": (ModelParams.deepfilternet3, 11), "df3_ll": (ModelParams.deepfilternet3_ll, 12)}


def _load(name, golden_dir):
    g = np.load(os.path.join(golden_dir, f"dfnet_lsnr_dropout_{name}.npz"))
    mk, seed = CASES[name]
    p = mk()
    sd = random_state_dict(p, seed, widths=widths_for(p))
    sd["enc.lsnr_fc.0.weight"] = (np.asarray(sd["enc.lsnr_fc.0.weight"]) * np.float32(g["lsnr_fc_gain"])).astype(np.float32)
    sd["enc.lsnr_fc.0.bias"] = np.full((1,), g["lsnr_fc_bias"], np.float32)
    return g, p, sd


@pytest.mark.parametrize("name", list(CASES))
def test_stream_oracle_compaction_matches_reference(name, golden_dir):
    g, p, sd = _load(name, golden_dir)
    sdt = {k: torch.as_tensor(v) for k, v in sd.items()}
    fe = O.pad_feat(torch.from_numpy(g["feat_erb"]), p.conv_lookahead)
    fs = O.pad_feat(torch.from_numpy(g["feat_spec"]).squeeze(1).permute(0, 3, 1, 2), p.conv_lookahead)
    enc = O.dfnet_encoder(p, sdt, fe, fs)
    lsnr = enc["lsnr"][0, :, 0].numpy()
    assert np.abs(lsnr - g["lsnr"][0, :, 0]).max() < 1e-3
    flags = [S.apply_stages(float(v), (-10.0, INF, INF)) for v in lsnr]
    kept = np.array([f[0] for f in flags])
    assert np.array_equal(kept, g["kept"]) and all(f[2] == f[0] and f[1] != f[0] for f in flags)
    gains, coefs, idx_g, idx_d = S.decode_stages(p, sdt, enc, flags)
    assert idx_g == idx_d == list(np.flatnonzero(kept))
    # mask: the reference's m is zero on the dropped frames (the oracle's zero mask), the decoder output elsewhere
    assert np.abs(gains[0].numpy() - g["m"][0, 0]).max() < 2e-5
    assert np.all(gains[0].numpy()[~kept] == 0) and np.all(g["m"][0, 0][~kept] == 0)
    # coefficients: [1, O, T, F', 2] in the reference, zeros on the dropped frames
    ref_c = g["df_coefs"][0][:, kept]
    got_c = torch.view_as_real(coefs[0]).numpy()
    assert np.abs(got_c - ref_c).max() < 3e-5 * max(1.0, np.abs(ref_c).max())
    assert np.all(g["df_coefs"][0][:, ~kept] == 0)
    # and the whole hop-by-hop oracle on the audio: the enhanced spectrum of the reference, synthesised, delayed by the lookahead
    y, ls, info = S.process_stream(p, sdt, g["audio"][0], thresholds=(-10.0, INF, INF))
    st = L.DF(p.sr, p.fft_size, p.hop_size, p.nb_erb, p.min_nb_freqs)
    spec_e = np.ascontiguousarray(g["spec_e"][:, 0, ..., 0] + 1j * g["spec_e"][:, 0, ..., 1]).astype(np.complex64)
    d = p.df_lookahead
    T = spec_e.shape[1]
    ref = st.synthesis(np.ascontiguousarray(np.concatenate([np.zeros((1, d, spec_e.shape[2]), np.complex64), spec_e[:, : T - d]], axis=1)))
    assert rms(y - ref[0]) < 2e-6 * max(1.0, rms(ref)) + 1e-7, rms(y - ref[0])


@pytest.mark.parametrize("name", list(CASES))
def test_gated_runtime_matches_reference_lsnr_dropout(backend, name, golden_dir):
    from deepfilternet_amd.enhance import init_df
    from deepfilternet_amd.streaming import DfStream

    if backend == "emu" and name == "df3":
        pytest.skip("the interpreter runs the no-lookahead model; the lookahead-2 model runs on the GPU")
    g, p, sd = _load(name, golden_dir)
    T = (int(g["T"]) if __import__("os").environ.get("DFX_EMU_ALL") == "1" else 12) if backend == "emu" else int(g["T"])   # one gated hop per pass: keep the interpreter run short
    d, hop = p.df_lookahead, p.hop_size
    model, df_state, _, _ = init_df(params=p, state_dict=sd, epoch="none")
    kept = g["kept"]
    spec = torch.view_as_complex(torch.from_numpy(np.ascontiguousarray(g["spec"][0, 0])))          # [T, F]
    # ---- raw path (df_process_frame_raw): gains / coefficients of net position k - lookahead at call k
    rt = DfStream(model, df_state, streams=1, max_frames=1, gating=True, thresholds=(-10.0, INF, INF))
    for k in range(T):
        lsnr, gains, coefs, stages = rt.process_raw(spec[k: k + 1])
        q = k - d
        if q < 0:
            continue
        st = int(stages[0])
        assert abs(float(lsnr[0]) - float(g["lsnr"][0, q, 0])) < 2e-3, (k, float(lsnr[0]), float(g["lsnr"][0, q, 0]))
        assert (st & 2) == 2                                      # gains exist on every frame: the mask, or zeros (tract.rs:485-486)
        assert bool(st & 8) == bool(kept[q]), (k, st)
        assert np.abs(gains[0].numpy() - g["m"][0, 0, q]).max() < 3e-5
        if kept[q]:
            ref_c = g["df_coefs"][0, :, q]                        # [O, F', 2]
            assert np.abs(torch.view_as_real(coefs[0]).numpy() - ref_c).max() < 5e-5 * max(1.0, np.abs(ref_c).max())
    # ---- audio path (df_process_frame): ISTFT of the reference's enhanced spectrum, delayed by the lookahead
    rt2 = DfStream(model, df_state, streams=1, max_frames=1, gating=True, thresholds=(-10.0, INF, INF))
    x = torch.from_numpy(g["audio"][:, : T * hop])
    y = torch.cat([rt2.process(x[:, k * hop:(k + 1) * hop]) for k in range(T)], dim=1).numpy()
    spec_e = np.ascontiguousarray(g["spec_e"][:, 0, ..., 0] + 1j * g["spec_e"][:, 0, ..., 1]).astype(np.complex64)
    stl = L.DF(p.sr, p.fft_size, p.hop_size, p.nb_erb, p.min_nb_freqs)
    ref = stl.synthesis(np.ascontiguousarray(np.concatenate([np.zeros((1, d, spec_e.shape[2]), np.complex64), spec_e[:, : T - d]], axis=1)))
    assert rms(y - ref) < 1e-4 * max(rms(ref), 1e-3), rms(y - ref)     # north-star bar; measured ~1e-6
    assert rms(y - ref) < 5e-6
