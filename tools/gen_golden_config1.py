#!/usr/bin/env python3
"""BASELINE.json configs[0]: DeepFilterNet2 ``enhance()`` on assets/noisy_snr0.wav, CPU PyTorch + pyDF, batch 1.

Runs the REFERENCE's own ``df.enhance.enhance`` with the reference's own DeepFilterNet2 model (``df.deepfilternet2``, seeded
weights: the pretrained zips are missing blobs) on a cut of the reference's own test asset, with ``libdf`` provided by the C oracle,
and records every array that crosses the pyDF boundary (the drop-in boundary of this repo):

    DF.analysis in/out, erb() out, erb_norm() out, unit_norm() out, DF.synthesis in/out, and enhance()'s return value.

tests/test_config1.py replays that boundary traffic through deepfilternet_amd.libdf (the GPU-backed pyDF replacement).
Build container only (needs /root/reference); the fixture is committed as tests/golden/config1_df2.npz.
"""
from __future__ import annotations

import os
import sys
import wave

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from tools.ref_import import REFERENCE_ROOT, install_shims, load_reference_config  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden", "config1_df2.npz")
START_S, LEN_S = 1.0, 0.75   # a stretch with speech onset; 36000 samples -> 77 frames with enhance()'s padding


def read_wav(path):
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2 and w.getnchannels() == 1
        sr = w.getframerate()
        x = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0   # io.py:25-57 (int16 -> [-1, 1))
    return x, sr


def build_df2(seed=0):
    import torch

    # DF2 defaults crash in the reference's own output layer: the shipped DF2 config uses grouped linears
    load_reference_config({("deepfilternet", "DF_OUTPUT_LAYER"): "groupedlinear", ("train", "MODEL"): "deepfilternet2"})
    import libdf
    from df.deepfilternet2 import ModelParams, init_model

    p = ModelParams()
    df_state = libdf.DF(sr=p.sr, fft_size=p.fft_size, hop_size=p.hop_size, nb_bands=p.nb_erb, min_nb_erb_freqs=p.min_nb_freqs)
    torch.manual_seed(seed)
    model = init_model(df_state).eval()
    g = torch.Generator().manual_seed(seed + 1)
    for m in model.modules():   # non-trivial BatchNorm statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
    return model, df_state, p


class Recorder:
    """pyDF-shaped proxy around the oracle's DF that keeps what crossed the boundary."""

    def __init__(self, inner, rec):
        self._i, self._r = inner, rec

    def analysis(self, x, *a, **k):
        self._r["analysis_in"] = np.array(x, copy=True)
        y = self._i.analysis(x, *a, **k)
        self._r["analysis_out"] = np.array(y, copy=True)
        return y

    def synthesis(self, x, *a, **k):
        self._r["synthesis_in"] = np.array(x, copy=True)   # pyDF mutates its input: copy first
        y = self._i.synthesis(x, *a, **k)
        self._r["synthesis_out"] = np.array(y, copy=True)
        return y

    def __getattr__(self, n):
        return getattr(self._i, n)


def main():
    import torch

    install_shims()
    model, df_state, p = build_df2()
    import importlib

    E = importlib.import_module("df.enhance")   # `df.enhance` the attribute is the function re-exported by df/__init__.py
    import libdf

    rec = {}
    for name in ("erb", "erb_norm", "unit_norm"):
        fn = getattr(libdf, name)

        def wrap(*a, _fn=fn, _name=name, **k):
            if _name == "erb_norm":
                rec["erb_out"] = np.array(a[0], copy=True)        # erb()'s output before erb_norm works on it in place
            y = _fn(*a, **k)
            rec[_name + "_out"] = np.array(y, copy=True)
            if _name == "unit_norm":
                rec["unit_norm_in"] = np.array(a[0], copy=True)
            return y

        setattr(E, name, wrap)
    x, sr = read_wav(os.path.join(REFERENCE_ROOT, "assets", "noisy_snr0.wav"))
    assert sr == p.sr == 48000
    x = x[int(START_S * sr): int((START_S + LEN_S) * sr)]
    audio = torch.from_numpy(x[None].copy())
    y = E.enhance(model, Recorder(df_state, rec), audio)          # enhance.py:206-250, pad=True
    y12 = E.enhance(model, Recorder(df_state, {}), audio, atten_lim_db=12.0)
    out = {k: v for k, v in rec.items() if k != "erb_out" or True}
    out.update(audio=x[None], enhanced=y.numpy(), enhanced_lim12=y12.numpy(), alpha=np.float32(E.get_norm_alpha(False)),
               erb_widths=np.asarray(df_state.erb_widths(), dtype=np.uint64),
               meta=np.array([p.sr, p.fft_size, p.hop_size, p.nb_erb, p.nb_df, p.min_nb_freqs], dtype=np.int64))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: (v.shape, v.dtype) for k, v in out.items()}, os.path.getsize(OUT))
    print("rms in", float(np.sqrt(np.mean(x ** 2))), "rms out", float(np.sqrt(np.mean(y.numpy() ** 2))))


if __name__ == "__main__":
    main()
