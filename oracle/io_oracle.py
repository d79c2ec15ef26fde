"""CPU restatement of the steps either side of enhance() in the reference's file loop — TEST INFRASTRUCTURE ONLY.

  * df/io.py:114-116 ``resample`` -> ``torchaudio.functional.resample`` with the parameter sets of io.py:92-111.  torchaudio (the
    2.0 - 2.2 releases: the ones whose ``resampling_method`` names are ``sinc_interp_hann`` / ``sinc_interp_kaiser``, the branch io.py:10-14
    takes when ``from torchaudio import AudioMetaData`` succeeds; ``torchaudio/functional/functional.py`` of v2.1.0 is the text restated here:
    float64 index grid when no dtype is passed, kernel rounded to float32 at the end) is a third-party dependency that is absent from
    this image, so its published algorithm is restated here (functional.py ``_get_sinc_resample_kernel`` /
    ``_apply_sinc_resample_kernel``): polyphase windowed-sinc bank built in float64 and rounded to float32, zero padding
    (width, width + orig), strided correlation, output trimmed to ceil(new * length / orig).  **Parity unpinned**: no torchaudio
    here to check against and the reference has no test vector for it; the restatement is checked by properties instead
    (tests/test_io.py: DC gain, sine reproduction below the cutoff, identity for equal rates, bank symmetry).
  * df/io.py:79-80 ``(audio * (1 << 15)).to(torch.int16)`` and torchaudio.load's ``/ 32768`` normalisation: torch itself.
"""
from __future__ import annotations

import math

import numpy as np

PARAMS = {   # io.py:92-111
    "sinc_fast": dict(kaiser=False, lowpass_filter_width=16, rolloff=0.99, beta=None),
    "sinc_best": dict(kaiser=False, lowpass_filter_width=64, rolloff=0.99, beta=None),
    "kaiser_fast": dict(kaiser=True, lowpass_filter_width=16, rolloff=0.85, beta=8.555504641634386),
    "kaiser_best": dict(kaiser=True, lowpass_filter_width=16, rolloff=0.9475937167399596, beta=14.769656459379492),
}


def sinc_resample_kernel(orig_sr: int, new_sr: int, method: str = "sinc_fast"):
    """-> (W float32 [new, 2*width + orig], width, orig, new) after dividing the rates by their gcd."""
    p = PARAMS[method]
    g = math.gcd(int(orig_sr), int(new_sr))
    orig, new = int(orig_sr) // g, int(new_sr) // g
    lpw = p["lowpass_filter_width"]
    base_freq = min(orig, new) * p["rolloff"]
    width = math.ceil(lpw * orig / base_freq)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = t * base_freq
    t = np.clip(t, -lpw, lpw)
    if p["kaiser"]:
        window = np.i0(p["beta"] * np.sqrt(1 - (t / lpw) ** 2)) / np.i0(p["beta"])
    else:
        window = np.cos(t * math.pi / lpw / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t))
    k = k * window * scale
    return k.astype(np.float32), width, orig, new


def resample(x: np.ndarray, orig_sr: int, new_sr: int, method: str = "sinc_fast") -> np.ndarray:
    """x float32 [..., T] -> float32 [..., ceil(new*T/orig)] (float64 accumulation)."""
    if int(orig_sr) == int(new_sr):
        return x
    W, width, orig, new = sinc_resample_kernel(orig_sr, new_sr, method)
    shape = x.shape
    x2 = x.reshape(-1, shape[-1]).astype(np.float64)
    T = shape[-1]
    out_len = -(-new * T // orig)
    frames = -(-out_len // new)
    K = W.shape[1]
    need = (frames - 1) * orig + K
    xp = np.zeros((x2.shape[0], max(need, width + T + width + orig)), dtype=np.float64)
    xp[:, width: width + T] = x2
    win = np.lib.stride_tricks.sliding_window_view(xp, K, axis=1)[:, ::orig][:, :frames]      # [B, frames, K]
    y = np.einsum("bnk,jk->bnj", win, W.astype(np.float64)).reshape(x2.shape[0], frames * new)[:, :out_len]
    return y.astype(np.float32).reshape(*shape[:-1], out_len)
