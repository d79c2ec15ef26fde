"""Host-side mirror of ``df/io.py`` (load_audio :25-57, save_audio :60-84, get_resample_params :92-111, resample :114-116) with
the sample work on the MI355X: the file is parsed on the host (RIFF/WAVE PCM16 via the standard library — the reference goes
through torchaudio, which is not a dependency here), the int16 -> float scaling, the sample-rate conversion and the float -> int16
encoding run as HIP kernels (csrc/dfx_io.hip), so a file -> file loop like ``enhance.main`` (enhance.py:73-89) keeps its audio on the
device between decode and encode:

    audio, meta = load_audio("noisy.wav", sr=48000)          # [C, T] float32 on the GPU, resampled if the file is not 48 kHz
    enhanced = enhance(model, df_state, audio)
    save_audio("noisy.wav", resample(enhanced, 48000, meta.sample_rate), meta.sample_rate, suffix="DeepFilterNet3")
"""
from __future__ import annotations

import ctypes as C
import os
import wave
from dataclasses import dataclass
from functools import lru_cache
from typing import Any, Dict, Optional, Tuple, Union

import numpy as np
import torch

from . import _lib

TA_RESAMPLE_SINC = "sinc_interp_hann"      # io.py:13-14
TA_RESAMPLE_KAISER = "sinc_interp_kaiser"


@dataclass
class AudioMetaData:
    """The fields of torchaudio's AudioMetaData that the reference reads (enhance.py:80-88)."""
    sample_rate: int
    num_frames: int
    num_channels: int
    bits_per_sample: int = 16
    encoding: str = "PCM_S"


def get_resample_params(method: str) -> Dict[str, Any]:
    """io.py:92-111, verbatim parameter sets."""
    params = {
        "sinc_fast": {"resampling_method": TA_RESAMPLE_SINC, "lowpass_filter_width": 16},
        "sinc_best": {"resampling_method": TA_RESAMPLE_SINC, "lowpass_filter_width": 64},
        "kaiser_fast": {"resampling_method": TA_RESAMPLE_KAISER, "lowpass_filter_width": 16, "rolloff": 0.85,
                        "beta": 8.555504641634386},
        "kaiser_best": {"resampling_method": TA_RESAMPLE_KAISER, "lowpass_filter_width": 16, "rolloff": 0.9475937167399596,
                        "beta": 14.769656459379492},
    }
    assert method in params.keys(), f"method must be one of {list(params.keys())}"
    return params[method]


class _Resampler:
    def __init__(self, orig_sr: int, new_sr: int, method: str):
        p = get_resample_params(method)
        h = C.c_void_p()
        kaiser = p["resampling_method"] == TA_RESAMPLE_KAISER
        _lib.check(_lib.lib().dfx_resampler_create(int(orig_sr), int(new_sr), int(p["lowpass_filter_width"]), float(p.get("rolloff", 0.99)),
                                                   int(kaiser), float(p.get("beta", 14.769656459379492)), C.byref(h)))
        self.h = h

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                _lib.lib().dfx_resampler_free(h)
            except Exception:  # noqa: BLE001
                pass


@lru_cache(maxsize=16)
def _resampler(orig_sr: int, new_sr: int, method: str, lib_path: str) -> _Resampler:
    return _Resampler(orig_sr, new_sr, method)


def resample(audio: torch.Tensor, orig_sr: int, new_sr: int, method: str = "sinc_fast") -> torch.Tensor:
    """io.py:114-116.  audio [..., T] float32 -> [..., ceil(new_sr * T / orig_sr)] on the device."""
    if int(orig_sr) == int(new_sr):
        return audio                                      # torchaudio.functional.resample returns its input here
    r = _resampler(int(orig_sr), int(new_sr), method, _lib.library_path())
    x = audio.to(_lib.device(), torch.float32)
    shape = x.shape
    x = x.reshape(-1, shape[-1]).contiguous()
    B, T = x.shape
    out_len = int(_lib.lib().dfx_resampler_out_len(r.h, T))
    y = torch.empty((B, out_len), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().dfx_resample(r.h, _lib.ptr(x), B, T, T, _lib.ptr(y), out_len, _lib.stream()))
    return y.reshape(*shape[:-1], out_len)


def pcm16_to_float(pcm: torch.Tensor) -> torch.Tensor:
    """int16 samples -> float32 in [-1, 1) (what torchaudio.load(normalize=True) returns)."""
    x = pcm.to(_lib.device()).contiguous()
    assert x.dtype == torch.int16
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().dfx_pcm16_to_f32(_lib.ptr(x), x.numel(), _lib.ptr(out), _lib.stream()))
    return out


def float_to_pcm16(audio: torch.Tensor) -> torch.Tensor:
    """save_audio's ``(audio * (1 << 15)).to(torch.int16)`` (io.py:79-80)."""
    x = audio.to(_lib.device(), torch.float32).contiguous()
    out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    _lib.check(_lib.lib().dfx_f32_to_pcm16(_lib.ptr(x), x.numel(), _lib.ptr(out), _lib.stream()))
    return out


def load_audio(file: str, sr: Optional[int] = None, verbose: bool = True, **kwargs) -> Tuple[torch.Tensor, AudioMetaData]:
    """io.py:25-57: audio [C, T] float32 (on the device), resampled to ``sr`` when given; ``method=`` selects the resampler set.
    ``pcm16=True`` (extension): a file that needs no resampling is handed back as its int16 samples (on the device) — ``enhance()`` takes
    them as they are."""
    method = kwargs.pop("method", "sinc_fast")
    want_pcm = bool(kwargs.pop("pcm16", False))
    with wave.open(file, "rb") as w:
        if w.getsampwidth() != 2 or w.getcomptype() != "NONE":
            raise RuntimeError(f"{file}: only 16-bit PCM RIFF/WAVE files are decoded here")
        ch, n, orig_sr = w.getnchannels(), w.getnframes(), w.getframerate()
        frames = kwargs.get("num_frames", -1)
        if frames is not None and frames > 0 and sr is not None:
            frames *= orig_sr // sr                       # io.py:46-47
        off = kwargs.get("frame_offset", 0)
        if off:
            w.setpos(min(off, n))
        raw = w.readframes(n - off if frames is None or frames <= 0 else min(frames, n - off))
    info = AudioMetaData(sample_rate=orig_sr, num_frames=n, num_channels=ch)
    pcm = torch.from_numpy(np.frombuffer(raw, dtype="<i2").reshape(-1, ch).T.copy())   # interleaved -> [C, T]
    if want_pcm and (sr is None or orig_sr == sr):
        return pcm.to(_lib.device()).contiguous(), info
    audio = pcm16_to_float(pcm)
    if sr is not None and orig_sr != sr:
        if verbose:
            import warnings

            warnings.warn(f"Audio sampling rate does not match model sampling rate ({orig_sr}, {sr}). Resampling...")
        audio = resample(audio, orig_sr, sr, method=method)
    return audio.contiguous(), info


def save_audio(file: str, audio: Union[torch.Tensor, np.ndarray], sr: int, output_dir: Optional[str] = None,
               suffix: Optional[str] = None, log: bool = False, dtype=torch.int16) -> str:
    """io.py:60-84 (16-bit PCM only: the reference's float32 option needs torchaudio's encoder)."""
    outpath = file
    if suffix is not None:
        base, ext = os.path.splitext(file)
        outpath = base + f"_{suffix}" + ext
    if output_dir is not None:
        outpath = os.path.join(output_dir, os.path.basename(outpath))
    if dtype != torch.int16:
        raise NotImplementedError("save_audio: only 16-bit PCM output")
    audio = torch.as_tensor(audio)
    if audio.ndim == 1:
        audio = audio.unsqueeze(0)
    pcm = audio if audio.dtype == torch.int16 else float_to_pcm16(audio)
    data = pcm.cpu().numpy().T.astype("<i2", copy=False)  # [T, C] interleaved
    with wave.open(outpath, "wb") as w:
        w.setnchannels(data.shape[1])
        w.setsampwidth(2)
        w.setframerate(int(sr))
        w.writeframes(np.ascontiguousarray(data).tobytes())
    return outpath
