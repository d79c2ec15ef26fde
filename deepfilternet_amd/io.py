"""Host-side mirror of ``df/io.py`` (load_audio :25-57, save_audio :60-84, get_resample_params :92-111, resample :114-116) with
the sample work on the MI355X: the file is parsed on the host (a RIFF/WAVE reader / writer of its own, below — the reference goes
through torchaudio, which is not a dependency here; 8 / 16 / 24 / 32-bit integer PCM and 32 / 64-bit IEEE float, plain or
WAVE_FORMAT_EXTENSIBLE), the int16 -> float scaling, the sample-rate conversion and the float -> int16 encoding run as HIP kernels
(csrc/dfx_io.hip), so a file -> file loop like ``enhance.main`` (enhance.py:73-89) keeps its audio on the device between decode and
encode (the rarer sample formats are scaled with torch ops on the device):

    audio, meta = load_audio("noisy.wav", sr=48000)          # [C, T] float32 on the GPU, resampled if the file is not 48 kHz
    enhanced = enhance(model, df_state, audio)
    save_audio("noisy.wav", resample(enhanced, 48000, meta.sample_rate), meta.sample_rate, suffix="DeepFilterNet3")
"""
from __future__ import annotations

import ctypes as C
import os
import struct
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


_WAVE_PCM, _WAVE_FLOAT, _WAVE_EXTENSIBLE = 1, 3, 0xFFFE


def _read_riff(file: str, frame_offset: int = 0, num_frames: int = -1):
    """-> (samples [T, C] numpy array in the file's own sample type (24-bit: int32, sign-extended), sample_rate, total frames, bits, float?).
    RIFF/WAVE with a 'fmt ' chunk of tag 1 (integer PCM), 3 (IEEE float) or 0xFFFE (extensible: the sub-format's first two bytes)."""
    with open(file, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise RuntimeError(f"{file}: not a RIFF/WAVE file")
        fmt = None
        while True:
            ch = f.read(8)
            if len(ch) < 8:
                raise RuntimeError(f"{file}: no 'data' chunk")
            cid, size = ch[:4], struct.unpack("<I", ch[4:])[0]
            if cid == b"fmt ":
                body = f.read(size + (size & 1))
                if size < 16 or len(body) < 16:
                    raise RuntimeError(f"{file}: 'fmt ' chunk of {size} bytes (16 at least)")
                tag, nch, rate, _, align, bits = struct.unpack("<HHIIHH", body[:16])
                if tag == _WAVE_EXTENSIBLE and size >= 26:
                    valid = struct.unpack("<H", body[18:20])[0]   # wValidBitsPerSample: fewer than the container (24 in 32) would need their own scale
                    if valid not in (0, bits):
                        raise RuntimeError(f"{file}: extensible WAVE with {valid} valid bits in {bits}-bit containers is not decoded")
                    tag = struct.unpack("<H", body[24:26])[0]
                fmt = (tag, nch, rate, align, bits)
            elif cid == b"data":
                if fmt is None:
                    raise RuntimeError(f"{file}: 'data' chunk before 'fmt '")
                tag, nch, rate, align, bits = fmt
                if tag not in (_WAVE_PCM, _WAVE_FLOAT) or (tag == _WAVE_PCM and bits not in (8, 16, 24, 32)) or (tag == _WAVE_FLOAT and bits not in (32, 64)):
                    raise RuntimeError(f"{file}: unsupported WAVE sample format (tag {tag}, {bits} bits); integer PCM 8/16/24/32 and IEEE float 32/64 are decoded")
                bps = bits // 8
                if align != nch * bps or nch < 1:
                    raise RuntimeError(f"{file}: inconsistent 'fmt ' chunk (block align {align}, {nch} channels x {bps} bytes)")
                here = f.tell()
                f.seek(0, 2)
                size = min(size, f.tell() - here)      # (a streamed file may carry 0xFFFFFFFF / a stale size)
                total = size // align
                off = min(max(int(frame_offset or 0), 0), total)
                n = total - off if num_frames is None or num_frames <= 0 else min(int(num_frames), total - off)
                f.seek(here + off * align)
                raw = f.read(n * align)
                n = len(raw) // align
                raw = raw[: n * align]
                if tag == _WAVE_FLOAT:
                    x = np.frombuffer(raw, dtype="<f4" if bits == 32 else "<f8")
                elif bits == 8:
                    x = np.frombuffer(raw, dtype=np.uint8)
                elif bits == 16:
                    x = np.frombuffer(raw, dtype="<i2")
                elif bits == 32:
                    x = np.frombuffer(raw, dtype="<i4")
                else:   # 24-bit little endian -> int32 with the sign extended
                    b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
                    x = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
                    x = (x ^ 0x800000) - 0x800000
                return x.reshape(n, nch), rate, total, bits, tag == _WAVE_FLOAT
            else:
                f.seek(size + (size & 1), 1)


def _write_riff(file: str, data: np.ndarray, sr: int, is_float: bool) -> None:
    """data [T, C]: int16 -> 16-bit PCM (tag 1), float32 -> 32-bit IEEE float (tag 3, with the 'fact' chunk non-PCM formats carry)."""
    T, nch = data.shape
    bps = 4 if is_float else 2
    payload = np.ascontiguousarray(data.astype("<f4" if is_float else "<i2", copy=False)).tobytes()
    pad = b"\x00" if len(payload) & 1 else b""
    if is_float:
        fmt = struct.pack("<HHIIHHH", _WAVE_FLOAT, nch, int(sr), int(sr) * nch * bps, nch * bps, 8 * bps, 0)
        extra = b"fact" + struct.pack("<II", 4, T)
    else:
        fmt = struct.pack("<HHIIHH", _WAVE_PCM, nch, int(sr), int(sr) * nch * bps, nch * bps, 8 * bps)
        extra = b""
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + extra + b"data" + struct.pack("<I", len(payload)) + payload + pad
    with open(file, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def load_audio(file: str, sr: Optional[int] = None, verbose: bool = True, **kwargs) -> Tuple[torch.Tensor, AudioMetaData]:
    """io.py:25-57: audio [C, T] float32 (on the device), resampled to ``sr`` when given; ``method=`` selects the resampler set.
    Sample formats as torchaudio.load(normalize=True) scales them: int16 / 32768 (a HIP kernel), 24-bit / 2^23, int32 / 2^31,
    uint8 (x - 128) / 128, float32 as stored, float64 rounded to float32.
    ``pcm16=True`` (extension): a 16-bit file that needs no resampling is handed back as its int16 samples (on the device) — ``enhance()``
    takes them as they are."""
    method = kwargs.pop("method", "sinc_fast")
    want_pcm = bool(kwargs.pop("pcm16", False))
    frames = kwargs.get("num_frames", -1)
    off = kwargs.get("frame_offset", 0)
    if frames is not None and frames > 0 and sr is not None:
        _, rate0, _, _, _ = _read_riff(file, 0, 1)        # io.py:46-47 scales num_frames by the file's own rate: read it first
        if rate0 < sr:   # (the reference's `num_frames *= rate0 // sr` is 0 there, which torchaudio reads as "nothing": say so instead of loading the whole file)
            raise RuntimeError(f"{file}: num_frames with a file rate ({rate0}) below the requested rate ({sr}) selects no frames (df/io.py:46-47)")
        frames *= rate0 // sr
    x, orig_sr, n, bits, is_float = _read_riff(file, off, frames)
    ch = x.shape[1]
    info = AudioMetaData(sample_rate=orig_sr, num_frames=n, num_channels=ch, bits_per_sample=bits,
                         encoding="PCM_F" if is_float else ("PCM_U" if bits == 8 else "PCM_S"))
    xt = torch.from_numpy(np.array(x.T, order="C"))       # interleaved -> [C, T] (a writable copy)
    if x.dtype == np.dtype("<i2"):
        if want_pcm and (sr is None or orig_sr == sr):
            return xt.to(_lib.device()).contiguous(), info
        audio = pcm16_to_float(xt)
    else:
        xd = xt.to(_lib.device())
        if is_float:
            audio = xd.to(torch.float32)
        elif bits == 8:
            audio = (xd.to(torch.float32) - 128.0) / 128.0
        else:
            audio = xd.to(torch.float32) / float(1 << (bits - 1))
    if sr is not None and orig_sr != sr:
        if verbose:
            import warnings

            warnings.warn(f"Audio sampling rate does not match model sampling rate ({orig_sr}, {sr}). Resampling...")
        audio = resample(audio, orig_sr, sr, method=method)
    return audio.contiguous(), info


def save_audio(file: str, audio: Union[torch.Tensor, np.ndarray], sr: int, output_dir: Optional[str] = None,
               suffix: Optional[str] = None, log: bool = False, dtype=torch.int16) -> str:
    """io.py:60-84: ``dtype=torch.int16`` (default) writes 16-bit PCM (float input scaled by 2^15 and truncated, io.py:79-80, on the
    device), ``dtype=torch.float32`` a 32-bit IEEE-float WAVE (int16 input divided by 2^15, io.py:81-82) — the two files torchaudio.save
    writes for an int16 / a float32 tensor."""
    outpath = file
    if suffix is not None:
        base, ext = os.path.splitext(file)
        outpath = base + f"_{suffix}" + ext
    if output_dir is not None:
        outpath = os.path.join(output_dir, os.path.basename(outpath))
    if dtype not in (torch.int16, torch.float32):
        raise ValueError(f"save_audio: dtype must be torch.int16 or torch.float32, not {dtype}")
    audio = torch.as_tensor(audio)
    if audio.ndim == 1:
        audio = audio.unsqueeze(0)
    if dtype == torch.int16 and audio.dtype != torch.int16:
        audio = float_to_pcm16(audio)
    if dtype == torch.float32 and audio.dtype != torch.float32:
        audio = audio.to(torch.float32) / (1 << 15)
    if audio.dtype not in (torch.int16, torch.float32):   # (torchaudio.save takes what it is given: only these two reach it from the branches above)
        audio = audio.to(torch.float32)
    _write_riff(outpath, audio.cpu().numpy().T, int(sr), audio.dtype == torch.float32)
    return outpath
