#!/usr/bin/env python3
"""Throughput of the steps either side of enhance() (csrc/dfx_io.hip) on one MI355X: PCM16 -> float, 44.1 kHz -> 48 kHz sinc
resampling (df.io.resample, "sinc_fast"), float -> PCM16, on a batch of 256 clips x 10 s.  One JSON line per step."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    from deepfilternet_amd import _lib
    from deepfilternet_amd import io as dio
    import ctypes as C

    dev = _lib.device()
    B, sr, new = 256, 44100, 48000
    T = sr * 10
    pcm = torch.randint(-20000, 20000, (B, T), dtype=torch.int16, device=dev)
    ms = timed(lambda: dio.pcm16_to_float(pcm))
    print(json.dumps({"step": "pcm16_to_f32", "ms": ms, "GB/s": B * T * 6 / ms / 1e6}))
    x = dio.pcm16_to_float(pcm)
    ph, taps = C.c_int(), C.c_int()
    _lib.check(_lib.lib().dfx_resampler_kernel(sr, new, 16, 0.99, 0, 0.0, C.byref(ph), C.byref(taps), None, None, 0))   # sinc_fast
    n, ntaps = ph.value, taps.value
    y = dio.resample(x, sr, new)
    ms = timed(lambda: dio.resample(x, sr, new))
    flops = 2.0 * y.numel() * ntaps
    print(json.dumps({"step": "resample 44100->48000 sinc_fast", "ms": ms, "taps": ntaps, "phases": n,
                      "TFLOP/s_fp32_valu": flops / ms / 1e9, "GB/s_algorithmic": (x.numel() + y.numel()) * 4 / ms / 1e6,
                      "audio_seconds_per_second": B * 10 / (ms / 1e3)}))
    ms = timed(lambda: dio.float_to_pcm16(y))
    print(json.dumps({"step": "f32_to_pcm16", "ms": ms, "GB/s": y.numel() * 6 / ms / 1e6}))


if __name__ == "__main__":
    main()
