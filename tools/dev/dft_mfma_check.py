#!/usr/bin/env python3
"""Dev: numeric model of dfx_fft480_mfma (csrc/dfx_dsp_kernels.h) — the 480-point transform as two chained fp16-split matrix products, 480 = 16 x 30,
with the kernel's scales (frame peak -> just below 2^14, matrices x 2^13, twiddle factors x 2^-17) — against numpy's double-precision FFT.
    python tools/dev/dft_mfma_check.py"""
import numpy as np

rng = np.random.default_rng(0)


def split(x):
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16).astype(np.float32)
    return hi, (x - hi).astype(np.float16).astype(np.float32)


def mm3(Ah, Al, Bh, Bl):   # hi hi + hi lo + lo hi, fp32 accumulate
    return (Al @ Bh + Ah @ Bl + Ah @ Bh).astype(np.float32)


for SG in (-1, 1):
    z = ((rng.standard_normal(480) + 1j * rng.standard_normal(480)) * 0.3).astype(np.complex64)
    ref = np.fft.fft(z.astype(np.complex128)) if SG < 0 else np.fft.ifft(z.astype(np.complex128)) * 480
    e = 14 - np.frexp(max(np.abs(z.real).max(), np.abs(z.imag).max()))[1]
    A1 = np.zeros((32, 32), np.float32)           # rows n2 (30, 31 repeat 29), k = (re | im, n1)
    for n2 in range(32):
        for k in range(32):
            v = z[30 * (k & 15) + min(n2, 29)]
            A1[n2, k] = (v.real if k < 16 else v.imag) * 2.0 ** e
    n1, k1 = np.arange(16)[:, None], np.arange(16)[None, :]
    W16 = np.exp(SG * 2j * np.pi * n1 * k1 / 16)
    B1r, B1i = np.vstack([W16.real, -W16.imag]) * 8192, np.vstack([W16.imag, W16.real]) * 8192
    Yr, Yi = mm3(*split(A1), *split(B1r)), mm3(*split(A1), *split(B1i))          # Y * 2^(e + 13), [n2][k1]
    tw = (np.exp(SG * 2j * np.pi * np.arange(32)[:, None] * k1 / 480) * 2.0 ** -17).astype(np.complex64)
    Ypr, Ypi = (Yr * tw.real - Yi * tw.imag).astype(np.float32), (Yr * tw.imag + Yi * tw.real).astype(np.float32)
    k2, n2 = np.arange(32)[:, None], np.arange(32)[None, :]
    ok = (k2 < 30) & (n2 < 30)
    Wr, Wi = np.where(ok, np.cos(2 * np.pi * k2 * n2 / 30), 0) * 8192, np.where(ok, SG * np.sin(2 * np.pi * k2 * n2 / 30), 0) * 8192
    Xr = (mm3(*split(Wr), *split(Ypr)) - mm3(*split(Wi), *split(Ypi))) * 2.0 ** (-e - 9)
    Xi = (mm3(*split(Wi), *split(Ypr)) + mm3(*split(Wr), *split(Ypi))) * 2.0 ** (-e - 9)
    X = np.array([Xr[k // 16, k % 16] + 1j * Xi[k // 16, k % 16] for k in range(480)])
    f32 = np.fft.fft(z) if SG < 0 else np.fft.ifft(z) * 480   # (numpy computes in double; the radix passes in fp32 are ~3e-8 .. 1e-7 of the peak)
    print(f"sg {SG:+d}: max |Y''| {max(np.abs(Ypr[:30]).max(), np.abs(Ypi[:30]).max()):.0f} (f16 range 65504); error {np.abs(X - ref).max() / np.abs(ref).max():.2e} of the peak")
