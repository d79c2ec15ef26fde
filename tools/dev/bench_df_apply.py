"""DF-apply kernel alone (BASELINE.json config 5 shape): hipEvent timing vs algorithmic bytes, plus a torch copy of the
same byte count as the achievable-bandwidth yardstick.  Usage: python tools/dev/bench_df_apply.py [order] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepfilternet_amd import _lib  # noqa: E402
from deepfilternet_amd import libdf  # noqa: E402

O = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
layout = int(sys.argv[3]) if len(sys.argv) > 3 else 2
B, T, F, nd, E = 256, 1002, 481, 96, 32
dev = _lib.device()
df = libdf.DF(48000, 960, 480, 32, 2)
g = torch.Generator(device=dev).manual_seed(0)
spec = torch.randn((B, T, F, 2), device=dev, generator=g)
shape = {0: (B, O, T, nd, 2), 1: (B, T, nd, O, 2), 2: (B, T, O, nd, 2)}[layout]
coefs = torch.randn(shape, device=dev, generator=g) * 0.3
gains = torch.rand((B, T, E), device=dev, generator=g)
out = torch.empty_like(spec)
L = _lib.lib()


def run():
    _lib.check(L.dfx_df_apply(_lib.ptr(spec), _lib.ptr(coefs), layout, _lib.ptr(gains), df.bands_handle, B, T, F, nd, O, 2 if O > 2 else 0,
                              0.0, 0.0, _lib.ptr(out), _lib.stream()))


for _ in range(3):
    run()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(iters):
    run()
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / iters
alg = (F * 8 + nd * O * 8 + E * 4 + F * 8) * B * T
print(f"df_apply O={O} layout={layout}: {ms:.4f} ms  algorithmic {alg/1e9:.3f} GB -> {alg/ms/1e6:.1f} GB/s ({alg/ms/1e6/8000:.3f} of 8 TB/s)")
# yardstick: device copy moving the same number of bytes (read n/2 + write n/2)
n = alg // 2 // 4
a = torch.empty(n, device=dev)
b = torch.empty(n, device=dev)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
ev[0].record()
for _ in range(iters):
    b.copy_(a)
ev[1].record()
torch.cuda.synchronize()
ms2 = ev[0].elapsed_time(ev[1]) / iters
print(f"torch copy of the same bytes: {ms2:.4f} ms -> {alg/ms2/1e6:.1f} GB/s")
