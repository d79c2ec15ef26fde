import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepfilternet_amd import _lib, libdf
B, T, F, E = 256, 1002, 481, 32
dev = _lib.device()
df = libdf.DF(48000, 960, 480, 32, 2)
g = torch.Generator(device=dev).manual_seed(0)
spec = torch.randn((B, T, F, 2), device=dev, generator=g)
gains = torch.rand((B, T, E), device=dev, generator=g)
out = torch.empty_like(spec)
L = _lib.lib()
def bench(nd, O, la, use_gains, layout=2, iters=20, label=""):
    shape = {0: (B, O, T, nd, 2), 1: (B, T, nd, O, 2), 2: (B, T, O, nd, 2)}[layout]
    coefs = torch.randn(shape, device=dev, generator=g) * 0.3
    def run():
        _lib.check(L.dfx_df_apply(_lib.ptr(spec), _lib.ptr(coefs), layout, _lib.ptr(gains) if use_gains else None, df.bands_handle if use_gains else None,
                                  B, T, F, nd, O, la, 0.0, 0.0, _lib.ptr(out), _lib.stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters): run()
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / iters
    alg = (F * 8 + nd * O * 8 + (E * 4 if use_gains else 0) + F * 8) * B * T
    print(f"{label:28s} nd={nd} O={O} la={la} gains={use_gains} layout={layout}: {ms:.4f} ms  {alg/1e9:.3f} GB -> {alg/ms/1e6:.0f} GB/s")
bench(2, 1, 0, True, label="~pure stream + gains")
bench(2, 1, 0, False, label="~pure copy")
bench(96, 1, 0, True, label="1 tap")
bench(96, 5, 2, True, label="5 taps (DF3)")
bench(96, 5, 0, True, label="5 taps causal")
bench(96, 5, 2, False, label="5 taps no gains")
bench(96, 10, 2, True, label="10 taps")
bench(96, 5, 2, True, layout=0, label="5 taps BOTF")
