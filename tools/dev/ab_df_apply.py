"""Dev: the same dfx_df_apply call timed on two builds of the library (argv: lib paths), alternating, same box."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepfilternet_amd import _lib, libdf
B, T, F, E, nd, O, la = 256, 1002, 481, 32, 96, 5, 2
dev = _lib.device()
df = libdf.DF(48000, 960, 480, 32, 2)
g = torch.Generator(device=dev).manual_seed(0)
spec = torch.randn((B, T, F, 2), device=dev, generator=g)
gains = torch.rand((B, T, E), device=dev, generator=g)
coefs = torch.randn((B, O, T, nd, 2), device=dev, generator=g) * 0.3
out = torch.empty_like(spec)
libs = []
for p in sys.argv[1:]:
    L = ctypes.CDLL(os.path.abspath(p))
    L.dfx_df_apply.restype = ctypes.c_int
    L.dfx_df_apply.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    L.dfx_bands_create.restype = ctypes.c_int
    h = ctypes.c_void_p()
    libs.append((p, L, None))
def run(L, bands):
    rc = L.dfx_df_apply(spec.data_ptr(), coefs.data_ptr(), 0, gains.data_ptr(), bands, B, T, F, nd, O, la, 0.0, 0.0, out.data_ptr(), None)
    assert rc == 0, rc
alg = (F * 8 + nd * O * 8 + E * 4 + F * 8) * B * T
for rep in range(3):
    for p, L, st in libs:
        # each library needs its own band table handle: build one from the erb widths through its own API
        w = (ctypes.c_uint64 * 32)(*[int(x) for x in df.erb_widths()])
        bh = ctypes.c_void_p()
        assert L.dfx_bands_create(w, 32, ctypes.byref(bh)) == 0
        for _ in range(3): run(L, bh)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(L, bh)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{os.path.basename(p):20s} {ms:.4f} ms  {alg/ms/1e6:.0f} GB/s")
