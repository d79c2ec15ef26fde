"""Dev: where the workgroups of the persistent GRU launch (dfx_k_gru_seq) spend their time — waiting for a chunk's input projection
vs running the chunk's steps.  DFX_SEQ_TRACE=1 python tools/dev/seq_trace.py"""
import ctypes as C
import os
import sys

os.environ["DFX_SEQ_TRACE"] = "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from bench import synth_audio
from deepfilternet_amd import _lib
from deepfilternet_amd.config import ModelParams
from deepfilternet_amd.enhance import enhance, init_df
from deepfilternet_amd.state_dict import random_state_dict

p = ModelParams.deepfilternet3()
model, df_state, _, _ = init_df(params=p, state_dict=random_state_dict(p, 0), epoch="none")
x = synth_audio(256, 480000, 100, torch.device("cuda"))
for _ in range(4):
    enhance(model, df_state, x)
torch.cuda.synchronize()
L = C.CDLL(_lib.library_path())
buf = np.zeros(8 * 64 * 96 * 3, np.uint64)
dims = (C.c_int * 3)()
L.dfx_model_seq_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
rc = L.dfx_model_seq_trace(model.handle, buf.ctypes.data, buf.size, dims)
assert rc == 0, rc
nl, ng, K = dims[0], dims[1], dims[2]
t = buf[: nl * ng * K * 3].reshape(nl, ng, K, 3).astype(np.float64) / 100.0   # microseconds
t0 = t[..., 0].min()
t -= t0
def bounds(T=1002, K0=int(os.environ.get("DFX_SEQ_CHUNKS", "16" if os.environ.get("DFX_SEQ_FOLLOW", "2") == "0" else "12")), ramp0=int(os.environ.get("DFX_SEQ_RAMP", "0"))):
    body, sizes, down, left = max(-(-T // K0), 32), [], [], T
    r = ramp0
    while ramp0 > 0 and r < body and left > 4 * body:
        sizes.append(r); left -= r; r *= 2
    r = ramp0
    while ramp0 > 0 and r < body and left > 3 * body:
        down.append(r); left -= r; r *= 2
    nbody = max(1, min(-(-left // body), 96 - len(sizes) - len(down)))
    sizes += [left * (i + 1) // nbody - left * i // nbody for i in range(nbody)]
    return np.array(sizes + down[::-1], dtype=np.float64)


steps = bounds()
assert len(steps) == K, (len(steps), K)
print("chunk sizes", steps.astype(int).tolist())
print(f"layers {nl} groups {ng} chunks {K}; phase length {t[..., 2].max() / 1e3:.3f} ms")
for l in range(nl):
    wait = (t[l, :, :, 1] - t[l, :, :, 0]).mean(axis=0)
    run = (t[l, :, :, 2] - t[l, :, :, 1]).mean(axis=0)
    if K > 16:   # many chunks: one summary line per layer
        mid = slice(K // 4, K - K // 4)
        print(f"layer {l}: first chunk starts {t[l, :, 0, 1].mean() / 1e3:6.3f} ms, last ends {t[l, :, -1, 2].mean() / 1e3:6.3f} ms | waits after the first: "
              f"{wait[1:].sum():6.0f} us in all | us/step over the middle half: {(run[mid].sum() / steps[mid].sum()):5.2f} (first chunk {run[0] / steps[0]:5.2f}, last {run[-1] / steps[-1]:5.2f})")
        continue
    print(f"layer {l}: first chunk starts {t[l, :, 0, 1].mean() / 1e3:6.3f} ms, last ends {t[l, :, -1, 2].mean() / 1e3:6.3f} ms | "
          f"wait per chunk (us): {' '.join(f'{v:5.0f}' for v in wait)} | us/step: {' '.join(f'{v:5.2f}' for v in run / steps)}")
