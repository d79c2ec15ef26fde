"""Dev: the two STFT kernels of enhance() timed alone (stream-level concurrency off: nothing else on the chip), per launch, at the bench size.
    [DFX_LIBRARY=...] python tools/dev/stft_kern.py [launches]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import synth_audio
from deepfilternet_amd import _lib
from deepfilternet_amd.config import ModelParams
from deepfilternet_amd.enhance import enhance, init_df
from deepfilternet_amd.state_dict import random_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
p = ModelParams.deepfilternet3()
model, df_state, _, _ = init_df(params=p, state_dict=random_state_dict(p, 0), epoch="none")
x = synth_audio(256, 480000, 100, torch.device("cuda"))
model.set_streams(False)
for _ in range(2):
    enhance(model, df_state, x)
torch.cuda.synchronize()
_lib.prof_enable(["dfx_k_analysis", "dfx_k_synthesis"])
ana, syn = [], []
for _ in range(n):
    _lib.prof_reset()
    enhance(model, df_state, x)
    torch.cuda.synchronize()
    r = _lib.prof_read()
    ana.append(r["dfx_k_analysis"][0] / r["dfx_k_analysis"][1])
    syn.append(r["dfx_k_synthesis"][0] / r["dfx_k_synthesis"][1])
_lib.prof_enable(None)
fmt = lambda v: f"min {min(v):.4f} median {statistics.median(v):.4f} max {max(v):.4f}"
print(f"{os.environ.get('DFX_LIBRARY', 'default'):45s} analysis {fmt(ana)} | finishing {fmt(syn)}")
