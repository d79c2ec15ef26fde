"""Dev: the finishing kernel of enhance() with the attenuation limit / post filter compiled in (dfx_k_synthesis_rows<5, true>), timed alone.
    [DFX_LIBRARY=...] python tools/dev/stft_kern_pf.py [launches]"""
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
    enhance(model, df_state, x, atten_lim_db=12.0)
torch.cuda.synchronize()
_lib.prof_enable(["dfx_k_synthesis"])
syn = []
for _ in range(n):
    _lib.prof_reset()
    enhance(model, df_state, x, atten_lim_db=12.0)
    torch.cuda.synchronize()
    r = _lib.prof_read()
    syn.append(r["dfx_k_synthesis"][0] / r["dfx_k_synthesis"][1])
_lib.prof_enable(None)
print(f"{os.environ.get('DFX_LIBRARY', 'default'):45s} finishing (atten_lim 12 dB) min {min(syn):.4f} median {statistics.median(syn):.4f} max {max(syn):.4f}")
