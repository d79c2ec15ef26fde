"""Dev: the bits of enhance() under the engine switches of the environment — one digest per call.
    DFX_SEQ_FOLLOW=0 python tools/dev/follow_check.py [calls [clips [samples]]]     (compare the digests with those of a run without the switch;
    default: the bench size, 256 clips x 480000 samples)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import synth_audio
from deepfilternet_amd.config import ModelParams
from deepfilternet_amd.enhance import enhance, init_df
from deepfilternet_amd.state_dict import random_state_dict

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 6
p = ModelParams.deepfilternet3()
model, df_state, _, _ = init_df(params=p, state_dict=random_state_dict(p, 0), epoch="none")
clips = int(sys.argv[2]) if len(sys.argv) > 2 else 256
samples = int(sys.argv[3]) if len(sys.argv) > 3 else 480000
x = synth_audio(clips, samples, 100, torch.device("cuda"))
digests = []
for i in range(calls):
    y = enhance(model, df_state, x)
    torch.cuda.synchronize()
    digests.append(hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16])
print("digests", " ".join(digests))
print("all equal", len(set(digests)) == 1)
