"""Soak: N full-size enhance() passes on one handle (and, with --pcm16, the 16-bit path); every output must carry the digest of the first and be finite, the
model's fault words must stay clear.  The bits of a pass do not depend on what ran before it or on which box it runs: a digest that differs is a fault.
    python tools/soak.py [--passes 300] [--batch 256] [--seconds 10]"""
import argparse
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_audio
from deepfilternet_amd.config import ModelParams
from deepfilternet_amd.enhance import enhance, init_df
from deepfilternet_amd.state_dict import random_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=300)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--seconds", type=float, default=10.0)
a = ap.parse_args()
p = ModelParams.deepfilternet3()
model, df_state, _, _ = init_df(params=p, state_dict=random_state_dict(p, 0), epoch="none")
x = synth_audio(a.batch, int(a.seconds * p.sr), 100, torch.device("cuda"))
first, bad, t0 = None, 0, time.time()
for i in range(a.passes):
    y = enhance(model, df_state, x)
    torch.cuda.synchronize()
    if not bool(torch.isfinite(y).all()):
        bad += 1
        print(f"pass {i}: non-finite output")
        continue
    d = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:16]
    if first is None:
        first = d
    elif d != first:
        bad += 1
        print(f"pass {i}: digest {d} != {first}")
print(f"{a.passes} passes of {a.batch} x {a.seconds:g} s in {time.time() - t0:.1f} s (incl. digests on the host): digest {first}, {bad} bad")
sys.exit(1 if bad else 0)
