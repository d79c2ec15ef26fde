"""Dev: one model handle while an unrelated kernel stream (torch matmuls from another host thread) shares the GPU."""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from deepfilternet_amd.config import ModelParams
from deepfilternet_amd.enhance import enhance, init_df
from deepfilternet_amd.state_dict import random_state_dict

p = ModelParams.deepfilternet3()
x = torch.from_numpy((0.1 * np.random.default_rng(1).standard_normal((256, 96000))).astype(np.float32)).cuda()
model, st = init_df(params=p, state_dict=random_state_dict(p, 0), epoch="none")[:2]
ref = enhance(model, st, x).cpu()
torch.cuda.synchronize()
stop = False
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048


def noise():
    s = torch.cuda.Stream()
    a = torch.randn(n, n, device="cuda")
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(20):
                a = torch.tanh(a @ a * 1e-3)
            s.synchronize()


t = threading.Thread(target=noise)
t.start()
bad, errs = 0, []
try:
    for i in range(12):
        y = enhance(model, st, x).cpu()
        if not torch.equal(y, ref):
            bad += 1
            d = (y - ref).abs()
            print(f"  call {i}: max diff {float(d.max()):.3e}, clips {(d.amax(dim=1) > 0).nonzero().flatten().tolist()[:12]}")
    model.check()
except Exception as e:   # noqa: BLE001
    errs.append(repr(e)[:200])
stop = True
t.join()
print("wrong calls", bad, "of 12; errors", errs)
