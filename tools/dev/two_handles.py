"""Dev: two model handles driven from two host threads at once (the follower guard: only one handle's pass takes followers at a time)."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.from_numpy((0.1 * np.random.default_rng(1).standard_normal((B, 96000))).astype(np.float32)).cuda()
models = [init_df(params=p, state_dict=random_state_dict(p, 0), epoch="none")[:2] for _ in range(2)]
ref = enhance(models[0][0], models[0][1], x).cpu()
torch.cuda.synchronize()
outs, errs = [[], []], []


def work(i):
    try:
        for _ in range(6):
            outs[i].append(enhance(models[i][0], models[i][1], x).cpu())
        models[i][0].check()
    except Exception as e:   # noqa: BLE001
        errs.append(repr(e)[:300])


ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
[t.start() for t in ts]
[t.join() for t in ts]
print("errors", errs)
print("equal", all(torch.equal(o, ref) for oo in outs for o in oo), [len(o) for o in outs])
for i in range(2):
    for j, o in enumerate(outs[i]):
        if not torch.equal(o, ref):
            d = (o - ref).abs()
            bad = (d.amax(dim=1) > 0).nonzero().flatten().tolist()
            cols = (d.amax(dim=0) > 0).nonzero().flatten()
            print(f"  handle {i} call {j}: max diff {float(d.max()):.3e}, clips {bad[:20]}{'...' if len(bad) > 20 else ''}, samples {int(cols.min())}..{int(cols.max())} ({len(cols)})")
