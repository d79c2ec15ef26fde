"""Dev: enhance() with the GRU recurrences on pairs of CUs (default) against one CU per 16 clips (DFX_GRU_PAIR=0), bit for bit, over batch sizes with odd group
counts / partial groups and several lengths.  Each configuration in a subprocess (the switch is read when the handle is created)."""
import os, subprocess, sys
CODE = r'''
import os, sys, hashlib, numpy as np, torch
sys.path.insert(0, os.getcwd())
from deepfilternet_amd.config import ModelParams
from deepfilternet_amd.enhance import enhance, init_df
from deepfilternet_amd.state_dict import random_state_dict
p = ModelParams.deepfilternet3()
model, st, _, _ = init_df(params=p, state_dict=random_state_dict(p, 0), epoch="none")
out = []
for B, n in [(17, 48000), (31, 96000), (33, 144000), (48, 480000), (100, 240000), (255, 480000), (256, 480000), (257, 120000)]:
    x = torch.from_numpy((0.1 * np.random.default_rng(B).standard_normal((B, n))).astype(np.float32)).cuda()
    ys = [enhance(model, st, x) for _ in range(2)]
    torch.cuda.synchronize()
    assert torch.equal(ys[0], ys[1]) and bool(torch.isfinite(ys[0]).all())
    out.append(hashlib.sha1(ys[0].cpu().numpy().tobytes()).hexdigest()[:16])
model.check()
print(" ".join(out))
'''
res = {}
for pair in ("1", "0"):
    env = dict(os.environ, DFX_GRU_PAIR=pair)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=600)
    res[pair] = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED: " + r.stderr[-400:]
    print("DFX_GRU_PAIR=" + pair, res[pair])
print("equal" if res["1"] == res["0"] and not res["1"].startswith("FAILED") else "DIFFERENT")
