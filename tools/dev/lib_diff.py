"""Dev: enhance() of the same input under two builds of the library (each in its own process), compared sample by sample.
    python tools/dev/lib_diff.py <libA.so|-> <libB.so|-> [clips [samples]]"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, REPO)
    import numpy as np
    import torch
    from bench import synth_audio
    from deepfilternet_amd.config import ModelParams
    from deepfilternet_amd.enhance import enhance, init_df
    from deepfilternet_amd.state_dict import random_state_dict

    out, clips, samples = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    p = ModelParams.deepfilternet3()
    model, df_state, _, _ = init_df(params=p, state_dict=random_state_dict(p, 0), epoch="none")
    x = synth_audio(clips, samples, 100, torch.device("cuda"))
    y = enhance(model, df_state, x)
    torch.cuda.synchronize()
    np.save(out, y.cpu().numpy())
    sys.exit(0)

import numpy as np

la, lb = sys.argv[1], sys.argv[2]
clips = int(sys.argv[3]) if len(sys.argv) > 3 else 37
samples = int(sys.argv[4]) if len(sys.argv) > 4 else 100001
ys = []
for i, l in enumerate((la, lb)):
    env = dict(os.environ)
    if l != "-":
        env["DFX_LIBRARY"] = l
    f = f"/tmp/lib_diff_{i}.npy"
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", f, str(clips), str(samples)], env=env, check=True)
    ys.append(np.load(f))
a, b = ys
d = np.abs(a.astype(np.float64) - b.astype(np.float64))
print("shape", a.shape, "rms", float(np.sqrt((a.astype(np.float64) ** 2).mean())), "max abs diff", float(d.max()), "differing samples", int((a != b).sum()), "of", a.size)
if d.max() > 0:
    r, c = np.unravel_index(np.argmax(d), d.shape)
    print("worst at clip", r, "sample", c, "frame", c // 480, "pos in hop", c % 480, a[r, c], b[r, c])
    fr = np.nonzero((a != b).any(axis=0))[0] // 480
    print("frames with differences:", len(np.unique(fr)), "first", np.unique(fr)[:12])
    pos = np.bincount(np.nonzero(a != b)[1] % 480, minlength=480)
    print("positions in the hop with most differences:", np.argsort(-pos)[:8], pos.max(), pos.min())
