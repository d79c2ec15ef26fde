"""PMC run for enhance()'s finishing kernel dfx_k_synthesis_rows (run under `rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv`).
Dispatches: 4 x the calibration form of dfx_k_df_apply_rows (nb_df=2, order=1, no gains: a pure stream with known byte counts, as in
tools/dev/pmc_df_apply.py), then 3 x enhance() at config-2 size (each ends with one dfx_k_synthesis_rows launch)."""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from bench import synth_audio  # noqa: E402
from deepfilternet_amd import _lib, libdf  # noqa: E402
from deepfilternet_amd.config import ModelParams  # noqa: E402
from deepfilternet_amd.enhance import enhance, init_df  # noqa: E402
from deepfilternet_amd.state_dict import random_state_dict  # noqa: E402

B, T, F, Fs, E = 256, 1002, 481, 488, 32
dev = _lib.device()
df = libdf.DF(48000, 960, 480, 32, 2)
g = torch.Generator(device=dev).manual_seed(0)
spec = torch.randn((B, T, Fs, 2), device=dev, generator=g)
out = torch.empty_like(spec)
coefs = torch.randn((B, 1, T, 2, 2), device=dev, generator=g) * 0.3
L = _lib.lib()
for _ in range(4):
    _lib.check(L.dfx_df_apply_strided(_lib.ptr(spec), Fs, _lib.ptr(coefs), 0, None, None, B, T, F, 2, 1, 0, 0.0, 0.0, _lib.ptr(out), Fs, _lib.stream()))
torch.cuda.synchronize()
del spec, out, coefs
p = ModelParams.deepfilternet3()
model, df_state, _, _ = init_df(params=p, state_dict=random_state_dict(p, 0), epoch="none")
x = synth_audio(B, 480000, 100, dev)
for _ in range(3):
    enhance(model, df_state, x)
torch.cuda.synchronize()
model.check()
