"""dev: repeat the fused ERB-feature check (tests/test_dsp_kernels.py::test_fused_erb_feature_band_layouts, 960 / 240 / 64 bands) many times per
transform form and report every mismatch (a flaky failure was seen once on the GPU with the matrix-pipe transform)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from deepfilternet_amd import libdf as D
from deepfilternet_amd.enhance import _norm_alpha, df_features
from oracle import libdf_oracle as L
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for (N, H, nb, minf) in ((960, 240, 64, 1), (960, 480, 32, 2), (960, 480, 80, 1)):
    rng = np.random.default_rng(N + nb)
    x = (0.2 * rng.standard_normal((2, H * 9))).astype(np.float32)
    o = L.DF(48000, N, H, nb, minf)
    S = o.analysis(x)
    FE = L.erb_norm(L.erb(S, o.erb_widths()), _norm_alpha(D.DF(48000, N, H, nb, minf)))
    for seg in ("0", "1"):
        os.environ["DFX_ERB_SEGMENTS"] = seg
        for mf in ("1", "0"):
            os.environ["DFX_FFT_MFMA"] = mf   # ("1" selects the matrix-pipe transform, anything else the radix passes)
            nbad = 0
            for r in range(reps):
                d = D.DF(48000, N, H, nb, minf)
                sp, fe, _ = df_features(torch.from_numpy(x), d, 96)
                fe = fe.squeeze(1).cpu().numpy()
                sc = torch.view_as_complex(sp.squeeze(1).cpu()).numpy()
                if np.abs(fe - FE).max() > 2e-5 or np.abs(sc - S).max() > 1e-6:
                    nbad += 1
                    if nbad <= 3:
                        bad = np.argwhere(np.abs(fe - FE) > 2e-5)
                        print("   MISMATCH", (N, H, nb), "seg", seg, "mf", mf, "rep", r, "fe err", np.abs(fe - FE).max(), "at", bad[:6].tolist(), "spec err", np.abs(sc - S).max(),
                              np.unravel_index(np.abs(sc - S).argmax(), S.shape))
            print((N, H, nb), "segments", seg, "mfma", mf, "bad", nbad, "of", reps)
