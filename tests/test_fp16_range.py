"""Range of the fp16-split ("fp16x3") matrix kernels — the default arithmetic of the GRU projections / recurrences and of the fused
DF-encoder convolutions (x = hi + lo in f16, three f16 MFMA products, fp32 accumulation; DESIGN.md §4, docs/measurements.md §5a).

f16 covers 6e-5 .. 65504: inputs far outside O(1) must either keep fp32-like accuracy or fail loudly, never return garbage quietly:
  * the GRU input projections scale every activation row by a power of two before the split (any finite row is safe);
  * the GRU state is bounded by construction (|h| < 1);
  * the fused df_conv0 -> df_conv1 / df_convp kernels — and, since round 5, the ERB decoder tail's erb_conv0 / conv0_out matrix-op forms —
    track the largest magnitude they split and report >= 6e4 through dfx_model_check() (DfNet.check()); DFX_EXACT_FP32=1 selects the exact
    fp32 kernels, which have no such limit.
Features scaled by 1e-4 .. 1e+4 exercise all of this against the torch oracle."""
import numpy as np
import pytest
import torch

from deepfilternet_amd.config import ModelParams
from deepfilternet_amd.state_dict import random_state_dict
from oracle import dfnet_oracle as O
from tests.helpers import emu_subset, widths_for


def _inputs(p, B, T, scale, seed=0):
    rng = np.random.default_rng(seed)
    spec = torch.from_numpy((0.05 * rng.standard_normal((B, 1, T, p.freq_bins, 2))).astype(np.float32))
    fe = torch.from_numpy((scale * 0.5 * rng.standard_normal((B, 1, T, p.nb_erb))).astype(np.float32))
    fs = torch.from_numpy((scale * rng.standard_normal((B, 1, T, p.nb_df, 2))).astype(np.float32))
    return spec, fe, fs


def _close(a, b, tol, what):
    a, b = np.asarray(a), np.asarray(b)
    err = np.abs(a - b).max()
    assert np.isfinite(a).all() and err <= tol * max(1.0, np.abs(b).max()), (what, err)


@pytest.mark.parametrize("scale", [1e-4, 1e-2, 1e2, 1e4])
def test_scaled_activations(backend, scale, monkeypatch):
    from deepfilternet_amd import _lib
    from deepfilternet_amd.model import DfNet

    if emu_subset(backend) and scale in (1e-2, 1e2):
        pytest.skip("interpreter subset: the two extreme scales run here, all four on the GPU (DFX_EMU_ALL=1 runs all)")
    p = ModelParams.deepfilternet3()
    sd = random_state_dict(p, 7, widths=widths_for(p))
    sdt = {k: torch.as_tensor(v) for k, v in sd.items()}
    B, T = (2, 7) if backend == "emu" else (5, 37)
    spec, fe, fs = _inputs(p, B, T, scale)
    ref = O.dfnet_forward(p, sdt, widths_for(p), spec, fe, fs)
    c0_max = float(ref["c0"].abs().max())          # the largest value the fused DF-encoder kernels have to split
    # round 5: conv0_out's operand (d1 + conv0p(e0)) and erb_conv0's feature patch are split inside dfx_k_erb_tail as well (matrix-pipe forms)
    co_in = O.conv_norm_act(ref["e0"], sdt, "erb_dec.conv0p", p.conv_ch, p.conv_ch, (1, 1)) + ref["d1"]
    c0_max = max(c0_max, float(co_in.abs().max()), float(fe.abs().max()))
    model = DfNet(p, sd)
    if c0_max >= 6.0e4:
        # outside the f16 range: the guard must have fired — loudly: from the call itself if its pass is already over (the interpreter
        # is synchronous), from check() otherwise
        with pytest.raises(_lib.DfxError, match="fp16-split"):
            model(spec, fe, fs)
            model.check()
        try:                                        # (a kernel late in the same pass — the decoder tail — may raise the word again after the
            model.check()                           #  first report: one more report belongs to the same pass)
        except _lib.DfxError:
            pass
        model.check()                               # the error word is cleared by the report
    else:
        spec_e, m, lsnr, coefs = model(spec, fe, fs)
        model.check()
        _close(m.cpu(), ref["m"], 3e-5, "mask")
        _close(lsnr.cpu(), ref["lsnr"], 3e-5, "lsnr")
        _close(coefs.cpu(), ref["df_coefs"], 5e-5, "df_coefs")
        _close(spec_e.cpu(), ref["spec_e"], 5e-5, "spec_e")
    # the exact fp32 kernels have no range limit: they match the oracle at every scale (at 1e4 the pre-activations of the sigmoids /
    # tanhs are ~1e4 themselves, so fp32 rounding of EITHER implementation moves the few unsaturated outputs by ~1e-4)
    monkeypatch.setenv("DFX_EXACT_FP32", "1")
    exact = DfNet(p, sd)
    spec_e2, m2, lsnr2, coefs2 = exact(spec, fe, fs)
    exact.check()
    k = 30.0 if scale >= 1e3 else 1.0
    _close(m2.cpu(), ref["m"], k * 3e-5, "mask (exact)")
    _close(lsnr2.cpu(), ref["lsnr"], k * 3e-5, "lsnr (exact)")
    _close(coefs2.cpu(), ref["df_coefs"], k * 1e-4, "df_coefs (exact)")
    _close(spec_e2.cpu(), ref["spec_e"], k * 1e-4, "spec_e (exact)")


def test_projection_rows_of_any_magnitude(backend):
    """dfx_k_proj256_h3 scales each row by a power of two before the f16 split: rows of 1e-6 and rows of 1e+6 in one batch keep fp32-like
    accuracy relative to their own scale (without the scaling the tiny rows lose their lo halves to the f16 subnormals and the large
    ones overflow to inf).  Exercised through the GRU stack of the ERB decoder: emb rows of wildly different magnitudes."""
    from deepfilternet_amd.model import DfNet

    p = ModelParams.deepfilternet3()
    sd = random_state_dict(p, 8, widths=widths_for(p))
    # make linear_in pass magnitudes through (positive weights keep ReLU open), so the GRU projection sees rows ~ scale
    sdt = {k: torch.as_tensor(v) for k, v in sd.items()}
    B, T = 3, 5
    spec, fe, fs = _inputs(p, B, T, 1.0, seed=3)
    fe[0] *= 1e-5
    fs[0] *= 1e-5
    fe[2] *= 3e2
    fs[2] *= 3e2
    ref = O.dfnet_forward(p, sdt, widths_for(p), spec, fe, fs)
    assert float(ref["c0"].abs().max()) < 6.0e4
    model = DfNet(p, sd)
    spec_e, m, lsnr, coefs = model(spec, fe, fs)
    model.check()
    for b in range(B):   # per clip: every clip is accurate relative to its own outputs
        _close(m[b].cpu(), ref["m"][b], 3e-5, f"mask[{b}]")
        _close(lsnr[b].cpu(), ref["lsnr"][b], 3e-5, f"lsnr[{b}]")
        _close(coefs[b].cpu(), ref["df_coefs"][b], 5e-5, f"df_coefs[{b}]")
