"""Streaming (dfx_stream_*): the frame loop of the reference's DfTract::process (libDF/src/tract.rs:509-642).  The oracle is the
batch path, which is pinned against the reference's goldens: the concatenated streaming output equals enhance(pad=False) of the
whole signal delayed by the model's lookahead, however the signal is cut into calls; the first `lookahead` hops are silence."""
import numpy as np
import pytest
import torch

from oracle import dfnet_oracle as O
from tests.helpers import emu_subset, named_params, rms, torch_sd


def _run_stream(rt, x, cuts):
    hop = rt.frame_length
    out, pos = [], 0
    for n in cuts:
        out.append(rt.process(x[:, pos * hop:(pos + n) * hop]))
        pos += n
    assert pos * hop == x.shape[1]
    return torch.cat(out, dim=1)


@pytest.mark.parametrize("name", ["defaults", "df3", "pf32"])
def test_stream_equals_batch_delayed(backend, name):
    from deepfilternet_amd.enhance import enhance, init_df
    from deepfilternet_amd.streaming import DfStream

    if backend == "emu" and name != "pf32":
        pytest.skip("the interpreter is slow: it covers the conv_ch=32 model (kt=3, lookahead 1); the GPU run covers all three")
    p = named_params(name)
    model, df_state, _, _ = init_df(params=p, epoch="none", seed=9)
    hop, T = 480, (9 if backend == "emu" else 23)   # the interpreter is slow: fewer hops and cut patterns there
    rng = np.random.default_rng(2)
    x = torch.from_numpy((0.1 * rng.standard_normal((3, hop * T))).astype(np.float32))
    ref = enhance(model, df_state, x, pad=False)                      # batch path (itself checked against the oracle below)
    assert rms(ref.numpy() - O.enhance(p, torch_sd(p, 9), x.numpy(), pad=False)) < 2e-6
    rt = DfStream(model, df_state, streams=3, max_frames=7)
    d = rt.delay_frames
    assert d == p.df_lookahead and rt.frame_length == hop
    for cuts in (([1, 1, 1, 3, 1, 2],) if backend == "emu" else ([1] * T, [7, 7, 7, 2], [3, 1, 5, 2, 7, 1, 4])):
        rt.reset()
        y = _run_stream(rt, x, cuts)
        assert y.shape == x.shape
        if d:
            assert float(y[:, : d * hop].abs().max()) == 0.0  # warm-up hops are silence
        err = rms((y[:, d * hop:] - ref[:, : (T - d) * hop]).numpy())
        assert err < 1e-6, (cuts, err)
    model.check()


def test_stream_controls(backend):
    from deepfilternet_amd.enhance import enhance, init_df
    from deepfilternet_amd.streaming import DfStream

    if emu_subset(backend):
        pytest.skip("interpreter subset: the controls are exercised through test_streaming_gated.py there (DFX_EMU_ALL=1 runs this too)")
    p = named_params("pf32")
    model, df_state, _, _ = init_df(params=p, epoch="none", seed=3)
    hop, T = 480, (8 if backend == "emu" else 12)
    rng = np.random.default_rng(4)
    x = torch.from_numpy((0.1 * rng.standard_normal((2, hop * T))).astype(np.float32))
    rt = DfStream(model, df_state, streams=2, max_frames=4)
    d = rt.delay_frames
    # attenuation limit: same mix as enhance(atten_lim_db=...)
    rt.set_atten_lim(12.0)
    y = _run_stream(rt, x, [4] * (T // 4))
    ref = enhance(model, df_state, x, pad=False, atten_lim_db=12.0)
    assert rms((y[:, d * hop:] - ref[:, : (T - d) * hop]).numpy()) < 1e-6
    # |dB| < 0.01: the reference passes the input through untouched and undelayed (tract.rs:540-543)
    rt.reset()
    rt.set_atten_lim(0.0)
    assert torch.equal(rt.process(x[:, : 4 * hop]), x[:, : 4 * hop])
    # >= 100 dB switches the limit off again
    rt.reset()
    rt.set_atten_lim(100.0)
    y = _run_stream(rt, x, [4] * (T // 4))
    ref = enhance(model, df_state, x, pad=False)
    assert rms((y[:, d * hop:] - ref[:, : (T - d) * hop]).numpy()) < 1e-6
    lsnr = rt.process(x[:, : hop], return_lsnr=True)[1]
    assert lsnr.shape == (2, 1) and bool(torch.isfinite(lsnr).all())
    with pytest.raises(ValueError):
        rt.process(x[:, : 5 * hop])      # more than max_frames
    with pytest.raises(ValueError):
        rt.process(x[:, : hop + 1])      # not a whole number of hops
