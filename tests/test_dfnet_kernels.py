"""DeepFilterNet3 forward on the HIP engine vs the reference's own PyTorch modules (tests/golden/dfnet_*.npz, generated
from /root/reference) and vs the torch oracle on other shapes.  'emu' = kernel sources on the CPU SIMT interpreter."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from deepfilternet_amd.state_dict import random_state_dict, state_dict_manifest
from oracle import dfnet_oracle as O
from tests.helpers import GOLDEN_SEEDS, named_params, torch_sd, widths_for

CFGS = ["defaults", "df3", "pf32"]


def _model(name, seed):
    from deepfilternet_amd.model import DfNet

    p = named_params(name)
    return p, DfNet(p, random_state_dict(p, seed))


def _cmp(a, b, tol, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max()
    assert err <= tol * max(1.0, np.abs(b).max()), (what, err)


@pytest.mark.parametrize("name", CFGS)
def test_c_manifest_matches_reference_names(backend, name):
    from deepfilternet_amd.model import make_cfg, tensor_manifest

    p = named_params(name)
    man = state_dict_manifest(p)
    got = tensor_manifest(make_cfg(p))
    for n, shape, _ in got:
        assert n in man and tuple(man[n]) == tuple(shape), n
    skip = ("erb_fb", "mask.erb_inv_fb", "df_dec.df_fc_a")
    want = [k for k in man if not k.endswith("num_batches_tracked") and not k.startswith(skip)]
    assert [n for n, _, _ in got] == want


@pytest.mark.parametrize("name", CFGS)
def test_forward_matches_reference_golden(backend, name, golden_dir):
    g = np.load(os.path.join(golden_dir, f"dfnet_{name}.npz"))
    p, model = _model(name, GOLDEN_SEEDS[name])
    spec_e, m, lsnr, coefs = model(torch.from_numpy(g["spec"]), torch.from_numpy(g["feat_erb"]), torch.from_numpy(g["feat_spec"]))
    _cmp(m.cpu(), g["m"], 2e-5, "mask")
    _cmp(lsnr.cpu(), g["lsnr"], 2e-5, "lsnr")
    _cmp(coefs.cpu(), g["df_coefs"], 3e-5, "df_coefs")
    _cmp(spec_e.cpu(), g["spec_e"], 3e-5, "spec_e")
    assert coefs.shape == (g["spec"].shape[0], p.df_order, g["spec"].shape[2], p.nb_df, 2)


@pytest.mark.parametrize("name,B,T", [("df3", 3, 21), ("defaults", 5, 9), ("pf32", 1, 1), ("df3_o10", 2, 13)])
def test_forward_matches_oracle_other_shapes(backend, name, B, T):
    """ragged sizes: B not a multiple of the GRU row tile, T not a multiple of the time tiles, single frame."""
    p, model = _model(name, 11)
    sd = torch_sd(p, 11)
    rng = np.random.default_rng(B + T)
    spec = torch.from_numpy((0.05 * rng.standard_normal((B, 1, T, p.freq_bins, 2))).astype(np.float32))
    fe = torch.from_numpy((0.5 * rng.standard_normal((B, 1, T, p.nb_erb))).astype(np.float32))
    fs = torch.from_numpy(rng.standard_normal((B, 1, T, p.nb_df, 2)).astype(np.float32))
    ref = O.dfnet_forward(p, sd, widths_for(p), spec, fe, fs)
    spec_e, m, lsnr, coefs = model(spec, fe, fs)
    _cmp(m.cpu(), ref["m"], 3e-5, "mask")
    _cmp(lsnr.cpu(), ref["lsnr"], 3e-5, "lsnr")
    _cmp(coefs.cpu(), ref["df_coefs"], 5e-5, "df_coefs")
    _cmp(spec_e.cpu(), ref["spec_e"], 5e-5, "spec_e")


def test_unsupported_configs_fail_loudly(backend):
    from deepfilternet_amd.model import DfNet

    p = named_params("defaults")
    p.conv_ch = 48
    with pytest.raises(Exception, match="conv_ch|unsupported"):
        DfNet(p, random_state_dict(p, 0))
    p = named_params("defaults")
    p.emb_hidden_dim = 128
    with pytest.raises(NotImplementedError, match="hidden size"):
        DfNet(p, {})
    p = named_params("defaults")
    sd = random_state_dict(p, 0)
    del sd["enc.df_fc_emb.0.weight"]
    with pytest.raises(KeyError, match="df_fc_emb"):
        DfNet(p, sd)
