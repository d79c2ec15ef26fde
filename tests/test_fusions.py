"""The fused kernels of the GRU phase's side work (docs/measurements.md §5f: the fan-outs around the embedding, the ERB decoder tail) against the
oracle, on the DeepFilterNet3 shape whose group structure the fusions are written for, in the serial, the event-pipelined and (on the GPU) the
persistent form of the GRU phase, with the skip connections that change what is fused.  (Up to round 5 every fusion also had a switch that
restored the separate kernels; round 6 removed those switches.)"""
import numpy as np
import pytest
import torch

from deepfilternet_amd.config import ModelParams
from deepfilternet_amd.state_dict import random_state_dict
from oracle import dfnet_oracle as O
from tests.helpers import emu_subset, rms, widths_for


def _params(variant: str) -> ModelParams:
    p = ModelParams.deepfilternet3()
    if variant == "skips":      # emb needed outside the fan-out kernel (ERB decoder skip), a residual into emb (encoder skip)
        p.emb_gru_skip_enc, p.emb_gru_skip = "groupedlinear", "groupedlinear"
    elif variant == "idskip":
        p.emb_gru_skip_enc, p.emb_gru_skip = "identity", "identity"
    elif variant == "noskip_df":
        p.df_gru_skip = "none"
    elif variant != "df3":
        raise KeyError(variant)
    return p


def _run(p, sd, x, env, monkeypatch, pipeline, mask_only=False):
    from deepfilternet_amd.enhance import enhance, init_df

    for k in ("DFX_STREAMS", "DFX_GRU_SEQ"):
        monkeypatch.delenv(k, raising=False)
    for kv in env:
        k, v = kv.split("=")
        monkeypatch.setenv(k, v)
    model, df_state, _, _ = init_df(params=p, state_dict=sd, epoch="none", mask_only=mask_only)
    if pipeline:
        model.set_pipeline(time_chunks=3, min_chunk_frames=2)
    y = enhance(model, df_state, x)
    model.check()
    return y


@pytest.mark.parametrize("form", ["serial", "pipelined", "persistent"])
@pytest.mark.parametrize("variant", ["df3", "skips", "idskip", "noskip_df", "mask_only"])
def test_fused_side_kernels_match_the_oracle_in_every_form_of_the_phase(backend, variant, form, monkeypatch):
    if backend == "emu" and form == "persistent":
        pytest.skip("the persistent GRU launch needs the GPU (the interpreter runs the event form)")
    if emu_subset(backend) and (variant in ("idskip", "noskip_df") or (form == "serial" and variant != "df3")):
        pytest.skip("interpreter subset (DFX_EMU_ALL=1 runs it); all cases run on the GPU")
    mask_only = variant == "mask_only"
    p = _params("df3" if mask_only else variant)
    sd = random_state_dict(p, 17, widths=widths_for(p))
    rng = np.random.default_rng(4)
    B, T = (2, 480 * 9 + 5) if backend == "emu" else ((33, 480 * 70 + 11) if form == "persistent" else (3, 480 * 23 + 5))
    x = torch.from_numpy((0.1 * rng.standard_normal((B, T))).astype(np.float32))
    base = {"serial": ["DFX_STREAMS=0"], "pipelined": ["DFX_GRU_SEQ=0"], "persistent": []}[form]
    y = _run(p, sd, x, base, monkeypatch, form == "pipelined", mask_only)
    y_serial = y if form == "serial" else _run(p, sd, x, ["DFX_STREAMS=0"], monkeypatch, False, mask_only)
    assert rms((y - y_serial).numpy()) < 1e-6, (variant, form)
    if not mask_only:   # (the mask-only oracle comparison lives in tests/test_config_options.py)
        rows = slice(0, 2)
        ref = O.enhance(p, {k: torch.as_tensor(v) for k, v in sd.items()}, x[rows].numpy())
        assert rms(y[rows].numpy() - ref) < 2e-6, (variant, form)
