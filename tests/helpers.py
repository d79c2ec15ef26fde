"""Shared test helpers: named model configs, seeded state-dicts and inputs (same recipes as tools/gen_golden.py)."""
import numpy as np
import torch

from deepfilternet_amd.config import ModelParams
from deepfilternet_amd.state_dict import random_state_dict
from oracle import libdf_oracle as L


def named_params(name: str) -> ModelParams:
    if name == "defaults":
        return ModelParams.defaults()
    if name == "df3":
        return ModelParams.deepfilternet3()
    if name in ("pf32", "pf32_nopf"):
        p = ModelParams.defaults()
        p.mask_pf, p.df_lookahead, p.conv_lookahead = name == "pf32", 1, 1
        p.df_gru_skip, p.df_pathway_kernel_size_t, p.conv_ch = "identity", 3, 32
        return p
    if name == "df3_o10":   # BASELINE.json configs[4]: deep filter of order 10 (the tiled df_convp path: 2 * df_order > 16)
        p = ModelParams.deepfilternet3()
        p.df_order, p.df_lookahead, p.conv_lookahead = 10, 3, 3
        return p
    raise KeyError(name)


GOLDEN_SEEDS = {"defaults": 0, "df3": 1, "pf32": 2}


def widths_for(p: ModelParams) -> np.ndarray:
    return L.erb_fb_widths(p.sr, p.fft_size, p.nb_erb, p.min_nb_freqs)


def torch_sd(p: ModelParams, seed: int):
    sd = random_state_dict(p, seed, widths=widths_for(p))
    return {k: torch.as_tensor(v) for k, v in sd.items()}


def rms(a):
    a = np.asarray(a)
    return float(np.sqrt(np.mean(np.abs(a) ** 2)))


def emu_subset(backend: str) -> bool:
    """True when a test case should be skipped on the CPU interpreter to keep the CPU suite within a few minutes: the case still runs
    on the GPU (-m gpu), and on the interpreter too with DFX_EMU_ALL=1."""
    import os

    return backend == "emu" and os.environ.get("DFX_EMU_ALL", "0") != "1"
