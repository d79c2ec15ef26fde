"""Model / DSP hyper-parameters with the reference's INI keys, sections, defaults and env-var override rule.

Mirrors (does not import) the reference's config machinery for the options the enhance() hot path reads:
  * ``[df]``            DeepFilterNet/df/config.py:12-39          (DfParams)
  * ``[deepfilternet]`` DeepFilterNet/df/deepfilternet3.py:25-77  (ModelParams)
  * ``[train] model``   DeepFilterNet/df/model.py:9-15
  * lookup order        DeepFilterNet/df/config.py:104-141: an environment variable named like the upper-case option
    wins over the INI value, the INI value wins over the default.
  * legacy key moves    DeepFilterNet/df/config.py:171-206 (``df_order``/``df_lookahead`` from [deepfilternet] to [df]).
"""
from __future__ import annotations

import math
import os
from configparser import ConfigParser
from dataclasses import dataclass, field, fields
from typing import Dict, Optional, Tuple


def _csv_int(s) -> Tuple[int, ...]:
    if isinstance(s, (tuple, list)):
        return tuple(int(v) for v in s)
    return tuple(int(v) for v in str(s).replace(" ", "").split(",") if v != "")


def _bool(s) -> bool:
    if isinstance(s, bool):
        return s
    v = str(s).lower()
    if v in {"true", "yes", "y", "on", "1"}:
        return True
    if v in {"false", "no", "n", "off", "0"}:
        return False
    raise ValueError("Parse error")


# (attribute, INI option, section, cast, default)
_OPTIONS = [
    ("sr", "SR", "df", int, 48_000),
    ("fft_size", "FFT_SIZE", "df", int, 960),
    ("hop_size", "HOP_SIZE", "df", int, 480),
    ("nb_erb", "NB_ERB", "df", int, 32),
    ("nb_df", "NB_DF", "df", int, 96),
    ("norm_tau", "NORM_TAU", "df", float, 1.0),
    ("lsnr_max", "LSNR_MAX", "df", int, 35),
    ("lsnr_min", "LSNR_MIN", "df", int, -15),
    ("min_nb_freqs", "MIN_NB_ERB_FREQS", "df", int, 2),
    ("df_order", "DF_ORDER", "df", int, 5),
    ("df_lookahead", "DF_LOOKAHEAD", "df", int, 0),
    ("pad_mode", "PAD_MODE", "df", str, "input"),
    ("conv_lookahead", "CONV_LOOKAHEAD", "deepfilternet", int, 0),
    ("conv_ch", "CONV_CH", "deepfilternet", int, 16),
    ("conv_depthwise", "CONV_DEPTHWISE", "deepfilternet", _bool, True),
    ("convt_depthwise", "CONVT_DEPTHWISE", "deepfilternet", _bool, True),
    ("conv_kernel", "CONV_KERNEL", "deepfilternet", _csv_int, (1, 3)),
    ("convt_kernel", "CONVT_KERNEL", "deepfilternet", _csv_int, (1, 3)),
    ("conv_kernel_inp", "CONV_KERNEL_INP", "deepfilternet", _csv_int, (3, 3)),
    ("emb_hidden_dim", "EMB_HIDDEN_DIM", "deepfilternet", int, 256),
    ("emb_num_layers", "EMB_NUM_LAYERS", "deepfilternet", int, 2),
    ("emb_gru_skip_enc", "EMB_GRU_SKIP_ENC", "deepfilternet", str, "none"),
    ("emb_gru_skip", "EMB_GRU_SKIP", "deepfilternet", str, "none"),
    ("df_hidden_dim", "DF_HIDDEN_DIM", "deepfilternet", int, 256),
    ("df_gru_skip", "DF_GRU_SKIP", "deepfilternet", str, "none"),
    ("df_pathway_kernel_size_t", "DF_PATHWAY_KERNEL_SIZE_T", "deepfilternet", int, 1),
    ("enc_concat", "ENC_CONCAT", "deepfilternet", _bool, False),
    ("df_num_layers", "DF_NUM_LAYERS", "deepfilternet", int, 3),
    ("df_n_iter", "DF_N_ITER", "deepfilternet", int, 1),
    ("lin_groups", "LINEAR_GROUPS", "deepfilternet", int, 1),
    ("enc_lin_groups", "ENC_LINEAR_GROUPS", "deepfilternet", int, 16),
    ("mask_pf", "MASK_PF", "deepfilternet", _bool, False),
    ("pf_beta", "PF_BETA", "deepfilternet", float, 0.02),
    ("lsnr_dropout", "LSNR_DROPOUT", "deepfilternet", _bool, False),
    ("model", "MODEL", "train", str, "deepfilternet3"),
]


@dataclass
class ModelParams:
    """DfParams + deepfilternet3.ModelParams as one flat record (same attribute names as the reference)."""

    sr: int = 48_000
    fft_size: int = 960
    hop_size: int = 480
    nb_erb: int = 32
    nb_df: int = 96
    norm_tau: float = 1.0
    lsnr_max: int = 35
    lsnr_min: int = -15
    min_nb_freqs: int = 2
    df_order: int = 5
    df_lookahead: int = 0
    pad_mode: str = "input"
    conv_lookahead: int = 0
    conv_ch: int = 16
    conv_depthwise: bool = True
    convt_depthwise: bool = True
    conv_kernel: Tuple[int, ...] = (1, 3)
    convt_kernel: Tuple[int, ...] = (1, 3)
    conv_kernel_inp: Tuple[int, ...] = (3, 3)
    emb_hidden_dim: int = 256
    emb_num_layers: int = 2
    emb_gru_skip_enc: str = "none"
    emb_gru_skip: str = "none"
    df_hidden_dim: int = 256
    df_gru_skip: str = "none"
    df_pathway_kernel_size_t: int = 1
    enc_concat: bool = False
    df_num_layers: int = 3
    df_n_iter: int = 1
    lin_groups: int = 1
    enc_lin_groups: int = 16
    mask_pf: bool = False
    pf_beta: float = 0.02
    lsnr_dropout: bool = False
    model: str = "deepfilternet3"
    # set when the artefact states the smoothing factor itself (config.ini `norm_alpha` of an exported model, tract.rs:279-284)
    norm_alpha_value: Optional[float] = None

    # ---- constructors -------------------------------------------------------------------------------------------
    @classmethod
    def from_ini(cls, path: Optional[str], env: Optional[Dict[str, str]] = None, must_exist: bool = False):
        """config.load() + ModelParams(): INI file with the reference's env-var override (config.py:119-122)."""
        parser = ConfigParser()
        if path is not None and os.path.isfile(path):
            with open(path) as f:
                parser.read_file(f)
        elif must_exist:
            raise ValueError(f"No config file found at '{path}'.")
        _fix_legacy_sections(parser)
        env = os.environ if env is None else env
        kw = {}
        for attr, opt, section, cast, default in _OPTIONS:
            if opt in env:
                val = env[opt]
            elif parser.has_option(section, opt):
                val = parser.get(section, opt)
            elif parser.has_option("settings", opt):
                val = parser.get("settings", opt)
            else:
                val = default
            kw[attr] = cast(val)
        p = cls(**kw)
        p.df_gru_skip = p.df_gru_skip.lower()   # deepfilternet3.py:303 lower-cases this option only
        return p

    @classmethod
    def defaults(cls) -> "ModelParams":
        """The code defaults (config.use_defaults()); NOT the shipped DeepFilterNet3."""
        return cls()

    @classmethod
    def deepfilternet3(cls) -> "ModelParams":
        """Hyper-parameters of the published DeepFilterNet3 checkpoint ("pretrained-shape"): the
        shipped config.ini is a missing blob in this environment, so these values are recalled from upstream; they
        reproduce the published 2.13 M parameters when instantiated with the reference's modules."""
        return cls(conv_ch=64, emb_num_layers=3, df_num_layers=2, lin_groups=16, enc_lin_groups=32,
                   df_pathway_kernel_size_t=5, conv_lookahead=2, df_lookahead=2, df_gru_skip="groupedlinear")

    @classmethod
    def deepfilternet3_ll(cls) -> "ModelParams":
        """Low-latency variant used by BASELINE.json config 4: DeepFilterNet3 without lookahead (ladspa/README.md:3)."""
        p = cls.deepfilternet3()
        p.conv_lookahead = 0
        p.df_lookahead = 0
        return p

    # ---- derived ------------------------------------------------------------------------------------------------
    @property
    def freq_bins(self) -> int:
        return self.fft_size // 2 + 1

    @property
    def emb_dim(self) -> int:
        """conv_ch * nb_erb // 4 (deepfilternet3.py:125-127): width of the embedding fed to the GRUs."""
        return self.conv_ch * self.nb_erb // 4

    def norm_alpha(self) -> float:
        """DeepFilterNet/df/utils.py:111-127 get_norm_alpha(): exp(-hop/sr/tau) rounded to >=3 digits, < 1."""
        if self.norm_alpha_value is not None:
            return float(self.norm_alpha_value)
        a_ = math.exp(-(self.hop_size / self.sr) / self.norm_tau)
        precision, a = 3, 1.0
        while a >= 1.0:
            a = round(a_, precision)
            precision += 1
        return a

    def to_ini(self) -> str:
        lines: Dict[str, list] = {}
        for attr, opt, section, cast, default in _OPTIONS:
            v = getattr(self, attr)
            if isinstance(v, tuple):
                v = ",".join(str(x) for x in v)
            lines.setdefault(section, []).append(f"{opt.lower()} = {v}")
        return "\n".join(f"[{s}]\n" + "\n".join(ls) + "\n" for s, ls in lines.items())

    def to_cfg(self):
        """This configuration as the C library's ``dfx_model_cfg`` (include/dfx.h), unchecked."""
        from . import _lib

        skip = {"none": 0, "identity": 1, "groupedlinear": 2}
        c = _lib.ModelCfg()
        c.sr, c.fft_size, c.hop_size, c.nb_erb, c.nb_df = self.sr, self.fft_size, self.hop_size, self.nb_erb, self.nb_df
        c.min_nb_freqs, c.df_order, c.df_lookahead = self.min_nb_freqs, self.df_order, self.df_lookahead
        c.lsnr_min, c.lsnr_max = int(self.lsnr_min), int(self.lsnr_max)
        c.conv_lookahead, c.conv_ch = self.conv_lookahead, self.conv_ch
        c.emb_hidden_dim, c.emb_num_layers = self.emb_hidden_dim, self.emb_num_layers
        c.df_hidden_dim, c.df_num_layers = self.df_hidden_dim, self.df_num_layers
        c.df_gru_skip = skip[self.df_gru_skip]
        c.df_pathway_kernel_size_t = self.df_pathway_kernel_size_t
        c.lin_groups, c.enc_lin_groups = self.lin_groups, self.enc_lin_groups
        c.mask_pf, c.pf_beta, c.norm_alpha = int(self.mask_pf), float(self.pf_beta), float(self.norm_alpha())
        c.emb_gru_skip_enc, c.emb_gru_skip, c.enc_concat = skip[self.emb_gru_skip_enc], skip[self.emb_gru_skip], int(bool(self.enc_concat))
        return c

    def check_supported(self) -> None:
        """The HIP engine covers the DeepFilterNet3 family; anything else is refused loudly (never a silent fallback).

        Two halves, one source of truth each: what the C library cannot see (options that are not part of ``dfx_model_cfg``: the model
        type, kernel shapes, training-only switches) is checked here; every shape the kernels are or are not instantiated for (conv_ch,
        band / bin counts, group tilings, hidden sizes, filter order) is asked of the library itself (``dfx_model_blob_floats`` runs the
        same ``check_cfg`` as ``dfx_model_create``), so Python and C refuse exactly the same configurations with the same message."""
        def need(cond, msg):
            if not cond:
                raise NotImplementedError(f"deepfilternet_amd: unsupported configuration: {msg}")

        need(self.model == "deepfilternet3", f"model={self.model!r} (only deepfilternet3 is on the HIP path)")
        need(self.conv_depthwise and self.convt_depthwise, "non-depthwise convolutions")
        need(tuple(self.conv_kernel) == (1, 3) and tuple(self.convt_kernel) == (1, 3), "conv kernels other than (1,3)")
        need(tuple(self.conv_kernel_inp) == (3, 3), "conv_kernel_inp other than (3,3)")
        for opt in ("emb_gru_skip_enc", "emb_gru_skip", "df_gru_skip"):
            need(getattr(self, opt) in ("none", "identity", "groupedlinear"), f"{opt}={getattr(self, opt)!r}")
        need(self.df_n_iter == 1, "df_n_iter != 1")
        need(not self.lsnr_dropout, "lsnr_dropout")
        need(self.conv_lookahead >= self.df_lookahead or self.conv_lookahead == 0, "conv_lookahead < df_lookahead")
        # pad_mode is a DfParams option that deepfilternet3.py never reads (only deepfilternet.py / deepfilternet2.py do): any value is fine
        import ctypes

        from . import _lib

        try:
            L = _lib.lib()
        except Exception as e:  # noqa: BLE001  (no built / loadable libdfx.so on this host: say so instead of a bare loader error)
            raise RuntimeError("deepfilternet_amd: cannot validate the configuration's kernel shapes without the native library "
                               f"(csrc/libdfx.so; build it with `python -m deepfilternet_amd.build`): {e}") from e
        cfg, n = self.to_cfg(), ctypes.c_int64()
        rc = L.dfx_model_blob_floats(ctypes.byref(cfg), ctypes.byref(n))
        if rc == _lib.DFX_ERR_INVALID_ARG:   # contradictory options (e.g. enc_concat with emb_gru_skip_enc): an invalid value, not a missing kernel
            raise ValueError(f"deepfilternet_amd: invalid configuration: {L.dfx_last_error().decode()}")
        if rc != _lib.DFX_OK:
            raise NotImplementedError(f"deepfilternet_amd: unsupported configuration: {L.dfx_last_error().decode()}")


def _fix_legacy_sections(parser: ConfigParser) -> None:
    """DeepFilterNet/df/config.py:171-206 (_fix_df / _fix_clc), the parts that touch inference options."""
    if not parser.has_section("df") and parser.has_section("clc"):
        parser["df"] = parser["clc"]
        parser.remove_section("clc")
    if parser.has_section("deepfilternet") and parser.has_section("df"):
        for key in ("df_order", "df_lookahead"):
            if key in parser["deepfilternet"]:
                parser["df"][key] = parser["deepfilternet"][key]
                del parser["deepfilternet"][key]
