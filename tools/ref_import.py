"""Import the *reference's own* Python package (``/root/reference/DeepFilterNet/df``) in this container.

Only used by tools/gen_golden.py (fixture generation) and by tests marked ``needs_reference``; nothing that runs
on the GPU box may depend on it (``/root/reference`` does not exist there).

The reference needs three modules that are not installed here:
  * ``loguru``     -> no-op logger (``logger.level(name).no`` is read at import time, logger.py:17-18)
  * ``torchaudio`` -> only the names ``AudioMetaData`` / ``functional.resample`` are touched at import (io.py:5,11,88)
  * ``libdf``      -> the C oracle's pyDF-shaped front end (oracle/libdf_oracle.py); the Rust original cannot be built.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "DeepFilterNet", "df"))


class _Lvl:
    no = 30
    name = "INFO"


class _Logger:
    def level(self, *a, **k):
        return _Lvl()

    def __getattr__(self, name):
        def _noop(*a, **k):
            return None

        return _noop


def install_shims() -> None:
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference tree
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    if _REPO not in sys.path:
        sys.path.insert(0, _REPO)
    if "loguru" not in sys.modules:
        m = types.ModuleType("loguru")
        m.logger = _Logger()
        sys.modules["loguru"] = m
    if "torchaudio" not in sys.modules:
        ta = types.ModuleType("torchaudio")
        ta.AudioMetaData = type("AudioMetaData", (), {})
        fn = types.ModuleType("torchaudio.functional")
        fn.resample = lambda x, *a, **k: x
        ta.functional = fn
        be = types.ModuleType("torchaudio.backend")
        bc = types.ModuleType("torchaudio.backend.common")
        bc.AudioMetaData = ta.AudioMetaData
        be.common = bc
        ta.backend = be
        sys.modules.update({"torchaudio": ta, "torchaudio.functional": fn, "torchaudio.backend": be,
                            "torchaudio.backend.common": bc})
    if "libdf" not in sys.modules:
        from oracle import libdf_oracle

        sys.modules["libdf"] = libdf_oracle
    p = os.path.join(REFERENCE_ROOT, "DeepFilterNet")
    if p not in sys.path:
        sys.path.insert(0, p)


def load_reference_config(overrides: dict | None = None):
    """config.use_defaults() + explicit overrides in the reference's own config singleton.

    overrides: {(section, option): value}.  Environment variables named like an option would win over the INI values
    (config.py:119-122), so they are removed first.
    """
    install_shims()
    from df.config import config  # type: ignore

    for k in list(os.environ):
        if k in _REF_OPTION_NAMES:
            del os.environ[k]
    config.use_defaults(allow_reload=True)  # fresh ConfigParser
    for (section, option), value in (overrides or {}).items():
        section = section.lower()
        if not config.parser.has_section(section):
            config.parser.add_section(section)
        config.parser.set(section, option.lower(), str(value))
    return config


_REF_OPTION_NAMES = {
    "SR", "FFT_SIZE", "HOP_SIZE", "NB_ERB", "NB_DF", "NORM_TAU", "LSNR_MAX", "LSNR_MIN", "MIN_NB_ERB_FREQS",
    "DF_ORDER", "DF_LOOKAHEAD", "PAD_MODE", "CONV_LOOKAHEAD", "CONV_CH", "CONV_DEPTHWISE", "CONVT_DEPTHWISE",
    "CONV_KERNEL", "CONVT_KERNEL", "CONV_KERNEL_INP", "EMB_HIDDEN_DIM", "EMB_NUM_LAYERS", "EMB_GRU_SKIP_ENC",
    "EMB_GRU_SKIP", "DF_HIDDEN_DIM", "DF_GRU_SKIP", "DF_PATHWAY_KERNEL_SIZE_T", "ENC_CONCAT", "DF_NUM_LAYERS",
    "DF_N_ITER", "LINEAR_GROUPS", "ENC_LINEAR_GROUPS", "MASK_PF", "PF_BETA", "LSNR_DROPOUT", "MODEL", "DEVICE",
}
