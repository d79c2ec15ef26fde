"""deepfilternet_amd — MI355X-native engine for DeepFilterNet's enhance() hot path (HIP kernels behind a C ABI).

Public surface mirrors the reference (DeepFilterNet/df/__init__.py:1-6 + the pyDF ``libdf`` module):
    from deepfilternet_amd import init_df, enhance, df_features, ModelParams
    from deepfilternet_amd import libdf          # DF, erb, erb_inv, erb_norm, unit_norm, unit_norm_init
"""
import os as _os

# every GRU layer / branch of the forward pass runs on its own HIP stream; ROCm multiplexes streams onto 4 hardware queues
# unless told otherwise, which serialises them again (must be set before the HIP runtime initialises)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

from .config import ModelParams  # noqa: F401,E402

__version__ = "0.1.0"
__all__ = ["ModelParams", "init_df", "enhance", "enhance_files", "df_features", "DfNet", "libdf", "export_dfx"]


def __getattr__(name):
    # lazy: importing the package must not require torch/HIP until something is used
    if name in ("init_df", "enhance", "enhance_files", "df_features"):
        import importlib

        # (not `from . import enhance`: that form asks the package for the attribute first, i.e. re-enters this function)
        return getattr(importlib.import_module(".enhance", __name__), name)
    if name == "DfNet":
        from .model import DfNet

        return DfNet
    if name == "export_dfx":
        from .model import export_dfx

        return export_dfx
    if name == "libdf":
        import importlib

        return importlib.import_module(".libdf", __name__)
    raise AttributeError(name)
