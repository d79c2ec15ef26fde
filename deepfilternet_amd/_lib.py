"""ctypes binding of libdfx.so (include/dfx.h).  PyTorch is used only as plumbing: device memory, streams.

There is NO CPU fallback: if the HIP library cannot be loaded, or no MI355X is visible, every compute entry point raises.
Unit tests may point the binding at the SIMT-interpreter build of the same sources with :func:`use_library` (explicit,
test-only); the package itself never does that.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFX_LIBRARY: another build of the same library (dev: kernel-parameter variants under tools/dev/_build/); never a different backend
DEFAULT_LIB = os.environ.get("DFX_LIBRARY") or os.path.join(_HERE, "csrc", "libdfx.so")

DFX_OK = 0
DFX_ERR_INVALID_ARG, DFX_ERR_UNSUPPORTED, DFX_ERR_HIP, DFX_ERR_NO_DEVICE, DFX_ERR_ALLOC = 1, 2, 3, 4, 5   # include/dfx.h:31-35
_ERRNAMES = {1: "invalid argument", 2: "unsupported configuration", 3: "HIP runtime error", 4: "no HIP device",
             5: "allocation failed"}


class DfxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"DF transform error: {msg} [{_ERRNAMES.get(code, code)}]")
        self.code = code


class ModelCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "sr", "fft_size", "hop_size", "nb_erb", "nb_df", "min_nb_freqs", "df_order", "df_lookahead", "lsnr_min",
        "lsnr_max", "conv_lookahead", "conv_ch", "emb_hidden_dim", "emb_num_layers", "df_hidden_dim", "df_num_layers",
        "df_gru_skip", "df_pathway_kernel_size_t", "lin_groups", "enc_lin_groups", "mask_pf")] + [
        ("pf_beta", C.c_float), ("norm_alpha", C.c_float)] + [(n, C.c_int32) for n in ("emb_gru_skip_enc", "emb_gru_skip", "enc_concat")]


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_fp = C.c_void_p  # data pointers are passed as raw addresses (device or, in the interpreter build, host)
_u64p = C.POINTER(C.c_uint64)
_f32p = C.POINTER(C.c_float)

# name -> (restype, argtypes)
SIGNATURES = {
    "dfx_version": (_i, []),
    "dfx_status_string": (C.c_char_p, [_i]),
    "dfx_last_error": (C.c_char_p, []),
    "dfx_device_count": (_i, []),
    "dfx_is_emulator": (_i, []),
    "dfx_bands_create": (_i, [_u64p, _i, C.POINTER(_vp)]),
    "dfx_bands_free": (None, [_vp]),
    "dfx_bands_nb": (_i, [_vp]),
    "dfx_bands_nfreqs": (_i, [_vp]),
    "dfx_state_create": (_i, [_i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "dfx_state_free": (None, [_vp]),
    "dfx_state_sr": (_i, [_vp]),
    "dfx_state_fft_size": (_i, [_vp]),
    "dfx_state_hop_size": (_i, [_vp]),
    "dfx_state_nb_erb": (_i, [_vp]),
    "dfx_state_wnorm": (_f, [_vp]),
    "dfx_state_bands": (_vp, [_vp]),
    "dfx_state_erb_widths": (_i, [_vp, _u64p]),
    "dfx_state_fft_window": (_i, [_vp, _f32p]),
    "dfx_erb_fb": (_i, [_i, _i, _i, _i, _u64p]),
    "dfx_analysis": (_i, [_vp, _fp, _i64, _i64, _i64, _fp, _fp, _fp, _vp]),
    "dfx_synthesis": (_i, [_vp, _fp, _i64, _i64, _fp, _fp, _fp, _i64, _vp]),
    "dfx_erb": (_i, [_vp, _fp, _i64, _i, _fp, _vp]),
    "dfx_erb_inv": (_i, [_vp, _fp, _i64, _fp, _vp]),
    "dfx_erb_norm": (_i, [_fp, _i64, _i64, _i, _f, _fp, _vp]),
    "dfx_unit_norm": (_i, [_fp, _i64, _fp, _i64, _i64, _i, _f, _fp, _vp]),
    "dfx_unit_norm_init": (_i, [_i, _f32p]),
    "dfx_features": (_i, [_vp, _fp, _i64, _i64, _i64, _i, _f, _fp, _fp, _fp, _vp]),
    "dfx_df_apply": (_i, [_fp, _fp, _i, _fp, _vp, _i64, _i64, _i, _i, _i, _i, _f, _f, _fp, _vp]),
    "dfx_df_apply_strided": (_i, [_fp, _i64, _fp, _i, _fp, _vp, _i64, _i64, _i, _i, _i, _i, _f, _f, _fp, _i64, _vp]),
    "dfx_model_tensor_count": (_i, [C.POINTER(ModelCfg), C.POINTER(_i)]),
    "dfx_model_tensor_info": (_i, [C.POINTER(ModelCfg), _i, C.c_char_p, _i, C.POINTER(_i64), C.POINTER(_i),
                                   C.POINTER(_i64)]),
    "dfx_model_blob_floats": (_i, [C.POINTER(ModelCfg), C.POINTER(_i64)]),
    "dfx_model_create": (_i, [C.POINTER(ModelCfg), _f32p, C.POINTER(_vp)]),
    "dfx_model_free": (None, [_vp]),
    "dfx_model_save_file": (_i, [C.POINTER(ModelCfg), _f32p, C.c_char_p]),
    "dfx_model_load_file": (_i, [C.c_char_p, C.POINTER(_vp)]),
    "dfx_onnx_targz_read": (_i, [C.c_char_p, C.POINTER(ModelCfg), _f32p, _i64, C.POINTER(_i64)]),
    "dfx_model_cfg_get": (_i, [_vp, C.POINTER(ModelCfg)]),
    "dfx_model_set_streams": (_i, [_vp, _i]),
    "dfx_model_set_run_df": (_i, [_vp, _i]),
    "dfx_model_set_pipeline": (_i, [_vp, _i, _i, _i]),
    "dfx_model_check": (_i, [_vp]),
    "dfx_model_poll": (_i, [_vp]),
    "dfx_model_query": (_i, [_vp, _i, C.POINTER(_i64)]),
    "dfx_model_workspace_bytes": (_i, [_vp, _i64, _i64, C.POINTER(_i64)]),
    "dfx_model_forward": (_i, [_vp, _vp, _fp, _fp, _fp, _i64, _i64, _f, _fp, _fp, _fp, _fp, _fp, _i64, _vp]),
    "dfx_enhance_workspace_bytes": (_i, [_vp, _vp, _i64, _i64, _i, C.POINTER(_i64)]),
    "dfx_enhance": (_i, [_vp, _vp, _fp, _i64, _i64, _i, _f, _fp, _fp, _i64, _vp]),
    "dfx_enhance_pcm16": (_i, [_vp, _vp, _vp, _i64, _i64, _i, _f, _vp, _fp, _i64, _vp]),
    "dfx_mf_filter": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _f, _f, _i64, _i64, _i, _i, _fp, _vp]),
    "dfx_pcm16_to_f32": (_i, [_vp, _i64, _fp, _vp]),
    "dfx_f32_to_pcm16": (_i, [_fp, _i64, _vp, _vp]),
    "dfx_resampler_create": (_i, [_i, _i, _i, C.c_double, _i, C.c_double, C.POINTER(_vp)]),
    "dfx_resampler_free": (None, [_vp]),
    "dfx_resampler_out_len": (_i64, [_vp, _i64]),
    "dfx_resampler_kernel": (_i, [_i, _i, _i, C.c_double, _i, C.c_double, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _f32p, _i64]),
    "dfx_resample": (_i, [_vp, _fp, _i64, _i64, _i64, _fp, _i64, _vp]),
    "dfx_stream_create": (_i, [_vp, _vp, _i64, _i, C.POINTER(_vp)]),
    "dfx_stream_free": (None, [_vp]),
    "dfx_stream_reset": (_i, [_vp, _vp]),
    "dfx_stream_frame_length": (_i, [_vp]),
    "dfx_stream_delay_frames": (_i, [_vp]),
    "dfx_stream_set_atten_lim": (_i, [_vp, _f]),
    "dfx_stream_set_post_filter_beta": (_i, [_vp, _f]),
    "dfx_stream_set_channels": (_i, [_vp, _i, _i]),
    "dfx_stream_set_gating": (_i, [_vp, _i]),
    "dfx_stream_set_thresholds": (_i, [_vp, _f, _f, _f]),
    "dfx_stream_process": (_i, [_vp, _fp, _i64, _fp, _fp, _vp]),
    "dfx_stream_process_raw": (_i, [_vp, _fp, _fp, _fp, _vp, _fp, _vp]),
    "dfx_prof_kernel_count": (_i, []),
    "dfx_prof_kernel_name": (C.c_char_p, [_i]),
    "dfx_prof_enable": (_i, [C.c_uint32]),
    "dfx_prof_reset": (_i, []),
    "dfx_prof_read": (_i, [_i, C.POINTER(C.c_double), C.POINTER(_i64)]),
}

_LIB: Optional[C.CDLL] = None
_LIB_PATH: Optional[str] = None
_EMU = False


def _bind(lib: C.CDLL) -> None:
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype, fn.argtypes = res, args
    if missing:
        raise RuntimeError(f"libdfx is missing symbols declared in include/dfx.h: {missing}")


def use_library(path: str) -> None:
    """Load a specific build of the library.  The package default is csrc/libdfx.so (HIP, gfx950)."""
    global _LIB, _LIB_PATH, _EMU
    lib = C.CDLL(path)
    _bind(lib)
    _LIB, _LIB_PATH = lib, path
    _EMU = bool(lib.dfx_is_emulator())


def lib() -> C.CDLL:
    if _LIB is None:
        if not os.path.exists(DEFAULT_LIB):
            # build in-tree on first use (needs hipcc); never silently substitute anything else
            from . import build as _build

            try:
                _build.build()
            except Exception as e:  # noqa: BLE001
                raise RuntimeError(
                    "deepfilternet_amd: the HIP extension csrc/libdfx.so is missing and could not be built "
                    f"({e}).  There is no CPU fallback.") from e
        use_library(DEFAULT_LIB)
    return _LIB  # type: ignore[return-value]


def is_emulator() -> bool:
    lib()
    return _EMU


def library_path() -> Optional[str]:
    return _LIB_PATH


def device() -> torch.device:
    """Device on which buffers handed to libdfx must live."""
    if is_emulator():
        return torch.device("cpu")
    if not torch.cuda.is_available() or lib().dfx_device_count() <= 0:
        raise RuntimeError("deepfilternet_amd: no MI355X / HIP device visible; libdfx has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream() -> C.c_void_p:
    if is_emulator():
        return C.c_void_p(0)
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def kernel_ids() -> dict:
    """{kernel name: id} of the kernels dfx_prof_* can time."""
    L = lib()
    return {L.dfx_prof_kernel_name(i).decode(): i for i in range(L.dfx_prof_kernel_count())}


def prof_enable(names=None) -> None:
    """Time the named kernels (None/empty = off, "all" = every kernel) with hipEvents on their launch stream."""
    ids = kernel_ids()
    if not names:
        mask = 0
    elif names == "all":
        mask = (1 << len(ids)) - 1
    else:
        mask = 0
        for n in names:
            mask |= 1 << ids[n]
    check(lib().dfx_prof_enable(mask))


def prof_reset() -> None:
    check(lib().dfx_prof_reset())


def prof_read() -> dict:
    """{kernel name: (total_ms, launches)} since the last reset (synchronises the recorded events)."""
    L = lib()
    out = {}
    for n, i in kernel_ids().items():
        ms, cnt = C.c_double(), C.c_int64()
        check(L.dfx_prof_read(i, C.byref(ms), C.byref(cnt)))
        if cnt.value:
            out[n] = (ms.value, cnt.value)
    return out


def check(rc: int) -> None:
    if rc != DFX_OK:
        raise DfxError(rc, lib().dfx_last_error().decode())


def ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def on_device(t: torch.Tensor) -> bool:
    d = device()
    return t.device.type == d.type and (d.type == "cpu" or t.device.index == d.index)
