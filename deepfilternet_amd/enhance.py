"""``init_df`` / ``df_features`` / ``enhance`` with the reference's signatures (DeepFilterNet/df/enhance.py:101-250),
running entirely on the MI355X through libdfx.so.

* ``df_features`` and ``enhance`` accept the same arguments and return tensors of the same shape/dtype as the reference.
* ``enhance`` keeps everything on the device: audio -> (pad) -> STFT + ERB/complex features -> DeepFilterNet3 ->
  deep filter + ERB gains (+ post filter, + attenuation limit) -> ISTFT -> slice, one C call (``dfx_enhance``).
* The batch dimension is the channel dimension C of the ``[C, T]`` input, exactly as in the reference.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch

from . import _lib
from .config import ModelParams
from .libdf import DF, erb, erb_norm, unit_norm
from .model import DfNet, read_cp
from .state_dict import random_state_dict

PRETRAINED_MODELS = ("DeepFilterNet", "DeepFilterNet2", "DeepFilterNet3")
DEFAULT_MODEL = "DeepFilterNet3"


def get_device() -> torch.device:
    """utils.py:20-29 get_device(); here always the HIP device libdfx runs on."""
    return _lib.device()


def init_df(
    model_base_dir: Optional[str] = None,
    post_filter: bool = False,
    log_level: str = "INFO",
    log_file: Optional[str] = "enhance.log",
    config_allow_defaults: bool = True,
    epoch: Union[str, int, None] = "best",
    default_model: str = DEFAULT_MODEL,
    mask_only: bool = False,
    *,
    params: Optional[ModelParams] = None,
    state_dict: Optional[Dict[str, "np.ndarray | torch.Tensor"]] = None,
    seed: int = 0,
) -> Tuple[DfNet, DF, str, int]:
    """enhance.py:101-187.  ``model_base_dir`` must contain ``config.ini`` and ``checkpoints/model_*.ckpt[.best]``.

    The pretrained models cannot be downloaded here (no network, enhance.py:253-273); passing one of their names raises.
    Extensions (keyword-only): ``params`` + ``state_dict`` build the model from memory; ``epoch="none"`` (or None) with no
    checkpoint initialises seeded synthetic weights (``seed``), which is what the benchmarks use.
    """
    load_cp = epoch is not None and not (isinstance(epoch, str) and epoch.lower() == "none")
    if params is None:
        if model_base_dir is None or model_base_dir in PRETRAINED_MODELS:
            raise FileNotFoundError(
                f"pretrained model '{model_base_dir or default_model}' is not available offline; pass a model directory "
                "with config.ini + checkpoints/, or params=/state_dict=")
        if os.path.isfile(model_base_dir) and model_base_dir.endswith(".tar.gz"):
            # extension: the reference's exported artefact (<model>_onnx.tar.gz, tract.rs:29-70) in place of a model directory
            from .model import read_onnx_targz

            p, state_dict = read_onnx_targz(model_base_dir)
        elif not os.path.isdir(model_base_dir):
            raise NotADirectoryError("Base directory not found at {}".format(model_base_dir))
        else:
            p = ModelParams.from_ini(os.path.join(model_base_dir, "config.ini"), must_exist=True)
    else:
        p = params
    if post_filter:
        p.mask_pf = True  # enhance.py:152-159
    df_state = DF(sr=p.sr, fft_size=p.fft_size, hop_size=p.hop_size, nb_bands=p.nb_erb, min_nb_erb_freqs=p.min_nb_freqs)
    ep = 0
    from_checkpoint = False
    if state_dict is None and load_cp and model_base_dir is not None and os.path.isdir(model_base_dir):
        state_dict, ep = read_cp(os.path.join(model_base_dir, "checkpoints"), epoch)
        if state_dict is None:
            raise FileNotFoundError("Could not find a checkpoint")  # reference: logger.error + exit(1)
        from_checkpoint = True
    if state_dict is None:
        state_dict = random_state_dict(p, seed)
    # enhance.py:172-175: init_model(df_state, run_df=not mask_only); checkpoints load like read_cp (non-strict, size mismatches dropped)
    model = DfNet(p, state_dict, df_state, run_df=not mask_only, strict=not from_checkpoint)
    suffix = os.path.basename(os.path.abspath(model_base_dir)) if model_base_dir else p.model
    if post_filter:
        suffix += "_pf"
    return model, df_state, suffix, ep


def df_features(audio: torch.Tensor, df: DF, nb_df: int, device=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """enhance.py:190-203: -> spec [C,1,T',F,2], erb_feat [C,1,T',E], spec_feat [C,1,T',nb_df,2] (on the HIP device)."""
    p_alpha = _norm_alpha(df)
    x = audio.to(_lib.device(), torch.float32).contiguous()
    Cn, T = x.shape
    Tf, F, E = T // df.hop_size(), df.fft_size() // 2 + 1, df.nb_erb()
    spec = torch.empty((Cn, Tf, F, 2), dtype=torch.float32, device=x.device)
    fe = torch.empty((Cn, Tf, E), dtype=torch.float32, device=x.device)
    fs = torch.empty((Cn, Tf, nb_df, 2), dtype=torch.float32, device=x.device)
    if Cn and Tf:
        _lib.check(_lib.lib().dfx_features(df.handle, _lib.ptr(x), Cn, T, x.stride(0), int(nb_df), float(p_alpha),
                                           _lib.ptr(spec), _lib.ptr(fe), _lib.ptr(fs), _lib.stream()))
    return spec.unsqueeze(1), fe.unsqueeze(1), fs.unsqueeze(1)


def _norm_alpha(df: DF, tau: float = 1.0) -> float:
    p = ModelParams(sr=df.sr(), hop_size=df.hop_size(), norm_tau=tau)
    return p.norm_alpha()


def _workspace_cap(model: DfNet, need: int) -> Optional[int]:
    """Bytes the enhance() workspace may take: DFX_WORKSPACE_CAP_GB if set, else 90 % of what the device has free plus the workspace this
    model already holds (None when that cannot be asked, e.g. on the CPU interpreter build).  The device is only asked when the answer can
    matter: a request that fits into the workspace the model already holds needs no allocation at all."""
    env = os.environ.get("DFX_WORKSPACE_CAP_GB")
    if env:
        return int(float(env) * (1 << 30))
    if _lib.device().type != "cuda":
        return None
    # the workspace this model already holds counts as available: DfNet.workspace() releases it before it allocates a larger one
    held = model._ws.numel() if getattr(model, "_ws", None) is not None else 0
    if need <= held:
        return held
    free, _ = torch.cuda.mem_get_info(_lib.device())
    return int(0.9 * (free + held))


@torch.no_grad()
def enhance(model: DfNet, df_state: DF, audio: torch.Tensor, pad: bool = True, atten_lim_db: Optional[float] = None
            ) -> torch.Tensor:
    """enhance.py:206-250.  audio [C, T] float32 (host or device) -> enhanced audio, same shape when ``pad``.

    The result lives on the device the input came from (a CPU tensor in -> CPU tensor out, like the reference).

    Extension: an ``int16`` tensor is taken as 16-bit PCM — the sample format the reference's file loop decodes and encodes around this
    call (enhance.py:73-89 -> io.py:25-84) — and comes back as ``int16``: torchaudio's ``x / 32768`` and save_audio's
    ``(audio * (1 << 15)).to(torch.int16)`` run inside the STFT kernel's loads and the ISTFT kernel's stores (``dfx_enhance_pcm16``): the same
    samples as ``float_to_pcm16(enhance(pcm16_to_float(audio)))``, half the bytes over PCIe and no conversion passes."""
    if not isinstance(model, DfNet):
        raise TypeError("enhance() of deepfilternet_amd needs a deepfilternet_amd.DfNet (see init_df)")
    src_dev = audio.device
    pcm16 = audio.dtype == torch.int16
    # a page-locked host batch moves by DMA without the driver's staging copies (and comes back into page-locked memory): the
    # transfer is then asynchronous on the launch stream, ~55 GB/s over PCIe Gen5 instead of ~10 for pageable memory
    pinned = src_dev.type == "cpu" and audio.is_pinned() and _lib.device().type == "cuda"
    x = audio.to(_lib.device(), torch.int16 if pcm16 else torch.float32, non_blocking=pinned).contiguous()
    if x.dim() != 2:
        raise ValueError("audio must have shape [C, T]")
    B, T = x.shape
    hop = df_state.hop_size()
    n_fft = df_state.fft_size()
    out_len = T if pad else ((T // hop) * hop)
    y = torch.empty((B, out_len), dtype=x.dtype, device=x.device)
    if B == 0 or out_len == 0:
        return y.to(src_dev)
    nbytes = C.c_int64()
    L = _lib.lib()
    lim_db = float(atten_lim_db) if atten_lim_db is not None else 0.0

    def ws_bytes(b: int) -> int:
        _lib.check(L.dfx_enhance_workspace_bytes(model.handle, df_state.handle, b, T, int(bool(pad)), C.byref(nbytes)))
        return int(nbytes.value)

    # The engine's scratch is ~82 MB per 10 s clip: a batch whose workspace does not fit what the device has free is
    # enhanced in sub-batches of whole clips, one after the other in the same workspace (clips are independent: same samples).
    # DFX_WORKSPACE_CAP_GB bounds the workspace explicitly (tests; shared devices).
    need = ws_bytes(B)
    cap = _workspace_cap(model, need)
    sub = B
    if cap is not None and need > cap:
        lo, hi = 1, B                      # largest sub-batch whose workspace fits (the requirement grows with the batch)
        if ws_bytes(1) > cap:
            raise MemoryError(f"enhance(): a single clip of {T} samples needs {ws_bytes(1)} bytes of workspace, {cap} available")
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if ws_bytes(mid) <= cap:
                lo = mid
            else:
                hi = mid - 1
        sub = lo if lo < 16 else lo - lo % 16   # whole groups of 16 clips (one GRU workgroup each) when there are that many
    ws = model.workspace(need if sub == B else ws_bytes(sub))
    for b0 in range(0, B, sub):
        n = min(sub, B - b0)
        _lib.check((L.dfx_enhance_pcm16 if pcm16 else L.dfx_enhance)(model.handle, df_state.handle, _lib.ptr(x[b0:b0 + n]), n, T, int(bool(pad)), lim_db,
                                                                     _lib.ptr(y[b0:b0 + n]), _lib.ptr(ws), ws.numel(), _lib.stream()))
    # No silent garbage: a kernel that found a fault (fp16-split range, flag-wait timeout) raised an error word of the model.  A host
    # result is only handed back after the device has finished, so its own pass is checked; a device result is asynchronous, and the
    # words are looked at without waiting — the C entry points do the same before they start the next pass, so a fault surfaces in the
    # next call at the latest (DFX_CHECK_EVERY_PASS=1: in dfx_enhance itself).
    if pinned:
        out = torch.empty(y.shape, dtype=y.dtype, pin_memory=True)
        out.copy_(y, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        model.poll()
        return out
    out = y.to(src_dev)
    model.poll()
    return out


def enhance_files(model: DfNet, df_state: DF, input_files, output_dir: Optional[str] = None, suffix: Optional[str] = None,
                  compensate_delay: bool = True, atten_lim_db: Optional[float] = None, method: str = "sinc_fast"):
    """The body of the reference's file loop, ``df.enhance.main`` (enhance.py:73-89), without its argument parser: every file is
    decoded, brought to the model's sampling rate, enhanced, brought back to its own rate and written next to the input (or into
    ``output_dir``) as ``<name>_<suffix>.wav``.  A file that already has the model's rate moves as 16-bit PCM all the way (the two
    conversions of df/io.py run inside the STFT / ISTFT kernels: ``enhance()`` on an int16 tensor); one that has to be resampled takes the
    float path (int16 -> float, resample, enhance, resample, float -> int16, all on the device).  Channels of a file are the batch, as in
    the reference.  Returns the written paths."""
    from .io import load_audio, resample, save_audio

    sr = df_state.sr()
    out = []
    for file in input_files:
        audio, meta = load_audio(file, sr=sr, verbose=False, method=method, pcm16=True)   # int16 when no resampling is needed
        enhanced = enhance(model, df_state, audio, pad=compensate_delay, atten_lim_db=atten_lim_db)
        if enhanced.dtype != torch.int16:
            enhanced = resample(enhanced, sr, meta.sample_rate, method=method)
        out.append(save_audio(file, enhanced, sr=meta.sample_rate, output_dir=output_dir, suffix=suffix))
    return out
