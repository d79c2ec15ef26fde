"""Frame-by-frame (streaming) enhancement: the host-side mirror of the reference's real-time runtime ``DfTract``
(libDF/src/tract.rs:509-642) and its C API ``df_create / df_process_frame / df_set_atten_lim / df_set_post_filter_beta / df_free``
(libDF/src/capi.rs:83-253), for many independent mono streams advanced in lockstep on one MI355X.

    rt = DfStream(model, df_state, streams=4096)           # df_create, once
    for hop in audio.split(rt.frame_length, dim=1):        # [streams, 480] per call
        out = rt.process(hop)                              # df_process_frame: the enhanced hop of `lookahead` calls ago

``process`` also accepts several hops per call (``[streams, n * hop]``, ``n <= max_frames``); the concatenated output does not depend
on how the signal is cut.  It equals ``enhance(model, df_state, audio, pad=False)`` delayed by ``delay_frames`` hops (the first
``delay_frames`` output hops are silence, like the reference's rolling buffers).

``gating=True`` switches on the reference runtime's per-frame decisions (tract.rs:513-525,658-672), taken independently by every
stream: stages are skipped according to the local SNR (``thresholds`` = min_db, max_db_erb, max_db_df; reference defaults
-10 / 30 / 20 dB), skipped decoders keep their state, and a stream that has been silent for more than five hops is answered with
zeros without being processed.

``channels=k`` makes every k consecutive rows the channels of one stream (``RuntimeParams::n_ch``): per-channel STFT / network state,
one ERB mask per stream (``reduce_mask`` = "mean" (reference default) | "max" | "none", tract.rs:96-118,868-902), one stage decision
per stream (taken from its first channel's local SNR).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from .libdf import DF
from .model import DfNet


class DfStream:
    def __init__(self, model: DfNet, df_state: DF, streams: int = 1, max_frames: int = 1, atten_lim_db: Optional[float] = None,
                 gating: bool = False, thresholds: Optional[Tuple[float, float, float]] = None, channels: int = 1,
                 reduce_mask: str = "mean"):
        if not isinstance(model, DfNet):
            raise TypeError("DfStream needs a deepfilternet_amd.DfNet (see init_df)")
        h = C.c_void_p()
        _lib.check(_lib.lib().dfx_stream_create(model.handle, df_state.handle, int(streams), int(max_frames), C.byref(h)))
        self._h = h
        self._model, self._df = model, df_state  # keep the handles the runtime points into alive
        self.streams, self.max_frames = int(streams), int(max_frames)
        if atten_lim_db is not None:
            self.set_atten_lim(atten_lim_db)
        if channels != 1:
            _lib.check(_lib.lib().dfx_stream_set_channels(self._h, int(channels), {"none": 0, "max": 1, "mean": 2}[reduce_mask]))
        self.channels = int(channels)
        if thresholds is not None:
            self.set_thresholds(*thresholds)
        if gating:
            self.set_gating(True)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().dfx_stream_free(h)
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass

    @property
    def frame_length(self) -> int:
        """df_get_frame_length (capi.rs:108): samples per hop."""
        return int(_lib.lib().dfx_stream_frame_length(self._h))

    @property
    def delay_frames(self) -> int:
        """Hops by which the output lags the input (the model's lookahead), on top of the STFT's fft-hop samples."""
        return int(_lib.lib().dfx_stream_delay_frames(self._h))

    def set_atten_lim(self, lim_db: float) -> None:
        """df_set_atten_lim (capi.rs:136, tract.rs:387-398): |dB| >= 100 = no limit, < 0.01 = pass the input through."""
        _lib.check(_lib.lib().dfx_stream_set_atten_lim(self._h, float(lim_db)))

    def set_post_filter_beta(self, beta: float) -> None:
        """df_set_post_filter_beta (capi.rs:146): 0 disables the post filter."""
        _lib.check(_lib.lib().dfx_stream_set_post_filter_beta(self._h, float(beta)))

    def set_gating(self, enable: bool) -> None:
        """DfTract::process's stage skipping and silent-input shortcut (tract.rs:513-525,658-672), per stream."""
        _lib.check(_lib.lib().dfx_stream_set_gating(self._h, int(bool(enable))))

    def set_thresholds(self, min_db_thresh: float, max_db_erb_thresh: float, max_db_df_thresh: float) -> None:
        """RuntimeParams::with_thresholds (tract.rs:160-170)."""
        _lib.check(_lib.lib().dfx_stream_set_thresholds(self._h, float(min_db_thresh), float(max_db_erb_thresh),
                                                        float(max_db_df_thresh)))

    def reset(self) -> None:
        _lib.check(_lib.lib().dfx_stream_reset(self._h, _lib.stream()))

    def process(self, frames: torch.Tensor, return_lsnr: bool = False):
        """df_process_frame (capi.rs:161) for every stream: ``frames`` [streams, n*hop] float32 -> enhanced [streams, n*hop]
        (on the device the input came from); with ``return_lsnr`` also the local SNR estimates [streams, n] in dB."""
        src_dev = frames.device
        x = frames.to(_lib.device(), torch.float32).contiguous()
        hop = self.frame_length
        if x.dim() != 2 or x.shape[0] != self.streams or x.shape[1] % hop or x.shape[1] == 0:
            raise ValueError(f"frames must have shape [{self.streams}, n*{hop}]")
        n = x.shape[1] // hop
        if n > self.max_frames:
            raise ValueError(f"at most max_frames={self.max_frames} hops per call")
        y = torch.empty_like(x)
        lsnr = torch.empty((self.streams, n), dtype=torch.float32, device=x.device) if return_lsnr else None
        _lib.check(_lib.lib().dfx_stream_process(self._h, _lib.ptr(x), n, _lib.ptr(y), _lib.ptr(lsnr), _lib.stream()))
        y = y.to(src_dev)
        self._model.poll()   # faults raised by kernels (invalid results) are never silent: see DfNet.poll
        return (y, lsnr.to(src_dev)) if return_lsnr else y

    def process_raw(self, spec: torch.Tensor):
        """df_process_frame_raw (capi.rs:172-210 -> DfTract::process_raw, tract.rs:441-507) for every stream: one spectral frame
        ``spec`` [streams, F] complex64 (or [streams, F, 2] float32) -> (lsnr [streams], gains [streams, nb_erb], coefs
        [streams, df_order, nb_df] complex64, stages [streams] uint8: bit value 2 = gains present, 8 = coefficients present — where the
        reference hands back NULL pointers the arrays here hold placeholders).  Needs ``gating=True``."""
        src_dev = spec.device
        x = torch.view_as_real(spec) if spec.is_complex() else spec
        x = x.to(_lib.device(), torch.float32).contiguous()
        p = self._model.p
        if x.dim() != 3 or x.shape[0] != self.streams or x.shape[1] != p.freq_bins or x.shape[2] != 2:
            raise ValueError(f"spec must have shape [{self.streams}, {p.freq_bins}] complex")
        gains = torch.empty((self.streams, p.nb_erb), dtype=torch.float32, device=x.device)
        coefs = torch.empty((self.streams, p.df_order, p.nb_df, 2), dtype=torch.float32, device=x.device)
        stages = torch.empty((self.streams,), dtype=torch.uint8, device=x.device)
        lsnr = torch.empty((self.streams,), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().dfx_stream_process_raw(self._h, _lib.ptr(x), _lib.ptr(gains), _lib.ptr(coefs), _lib.ptr(stages), _lib.ptr(lsnr),
                                                     _lib.stream()))
        out = lsnr.to(src_dev), gains.to(src_dev), torch.view_as_complex(coefs).to(src_dev), stages.to(src_dev)
        self._model.poll()
        return out
