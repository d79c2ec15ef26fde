"""`libdf`-compatible Python module backed by the HIP library (drop-in for the pyo3 module of the reference).

Same surface as pyDF (pyDF/src/lib.rs:14-310, stub pyDF/libdf.pyi:5-70): class ``DF`` and the free functions ``erb``,
``erb_inv``, ``erb_norm``, ``unit_norm``, ``unit_norm_init`` — same shapes, dtypes, exception types and the same
in-place side effect of ``erb_norm``.  numpy in -> numpy out exactly like pyDF (host<->device copies around one batched
kernel launch); additionally every function accepts a torch tensor that already lives on the GPU and then returns a
device tensor without any host round trip (that is what the fused ``enhance()`` builds on).

Differences, all deliberate and documented in INTEGRATION.md:
  * channels are processed as one batch on the GPU instead of sequentially on one core;
  * ``DF.synthesis`` does not clobber its input (the reference does);
  * without a MI355X these functions raise — there is no CPU implementation in this package.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Union

import numpy as np
import torch

from . import _lib

ArrayLike = Union[np.ndarray, torch.Tensor]


def _to_dev(a: ArrayLike, dtype: torch.dtype) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        t = a
    else:
        t = torch.from_numpy(a)
    if t.dtype != dtype:
        raise TypeError(f"expected dtype {dtype}, got {t.dtype}")
    return t.to(_lib.device()).contiguous()


def _ret(t: torch.Tensor, like: ArrayLike):
    if isinstance(like, torch.Tensor):
        return t
    return t.cpu().numpy()


def _check_contig(a: ArrayLike, what: str = "Input") -> None:
    # pyDF/src/lib.rs:59-64,94-99
    if isinstance(a, np.ndarray):
        ok = a.size > 0 and a.flags["C_CONTIGUOUS"]
    else:
        ok = a.numel() > 0 and a.is_contiguous()
    if not ok:
        raise RuntimeError(f"[df] {what} array empty or not contiguous.")


class _Bands:
    """Device-resident ERB band table (cached per widths tuple)."""

    _cache = {}

    def __init__(self, widths):
        w = np.ascontiguousarray(np.asarray(widths), dtype=np.uint64)
        h = C.c_void_p()
        _lib.check(_lib.lib().dfx_bands_create(w.ctypes.data_as(C.POINTER(C.c_uint64)), len(w), C.byref(h)))
        self.handle, self.nb, self.F = h, len(w), int(w.sum())

    @classmethod
    def get(cls, widths) -> "_Bands":
        key = (_lib.library_path(), tuple(int(v) for v in np.asarray(widths).tolist()))
        b = cls._cache.get(key)
        if b is None:
            b = cls._cache[key] = cls(widths)
        return b


class DF:
    """pyDF/src/lib.rs:14-136 — DeepFilter state used for analysis and synthesis."""

    def __init__(self, sr: int, fft_size: int, hop_size: int, nb_bands: int = 32, min_nb_erb_freqs: int = 1):
        h = C.c_void_p()
        rc = _lib.lib().dfx_state_create(int(sr), int(fft_size), int(hop_size), int(nb_bands), int(min_nb_erb_freqs),
                                         C.byref(h))
        _lib.check(rc)
        self._h = h
        self._sr, self._fft, self._hop, self._nb = int(sr), int(fft_size), int(hop_size), int(nb_bands)
        self._ana_mem: Optional[torch.Tensor] = None  # [fft-hop], None == zeros (after reset)
        self._syn_mem: Optional[torch.Tensor] = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        try:
            if h:
                _lib.lib().dfx_state_free(h)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    # -- handle access for the fused paths
    @property
    def handle(self) -> C.c_void_p:
        return self._h

    @property
    def bands_handle(self) -> C.c_void_p:
        return C.c_void_p(_lib.lib().dfx_state_bands(self._h))

    def analysis(self, input: ArrayLike, reset: bool = True) -> ArrayLike:
        """[C, T] float32 -> [C, T // hop, fft/2+1] complex64 (pyDF/src/lib.rs:41-72)."""
        if not isinstance(input, (np.ndarray, torch.Tensor)) or input.ndim != 2:
            raise TypeError("argument 'input': expected a 2-d float32 array")
        if (input.dtype != np.float32) if isinstance(input, np.ndarray) else (input.dtype != torch.float32):
            raise TypeError("argument 'input': expected a 2-d float32 array")
        Cn, T = input.shape
        hop, F, ML = self._hop, self._fft // 2 + 1, self._fft - self._hop
        Tf = T // hop
        if reset and Cn > 0:
            self._syn_mem = None   # pyDF resets the whole DFState per channel (lib.rs:56-58,155-158): analysis AND synthesis memory
        if Cn == 0 or Tf == 0:
            if Cn > 0:
                _check_contig(input)
            out = torch.zeros((Cn, Tf, F), dtype=torch.complex64, device=_lib.device())
            return _ret(out, input)
        _check_contig(input)
        x = _to_dev(input, torch.float32)
        out = torch.empty((Cn, Tf, F), dtype=torch.complex64, device=x.device)
        mem_out = torch.empty((Cn, ML), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        if reset or Cn == 1:
            mem_in = None
            if not reset and self._ana_mem is not None:
                mem_in = self._ana_mem.reshape(1, ML)
            _lib.check(L.dfx_analysis(self._h, _lib.ptr(x), Cn, T, x.stride(0), _lib.ptr(mem_in), _lib.ptr(mem_out),
                                      _lib.ptr(out), _lib.stream()))
        else:
            # pyDF keeps ONE DFState: without a reset, channel c continues from the memory channel c-1 left behind
            mem = self._ana_mem.reshape(1, ML) if self._ana_mem is not None else None
            for c in range(Cn):
                _lib.check(L.dfx_analysis(self._h, _lib.ptr(x[c:c + 1]), 1, T, x.stride(0), _lib.ptr(mem),
                                          _lib.ptr(mem_out[c:c + 1]), _lib.ptr(out[c:c + 1]), _lib.stream()))
                mem = mem_out[c:c + 1]
        self._ana_mem = mem_out[Cn - 1].clone()
        return _ret(out, input)

    def synthesis(self, input: ArrayLike, reset: bool = True) -> ArrayLike:
        """[C, T', F] complex64 -> [C, T' * hop] float32 (pyDF/src/lib.rs:74-107)."""
        if not isinstance(input, (np.ndarray, torch.Tensor)) or input.ndim != 3:
            raise TypeError("argument 'input': expected a 3-d complex64 array")
        if (input.dtype != np.complex64) if isinstance(input, np.ndarray) else (input.dtype != torch.complex64):
            raise TypeError("argument 'input': expected a 3-d complex64 array")
        _check_contig(input)
        Cn, Tf, F = input.shape
        if F != self._fft // 2 + 1:
            raise RuntimeError(f"DF shape error: expected {self._fft // 2 + 1} frequency bins, got {F}")
        hop, ML = self._hop, self._fft - self._hop
        if reset and Cn > 0:
            self._ana_mem = None   # lib.rs:88-90: the same reset from the synthesis side
        y = _to_dev(input, torch.complex64)
        yr = torch.view_as_real(y)
        out = torch.empty((Cn, Tf * hop), dtype=torch.float32, device=y.device)
        mem_out = torch.empty((Cn, ML), dtype=torch.float32, device=y.device)
        L = _lib.lib()
        if reset or Cn == 1:
            mem_in = None
            if not reset and self._syn_mem is not None:
                mem_in = self._syn_mem.reshape(1, ML)
            _lib.check(L.dfx_synthesis(self._h, _lib.ptr(yr), Cn, Tf, _lib.ptr(mem_in), _lib.ptr(mem_out), _lib.ptr(out),
                                       out.stride(0), _lib.stream()))
        else:
            mem = self._syn_mem.reshape(1, ML) if self._syn_mem is not None else None
            for c in range(Cn):
                _lib.check(L.dfx_synthesis(self._h, _lib.ptr(yr[c:c + 1]), 1, Tf, _lib.ptr(mem),
                                           _lib.ptr(mem_out[c:c + 1]), _lib.ptr(out[c:c + 1]), out.stride(0),
                                           _lib.stream()))
                mem = mem_out[c:c + 1]
        self._syn_mem = mem_out[Cn - 1].clone()
        return _ret(out, input)

    def erb_widths(self) -> np.ndarray:
        out = np.zeros(self._nb, dtype=np.uint64)
        _lib.check(_lib.lib().dfx_state_erb_widths(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def fft_window(self) -> np.ndarray:
        out = np.zeros(self._fft, dtype=np.float32)
        _lib.check(_lib.lib().dfx_state_fft_window(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def wnorm(self) -> float:
        return float(_lib.lib().dfx_state_wnorm(self._h))

    def sr(self) -> int:
        return self._sr

    def fft_size(self) -> int:
        return self._fft

    def hop_size(self) -> int:
        return self._hop

    def nb_erb(self) -> int:
        return self._nb

    def reset(self) -> None:
        self._ana_mem = None
        self._syn_mem = None


def _dtype_is(a: ArrayLike, np_dt, t_dt) -> bool:
    return a.dtype == (np_dt if isinstance(a, np.ndarray) else t_dt)


def erb(input: ArrayLike, erb_fb: Union[np.ndarray, List[int]], db: bool = True) -> ArrayLike:
    """ERB filterbank (+ dB).  complex64 [..., F] with 2-4 dims -> float32 [..., E]  (pyDF/src/lib.rs:142-192)."""
    if not _dtype_is(input, np.complex64, torch.complex64):
        raise TypeError("argument 'input': expected complex64")
    if input.ndim not in (2, 3, 4):
        raise ValueError(f"Dimension not supported for erb: {input.ndim}")
    bands = _Bands.get(erb_fb)
    if input.shape[-1] != bands.F:
        raise RuntimeError(f"DF shape error: {input.shape[-1]} frequency bins do not match the ERB widths ({bands.F})")
    x = _to_dev(input, torch.complex64)
    rows = int(np.prod(x.shape[:-1]))
    out = torch.empty(tuple(x.shape[:-1]) + (bands.nb,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().dfx_erb(bands.handle, _lib.ptr(torch.view_as_real(x)), rows, int(bool(db)), _lib.ptr(out),
                                  _lib.stream()))
    return _ret(out, input)


def erb_inv(input: ArrayLike, erb_fb: Union[np.ndarray, List[int]]) -> ArrayLike:
    """float32 [..., E] -> float32 [..., F]  (pyDF/src/lib.rs:194-250)."""
    if not _dtype_is(input, np.float32, torch.float32):
        raise TypeError("argument 'input': expected float32")
    bands = _Bands.get(erb_fb)
    if input.shape[-1] != bands.nb:
        raise ValueError(f"Number of erb bands do not match with input: {input.shape[-1]}, {bands.nb}")
    if input.ndim not in (2, 3, 4):
        raise ValueError(f"Dimension not supported for erb: {input.ndim}")
    x = _to_dev(input, torch.float32)
    rows = int(np.prod(x.shape[:-1]))
    out = torch.empty(tuple(x.shape[:-1]) + (bands.F,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().dfx_erb_inv(bands.handle, _lib.ptr(x), rows, _lib.ptr(out), _lib.stream()))
    return _ret(out, input)


def erb_norm(erb: ArrayLike, alpha: float, state: Optional[ArrayLike] = None) -> ArrayLike:
    """Exponential mean normalisation of [C, T, E] float32 — IN PLACE and returned (pyDF/src/lib.rs:252-274)."""
    if erb.ndim != 3 or not _dtype_is(erb, np.float32, torch.float32):
        raise TypeError("argument 'erb': expected a 3-d float32 array")
    _check_contig(erb)
    Cn, T, E = erb.shape
    x = _to_dev(erb, torch.float32)
    if isinstance(erb, torch.Tensor) and x.data_ptr() != erb.data_ptr():
        x = x.clone()
    st = None
    if state is not None:
        st = _to_dev(state, torch.float32).clone()  # `.to_owned()`: the caller's state is not updated
        if tuple(st.shape) != (Cn, E):
            raise RuntimeError(f"DF shape error: state shape {tuple(st.shape)} != {(Cn, E)}")
    _lib.check(_lib.lib().dfx_erb_norm(_lib.ptr(x), Cn, T, E, float(alpha), _lib.ptr(st), _lib.stream()))
    if isinstance(erb, np.ndarray):
        res = x.cpu().numpy()
        erb[...] = res  # the reference mutates its input (unsafe as_array_mut) and returns a copy
        return res
    if x.data_ptr() != erb.data_ptr():
        erb.copy_(x)
    return x.clone()


def unit_norm(spec: ArrayLike, alpha: float, state: Optional[ArrayLike] = None) -> ArrayLike:
    """Exponential unit normalisation of [C, T, F] complex64; works on a copy (pyDF/src/lib.rs:276-298)."""
    if spec.ndim != 3 or not _dtype_is(spec, np.complex64, torch.complex64):
        raise TypeError("argument 'spec': expected a 3-d complex64 array")
    Cn, T, F = spec.shape
    if isinstance(spec, torch.Tensor) and _lib.on_device(spec) and spec.stride(2) == 1 and spec.stride(0) == T * spec.stride(1):
        x, fstride = spec, spec.stride(1)  # e.g. spec[..., :nb_df]: read in place with a frame stride, no copy
    else:
        x = _to_dev(np.ascontiguousarray(spec) if isinstance(spec, np.ndarray) else spec.contiguous(), torch.complex64)
        fstride = F
    out = torch.empty((Cn, T, F), dtype=torch.complex64, device=_lib.device())
    st = None
    if state is not None:
        st = _to_dev(state, torch.float32).clone()
        if tuple(st.shape) != (Cn, F):
            raise RuntimeError(f"DF shape error: state shape {tuple(st.shape)} != {(Cn, F)}")
    if Cn and T:
        xr = torch.view_as_real(x)
        _lib.check(_lib.lib().dfx_unit_norm(_lib.ptr(xr), fstride, _lib.ptr(torch.view_as_real(out)), Cn, T, F,
                                            float(alpha), _lib.ptr(st), _lib.stream()))
    return _ret(out, spec)


def unit_norm_init(num_freq_bins: int) -> np.ndarray:
    """pyDF/src/lib.rs:300-309: linspace(1e-3, 1e-4, n) as [1, n] float32."""
    out = np.zeros((1, int(num_freq_bins)), dtype=np.float32)
    _lib.check(_lib.lib().dfx_unit_norm_init(int(num_freq_bins), out.ctypes.data_as(C.POINTER(C.c_float))))
    return out
