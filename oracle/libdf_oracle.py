"""`libdf`-shaped Python front end of the C oracle (TEST INFRASTRUCTURE ONLY).

Mirrors the pyo3 module of the reference, pyDF/src/lib.rs:14-310 (stub: pyDF/libdf.pyi:5-70): class ``DF`` and the
free functions ``erb``, ``erb_inv``, ``erb_norm``, ``unit_norm``, ``unit_norm_init`` with the same shapes, dtypes,
in-place side effects and exception types.  It is used (a) as the checker in tests, (b) as the ``libdf`` module when the
reference's own Python code is imported to generate golden vectors (tools/gen_golden.py) and (c) as bench.py's
cpu_baseline ("port").
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdf_oracle.so")


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("df_oracle.c", "df_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


def _load():
    lib = ctypes.CDLL(build())
    c = ctypes
    fp, u64p, vp = c.POINTER(c.c_float), c.POINTER(c.c_uint64), c.c_void_p
    lib.dfo_erb_fb.argtypes = [c.c_int, c.c_int, c.c_int, c.c_int, u64p]
    lib.dfo_erb_fb.restype = c.c_int
    lib.dfo_state_new.argtypes = [c.c_int] * 5
    lib.dfo_state_new.restype = vp
    lib.dfo_state_free.argtypes = [vp]
    lib.dfo_state_reset.argtypes = [vp]
    for n in ("sr", "fft_size", "hop_size", "nb_erb"):
        f = getattr(lib, "dfo_state_" + n)
        f.argtypes, f.restype = [vp], c.c_int
    lib.dfo_state_wnorm.argtypes, lib.dfo_state_wnorm.restype = [vp], c.c_float
    lib.dfo_state_window.argtypes = [vp, fp]
    lib.dfo_state_erb_widths.argtypes = [vp, u64p]
    lib.dfo_analysis.argtypes = [vp, fp, c.c_int64, c.c_int64, c.c_int, fp]
    lib.dfo_synthesis.argtypes = [vp, fp, c.c_int64, c.c_int64, c.c_int, fp]
    lib.dfo_erb.argtypes = [fp, c.c_int64, u64p, c.c_int, c.c_int, fp]
    lib.dfo_erb_inv.argtypes = [fp, c.c_int64, u64p, c.c_int, fp]
    lib.dfo_erb_norm.argtypes = [fp, c.c_int64, c.c_int64, c.c_int, c.c_float, fp]
    lib.dfo_unit_norm.argtypes = [fp, c.c_int64, c.c_int64, c.c_int, c.c_float, fp]
    lib.dfo_apply_band_gain.argtypes = [fp, c.c_int64, fp, u64p, c.c_int]
    lib.dfo_post_filter.argtypes = [fp, fp, c.c_int64, c.c_int, c.c_float]
    lib.dfo_unit_norm_init.argtypes = [c.c_int, fp]
    for n in dir(lib):
        pass
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _u64(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def _check_contig(a: np.ndarray, what: str = "Input"):
    # pyDF/src/lib.rs:59-64,94-99: as_slice() fails on empty or non-contiguous rows
    if a.size == 0 or not a.flags["C_CONTIGUOUS"]:
        raise RuntimeError(f"[df] {what} array empty or not contiguous.")


class DF:
    """pyDF/src/lib.rs:14-136.  One sequential DFState, reset before every channel."""

    def __init__(self, sr: int, fft_size: int, hop_size: int, nb_bands: int = 32, min_nb_erb_freqs: int = 1):
        h = lib().dfo_state_new(int(sr), int(fft_size), int(hop_size), int(nb_bands), int(min_nb_erb_freqs))
        if not h:
            # lib.rs:111 assert!(hop_size * 2 <= fft_size) -> Rust panic surfaces in Python as pyo3 PanicException
            raise RuntimeError("assertion failed: hop_size * 2 <= fft_size")
        self._h = ctypes.c_void_p(h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.dfo_state_free(h)

    def analysis(self, input: np.ndarray, reset: bool = True) -> np.ndarray:
        if not isinstance(input, np.ndarray) or input.dtype != np.float32 or input.ndim != 2:
            raise TypeError("argument 'input': expected a 2-d float32 numpy array")
        _check_contig(input)
        C, T = input.shape
        hop, F = self.hop_size(), self.fft_size() // 2 + 1
        out = np.zeros((C, T // hop, F), dtype=np.complex64)
        if out.size:
            lib().dfo_analysis(self._h, _fp(input), C, T, int(bool(reset)), _fp(out.view(np.float32)))
        return out

    def synthesis(self, input: np.ndarray, reset: bool = True) -> np.ndarray:
        if not isinstance(input, np.ndarray) or input.dtype != np.complex64 or input.ndim != 3:
            raise TypeError("argument 'input': expected a 3-d complex64 numpy array")
        _check_contig(input)
        C, Tf, F = input.shape
        if F != self.fft_size() // 2 + 1:
            raise RuntimeError("[df] Input array has wrong number of frequency bins.")
        out = np.zeros((C, Tf * self.hop_size()), dtype=np.float32)
        lib().dfo_synthesis(self._h, _fp(input.view(np.float32)), C, Tf, int(bool(reset)), _fp(out))
        return out

    def erb_widths(self) -> np.ndarray:
        out = np.zeros(self.nb_erb(), dtype=np.uint64)
        lib().dfo_state_erb_widths(self._h, _u64(out))
        return out

    def fft_window(self) -> np.ndarray:
        out = np.zeros(self.fft_size(), dtype=np.float32)
        lib().dfo_state_window(self._h, _fp(out))
        return out

    def wnorm(self) -> float:  # not in pyDF; DFState.wnorm (lib.rs:134)
        return float(lib().dfo_state_wnorm(self._h))

    def sr(self) -> int:
        return lib().dfo_state_sr(self._h)

    def fft_size(self) -> int:
        return lib().dfo_state_fft_size(self._h)

    def hop_size(self) -> int:
        return lib().dfo_state_hop_size(self._h)

    def nb_erb(self) -> int:
        return lib().dfo_state_nb_erb(self._h)

    def reset(self) -> None:
        lib().dfo_state_reset(self._h)


def _widths(erb_fb) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(erb_fb), dtype=np.uint64)


def erb(input: np.ndarray, erb_fb, db: bool = True) -> np.ndarray:
    """pyDF/src/lib.rs:142-192."""
    if input.dtype != np.complex64:
        raise TypeError("argument 'input': expected complex64")
    if input.ndim not in (2, 3, 4):
        raise ValueError(f"Dimension not supported for erb: {input.ndim}")
    w = _widths(erb_fb)
    x = np.ascontiguousarray(input)
    rows = int(np.prod(x.shape[:-1]))
    out = np.zeros(x.shape[:-1] + (len(w),), dtype=np.float32)
    if int(w.sum()) != x.shape[-1]:
        raise RuntimeError("DF shape error: frequency bins do not match erb widths")
    if rows:
        lib().dfo_erb(_fp(x.view(np.float32)), rows, _u64(w), len(w), int(bool(db)), _fp(out))
    return out


def erb_inv(input: np.ndarray, erb_fb) -> np.ndarray:
    """pyDF/src/lib.rs:194-250."""
    if input.dtype != np.float32:
        raise TypeError("argument 'input': expected float32")
    w = _widths(erb_fb)
    if input.shape[-1] != len(w):
        raise ValueError(f"Number of erb bands do not match with input: {input.shape[-1]}, {len(w)}")
    if input.ndim not in (2, 3, 4):
        raise ValueError(f"Dimension not supported for erb: {input.ndim}")
    x = np.ascontiguousarray(input)
    rows = int(np.prod(x.shape[:-1]))
    out = np.zeros(x.shape[:-1] + (int(w.sum()),), dtype=np.float32)
    if rows:
        lib().dfo_erb_inv(_fp(x), rows, _u64(w), len(w), _fp(out))
    return out


def erb_norm(erb: np.ndarray, alpha: float, state: Optional[np.ndarray] = None) -> np.ndarray:
    """pyDF/src/lib.rs:252-274: normalises *in place* (unsafe as_array_mut) and returns a copy."""
    if erb.dtype != np.float32 or erb.ndim != 3:
        raise TypeError("argument 'erb': expected a 3-d float32 numpy array")
    _check_contig(erb)
    C, T, E = erb.shape
    st = None
    if state is not None:
        st = np.array(state, dtype=np.float32, copy=True)  # .to_owned(): caller's state is not updated
    lib().dfo_erb_norm(_fp(erb), C, T, E, float(alpha), _fp(st) if st is not None else None)
    return erb.copy()


def unit_norm(spec: np.ndarray, alpha: float, state: Optional[np.ndarray] = None) -> np.ndarray:
    """pyDF/src/lib.rs:276-298: works on a copy."""
    if spec.dtype != np.complex64 or spec.ndim != 3:
        raise TypeError("argument 'spec': expected a 3-d complex64 numpy array")
    out = np.array(spec, dtype=np.complex64, order="C", copy=True)
    C, T, F = out.shape
    st = None
    if state is not None:
        st = np.array(state, dtype=np.float32, copy=True)
    if out.size:
        lib().dfo_unit_norm(_fp(out.view(np.float32)), C, T, F, float(alpha), _fp(st) if st is not None else None)
    return out


def unit_norm_init(num_freq_bins: int) -> np.ndarray:
    """pyDF/src/lib.rs:300-309."""
    out = np.zeros((1, int(num_freq_bins)), dtype=np.float32)
    lib().dfo_unit_norm_init(int(num_freq_bins), _fp(out))
    return out


# ---- helpers that are not part of pyDF but restate lib.rs functions used by the streaming / DfNet paths ----

def apply_band_gain(spec: np.ndarray, gains: np.ndarray, erb_fb) -> np.ndarray:
    """libDF/src/lib.rs:314-326 over all leading dims; returns a new array."""
    w = _widths(erb_fb)
    out = np.array(spec, dtype=np.complex64, order="C", copy=True)
    g = np.ascontiguousarray(gains, dtype=np.float32)
    rows = int(np.prod(out.shape[:-1]))
    lib().dfo_apply_band_gain(_fp(out.view(np.float32)), rows, _fp(g), _u64(w), len(w))
    return out


def post_filter(noisy: np.ndarray, enh: np.ndarray, beta: float) -> np.ndarray:
    """libDF/src/lib.rs:446-471 over all leading dims; returns a new array."""
    n = np.ascontiguousarray(noisy, dtype=np.complex64)
    out = np.array(enh, dtype=np.complex64, order="C", copy=True)
    rows = int(np.prod(out.shape[:-1]))
    lib().dfo_post_filter(_fp(n.view(np.float32)), _fp(out.view(np.float32)), rows, out.shape[-1], float(beta))
    return out


def erb_fb_widths(sr: int, fft_size: int, nb_bands: int, min_nb_freqs: int) -> np.ndarray:
    out = np.zeros(nb_bands, dtype=np.uint64)
    lib().dfo_erb_fb(sr, fft_size, nb_bands, min_nb_freqs, _u64(out))
    return out
