"""Host-side mirror of the reference's multi-frame filter modules (``df/multiframe.py``): same class names, constructor arguments
and ``forward`` signatures — ``MfWf`` (:221-321), ``MfMvdr`` (:324-413) and the plain deep filter ``DF`` (:139-180) — computing on
the MI355X through libdfx (``dfx_mf_filter`` / ``dfx_df_apply``).  Inference only; unlike the reference the input spectrogram is
not modified in place: a new tensor is returned."""
from __future__ import annotations

import torch

from . import _lib


class _MfBase:
    _op = 0

    def __init__(self, num_freqs: int, frame_size: int, lookahead: int = 0, cholesky_decomp: bool = False, inverse: bool = True,
                 enforce_constraints: bool = True, eps: float = 1e-8, dload: float = 1e-7):
        self.num_freqs, self.frame_size, self.lookahead = int(num_freqs), int(frame_size), int(lookahead)
        self.cholesky_decomp, self.inverse, self.enforce_constraints = bool(cholesky_decomp), bool(inverse), bool(enforce_constraints)
        self.eps, self.dload = float(eps), float(dload)

    def eval(self):
        return self

    def forward(self, spec: torch.Tensor, ifc: torch.Tensor, iR: torch.Tensor) -> torch.Tensor:
        """spec [B, 1, T, F, 2], ifc [B, T, F', N*2], iR [B, T, F', N*N*2] (float32) -> filtered spec [B, 1, T, F, 2]."""
        N, nb = self.frame_size, self.num_freqs
        dev = _lib.device()
        src = spec.device
        x = spec.to(dev, torch.float32).contiguous()
        B, C, T, F, two = x.shape
        if C != 1 or two != 2:
            raise ValueError("spec must have shape [B, 1, T, F, 2]")
        v = ifc.to(dev, torch.float32).contiguous()
        m = iR.to(dev, torch.float32).contiguous()
        if tuple(v.shape) != (B, T, nb, 2 * N) or tuple(m.shape) != (B, T, nb, 2 * N * N):
            raise ValueError(f"ifc / matrix must have shapes [B, T, {nb}, {2 * N}] / [B, T, {nb}, {2 * N * N}]")
        out = torch.empty_like(x)
        _lib.check(_lib.lib().dfx_mf_filter(_lib.ptr(x), _lib.ptr(v), _lib.ptr(m), self._op, N, self.lookahead, int(self.cholesky_decomp),
                                            int(self.inverse), int(self.enforce_constraints), self.eps, self.dload, B, T, F, nb,
                                            _lib.ptr(out), _lib.stream()))
        return out.to(src)

    __call__ = forward

    def get_r_factor(self):
        raise NotImplementedError


class MfWf(_MfBase):
    """Multi-frame Wiener filter (multiframe.py:221-321)."""
    _op = 0

    def get_r_factor(self):  # :270-279
        return {(True, True): 2e3, (True, False): 3e7, (False, True): 2e-4, (False, False): 5e-6}[(self.inverse, self.cholesky_decomp)]


class MfMvdr(_MfBase):
    """Multi-frame MVDR beamformer (multiframe.py:324-413)."""
    _op = 1

    def get_r_factor(self):  # :361-370
        return {(True, True): 2e4, (True, False): 3e8, (False, True): 5e-5, (False, False): 1e-6}[(self.inverse, self.cholesky_decomp)]
