"""torch-CPU restatement of the reference's multi-frame filter ops — TEST INFRASTRUCTURE ONLY.

``df/multiframe.py``: MultiFrameModule.spec_unfold :85-95, apply_coefs :103-107, MfWf.forward :282-321, MfMvdr.forward :373-413,
_tik_reg :436-452.  Pinned: tests/golden/mf_ops.npz holds the outputs of the reference's own modules (tools/gen_golden_mf.py) for
every (cholesky_decomp, inverse) combination the MF model can select; tests/test_mf.py checks this restatement against them.
The linear solve is ``torch.linalg.solve`` (LAPACK cgesv: LU with partial pivoting), as in the reference."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor


def tik_reg(mat: Tensor, reg: float = 1e-7, eps: float = 1e-8) -> Tensor:
    """multiframe.py:436-452: mat + (trace(mat).real * reg + eps) * I."""
    n = mat.size(-1)
    eye = torch.eye(n, dtype=mat.dtype)
    epsilon = torch.diagonal(mat, 0, -2, -1).sum(-1).real[..., None, None] * reg + eps
    return mat + epsilon * eye


@torch.no_grad()
def mf_filter(spec: Tensor, ifc: Tensor, mat: Tensor, *, mvdr: bool, num_freqs: int, frame_size: int, lookahead: int = 0,
              cholesky_decomp: bool = False, inverse: bool = True, enforce_constraints: bool = True, eps: float = 1e-8,
              dload: float = 1e-7) -> Tensor:
    """spec [B,1,T,F,2], ifc [B,T,F',N*2], mat [B,T,F',N*N*2] -> [B,1,T,F,2] (a new tensor; the reference writes into spec)."""
    N = frame_size
    x = torch.view_as_complex(spec.contiguous())                                   # [B,1,T,F]
    xu = F.pad(torch.view_as_real(x), (0, 0, 0, 0, N - 1 - lookahead, lookahead))  # pad time (:72-76)
    xu = torch.view_as_complex(xu.contiguous()).unfold(2, N, 1) if N > 1 else x.unsqueeze(-1)   # [B,1,T,F,N]
    m = torch.view_as_complex(mat.unflatten(3, (N, N, 2)).contiguous()).clone()    # [B,T,F',N,N]
    iu = torch.triu_indices(N, N, 1)
    if cholesky_decomp:
        if enforce_constraints:
            m[:, :, :, iu[0], iu[1]] = 0.0
        m = m.matmul(m.transpose(3, 4).conj())
    if enforce_constraints and not inverse and not cholesky_decomp:
        torch.diagonal(m, dim1=-1, dim2=-2).imag = 0.0
        m[:, :, :, iu[0], iu[1]] = m[:, :, :, iu[1], iu[0]].conj()
    v = torch.view_as_complex(ifc.unflatten(3, (N, 2)).contiguous())               # [B,T,F',N]
    if not inverse:
        num = torch.linalg.solve(tik_reg(m, dload, eps), v)
    else:
        num = torch.einsum("...nm,...m->...n", m, v)
    if mvdr:
        den = torch.einsum("...n,...n->...", v.conj(), num)
        w = num * v[..., -1, None].conj() / (den.real.unsqueeze(-1) + eps)
    else:
        w = num
    y = torch.einsum("...n,...n->...", xu[..., :num_freqs, :], w.unsqueeze(1))      # apply_coefs
    out = x.clone()
    out[..., :num_freqs] = y
    return torch.view_as_real(out)
