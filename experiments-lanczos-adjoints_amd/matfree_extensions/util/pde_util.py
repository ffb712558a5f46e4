"""PDE matrix-function helpers on the Arnoldi path -- MI355X build ("next" tier).

Covers the pieces of the reference's ``util/pde_util.py`` that are thin compositions of the hot path:
``expm_arnoldi`` (:257-268), ``solver_expm`` (:240-252), ``sampler_lanczos`` (:335-356) and the 5-point wave operator
(:18-20,126-157) expressed as a native CSR operator so that the non-symmetric Arnoldi forward/adjoint kernels run it
end to end.  The dense k x k ``expm`` / ``eigh`` are torch calls (k <= ~100: plumbing, differentiable).
"""

from __future__ import annotations

import numpy as np
import torch

from .. import arnoldi, lanczos
from ..operators import CsrOp


def expm_arnoldi(krylov_depth, *, max_squarings: int = 32, reortho="full", custom_vjp=True):
    """exp(dt A) y0 ~ (1/c) Q expm(dt H) e1  (util/pde_util.py:257-268).  ``max_squarings`` is accepted for signature
    parity (torch.linalg.matrix_exp chooses its own scaling)."""

    def expm(matvec, dt, y0_flat, *p):
        algorithm = arnoldi.hessenberg(matvec, krylov_depth, reortho=reortho, custom_vjp=custom_vjp)
        Q, H, _r, c = algorithm(y0_flat, *p)
        expmat = torch.linalg.matrix_exp(dt * H)
        out = (Q @ expmat[..., :, 0:1])[..., 0] / (c[..., None] if c.dim() else c)
        return out, {"num_matvecs": krylov_depth}

    return expm


def solver_expm(t0, t1, vector_field, /, expm):
    """util/pde_util.py:240-252 for flat states: y(t1) = exp((t1 - t0) A) y0 with A v = vector_field(v, *p)."""

    def solve(y0, *p):
        shape = y0.shape
        out, info = expm(lambda v, *q: vector_field(v.reshape(shape), *q).reshape(-1), t1 - t0, y0.reshape(-1), *p)
        return out.reshape(shape), info

    return solve


def sampler_lanczos(*, mean, cov_matvec, num, lanczos_rank):
    """util/pde_util.py:335-356: samples  mean + |eps| Q^T K^(1/2) Q eps/|eps|  from N(mean, A) via Lanczos."""

    def sample(key):
        tridiag = lanczos.tridiag(cov_matvec, lanczos_rank, reortho="full")
        if torch.is_tensor(key):
            eps = key
        else:
            gen = torch.Generator(device=mean.device)
            gen.manual_seed(int(key))
            eps = torch.randn((num, mean.numel()), dtype=mean.dtype, device=mean.device, generator=gen)
        norm = torch.linalg.vector_norm(eps, dim=-1, keepdim=True)
        u = eps / norm
        (Q, (diag, off)), _ = tridiag(u)  # Q (num, k, n)
        K = torch.diag_embed(diag) + torch.diag_embed(off, 1) + torch.diag_embed(off, -1)
        w, v = torch.linalg.eigh(K)
        w = torch.clamp_min(w, 0.0)
        factor = (v * torch.sqrt(w)[..., None, :]) @ v.transpose(-1, -2)
        coeff = torch.einsum("bkn,bn->bk", Q, u)
        return norm * torch.einsum("bkn,bk->bn", Q, torch.einsum("bkl,bl->bk", factor, coeff)) + mean.reshape(1, -1)

    return sample


def stencil_laplacian(dx):
    """util/pde_util.py:18-20."""
    return torch.tensor([[0.0, 1.0, 0.0], [1.0, -2.0, 1.0], [0.0, 1.0, 0.0]], dtype=torch.float64) / dx**2


def wave_operator(res: int, dx: float, *, boundary: str = "neumann", device=None, dtype=torch.float64):
    """The anisotropic wave right-hand side  d/dt (u, du) = (du, scale o Lap u)  (util/pde_util.py:126-143 with
    boundary_neumann :153-157 or boundary_dirichlet :146-150) as a NON-symmetric sparse operator on the flattened
    state (2 res^2,).  Returns (CsrOp, values_fn): ``values_fn(scale)`` maps the (res, res) positive coefficient field
    to the stored CSR values (differentiable), so  op(v, values_fn(scale))  is the matvec."""
    n2 = res * res
    idx = np.arange(n2).reshape(res, res)
    rows, cols, w, coef = [], [], [], []  # value = w * (scale[coef] if coef >= 0 else 1)
    for a in range(n2):  # upper-right identity block: d/dt u = du
        rows.append(a); cols.append(n2 + a); w.append(1.0); coef.append(-1)
    # assemble explicitly: convolve2d(stencil, pad(u)) with the reference's stencil [[0,1,0],[1,-2,1],[0,1,0]]/dx^2
    st = np.array([[0.0, 1.0, 0.0], [1.0, -2.0, 1.0], [0.0, 1.0, 0.0]]) / dx**2
    for i in range(res):
        for j in range(res):
            acc = {}
            for di in (-1, 0, 1):
                for dj in (-1, 0, 1):
                    sw = st[di + 1, dj + 1]
                    if sw == 0.0:
                        continue
                    ii, jj = i + di, j + dj
                    if 0 <= ii < res and 0 <= jj < res:
                        acc[idx[ii, jj]] = acc.get(idx[ii, jj], 0.0) + sw
                    elif boundary == "neumann":  # jnp.pad(mode="edge"): the padded value is the nearest interior cell
                        ic, jc = min(max(ii, 0), res - 1), min(max(jj, 0), res - 1)
                        acc[idx[ic, jc]] = acc.get(idx[ic, jc], 0.0) + sw
                    elif boundary != "dirichlet":
                        raise ValueError(boundary)
            for col, val in acc.items():
                rows.append(n2 + idx[i, j]); cols.append(col); w.append(val); coef.append(idx[i, j])
    rows, cols = np.asarray(rows), np.asarray(cols)
    op, wv, order = CsrOp.from_coo(rows, cols, np.asarray(w), 2 * n2, device)
    wv = wv.to(dtype)
    coef_t = torch.as_tensor(np.asarray(coef)[order.numpy()], device=device)

    def values_fn(scale):
        s = scale.reshape(-1)
        return torch.where(coef_t >= 0, wv * s[torch.clamp_min(coef_t, 0)], wv)

    return op, values_fn
