"""PDE matrix-function helpers on the Arnoldi path -- MI355X build ("next" tier).

Covers the pieces of the reference's ``util/pde_util.py`` that are thin compositions of the hot path:
``expm_arnoldi`` (:257-268), its dense baseline ``expm_pade`` (:271-280), ``solver_expm`` (:240-252), ``sampler_lanczos`` (:335-356),
the small helpers ``mesh_tensorproduct`` / ``loss_mse`` / ``loss_mse_relative`` (:14-15, :160-173) and the 5-point wave operator
(:18-20,126-157) expressed as a native CSR operator so that the non-symmetric Arnoldi forward/adjoint kernels run it
end to end.  The dense k x k ``expm`` / ``eigh`` are torch calls (k <= ~100: plumbing, differentiable).
"""

from __future__ import annotations

import numpy as np
import torch

from .. import arnoldi, lanczos
from ..operators import CsrOp, RowShardedOp, as_operator


def expm_arnoldi(krylov_depth, *, max_squarings: int = 32, reortho="full", custom_vjp=True):
    """exp(dt A) y0 ~ (1/c) Q expm(dt H) e1  (util/pde_util.py:257-268).  ``max_squarings`` is accepted for signature
    parity (torch.linalg.matrix_exp chooses its own scaling)."""

    def expm(matvec, dt, y0_flat, *p):
        algorithm = arnoldi.hessenberg(matvec, krylov_depth, reortho=reortho, custom_vjp=custom_vjp)
        Q, H, _r, c = algorithm(y0_flat, *p)
        op, _ = as_operator(matvec)
        if isinstance(op, RowShardedOp):  # y0, Q and the result are row shards; H and c are replicated but consumed per shard
            from ..distributed import sum_grad

            H, c = sum_grad(op.comm, H), sum_grad(op.comm, c)
        expmat = torch.linalg.matrix_exp(dt * H)
        # Q expm(dt H) e1 as a weighted sum over the (k, n) storage of the basis: elementwise kernels in both directions (as a
        # matmul on the transposed view the backward went to a 41 ms skinny rocBLAS GEMM at n = 2e6)
        Qkn = Q.transpose(-1, -2)
        out = (Qkn * expmat[..., :, 0:1]).sum(dim=-2) / (c[..., None] if c.dim() else c)
        return out, {"num_matvecs": krylov_depth}

    return expm


def expm_pade():
    """Dense baseline of expm_arnoldi (util/pde_util.py:271-280): materialise A through the matvec (the reference takes its Jacobian; the
    matvec is linear, so n applications to the identity are the same matrix) and apply torch.linalg.matrix_exp(dt A) to y0.  Small n only."""

    def expm(matvec, dt, y0_flat, *p):
        n = y0_flat.shape[0]
        op, _ = as_operator(matvec)
        cols = op(torch.eye(n, dtype=y0_flat.dtype, device=y0_flat.device), *p)  # row b = A e_b
        return torch.linalg.matrix_exp(dt * cols.t()) @ y0_flat

    return expm


def mesh_tensorproduct(x, y, /):
    """util/pde_util.py:14-15: jnp.stack(jnp.meshgrid(x, y)) -- 'xy' indexing, shape (2, len(y), len(x))."""
    return torch.stack(torch.meshgrid(x, y, indexing="xy"))


def loss_mse():
    """util/pde_util.py:160-164."""

    def loss(sol, /, *, targets):
        return torch.mean((sol - targets) ** 2)

    return loss


def loss_mse_relative(*, nugget, reduce=torch.mean):
    """util/pde_util.py:167-173."""

    def loss(sol, /, *, targets):
        mse_abs = (sol - targets) ** 2
        return reduce(mse_abs / (nugget + torch.abs(targets)))

    return loss


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
    import scipy.sparse

    if boundary not in ("neumann", "dirichlet"):
        raise ValueError(boundary)
    n2 = res * res
    idx = np.arange(n2).reshape(res, res)
    ii, jj = np.meshgrid(np.arange(res), np.arange(res), indexing="ij")
    st = np.array([[0.0, 1.0, 0.0], [1.0, -2.0, 1.0], [0.0, 1.0, 0.0]]) / dx**2  # stencil_laplacian (util/pde_util.py:18-20)
    r_l, c_l, w_l = [], [], []
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            sw = st[di + 1, dj + 1]
            if sw == 0.0:
                continue
            ni, nj = ii + di, jj + dj
            inside = (ni >= 0) & (ni < res) & (nj >= 0) & (nj < res)
            if boundary == "neumann":  # jnp.pad(mode="edge"): the padded value is the nearest interior cell
                ni, nj, keep = np.clip(ni, 0, res - 1), np.clip(nj, 0, res - 1), np.ones_like(inside)
            else:                      # zero padding: the term drops out
                keep = inside
            r_l.append(idx[keep])
            c_l.append(idx[ni[keep], nj[keep]])
            w_l.append(np.full(int(keep.sum()), sw))
    lap = scipy.sparse.coo_matrix((np.concatenate(w_l), (np.concatenate(r_l), np.concatenate(c_l))), shape=(n2, n2)).tocsr()
    lap.sum_duplicates()
    lap = lap.tocoo()
    # state (u, du): d/dt u = du (identity block), d/dt du = scale o Lap u;  value = w * (scale[coef] if coef >= 0 else 1)
    rows = np.concatenate([np.arange(n2), n2 + lap.row])
    cols = np.concatenate([n2 + np.arange(n2), lap.col])
    w = np.concatenate([np.ones(n2), lap.data])
    coef = np.concatenate([np.full(n2, -1), lap.row])
    op, wv, order = CsrOp.from_coo(rows, cols, w, 2 * n2, device)
    wv = wv.to(dtype)
    coef_t = torch.as_tensor(np.asarray(coef)[order.numpy()], device=device)
    # CSR order: rows 0 .. n2-1 hold the identity block (one entry each), rows n2 .. 2 n2 - 1 the Laplacian block, so the
    # entries that depend on scale[r] are the contiguous range crow[n2 + r] .. crow[n2 + r + 1]
    seg = (op.crow[n2:].to(torch.int64) - n2).contiguous()
    gather = torch.clamp_min(coef_t, 0)
    is_lap = coef_t >= 0

    class _Values(torch.autograd.Function):
        """values = w * scale[row] on the Laplacian block.  The backward is a per-row segment sum via a cumulative sum
        (torch's generic gather backward sorts all 6e6 indices: 50 ms at res = 1000)."""

        @staticmethod
        def forward(ctx, s):
            return torch.where(is_lap, wv * s[gather], wv)

        @staticmethod
        def backward(ctx, g):
            gw = (g * wv)[n2:].double()
            cs = torch.cat([gw.new_zeros(1), torch.cumsum(gw, 0)])
            return (cs[seg[1:]] - cs[seg[:-1]]).to(g.dtype)

    def values_fn(scale):
        return _Values.apply(scale.reshape(-1))

    return op, values_fn
