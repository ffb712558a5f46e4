"""Lanczos posterior samplers of the reference's ``util/bnn_util.py`` (:372-409) -- MI355X build ("next" tier, §8f-3).

Only the samplers are here: they are compositions of ``lanczos.tridiag(reortho="full")`` (the hot-path kernels) with a small dense
factorisation of the k x k tridiagonal matrix (torch, k <= ~100).  The network / GGN construction side of bnn_util (flax models,
``ggn_fun``) is outside SURVEY.md §8; ``ggn_fun`` / ``ggn_vp`` are whatever produces the matrix or the matvec.
"""

from __future__ import annotations

import torch

from .. import lanczos
from ..operators import DenseOp


def _normal(key, shape, like):
    if torch.is_tensor(key):
        return key  # explicit draws (the tests' way of fixing the randomness)
    gen = torch.Generator(device=like.device)
    gen.manual_seed(int(key))
    return torch.randn(shape, dtype=like.dtype, device=like.device, generator=gen)


def _dense_tridiag(diagonal, off_diagonal):
    """util/bnn_util.py:412-413 (batched)."""
    return torch.diag_embed(diagonal) + torch.diag_embed(off_diagonal, 1) + torch.diag_embed(off_diagonal, -1)


def sampler_cholesky(*, ggn_fun, num):
    """util/bnn_util.py:361-369: samples  variables + chol(ggn^{-1}) eps  (dense baseline)."""

    def sample(key, alpha, variables, x_train, y_train):
        ggn = ggn_fun(alpha, variables, x_train, y_train)
        ggn_inv_sqrt = torch.linalg.cholesky(torch.linalg.inv(ggn))
        eps = _normal(key, (num, *variables.shape), variables)
        return (ggn_inv_sqrt @ eps.T).T + variables[None, ...]

    return sample


def sampler_lanczos(*, ggn_fun, num, lanczos_rank):
    """util/bnn_util.py:372-389: per draw eps, Lanczos on the GGN started at eps, then  Q^T chol(T^{-1}) Q eps."""

    def sample(key, alpha, variables, x_train, y_train):
        ggn = ggn_fun(alpha, variables, x_train, y_train)
        tridiag = lanczos.tridiag(DenseOp().bind(ggn), lanczos_rank, reortho="full")
        eps = _normal(key, (num, *variables.shape), variables)
        (Q, (diag, off)), _ = tridiag(eps)  # Q (num, k, n)
        tri_inv_sqrt = torch.linalg.cholesky(torch.linalg.inv(_dense_tridiag(diag, off)))
        coeff = torch.einsum("bkn,bn->bk", Q, eps)
        return torch.einsum("bkn,bk->bn", Q, torch.einsum("bkl,bl->bk", tri_inv_sqrt, coeff)) + variables[None, ...]

    return sample


def lanczos_sampler(*, ggn_vp, num_samples, lanczos_rank, key, params_vec):
    """util/bnn_util.py:392-409: eigen-decomposition of the Lanczos tridiagonal matrix, eigenvalues below 1e-9 dropped."""
    eps = _normal(key, (num_samples, *params_vec.shape), params_vec)
    (Q, (diag, off)), _ = lanczos.tridiag(ggn_vp, lanczos_rank, reortho="full")(eps)
    w, v = torch.linalg.eigh(_dense_tridiag(diag, off))
    eigvecs = Q.transpose(-1, -2) @ v  # (num, n, k)
    small = w < 1e-9
    inv_eigvals = torch.where(small, torch.zeros_like(w), 1 / torch.where(small, torch.ones_like(w), w))
    sample = torch.sqrt(inv_eigvals) * eps[:, :lanczos_rank]
    return params_vec + torch.einsum("bnk,bk->bn", eigvecs, sample)
