"""Low-rank approximations: partial (pivoted) Cholesky and the preconditioner built from it -- MI355X build.

Same functional surface as the reference's ``matfree_extensions/low_rank.py``:

    cholesky_partial(rank=...)(lazy_kernel, n)        -> (chol (n, rank), {})
    cholesky_partial_pivot(rank=...)(lazy_kernel, n)  -> (chol (n, rank), {"success": bool})
    preconditioner(cholesky)(lazy_kernel, nrows)      -> (solve, info),   solve(v, s) = (s I + L L^T)^{-1} v

executed by libmfx (``mfx_partial_cholesky`` / ``mfx_precond_apply``).  Difference: the reference's ``lazy_kernel(i, j)``
element callable becomes an object with element access -- a bound ``DenseOp`` / ``KernelGramOp`` (``op.bind(...)``), a
dense matrix, or ``without_noise(bound_gram_op)`` for the noise-free kernel of likelihood_pdf_p
(util/gp_util.py:253-254).  Like the reference (low_rank.py:47-55,86-93), nothing here is differentiable.
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .operators import BoundOp, DenseOp, NativeOp


class without_noise:
    """Marks a bound kernel-Gram operator as 'kernel only': K(X, X) without the noise on its diagonal."""

    def __init__(self, bound):
        self.bound = bound


def _element_source(lazy_kernel):
    with_noise = True
    if isinstance(lazy_kernel, without_noise):
        lazy_kernel, with_noise = lazy_kernel.bound, False
    if torch.is_tensor(lazy_kernel):
        lazy_kernel = DenseOp().bind(lazy_kernel)
    if not (isinstance(lazy_kernel, BoundOp) and isinstance(lazy_kernel.op, NativeOp)):
        raise TypeError("lazy_kernel must be a dense matrix or a bound DenseOp / KernelGramOp (element access on the "
                        "device); arbitrary (i, j) callables are not supported in the MI355X build")
    return lazy_kernel, with_noise


def _factor(lazy_kernel, n, rank, pivot):
    if rank > n:
        raise ValueError(f"Rank exceeds n: {rank} >= {n}.")
    if rank < 1:
        raise ValueError(f"Rank must be positive, but {rank} < {1}.")
    bound, with_noise = _element_source(lazy_kernel)
    with torch.no_grad():
        cparams = tuple(q.detach() for q in bound.op.constrain(*bound.params))
        _lib.require_device(*cparams)
        ref = cparams[0]
        dt, dev = ref.dtype, ref.device
        if bound.op.size(*cparams) != n:
            raise ValueError(f"operator has {bound.op.size(*cparams)} rows, expected n = {n}")
        desc = bound.op.descriptor(cparams, dt, n)
        ws = _lib.workspace_pcg(desc, n, 1, rank, dev)
        lt = torch.empty((rank, n), dtype=dt, device=dev)
        pivots = torch.empty((rank,), dtype=torch.int64, device=dev)
        success = torch.empty((1,), dtype=torch.int32, device=dev)
        _lib.check(_lib.get().mfx_partial_cholesky(C.byref(desc), rank, int(pivot), int(with_noise), _lib.ptr(lt),
                                                   _lib.ptr(pivots), _lib.ptr(success), _lib.ptr(ws), ws.numel(),
                                                   _lib.stream_ptr(dev)))
    return lt, pivots, success


def cholesky_partial(*, rank: int):
    """low_rank.py:63-120."""

    def cholesky(lazy_kernel, n: int, /):
        lt, _piv, _ok = _factor(lazy_kernel, n, rank, pivot=False)
        return lt.t(), {}

    return cholesky


def cholesky_partial_pivot(*, rank: int):
    """low_rank.py:123-228 (rows returned in the original order, :227-228)."""

    def cholesky(matrix_element, n: int):
        lt, piv, ok = _factor(matrix_element, n, rank, pivot=True)
        return lt.t(), {"success": ok[0] != 0, "pivots": piv}

    return cholesky


class _NoGrad(torch.autograd.Function):
    """low_rank.py:47-55: 'Ensure that no one ever differentiates through here'."""

    @staticmethod
    def forward(ctx, pre, v, s):
        return pre._apply(v, s)

    @staticmethod
    def backward(ctx, _g):
        raise RuntimeError


class Preconditioner:
    """solve(v, s) = (s I + L L^T)^{-1} v by the Woodbury identity (low_rank.py:31-43)."""

    def __init__(self, chol):
        lt = chol.t()
        self.lt = lt if lt.is_contiguous() else lt.contiguous()  # (rank, n); a view of what the factorisation wrote
        self.rank, self.n = self.lt.shape
        assert self.rank <= self.n, (self.n, self.rank)  # tall, not wide (low_rank.py:26-29)
        self._gram = None

    def minv(self, s):
        """(s I + L^T L)^{-1} (rank, rank) and the device scalar s."""
        if self._gram is None:
            self._gram = (self.lt @ self.lt.t()).double()
        s = torch.as_tensor(s, dtype=self.lt.dtype, device=self.lt.device).detach().reshape(1)
        cap = self._gram + s.double() * torch.eye(self.rank, dtype=torch.float64, device=self.lt.device)
        minv = torch.cholesky_inverse(torch.linalg.cholesky(cap))
        return minv.to(self.lt.dtype).contiguous(), s.contiguous()

    def _apply(self, v, s):
        _lib.require_device(v)
        V = (v if v.dim() == 2 else v[None]).detach().contiguous()
        p, n = V.shape
        if n != self.n or V.dtype != self.lt.dtype:
            raise ValueError(f"preconditioner of size {self.n} ({self.lt.dtype}) applied to {tuple(v.shape)} ({v.dtype})")
        minv, sdev = self.minv(s)
        desc = _lib.Operator()
        desc.kind, desc.dtype, desc.n = _lib.OP_DENSE, _lib.dtype_code(V.dtype), n
        ws = _lib.workspace_pcg(desc, n, p, self.rank, V.device)
        z = torch.empty_like(V)
        _lib.check(_lib.get().mfx_precond_apply(desc.dtype, n, self.rank, _lib.ptr(self.lt), _lib.ptr(minv),
                                                _lib.ptr(sdev), _lib.ptr(V), n, _lib.ptr(z), n, p, _lib.ptr(ws),
                                                ws.numel(), _lib.stream_ptr(V.device)))
        return z if v.dim() == 2 else z[0]

    def __call__(self, v, s):
        if torch.is_grad_enabled() and (v.requires_grad or (torch.is_tensor(s) and s.requires_grad)):
            s_t = s if torch.is_tensor(s) else torch.as_tensor(s, dtype=v.dtype, device=v.device)
            return _NoGrad.apply(self, v, s_t)
        return self._apply(v, s)

    def bind(self, s):
        """P = lambda v: pre(v, s) (util/gp_util.py:265) in a form the native PCG loop recognises."""
        return BoundPreconditioner(self, s)


class BoundPreconditioner:
    def __init__(self, pre, s):
        self.pre, self.s = pre, s

    def __call__(self, v):
        return self.pre(v, self.s)


def preconditioner(cholesky, /):
    """low_rank.py:10-60."""

    def solve_with_preconditioner(lazy_kernel, /, nrows: int):
        chol, info = cholesky(lazy_kernel, nrows)
        return Preconditioner(chol), info

    return solve_with_preconditioner
