"""Conjugate gradient solvers -- MI355X build.

Same functional surface as the reference's ``matfree_extensions/cg.py``:

    cg_fixed_step(num_matvecs)(A, b)                         pcg_fixed_step(num_matvecs)(A, b, P)
    cg_adaptive(atol=, rtol=, maxiter=, miniter=)(A, b)      pcg_adaptive(...)(A, b, P)
        -> (x, info)   info = {"residual_abs", "residual_rel"[, "num_steps"]}

with the whole iteration (cg.py:27-58, 85-135) executed by libmfx (``mfx_pcg_solve``).  ``A`` is a native operator
(``op.bind(*params)``) or any callable ``A(v)``; ``b`` may be a batch (p, n) of right-hand sides (each is solved
independently, adaptive stopping included).  ``P`` is ``low_rank.Preconditioner.bind(s)`` (or ``None``): arbitrary Python
preconditioners are not supported inside the HIP loop.  ``A`` may be an ``operators.RowShardedOp`` (bound): ``b``, ``x`` and the
residuals are then this rank's rows, ``P`` is the same (replicated) preconditioner object on every rank -- its rows are taken
here -- and the loop runs in ``mfx_pcg_solve_sharded`` (not for the re-orthogonalising variant).

Differentiation is the rule of ``jax.lax.custom_linear_solve(..., symmetric=True)`` (cg.py:23-25): the cotangent of the
right-hand side is another solve with the same solver, the cotangent of the operator's parameters is the parameter sweep
with (-lambda, x); nothing flows through the preconditioner or the info dict.  ``cg_fixed_step_reortho`` /
``pcg_fixed_step_reortho`` (cg.py:140-219; "needs more work" according to the reference's own test) are reproduced as they are.
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .low_rank import BoundPreconditioner
from .operators import CallbackOp, RowShardedOp, _PtrRegistry, as_operator


def cg_fixed_step(*args, **kwargs):
    pcg_solve = pcg_fixed_step(*args, **kwargs)

    def cg(A, b):
        return pcg_solve(A, b, None)

    return cg


def pcg_fixed_step(num_matvecs: int, /):
    cfg = {"maxiter": int(num_matvecs), "miniter": 0, "atol": 1.0, "rtol": 0.0, "adaptive": False}

    def pcg(A, b, P):
        x, r, _steps = _solve(A, b, P, cfg)
        return x, {"residual_abs": r, "residual_rel": r / x.detach().abs()}

    return pcg


def cg_fixed_step_reortho(*args, **kwargs):
    pcg_solve = pcg_fixed_step_reortho(*args, **kwargs)

    def cg(A, b):
        return pcg_solve(A, b, None)

    return cg


def pcg_fixed_step_reortho(num_matvecs: int, /):
    """cg.py:151-219: every step re-orthogonalises the residual against the stored normalised residuals Q."""
    cfg = {"maxiter": int(num_matvecs), "miniter": 0, "atol": 1.0, "rtol": 0.0, "adaptive": False, "reortho": True}

    def pcg(A, b, P):
        x, r, Q = _solve(A, b, P, cfg)
        return x, {"residual_abs": r, "Q": Q.transpose(-1, -2)}

    return pcg


def cg_adaptive(**kwargs):
    pcg_solve = pcg_adaptive(**kwargs)

    def cg(A, b):
        return pcg_solve(A, b, None)

    return cg


def pcg_adaptive(*, atol: float, rtol, maxiter: int, miniter: int = 0):
    """atol and rtol follow allclose logic (cg.py:73-74)."""
    cfg = {"maxiter": int(maxiter), "miniter": int(miniter), "atol": float(atol), "rtol": float(rtol), "adaptive": True}

    def pcg(A, b, P):
        x, r, steps = _solve(A, b, P, cfg)
        return x, {"residual_abs": r, "residual_rel": r / x.detach().abs(), "num_steps": steps}

    return pcg


def _solve(A, b, P, cfg):
    op, bound = as_operator(A)
    if isinstance(op, RowShardedOp) and cfg.get("reortho", False):
        raise NotImplementedError("the re-orthogonalising PCG variant is not row-sharded in the MI355X build")
    params = tuple(bound) if bound is not None else ()
    if P is not None and not isinstance(P, BoundPreconditioner):
        raise TypeError("P must be None or low_rank.Preconditioner.bind(s): the PCG loop runs inside libmfx and "
                        "cannot call back into an arbitrary Python preconditioner")
    batched = b.dim() == 2
    B = b if batched else b[None]
    cparams = op.constrain(*params)
    x, r, steps = _PcgFn.apply(op, cfg, P, B, *cparams)  # steps: int64 (p,), or the basis Q (p, m, n) when re-orthogonalising
    if not batched:
        return x[0], r[0], steps[0]
    return x, r, steps


def _run_sharded(sop, cfg, P, B, cparams):
    """One mfx_pcg_solve_sharded call: B, x, r are this rank's rows; steps replicated."""
    lib = _lib.get()
    op, comm = sop.op, sop.comm
    B = B.contiguous()
    p, nrows = B.shape
    n = comm.n
    if nrows != comm.nrows:
        raise ValueError(f"row-sharded operator: expected this rank's {comm.nrows} rows of the right-hand side, got {nrows}")
    dt, dev = B.dtype, B.device
    desc = op.descriptor(cparams, dt, n)
    rank, lt, minv, shift = 0, None, None, None
    if P is not None:
        pre = P.pre
        if pre.n != n or pre.lt.dtype != dt:
            raise ValueError(f"preconditioner of size {pre.n} ({pre.lt.dtype}) used for a system of size {n} ({dt})")
        rank = pre.rank
        lt = pre.lt[:, comm.row0 : comm.row0 + comm.nrows].contiguous()  # this rank's columns of L^T
        minv, shift = pre.minv(P.s)  # from the whole L: replicated
    cm0 = _lib.Comm()
    cm0.rank, cm0.world, cm0.nloc = comm.rank, comm.world, comm.nloc
    ws = _lib.scratch(int(lib.mfx_pcg_sharded_workspace_bytes(C.byref(desc), C.byref(cm0), n, p, rank)), dev)
    x = torch.empty_like(B)
    r = torch.empty_like(B)
    steps = torch.empty((p,), dtype=torch.int64, device=dev)
    cm, keep = comm.struct(ws, tensors=(), plans=sop.plans)
    with _lib.busy(ws):
        rc = lib.mfx_pcg_solve_sharded(C.byref(desc), C.byref(cm), _lib.ptr(B), nrows, n, p, _lib.ptr(lt), rank, _lib.ptr(minv),
                                       _lib.ptr(shift), cfg["maxiter"], cfg["miniter"], cfg["atol"], cfg["rtol"],
                                       int(cfg["adaptive"]), _lib.ptr(x), _lib.ptr(r), _lib.ptr(steps), _lib.ptr(ws), ws.numel(),
                                       _lib.stream_ptr(dev))
    if keep[2]:
        raise keep[2][0]
    _lib.check(rc)
    return x, r, steps


def _run(op, cfg, P, B, cparams):
    """One mfx_pcg_solve call on detached tensors -> (x, r, steps)."""
    if isinstance(op, RowShardedOp):
        return _run_sharded(op, cfg, P, B, cparams)
    lib = _lib.get()
    B = B.contiguous()
    p, n = B.shape
    dt, dev = B.dtype, B.device
    keep = None
    if isinstance(op, CallbackOp):
        reg = _PtrRegistry()
        desc, keep, _ = op.make(cparams, dt, n, reg, want_grads=False)
    else:
        desc = op.descriptor(cparams, dt, n)
    rank, lt, minv, shift = 0, None, None, None
    if P is not None:
        pre = P.pre
        if pre.n != n or pre.lt.dtype != dt:
            raise ValueError(f"preconditioner of size {pre.n} ({pre.lt.dtype}) used for a system of size {n} ({dt})")
        rank, lt = pre.rank, pre.lt
        minv, shift = pre.minv(P.s)
    reortho = cfg.get("reortho", False)
    ws = _lib.workspace_pcg(desc, n, p, max(rank, cfg["maxiter"]) if reortho else rank, dev)
    x = torch.empty_like(B)
    r = torch.empty_like(B)
    if keep is not None:
        reg.add_bytes(ws, dt)
    if reortho:
        m = cfg["maxiter"]
        if m < 1:
            raise ValueError("pcg_fixed_step_reortho needs num_matvecs >= 1")
        Q = torch.zeros((p, m, n), dtype=dt, device=dev)
        with _lib.busy(ws):
            rc = lib.mfx_pcg_solve_reortho(C.byref(desc), _lib.ptr(B), n, n, p, _lib.ptr(lt), rank, _lib.ptr(minv),
                                           _lib.ptr(shift), m, _lib.ptr(x), _lib.ptr(r), _lib.ptr(Q), _lib.ptr(ws), ws.numel(),
                                           _lib.stream_ptr(dev))
        if keep is not None and keep[1]:
            raise keep[1][0]
        _lib.check(rc)
        return x, r, Q
    steps = torch.empty((p,), dtype=torch.int64, device=dev)
    with _lib.busy(ws):
        rc = lib.mfx_pcg_solve(C.byref(desc), _lib.ptr(B), n, n, p, _lib.ptr(lt), rank, _lib.ptr(minv), _lib.ptr(shift),
                               cfg["maxiter"], cfg["miniter"], cfg["atol"], cfg["rtol"], int(cfg["adaptive"]),
                               _lib.ptr(x), _lib.ptr(r), _lib.ptr(steps), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    if keep is not None and keep[1]:
        raise keep[1][0]
    _lib.check(rc)
    return x, r, steps


class _PcgFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, op, cfg, P, B, *cparams):
        tensors = [q for q in cparams if torch.is_tensor(q)]
        _lib.require_device(B, *tensors)
        x, r, steps = _run(op, cfg, P, B, cparams)
        ctx.op, ctx.cfg, ctx.P = op, cfg, P
        ctx.nontensor = [None if torch.is_tensor(q) else q for q in cparams]
        ctx.save_for_backward(x, *tensors)
        ctx.mark_non_differentiable(r, steps)
        return x, r, steps

    @staticmethod
    def backward(ctx, dx, _dr, _dsteps):
        x, *tensors = ctx.saved_tensors
        it = iter(tensors)
        cparams = tuple(next(it) if q is None else q for q in ctx.nontensor)
        op = ctx.op
        lam, _r, _s = _run(op, ctx.cfg, ctx.P, dx, cparams)  # symmetric=True: the transpose solve is the same solve
        p, n = x.shape
        if isinstance(op, RowShardedOp):  # this rank's rows of -lambda against ALL entries of x; partial sums -> one all-reduce
            nat, comm = op.op, op.comm
            desc = nat.descriptor(cparams, x.dtype, comm.n)
            desc.row0, desc.nrows = comm.row0, comm.nrows
            gstruct, grads = nat.new_grads(*cparams)
            ws = _lib.workspace(desc, comm.n, 1, p, x.device)
            L = (-lam).contiguous()
            xfull = comm.gather_rows(x)
            _lib.check(_lib.get().mfx_op_vjp_params(C.byref(desc), _lib.ptr(L), comm.nrows, _lib.ptr(xfull), comm.n, p,
                                                    C.byref(gstruct), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device)))
            if comm.world > 1 and grads:
                flat = torch.cat([g.reshape(-1) for g in grads])
                comm.all_reduce_(flat)
                off = 0
                for g in grads:
                    g.copy_(flat[off : off + g.numel()].reshape(g.shape))
                    off += g.numel()
        elif isinstance(op, CallbackOp):
            grads = []
            diff_idx = [i for i, q in enumerate(cparams) if torch.is_tensor(q) and q.is_floating_point()]
            acc = {i: torch.zeros_like(cparams[i]) for i in diff_idx}
            for bidx in range(p):
                with torch.enable_grad():
                    live = [q.detach().requires_grad_(True) if i in acc else q for i, q in enumerate(cparams)]
                    diff = [live[i] for i in diff_idx]
                    if diff:
                        out = op.fn(x[bidx].detach(), *live)
                        gs = torch.autograd.grad(out, diff, -lam[bidx], allow_unused=True)
                        for i, g in zip(diff_idx, gs):
                            if g is not None:
                                acc[i].add_(g)
            grads = [acc.get(i) for i in range(len(cparams))]
        else:
            desc = op.descriptor(cparams, x.dtype, n)
            gstruct, grads = op.new_grads(*cparams)
            ws = _lib.workspace(desc, n, 1, p, x.device)
            L = (-lam).contiguous()
            _lib.check(_lib.get().mfx_op_vjp_params(C.byref(desc), _lib.ptr(L), n, _lib.ptr(x), n, p, C.byref(gstruct),
                                                    _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device)))
        return (None, None, None, lam, *grads)
