"""Arnoldi (Hessenberg) factorisation with its custom adjoint -- MI355X build.

Same functional surface as the reference's ``matfree_extensions/arnoldi.py``:

    hessenberg(matvec, krylov_depth, /, *, reortho, custom_vjp=True, reortho_vjp="match")
        -> estimate(v, *params) -> (Q (n, k), H (k, k), r (n,), c ())

but on ``torch`` tensors that live on a ROCm device, with the forward loop (arnoldi.py:57-101) and the
adjoint scan (arnoldi.py:104-220) executed by libmfx's HIP kernels (``mfx_arnoldi_forward`` /
``mfx_arnoldi_adjoint``).  ``v`` may also be a batch (p, n): outputs gain a leading probe axis (this
replaces ``jax.vmap`` over probes, hutchinson.py:14).

Differences from the reference, all deliberate:
  * parameters must be explicit tensors (no closure conversion, arnoldi.py:22): pass a native
    operator (``operators.DenseOp`` ...) or any callable ``matvec(v, *params)``;
  * complex start vectors (tests/test_arnoldi/test_hessenberg_forward.py:10-37) run the FORWARD only
    (``mfx_arnoldi_forward_complex``): a complex Krylov vector takes two real basis slots (q, i q), the operator acts on
    interleaved (re, im) reals -- a complex ``DenseOp`` as its (2n, 2n) real form on the native dense kernel, any other
    callable through the callback operator.  The adjoint is real float32/float64 only (the reference's is not exercised
    on complex input either);
  * ``custom_vjp=False`` is the reference's baseline, back-propagation THROUGH the loop: the recurrence then runs as
    differentiable torch ops around the HIP operator (``_autodiff.py``) -- slow and memory-hungry by construction.
Reference quirk Q1 is reproduced: the forward pass re-orthogonalises unless ``reortho_vjp="none"``
(arnoldi.py:26,91), whatever ``reortho`` says; ``reortho`` only selects the adjoint's re-projection.
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .operators import CallbackOp, DenseOp, RowShardedOp, _PtrRegistry, as_operator


def hessenberg(
    matvec,
    krylov_depth,
    /,
    *,
    reortho: str,
    custom_vjp: bool = True,
    reortho_vjp: str = "match",
):
    reortho_expected = ["none", "full"]
    if reortho not in reortho_expected or not isinstance(reortho, str):
        msg = f"Unexpected input for {reortho}: either of {reortho_expected} expected."
        raise TypeError(msg)
    op, bound = as_operator(matvec)

    def estimate(v, *params):
        if bound is not None:
            params = tuple(bound) + tuple(params)
        batched = v.dim() == 2
        V = v if batched else v[None]
        sharded = isinstance(op, RowShardedOp)
        if V.is_complex():
            if sharded:
                raise NotImplementedError("complex Arnoldi on a row-sharded operator")
            if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in (V, *params)):
                raise NotImplementedError("complex Arnoldi is forward only: detach the inputs (the adjoint system is real)")
            second_pass = (reortho_vjp if reortho_vjp != "match" else reortho_vjp) != "none"  # Q1, as below
            Qkn, H, r, c = _complex_forward(op, int(krylov_depth), second_pass, V, params)
            Q = Qkn.transpose(-1, -2)
            return (Q, H, r, c) if batched else (Q[0], H[0], r[0], c[0])
        n = op.comm.n if sharded else V.shape[-1]
        if sharded and V.shape[-1] != op.comm.nrows:
            raise ValueError(f"row-sharded operator: expected this rank's {op.comm.nrows} rows of the start vector, got {V.shape[-1]}")
        if krylov_depth < 1 or krylov_depth > n:
            msg = f"Parameter depth {krylov_depth} is outside the expected range"
            raise ValueError(msg)
        # Q1 (arnoldi.py:26): the forward always sees `reortho_vjp`
        reortho_fwd = reortho_vjp if reortho_vjp != "match" else reortho_vjp
        second_pass = reortho_fwd != "none"
        wants_grad = torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in (V, *params))
        if not custom_vjp and not sharded and wants_grad:  # autodiff THROUGH the loop: the reference's baseline (_autodiff.py)
            from . import _autodiff

            Qb, H, r, c = _autodiff.batched(_autodiff.arnoldi_forward, V, op, int(krylov_depth), params, second_pass=second_pass)
            return (Qb, H, r, c) if batched else (Qb[0], H[0], r[0], c[0])
        cparams = op.constrain(*params)
        if sharded:
            Qkn, H, r, c = _ArnoldiShardedFn.apply(op, int(krylov_depth), second_pass, reortho, custom_vjp, V, *cparams)
        else:
            Qkn, H, r, c = _ArnoldiFn.apply(op, int(krylov_depth), second_pass, reortho, custom_vjp, V, *cparams)
        Q = Qkn.transpose(-1, -2)  # reference layout (n, k); storage stays (k, n)
        if not batched:
            return Q[0], H[0], r[0], c[0]
        return Q, H, r, c

    return estimate


def _complex_forward(op, k, second_pass, V, params):
    """Complex forward (arnoldi.py:57-101 with its conjugations): the operator in its real form on interleaved vectors."""
    lib = _lib.get()
    V = V.contiguous()
    p, n = V.shape
    if k < 1 or k > n:
        raise ValueError(f"Parameter depth {k} is outside the expected range")
    cdt, dev = V.dtype, V.device
    rdt = torch.float32 if cdt == torch.complex64 else torch.float64
    _lib.require_device(V)
    keep = None
    if isinstance(op, DenseOp):  # native dense kernel on the (2n, 2n) real form [[Re, -Im], [Im, Re]] per entry
        (A,) = op.constrain(*params)
        A = A.to(cdt)
        M = torch.stack([torch.stack([A.real, -A.imag], -1), torch.stack([A.imag, A.real], -1)], 1).reshape(2 * n, 2 * n)
        desc = op.descriptor((M.contiguous(),), rdt, 2 * n)
        hold = M
    else:
        fn = op.fn if isinstance(op, CallbackOp) else op

        def real_form(x, *ps):  # x: (2n,) reals = n interleaved complex entries
            y = fn(torch.view_as_complex(x.view(n, 2)), *ps)
            return torch.view_as_real(y.to(cdt).contiguous()).reshape(2 * n)

        reg = _PtrRegistry()
        desc, keep, _ = CallbackOp(real_form).make(tuple(params), rdt, 2 * n, reg, want_grads=False)
        hold = None
    Q = torch.empty((p, k, n), dtype=cdt, device=dev)
    H = torch.empty((p, k, k), dtype=cdt, device=dev)
    r = torch.empty((p, n), dtype=cdt, device=dev)
    c = torch.empty((p,), dtype=cdt, device=dev)
    ws = _lib.scratch(int(lib.mfx_complex_workspace_bytes(C.byref(desc), n, k, p)), dev)
    if keep is not None:
        for t in (V, Q, r):
            reg.add(torch.view_as_real(t))
        reg.add_bytes(ws, rdt)
    with _lib.busy(ws):
        rc = lib.mfx_arnoldi_forward_complex(C.byref(desc), _lib.ptr(V), n, k, p, int(second_pass), _lib.ptr(Q), _lib.ptr(H),
                                             _lib.ptr(r), _lib.ptr(c), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    if keep is not None and keep[1]:
        raise keep[1][0]
    _lib.check(rc)
    del hold
    return Q, H, r, c


class _ArnoldiFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, op, k, second_pass, reortho_bwd, differentiable, V, *cparams):
        tensors = [q for q in cparams if torch.is_tensor(q)]
        _lib.require_device(V, *tensors)
        lib = _lib.get()
        V = V.contiguous()
        p, n = V.shape
        dt, dev = V.dtype, V.device
        Q = torch.empty((p, k, n), dtype=dt, device=dev)
        H = torch.empty((p, k, k), dtype=dt, device=dev)
        r = torch.empty((p, n), dtype=dt, device=dev)
        c = torch.empty((p,), dtype=dt, device=dev)
        keep = None
        if isinstance(op, CallbackOp):
            reg = _PtrRegistry()
            desc, keep, _ = op.make(cparams, dt, n, reg, want_grads=False)
        else:
            desc = op.descriptor(cparams, dt, n)
        ws = _lib.workspace(desc, n, k, p, dev)
        if keep is not None:
            for t in (V, Q, r):
                reg.add(t)
            reg.add_bytes(ws, dt)
        with _lib.busy(ws):
            rc = lib.mfx_arnoldi_forward(C.byref(desc), _lib.ptr(V), n, k, p, int(second_pass), _lib.ptr(Q),
                                         _lib.ptr(H), _lib.ptr(r), _lib.ptr(c), _lib.ptr(ws), ws.numel(),
                                         _lib.stream_ptr(dev))
        if keep is not None and keep[1]:
            raise keep[1][0]
        _lib.check(rc)
        ctx.op, ctx.k, ctx.reortho, ctx.differentiable = op, k, reortho_bwd, differentiable
        ctx.nparams = len(cparams)
        ctx.nontensor = [None if torch.is_tensor(q) else q for q in cparams]
        ctx.save_for_backward(Q, H, r, c, *tensors)
        ctx.set_materialize_grads(False)
        return Q, H, r, c

    @staticmethod
    def backward(ctx, dQ, dH, dr, dc):
        if not ctx.differentiable:
            raise RuntimeError("hessenberg(custom_vjp=False) is not differentiable in the MI355X build; "
                               "use custom_vjp=True (the adjoint system).")
        Q, H, r, c, *tensors = ctx.saved_tensors
        it = iter(tensors)
        cparams = tuple(next(it) if q is None else q for q in ctx.nontensor)
        op, k, lib = ctx.op, ctx.k, _lib.get()
        p, _, n = Q.shape
        dt, dev = Q.dtype, Q.device
        dQ = None if dQ is None else dQ.contiguous()
        dr = None if dr is None else dr.contiguous()
        dc = None if dc is None else dc.contiguous()
        dH = torch.zeros_like(H) if dH is None else dH.contiguous()
        dv = torch.empty((p, n), dtype=dt, device=dev)
        Lam = torch.empty((p, k, n), dtype=dt, device=dev)
        keep = None
        if isinstance(op, CallbackOp):
            reg = _PtrRegistry()
            desc, keep, grads = op.make(cparams, dt, n, reg, want_grads=True)
            gptr = None
        else:
            desc = op.descriptor(cparams, dt, n)
            gstruct, grads = op.new_grads(*cparams)
            gptr = C.byref(gstruct)
        ws = _lib.workspace(desc, n, k, p, dev)
        if keep is not None:
            for t in (Q, r, Lam, dv, dQ, dr):
                reg.add(t)
            reg.add_bytes(ws, dt)
        with _lib.busy(ws):
            rc = lib.mfx_arnoldi_adjoint(C.byref(desc), n, k, p, _lib.ptr(Q), _lib.ptr(H), _lib.ptr(r), _lib.ptr(c),
                                         _lib.ptr(dQ), _lib.ptr(dH), _lib.ptr(dr), _lib.ptr(dc),
                                         _lib.REORTHO_FULL if ctx.reortho == "full" else _lib.REORTHO_NONE,
                                         _lib.ptr(dv), _lib.ptr(Lam), gptr, _lib.ptr(ws), ws.numel(),
                                         _lib.stream_ptr(dev))
        if keep is not None and keep[1]:
            raise keep[1][0]
        _lib.check(rc)
        return (None, None, None, None, None, dv, *grads)


class _ArnoldiShardedFn(torch.autograd.Function):
    """_ArnoldiFn on row shards (``mfx_arnoldi_forward_sharded`` / ``mfx_arnoldi_adjoint_sharded``): V, Q, r, dv hold this
    rank's rows; H and c are replicated; the parameter gradients are summed over the row group before they are returned."""

    @staticmethod
    def forward(ctx, sop, k, second_pass, reortho_bwd, differentiable, V, *cparams):
        op, comm = sop.op, sop.comm
        tensors = [q for q in cparams if torch.is_tensor(q)]
        _lib.require_device(V, *tensors)
        lib = _lib.get()
        V = V.contiguous()
        p, nrows = V.shape
        n = comm.n
        dt, dev = V.dtype, V.device
        Q = torch.empty((p, k, nrows), dtype=dt, device=dev)
        Qfull = torch.empty((p, k, n), dtype=dt, device=dev)
        H = torch.empty((p, k, k), dtype=dt, device=dev)
        r = torch.empty((p, nrows), dtype=dt, device=dev)
        c = torch.empty((p,), dtype=dt, device=dev)
        desc = op.descriptor(cparams, dt, n)
        cm0 = _lib.Comm()
        cm0.rank, cm0.world, cm0.nloc = comm.rank, comm.world, comm.nloc
        ws = _lib.scratch(int(lib.mfx_sharded_workspace_bytes(C.byref(desc), C.byref(cm0), n, k, p)), dev)
        cm, keep = comm.struct(ws, tensors=(Q, Qfull), plans=sop.plans)
        with _lib.busy(ws):
            rc = lib.mfx_arnoldi_forward_sharded(C.byref(desc), C.byref(cm), _lib.ptr(V), n, k, p, int(second_pass),
                                                 _lib.ptr(Q), _lib.ptr(Qfull), _lib.ptr(H), _lib.ptr(r), _lib.ptr(c),
                                                 _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        if keep[2]:
            raise keep[2][0]
        _lib.check(rc)
        ctx.sop, ctx.k, ctx.reortho, ctx.differentiable = sop, k, reortho_bwd, differentiable
        ctx.nontensor = [None if torch.is_tensor(q) else q for q in cparams]
        ctx.save_for_backward(Q, Qfull, H, r, c, *tensors)
        ctx.set_materialize_grads(False)
        return Q, H, r, c

    @staticmethod
    def backward(ctx, dQ, dH, dr, dc):
        if not ctx.differentiable:
            raise RuntimeError("hessenberg(custom_vjp=False) is not differentiable in the MI355X build; "
                               "use custom_vjp=True (the adjoint system).")
        Q, Qfull, H, r, c, *tensors = ctx.saved_tensors
        it = iter(tensors)
        cparams = tuple(next(it) if q is None else q for q in ctx.nontensor)
        op, comm, k, lib = ctx.sop.op, ctx.sop.comm, ctx.k, _lib.get()
        p, _, nrows = Q.shape
        n = comm.n
        dt, dev = Q.dtype, Q.device
        dQ = None if dQ is None else dQ.contiguous()
        dr = None if dr is None else dr.contiguous()
        dc = None if dc is None else dc.contiguous()
        dH = torch.zeros_like(H) if dH is None else dH.contiguous()
        dv = torch.empty((p, nrows), dtype=dt, device=dev)
        Lam = torch.empty((p, k, nrows), dtype=dt, device=dev)
        desc = op.descriptor(cparams, dt, n)
        gstruct, grads = op.new_grads(*cparams)
        cm0 = _lib.Comm()
        cm0.rank, cm0.world, cm0.nloc = comm.rank, comm.world, comm.nloc
        ws = _lib.scratch(int(lib.mfx_sharded_workspace_bytes(C.byref(desc), C.byref(cm0), n, k, p)), dev)
        cm, keep = comm.struct(ws, tensors=(Lam,), plans=ctx.sop.plans)
        with _lib.busy(ws):
            rc = lib.mfx_arnoldi_adjoint_sharded(C.byref(desc), C.byref(cm), n, k, p, _lib.ptr(Q), _lib.ptr(Qfull), _lib.ptr(H),
                                                 _lib.ptr(r), _lib.ptr(c), _lib.ptr(dQ), _lib.ptr(dH), _lib.ptr(dr), _lib.ptr(dc),
                                                 _lib.REORTHO_FULL if ctx.reortho == "full" else _lib.REORTHO_NONE,
                                                 _lib.ptr(dv), _lib.ptr(Lam), C.byref(gstruct), _lib.ptr(ws), ws.numel(),
                                                 _lib.stream_ptr(dev))
        if keep[2]:
            raise keep[2][0]
        _lib.check(rc)
        if comm.world > 1 and grads:  # partial sums over this rank's rows -> the complete gradient, ONE small all-reduce
            flat = torch.cat([g.reshape(-1) for g in grads])
            comm.all_reduce_(flat)
            off = 0
            for g in grads:
                g.copy_(flat[off : off + g.numel()].reshape(g.shape))
                off += g.numel()
        return (None, None, None, None, None, dv, *grads)
