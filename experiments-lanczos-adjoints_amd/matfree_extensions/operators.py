"""Native operator objects: the explicit-parameter replacement for the reference's matvec closures.

The reference passes ``matvec(v, *params)`` callables and relies on ``jax.closure_convert`` to make
captured hyper-parameters differentiable (arnoldi.py:21-23).  Here a *native operator* is a small
object that (i) is itself callable as ``op(v, *params)`` -- so it can be handed to
``lanczos.tridiag`` / ``arnoldi.hessenberg`` / ``lanczos.integrand_spd`` wherever the reference takes
a matvec -- and (ii) knows how to describe itself to libmfx (``struct mfx_operator``) so that the
whole Krylov loop, the transposed matvec and the parameter-gradient sweep run as HIP kernels.

  DenseOp()                         params = (A,)                 tests/test_lanczos/test_tridiag_forward.py:18
  CsrOp(crow, col, n)               params = (values,)            experiments/benchmarks/.../suite_sparse/benchmark.py:64-68
  RbfGramOp(X, noise_minval=...)    params = (raw_lengthscale, raw_outputscale, raw_noise)
                                                                  util/gp_util.py:151-201,225-226,525-549
  CallbackOp(fn)                    any Python ``fn(v, *params)`` (torch ops); per-step host control

``op.bind(*params)`` freezes the parameters into a zero-argument-parameter matvec (what the GP code
hands to ``krylov_logdet_slq``, util/gp_util.py:555-557) while keeping them differentiable.
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _check_dtype(desc, t, what):
    """The descriptor's dtype is the dtype of the VECTORS; operator data of another dtype would be misread by the kernels."""
    if _lib.dtype_code(t.dtype) != desc.dtype:
        want = "float32" if desc.dtype == _lib.MFX_F32 else "float64"
        raise TypeError(f"{what} must have the dtype of the vectors it is applied to ({want}), got {t.dtype}")


class NativeOp:
    kind = None

    # ---- differentiable map from user parameters to the tensors the kernels consume -------------
    def constrain(self, *params):
        return params

    def fill(self, desc: _lib.Operator, *cparams):
        raise NotImplementedError

    def new_grads(self, *cparams):
        """-> (OpGrads struct, tuple of zero-initialised gradient tensors aligned with cparams)."""
        raise NotImplementedError

    def descriptor(self, cparams, dtype, n):
        desc = _lib.Operator()
        desc.kind = self.kind
        desc.dtype = _lib.dtype_code(dtype)
        desc.n = n
        self.fill(desc, *cparams)
        return desc

    def size(self, *cparams):
        raise NotImplementedError

    # ---- op(v, *params): plain matvec through mfx_op_apply ---------------------------------------
    def __call__(self, v, *params):
        return _ApplyFn.apply(self, False, v, *self.constrain(*params))

    def bind(self, *params):
        return BoundOp(self, params)


class BoundOp:
    """An operator with its parameters attached: ``bound(v)`` is the matvec, and the Krylov drivers
    recover ``(op, params)`` to keep the parameters differentiable (explicit closure conversion)."""

    def __init__(self, op, params):
        self.op, self.params = op, tuple(params)

    def __call__(self, v):
        return self.op(v, *self.params)


class RowShardedOp:
    """A native operator whose rows -- and the rows of every Krylov vector -- are sharded over the ranks of a
    ``distributed.RowComm``.  Hand it to ``lanczos.tridiag(reortho="full")`` / ``arnoldi.hessenberg`` /
    ``lanczos.integrand_spd`` in place of the operator: vectors are then this rank's row shards (p, nrows), H and the
    SLQ values come out replicated, parameter gradients complete (summed over the row group)."""

    def __init__(self, op, comm, exchange="auto"):
        if not isinstance(op, NativeOp):
            raise TypeError("row sharding needs a native operator (DenseOp, CsrOp, RbfGramOp)")
        self.op, self.comm = op, comm
        # sparse operators: neighbour exchange of the entries this rank's rows read (the halo of a stencil) instead of
        # all-gathering the whole iterate; plans for A and A^T (collective: every rank of the row group builds them here)
        self.plans = None
        if exchange == "auto":
            exchange = isinstance(op, CsrOp) and comm.world > 1
        if exchange:
            if not isinstance(op, CsrOp):
                raise TypeError("the neighbour exchange needs a sparse (CSR) operator")
            self.plans = (comm.plan_exchange(op.crow, op.col), comm.plan_exchange(op.t_crow, op.t_col))

    def bind(self, *params):
        return BoundOp(self, params)

    def constrain(self, *params):
        return self.op.constrain(*params)

    def __call__(self, v, *params):
        """this rank's rows of A v from the row shard of v (gathers v first; not differentiable -- the Krylov drivers
        use the fused device path instead)"""
        with torch.no_grad():
            cparams = self.op.constrain(*params)
            V = v if v.dim() == 2 else v[None]
            full = self.comm.gather_rows(V.contiguous())
            desc = self.op.descriptor(cparams, V.dtype, self.comm.n)
            desc.row0, desc.nrows = self.comm.row0, self.comm.nrows
            ws = _lib.workspace(desc, self.comm.n, 1, V.shape[0], V.device)
            y = torch.empty_like(V)
            _lib.check(_lib.get().mfx_op_apply(C.byref(desc), _lib.ptr(full), self.comm.n, _lib.ptr(y), self.comm.nrows,
                                               V.shape[0], 0, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(V.device)))
        return y if v.dim() == 2 else y[0]


class _ApplyFn(torch.autograd.Function):
    """y = A(theta) x (or A^T x); backward = A^T dy and the parameter sweep with batch = p."""

    @staticmethod
    def forward(ctx, op, transpose, v, *cparams):
        _lib.require_device(v, *cparams)
        lib = _lib.get()
        V = (v if v.dim() == 2 else v[None]).contiguous()
        p, n = V.shape
        desc = op.descriptor(cparams, V.dtype, n)
        ws = _lib.workspace(desc, n, 1, p, V.device)
        y = torch.empty_like(V)
        _lib.check(
            lib.mfx_op_apply(C.byref(desc), _lib.ptr(V), n, _lib.ptr(y), n, p, int(transpose),
                             _lib.ptr(ws), ws.numel(), _lib.stream_ptr(V.device))
        )
        ctx.op, ctx.transpose, ctx.vdim = op, transpose, v.dim()
        ctx.save_for_backward(V, *cparams)
        return y if v.dim() == 2 else y[0]

    @staticmethod
    def backward(ctx, dy):
        V, *cparams = ctx.saved_tensors
        op, lib = ctx.op, _lib.get()
        DY = (dy if dy.dim() == 2 else dy[None]).contiguous()
        p, n = V.shape
        desc = op.descriptor(cparams, V.dtype, n)
        ws = _lib.workspace(desc, n, 1, p, V.device)
        dv = torch.empty_like(V)
        _lib.check(
            lib.mfx_op_apply(C.byref(desc), _lib.ptr(DY), n, _lib.ptr(dv), n, p, int(not ctx.transpose),
                             _lib.ptr(ws), ws.numel(), _lib.stream_ptr(V.device))
        )
        gstruct, grads = op.new_grads(*cparams)
        L, R = (DY, V) if not ctx.transpose else (V, DY)
        _lib.check(
            lib.mfx_op_vjp_params(C.byref(desc), _lib.ptr(L), n, _lib.ptr(R), n, p, C.byref(gstruct),
                                  _lib.ptr(ws), ws.numel(), _lib.stream_ptr(V.device))
        )
        return (None, None, dv if ctx.vdim == 2 else dv[0], *grads)


class DenseOp(NativeOp):
    """matvec(v, A) = A @ v."""

    kind = _lib.OP_DENSE

    def constrain(self, A):
        return (A.contiguous(),)

    def size(self, A):
        return A.shape[0]

    def fill(self, desc, A):
        _check_dtype(desc, A, "DenseOp: the matrix")
        if A.dim() != 2 or A.shape[0] != A.shape[1]:
            raise ValueError(f"DenseOp expects a square matrix, got {tuple(A.shape)}")
        desc.dense_a = A.data_ptr()
        desc.lda = A.stride(0)

    def new_grads(self, A):
        g = torch.zeros_like(A)
        s = _lib.OpGrads()
        s.dense_a = g.data_ptr()
        return s, (g,)


class CsrOp(NativeOp):
    """matvec(v, vals) = CSR(vals; crow, col) @ v; differentiable w.r.t. all stored values."""

    kind = _lib.OP_CSR

    def __init__(self, crow, col, n, device=None):
        device = device if device is not None else crow.device
        self.n = int(n)
        self.crow = crow.to(device=device, dtype=torch.int32).contiguous()
        self.col = col.to(device=device, dtype=torch.int32).contiguous()
        counts = (self.crow[1:] - self.crow[:-1]).to(torch.int64)
        self.row = torch.repeat_interleave(torch.arange(self.n, device=device, dtype=torch.int32), counts)
        self.nnz = int(self.col.numel())
        # CSR structure of A^T (for the Arnoldi adjoint's A^T lambda): stable sort of entries by column
        key = self.col.to(torch.int64) * self.n + self.row.to(torch.int64)
        perm = torch.argsort(key, stable=True)
        self.t_perm = perm.to(torch.int32).contiguous()
        self.t_col = self.row[perm].contiguous()
        tcounts = torch.bincount(self.col.to(torch.int64), minlength=self.n)
        self.t_crow = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(tcounts, 0)]).to(torch.int32)
        # longest row of A and of A^T (one host read at construction): decides fused-step vs 8-lanes-per-row kernels in libmfx
        self.max_row_nnz = int(max(int(counts.max()) if self.nnz else 0, int(tcounts.max()) if self.nnz else 0))

    @classmethod
    def from_coo(cls, row, col, vals, n, device):
        """COO (e.g. scipy.io.mmread's symmetric expansion, util/exp_util.py:35-42) -> (op, values)."""
        row = torch.as_tensor(row, dtype=torch.int64)
        col = torch.as_tensor(col, dtype=torch.int64)
        order = torch.argsort(row * n + col, stable=True)
        row, col = row[order], col[order]
        crow = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(torch.bincount(row, minlength=n), 0)])
        op = cls(crow.to(device), col.to(device), n, device=device)
        return op, torch.as_tensor(vals)[order].to(device), order

    def constrain(self, vals):
        return (vals.contiguous(),)

    def size(self, vals):
        return self.n

    def fill(self, desc, vals):
        _check_dtype(desc, vals, "CsrOp: the stored values")
        if vals.numel() != self.nnz:
            raise ValueError(f"CsrOp expects {self.nnz} values, got {vals.numel()}")
        desc.crow, desc.col, desc.row = self.crow.data_ptr(), self.col.data_ptr(), self.row.data_ptr()
        desc.val, desc.nnz, desc.max_row_nnz = vals.data_ptr(), self.nnz, self.max_row_nnz
        desc.t_crow, desc.t_col, desc.t_perm = self.t_crow.data_ptr(), self.t_col.data_ptr(), self.t_perm.data_ptr()

    def new_grads(self, vals):
        g = torch.zeros_like(vals)
        s = _lib.OpGrads()
        s.val = g.data_ptr()
        return s, (g,)


def softplus(x, beta=1.0, threshold=20.0):
    """torch-style thresholded softplus, as util/gp_util.py:187-201 mirrors it."""
    return torch.nn.functional.softplus(x, beta=beta, threshold=threshold)


class RbfGramOp(NativeOp):
    """(K(X, X) + noise I) v with the reference's scaled-kernel parametrisation, matrix-free.

    kernel: "rbf" (kernel_scaled_rbf, util/gp_util.py:151-184), "matern32" (:69-107), "matern12" (:110-148).

    params = (raw_lengthscale [() or (d,)], raw_outputscale (), raw_noise ());
    lengthscale = softplus(raw_l), outputscale = softplus(raw_s)         (util/gp_util.py:164-165)
    noise = noise_minval + softplus(raw_noise)                            (util/gp_util.py:187-201,222)

    precision: arithmetic of the fp32 Gram kernels for wide probe batches --
      "f16x3" (default)  matvec AND parameter-gradient GEMM emulated on the f16 matrix pipe (hi/lo split,
                         3 products, fp32 accumulate): 2.1x faster end to end, accuracy on par with "fp32";
      "f16x3-matvec"     split matvec, exact fp32 gradient GEMM: the most accurate mode against fp64;
      "fp32"             exact fp32 MFMA everywhere.
    fp64 operators ignore it.
    """

    kind = _lib.OP_RBF
    _MODES = {"fp32": _lib.RBF_FP32, "f16x3-matvec": _lib.RBF_F16X3_MATVEC, "f16x3": _lib.RBF_F16X3}

    _KERNELS = {"rbf": _lib.KERNEL_RBF, "matern12": _lib.KERNEL_MATERN12, "matern32": _lib.KERNEL_MATERN32}

    def __init__(self, X, noise_minval=0.0, precision="f16x3", kernel="rbf"):
        if X.dim() != 2:
            raise ValueError("RbfGramOp expects inputs of shape (n, d)")
        if precision not in self._MODES:
            raise ValueError(f"precision must be one of {sorted(self._MODES)}")
        if kernel not in self._KERNELS:
            raise ValueError(f"kernel must be one of {sorted(self._KERNELS)}")
        self.kernel = kernel
        self.X = X.contiguous()
        self.n, self.d = X.shape
        self.noise_minval = noise_minval
        self.precision = precision

    def constrain(self, raw_lengthscale, raw_outputscale, raw_noise):
        dt = self.X.dtype
        ls = softplus(raw_lengthscale).to(dt).reshape(-1)
        if ls.numel() not in (1, self.d):
            raise ValueError(f"raw_lengthscale must have shape () or ({self.d},)")
        s = softplus(raw_outputscale).to(dt).reshape(1)
        nz = (self.noise_minval + softplus(raw_noise)).to(dt).reshape(1)
        return (ls.contiguous(), s, nz)

    def size(self, *_):
        return self.n

    def fill(self, desc, ls, s, nz):
        if self.X.dtype != ls.dtype:
            raise TypeError("RbfGramOp: X and the hyper-parameters must share a dtype")
        _check_dtype(desc, self.X, "RbfGramOp: the inputs X")
        desc.x, desc.d = self.X.data_ptr(), self.d
        desc.ard = int(ls.numel() == self.d)
        desc.rbf_mode = self._MODES[self.precision]
        desc.kernel_fn = self._KERNELS[self.kernel]
        desc.lengthscale, desc.outputscale, desc.noise = ls.data_ptr(), s.data_ptr(), nz.data_ptr()

    def cross_apply(self, xnew, v, *params):
        """K(xnew, X) v (no noise term): the prior cross-covariance matvec of the posterior mean
        (util/gp_util.py:299-305).  xnew (m, d), v (n,) or (p, n) -> (m,) or (p, m).  Not differentiable."""
        with torch.no_grad():
            cparams = self.constrain(*params)
            _lib.require_device(xnew, v, *cparams)
            V = (v if v.dim() == 2 else v[None]).contiguous()
            xnew = xnew.to(self.X.dtype).contiguous()
            if xnew.dim() != 2 or xnew.shape[1] != self.d or V.shape[1] != self.n:
                raise ValueError(f"cross_apply: xnew {tuple(xnew.shape)}, v {tuple(v.shape)} do not match X {tuple(self.X.shape)}")
            m, p = xnew.shape[0], V.shape[0]
            desc = self.descriptor(cparams, V.dtype, self.n)
            lib = _lib.get()
            ws = _lib.scratch(int(lib.mfx_gram_cross_workspace_bytes(C.byref(desc), m)), V.device)
            y = torch.empty((p, m), dtype=V.dtype, device=V.device)
            _lib.check(lib.mfx_gram_cross_apply(C.byref(desc), _lib.ptr(xnew), m, _lib.ptr(V), self.n, _lib.ptr(y), m, p,
                                                _lib.ptr(ws), ws.numel(), _lib.stream_ptr(V.device)))
        return y if v.dim() == 2 else y[0]

    def new_grads(self, ls, s, nz):
        g = (torch.zeros_like(ls), torch.zeros_like(s), torch.zeros_like(nz))
        st = _lib.OpGrads()
        st.lengthscale, st.outputscale, st.noise = (t.data_ptr() for t in g)
        return st, g


class _PtrRegistry:
    """Maps raw device pointers handed to a callback back to torch views (no copies)."""

    def __init__(self):
        self.bufs = []

    def add(self, t):
        if t is not None:
            if not t.is_contiguous():
                raise ValueError("registered buffers must be contiguous")
            self.bufs.append((t.data_ptr(), t.numel() * t.element_size(), t.view(-1)))

    def add_bytes(self, t_u8, dtype):
        es = torch.empty((), dtype=dtype).element_size()
        usable = (t_u8.numel() // es) * es
        self.bufs.append((t_u8.data_ptr(), usable, t_u8[:usable].view(dtype)))

    def view(self, ptr, ld, p, n):
        for base, nbytes, flat in self.bufs:
            if base <= ptr < base + nbytes:
                off = (ptr - base) // flat.element_size()
                return flat[off:].as_strided((p, n), (ld, 1))
        raise RuntimeError("libmfx callback received an unknown pointer")


class CallbackOp:
    """Generic Python matvec ``fn(v, *params)`` (torch ops on device tensors).

    libmfx still drives the k-loop and runs every Krylov vector kernel; only the matvec (and, in the
    adjoint, its VJP through torch.autograd -- the analogue of jax.vjp at arnoldi.py:207-208 /
    lanczos.py:328-329) is delegated.  ``fn`` acts on ONE vector; probes are looped.
    """

    kind = _lib.OP_CALLBACK

    def __init__(self, fn):
        self.fn = fn

    def constrain(self, *params):
        return params

    def __call__(self, v, *params):
        return self.fn(v, *params)

    def bind(self, *params):
        return BoundOp(self, params)

    def make(self, params, dtype, n, registry, want_grads):
        """-> (descriptor, keepalive, grad accumulators or None)."""
        params = tuple(params)
        accum = [torch.zeros_like(q) if (want_grads and torch.is_tensor(q) and q.is_floating_point()) else None for q in params]
        fn = self.fn
        failure = []

        def trampoline(_ctx, mode, xp, ldx, auxp, ldaux, yp, ldy, p, nn, _stream):
            try:
                X = registry.view(xp, ldx, p, nn)
                Y = registry.view(yp, ldy, p, nn)
                if mode == 0:
                    with torch.no_grad():
                        for b in range(p):
                            Y[b].copy_(fn(X[b], *params))
                    return 0
                AUX = registry.view(auxp, ldaux, p, nn)
                for b in range(p):
                    with torch.enable_grad():
                        live = [q.detach().requires_grad_(True) if a is not None else q for q, a in zip(params, accum)]
                        diff = [q for q, a in zip(live, accum) if a is not None]
                        if mode == 1:  # y = A^T x ; d/dtheta x^T A(theta) aux      (arnoldi.py:207-209)
                            u = AUX[b].detach().clone().requires_grad_(True)
                            out = fn(u, *live)
                            grads = torch.autograd.grad(out, [u, *diff], X[b], allow_unused=True)
                            Y[b].copy_(grads[0])
                            pg = grads[1:]
                        else:  # y = A x ; d/dtheta aux^T A(theta) x                   (lanczos.py:328-329)
                            out = fn(X[b].detach(), *live)
                            Y[b].copy_(out.detach())
                            pg = torch.autograd.grad(out, diff, AUX[b], allow_unused=True) if diff else ()
                    it = iter(pg)
                    for a in accum:
                        if a is not None:
                            g = next(it)
                            if g is not None:
                                a.add_(g)
                return 0
            except Exception as exc:  # never let an exception cross the C boundary
                failure.append(exc)
                return 1

        cb = _lib.CALLBACK_T(trampoline)
        desc = _lib.Operator()
        desc.kind = _lib.OP_CALLBACK
        desc.dtype = _lib.dtype_code(dtype)
        desc.n = n
        desc.callback = cb
        return desc, (cb, failure), accum


KernelGramOp = RbfGramOp  # the operator covers the reference's three stationary kernels


def as_operator(matvec):
    """matvec argument of the reference API -> (operator, bound-params or None)."""
    if isinstance(matvec, BoundOp):
        return matvec.op, matvec.params
    if isinstance(matvec, (NativeOp, CallbackOp, RowShardedOp)):
        return matvec, None
    if callable(matvec):
        return CallbackOp(matvec), None
    raise TypeError(f"matvec must be callable or a native operator, got {type(matvec)}")
