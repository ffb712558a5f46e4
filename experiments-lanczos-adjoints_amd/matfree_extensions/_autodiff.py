"""``custom_vjp=False``: the forward recurrences written with differentiable torch ops, so that torch.autograd back-propagates
THROUGH the loop -- the baseline the reference compares its adjoints against (tests/test_arnoldi/test_hessenberg_adjoint.py,
tests/test_lanczos/test_tridiag_adjoint.py, benchmark.py:125-139 "backprop through the loop").

Not the product path: the operator application still goes through libmfx (``op(v, *params)`` is a HIP call with its own
autograd rule), but the Gram-Schmidt / three-term vector algebra here is plain torch on the device, one launch per
operation, with every intermediate kept for the backward pass -- which is exactly what makes it the slow, memory-hungry
baseline (arnoldi.py:57-101 and lanczos.py:215-285 are differentiated the same way by JAX when custom_vjp=False).
"""

from __future__ import annotations

import torch


def arnoldi_forward(matvec, k, v, params, *, second_pass=True):
    """arnoldi.py:57-101 for ONE vector v (n,) -> Q (n, k), H (k, k), r (n,), c ()"""
    cols, hcols = [], []
    length0 = torch.sqrt(v @ v)
    length, w = length0, v
    for i in range(k):
        q = w / length
        cols.append(q)
        Q = torch.stack(cols, dim=1)
        w = matvec(q, *params)
        h = Q.T @ w
        w = w - Q @ h
        if second_pass:
            w = w - Q @ (Q.T @ w)
        length = torch.sqrt(w @ w)
        hcols.append(torch.cat([h, length[None], w.new_zeros(k - i - 1)])[:k])  # Q2: no (k+1)-th row
    return torch.stack(cols, dim=1), torch.stack(hcols, dim=1), w, 1.0 / length0


def lanczos_forward(matvec, k, v, params):
    """lanczos.py:215-285 for ONE vector -> xs (k+1, n), alpha (k,), beta (k,)"""
    xs = [v / torch.linalg.vector_norm(v)]
    a, b = [], []
    prev, bprev = torch.zeros_like(v), v.new_zeros(())
    for i in range(k):
        w = matvec(xs[i], *params)
        ai = xs[i] @ w
        r = w - ai * xs[i] - bprev * prev
        bi = torch.linalg.vector_norm(r)
        xs.append(r / bi)
        a.append(ai)
        b.append(bi)
        prev, bprev = xs[i], bi
    return torch.stack(xs), torch.stack(a), torch.stack(b)


def batched(fn, V, *args, **kw):
    """apply a single-vector recurrence to every row of V (p, n) and stack the outputs"""
    outs = [fn(*args[:2], V[b], *args[2:], **kw) for b in range(V.shape[0])]
    return tuple(torch.stack([o[i] for o in outs]) for i in range(len(outs[0])))
