"""Independent, differentiable torch-CPU forward passes (test infrastructure).

These play the role that "plain autodiff through the forward loop" plays in the reference's tests
(tests/test_lanczos/test_tridiag_adjoint.py:12-50, tests/test_arnoldi/test_hessenberg_adjoint.py):
the custom adjoint must agree with autodiff of an un-adorned forward pass.  Written functionally
(no in-place ops) so torch.autograd can differentiate them.
"""

import torch


def arnoldi_forward(matvec, k, v, *params, second_pass=True):
    n = v.shape[0]
    cols, hcols = [], []
    length0 = torch.sqrt(v @ v)
    length, w = length0, v
    for i in range(k):
        q = w / length
        cols.append(q)
        Q = torch.stack(cols, dim=1)  # (n, i+1)
        w = matvec(q, *params)
        h = Q.T @ w
        w = w - Q @ h
        if second_pass:
            w = w - Q @ (Q.T @ w)
        length = torch.sqrt(w @ w)
        col = torch.cat([h, length[None], torch.zeros(k - i - 1, dtype=v.dtype)])[:k]
        hcols.append(col)
    Q = torch.stack(cols, dim=1)
    H = torch.stack(hcols, dim=1)
    assert Q.shape == (n, k)
    return Q, H, w, 1.0 / length0


def tridiag_full(matvec, k, v, *params):
    Q, H, r, _ = arnoldi_forward(matvec, k, v, *params)
    T = 0.5 * (H + H.T)
    rn = torch.linalg.norm(r)
    return (Q.T, (torch.diagonal(T), torch.diagonal(T, 1))), (r / rn, rn)


def tridiag_none(matvec, k, v, *params):
    xs = [v / torch.linalg.norm(v)]
    a, b = [], []
    prev, bprev = torch.zeros_like(v), torch.zeros((), dtype=v.dtype)
    for i in range(k):
        w = matvec(xs[i], *params)
        ai = xs[i] @ w
        r = w - ai * xs[i] - bprev * prev
        bi = torch.linalg.norm(r)
        xs.append(r / bi)
        a.append(ai)
        b.append(bi)
        prev, bprev = xs[i], bi
    X = torch.stack(xs)
    a, b = torch.stack(a), torch.stack(b)
    return (X[:-1], (a, b[:-1])), (X[-1], b[-1])


def integrand_spd(matfun, k, matvec, v0, *params, reortho="full"):
    scale = torch.linalg.norm(v0)
    u = v0 / scale
    fn = tridiag_full if reortho == "full" else tridiag_none
    (_, (diag, off)), _ = fn(matvec, k, u, *params)
    T = torch.diag(diag) + torch.diag(off, 1) + torch.diag(off, -1)
    lam, U = torch.linalg.eigh(T)
    return scale**2 * (U[0] @ (matfun(lam) * U[0]))


def flat_cat(tree):
    out = []

    def rec(t):
        if isinstance(t, (tuple, list)):
            for s in t:
                rec(s)
        else:
            out.append(torch.as_tensor(t).reshape(-1))

    rec(tree)
    return torch.cat(out)
