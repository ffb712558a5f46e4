"""CPU oracle for the Lanczos/Arnoldi-with-adjoint + SLQ + Hutchinson hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.

This is a NumPy (fp64 by default, dtype-preserving) *restatement* of the reference algorithm,
written from the reference's source text (file:line cited per function, relative to
``/root/reference/src/matfree_extensions``).  The reference is pure-Python JAX; JAX and ``matfree``
are not installable in the build container (``ModuleNotFoundError``), so the reference itself was
never executed.

Pinning status: the reference ships **no stored golden vectors** for this path; its
tests are known-answer *property* tests (decomposition identities, custom-VJP == autodiff-VJP,
k = n exactness).  This oracle is pinned against exactly those known-answer tests in
``tests/test_oracle_*.py`` (identities at the reference's tolerances; adjoints against an independent
torch-autograd differentiation of an independently written forward pass, and against central finite
differences).  PRNG-key dependent values (``jax.random`` streams, ``matfree.test_util`` matrices) are
**parity unpinned**: parity is defined per explicit probe matrix / explicit test matrix.

Reference quirks reproduced on purpose are marked ``# Q<n>``.
"""

from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------
# Operators: A(theta) v, A(theta)^T lam, and d/dtheta [cot^T A(theta) v]
# --------------------------------------------------------------------------------------


class DenseOp:
    """matvec(v, A) = A @ v  (tests/test_lanczos/test_tridiag_forward.py:18)."""

    def apply(self, v, A):
        return A @ v

    def apply_t(self, lam, A):
        return A.T @ lam

    def param_vjp(self, v, cot, A):
        # v, cot of shape (n,) or a stacked batch (b, n): sum_b cot_b v_b^T
        return (np.atleast_2d(cot).T @ np.atleast_2d(v),)


class DenseSymOp:
    """matvec(v, P) = (P + P^T) @ v  (tests/test_lanczos/test_tridiag_adjoint.py:20-21)."""

    def apply(self, v, P):
        return (P + P.T) @ v

    def apply_t(self, lam, P):
        return (P + P.T) @ lam

    def param_vjp(self, v, cot, P):
        o = np.atleast_2d(cot).T @ np.atleast_2d(v)
        return (o + o.T,)


class CooOp:
    """matvec(v, vals) = COO(vals, (row, col)) @ v  (benchmark.py:64-68, exp_util.py:35-42).

    The reference differentiates w.r.t. *all* stored values of the symmetric-expanded matrix.
    """

    def __init__(self, row, col, n):
        self.row = np.asarray(row, dtype=np.int64)
        self.col = np.asarray(col, dtype=np.int64)
        self.n = int(n)

    def apply(self, v, vals):
        out = np.zeros(self.n, dtype=np.result_type(v, vals))
        np.add.at(out, self.row, vals * v[self.col])
        return out

    def apply_t(self, lam, vals):
        out = np.zeros(self.n, dtype=np.result_type(lam, vals))
        np.add.at(out, self.col, vals * lam[self.row])
        return out

    def param_vjp(self, v, cot, vals):
        return ((np.atleast_2d(cot)[:, self.row] * np.atleast_2d(v)[:, self.col]).sum(0),)


def softplus(x, beta=1.0, threshold=20.0):
    """util/gp_util.py:187-201 (constraint_greater_than): torch-style thresholded softplus."""
    x = np.asarray(x)
    small = x * beta < threshold
    xs = np.where(small, x, 1.0)
    return np.where(small, np.log1p(np.exp(beta * xs)) / beta, x)


def softplus_grad(x, beta=1.0, threshold=20.0):
    x = np.asarray(x)
    small = x * beta < threshold
    xs = np.where(small, x, 1.0)
    return np.where(small, 1.0 / (1.0 + np.exp(-beta * xs)), 1.0)


def rbf_kernel_matrix(Xa, Xb, lengthscale, outputscale):
    """util/gp_util.py:160-176: k(x,y) = s * exp(-max(0, |x/l|^2 + |y/l|^2 - 2 (x/l).(y/l)) / 2)."""
    xa = Xa / lengthscale
    xb = Xb / lengthscale
    sq = (xa * xa).sum(-1)[:, None] + (xb * xb).sum(-1)[None, :] - 2.0 * (xa @ xb.T)
    sq = np.maximum(0.0, sq)
    return outputscale * np.exp(-0.5 * sq)


def scaled_sqdist(Xa, Xb, lengthscale, diag_offset=None):
    """max(0, |x/l|^2 + |y/l|^2 - 2 (x/l).(y/l)) -- the clamped expanded form shared by all three kernels.

    diag_offset: Xa = Xb[diag_offset : diag_offset + len(Xa)]; the distance of a point to itself is then set to its
    exact value 0.  (The expanded form leaves O(eps |x|^2) there, which the Matern kernels' sqrt turns into an
    O(sqrt(eps)) error of K_ii -- 3e-4 in fp32 -- in the reference as well; RBF is insensitive to it.)"""
    xa = Xa / lengthscale
    xb = Xb / lengthscale
    sq = (xa * xa).sum(-1)[:, None] + (xb * xb).sum(-1)[None, :] - 2.0 * (xa @ xb.T)
    sq = np.maximum(0.0, sq)
    if diag_offset is not None:
        sq[np.arange(len(Xa)), diag_offset + np.arange(len(Xa))] = 0.0
    return sq


def kernel_matrix(kind, Xa, Xb, lengthscale, outputscale, diag_offset=None, eps=None):
    """util/gp_util.py:160-176 (rbf), :69-107 (matern32: sqrt(3) x / l, (1 + r) exp(-r)), :110-148 (matern12: exp(-r));
    r = sqrt(clamped squared distance + eps(dtype)) (util/gp_util.py:99-100,140-141)."""
    if kind == "rbf":
        return rbf_kernel_matrix(Xa, Xb, lengthscale, outputscale)
    sq = scaled_sqdist(Xa, Xb, lengthscale, diag_offset)
    eps = np.finfo(np.result_type(Xa, Xb)).eps if eps is None else eps  # the reference takes eps of the COMPUTE dtype
    if kind == "matern32":
        r = np.sqrt(3.0 * sq + eps)
        return outputscale * (1.0 + r) * np.exp(-r)
    if kind == "matern12":
        r = np.sqrt(sq + eps)
        return outputscale * np.exp(-r)
    raise ValueError(kind)


def kernel_lengthscale_weight(kind, Xa, Xb, lengthscale, diag_offset=None, eps=None):
    """w with  dK_ij / d l_c = outputscale * w_ij * (x_ic - x_jc)^2 / l_c^3  (differentiating the formulas above)."""
    sq = scaled_sqdist(Xa, Xb, lengthscale, diag_offset)
    if kind == "rbf":
        return np.exp(-0.5 * sq)
    eps = np.finfo(np.result_type(Xa, Xb)).eps if eps is None else eps
    if kind == "matern32":
        return 3.0 * np.exp(-np.sqrt(3.0 * sq + eps))
    r = np.sqrt(sq + eps)
    return np.where(sq > 0.0, np.exp(-r) / r, 0.0)  # d max(0, s)/ds = 0 on the clamped side


class RbfGramOp:
    """K(X,X; raw_l, raw_s) + noise I, matrix-free semantics of util/gp_util.py:225-226,525-549.

    `kernel` selects the reference's kernel family: "rbf" (kernel_scaled_rbf), "matern32", "matern12".

    params = (raw_lengthscale [() or (d,)], raw_outputscale (), raw_noise ());
    lengthscale = softplus(raw_l), outputscale = softplus(raw_s), noise = minval + softplus(raw_noise).
    Row-chunked so that n up to a few 1e4 stays in memory.
    """

    def __init__(self, X, noise_minval=0.0, chunk=2048, cache_limit=4096, kernel="rbf", eps=None):
        self.kernel = kernel
        self.eps = eps  # Matern: eps of the dtype the compared implementation computes in (default: dtype of X)
        self.X = np.asarray(X)
        self.n, self.d = self.X.shape
        self.noise_minval = noise_minval
        self.chunk = chunk
        self.cache_limit = cache_limit  # n up to which the dense K is kept between calls (test speed only)
        self._cache = (None, None)

    def _dense_k(self, ls, s):
        if self.n > self.cache_limit:
            return None
        key = (np.asarray(ls).tobytes(), np.asarray(s).tobytes())
        if self._cache[0] != key:
            self._cache = (key, kernel_matrix(self.kernel, self.X, self.X, ls, s, diag_offset=0, eps=self.eps))
        return self._cache[1]

    def constrained(self, raw_l, raw_s, raw_noise):
        return softplus(raw_l), softplus(raw_s), self.noise_minval + softplus(raw_noise)

    def apply(self, v, raw_l, raw_s, raw_noise):
        ls, s, noise = self.constrained(raw_l, raw_s, raw_noise)
        Kd = self._dense_k(ls, s)
        if Kd is not None:
            return v @ Kd.T + noise * v
        out = np.empty(v.shape, dtype=np.result_type(v, self.X))
        for a in range(0, self.n, self.chunk):
            b = min(self.n, a + self.chunk)
            K = kernel_matrix(self.kernel, self.X[a:b], self.X, ls, s, diag_offset=a, eps=self.eps)
            out[..., a:b] = v @ K.T  # supports v of shape (n,) or (p, n)
        return out + noise * v

    def apply_t(self, lam, raw_l, raw_s, raw_noise):
        return self.apply(lam, raw_l, raw_s, raw_noise)  # symmetric

    def param_vjp(self, v, cot, raw_l, raw_s, raw_noise):
        """d/d(raw) of sum_b cot_b^T A v_b ; v, cot of shape (n,) or (p, n)."""
        ls, s, noise = self.constrained(raw_l, raw_s, raw_noise)
        V = np.atleast_2d(v)
        C = np.atleast_2d(cot)
        ard = np.ndim(raw_l) > 0
        g_l = np.zeros(self.d if ard else (), dtype=np.float64)
        g_s = 0.0
        Kd = self._dense_k(ls, s)
        for a in range(0, self.n, self.chunk):
            b = min(self.n, a + self.chunk)
            K = Kd[a:b] if Kd is not None else kernel_matrix(self.kernel, self.X[a:b], self.X, ls, s, diag_offset=a, eps=self.eps)
            S = C[:, a:b].T @ V  # (chunk, n): sum_b cot_b[i] v_b[j]
            g_s += (S * K).sum() / s
            W = S * (s * kernel_lengthscale_weight(self.kernel, self.X[a:b], self.X, ls, diag_offset=a, eps=self.eps))
            if ard:
                for c in range(self.d):
                    diff2 = (self.X[a:b, c][:, None] - self.X[None, :, c]) ** 2
                    g_l[c] += (W * diff2).sum() / ls[c] ** 3
            else:
                sqn = (self.X * self.X).sum(-1)
                diff2 = np.maximum(0.0, sqn[a:b, None] + sqn[None, :] - 2.0 * self.X[a:b] @ self.X.T)
                diff2[np.arange(b - a), np.arange(a, b)] = 0.0  # exact zero distance of a point to itself
                g_l += (W * diff2).sum() / ls**3
        g_noise = float((C * V).sum())
        return (
            g_l * softplus_grad(raw_l),
            g_s * softplus_grad(raw_s),
            g_noise * softplus_grad(raw_noise),
        )


# --------------------------------------------------------------------------------------
# A5: Arnoldi forward (arnoldi.py:57-101)
# --------------------------------------------------------------------------------------


def arnoldi_forward(op, k, v, *params, reortho="full", reortho_vjp="match"):
    """Returns Q (n,k), H (k,k), r (n,) un-normalised, c = 1/|v|.

    Classical Gram-Schmidt against all (zero-padded) columns, optional second pass whose
    coefficients are NOT added to h (arnoldi.py:87-92).  Complex v: the conjugations of arnoldi.py:66,87,92,95
    (identity on real input; the lengths then carry a zero imaginary part, as in the reference).
    """
    if reortho not in ("none", "full"):
        raise TypeError(f"Unexpected input for {reortho}: either of ['none', 'full'] expected.")
    n = v.shape[0]
    if k < 1 or k > n:
        raise ValueError(f"Parameter depth {k} is outside the expected range")
    # Q1 (arnoldi.py:26): the forward pass sees `reortho_vjp`, never `reortho`; with the default
    # "match" that string is != "none", hence the second pass always runs.
    second_pass = reortho_vjp != "none"
    Q = np.zeros((n, k), dtype=v.dtype)
    H = np.zeros((k, k), dtype=v.dtype)
    length0 = np.sqrt(np.vdot(v, v))
    length = length0
    w = v
    for i in range(k):
        q = w / length
        Q[:, i] = q
        w = op.apply(q, *params)
        h = Q.conj().T @ w
        w = w - Q @ h
        if second_pass:
            w = w - Q @ (Q.conj().T @ w)
        length = np.sqrt(np.vdot(w, w))
        if i + 1 < k:  # Q2 (arnoldi.py:98): the out-of-range write at i = k-1 is dropped
            h[i + 1] = length
        H[:, i] = h
    return Q, H, w, 1.0 / length0


# --------------------------------------------------------------------------------------
# A6: Arnoldi adjoint (arnoldi.py:104-220)
# --------------------------------------------------------------------------------------


def arnoldi_adjoint(op, params, *, Q, H, r, c, dQ, dH, dr, dc, reortho="full"):
    """VJP of arnoldi_forward.  Returns (dv, dparams-tuple).  Valid for non-symmetric A."""
    n, k = Q.shape
    tril = np.tril(np.ones((k, k)))
    lower_mask = tril - 0.5 * np.eye(k)  # arnoldi.py:112-117
    ps_mask = np.tril(np.ones((k, k)), 1)  # arnoldi.py:133

    eta = dH[:, -1] - Q.T @ dr  # arnoldi.py:120
    lam = dr + Q @ eta  # arnoldi.py:121
    Lam = np.zeros_like(Q)
    Gam = np.zeros((k, k), dtype=Q.dtype)
    Pi_xi = dQ.T + np.outer(eta, r)  # arnoldi.py:127   (k, n)
    e1 = np.zeros(k)
    e1[0] = 1.0
    Pi_gamma = -dc * c * np.outer(e1, e1) + H @ dH.T - dQ.T @ Q  # arnoldi.py:128

    beta_minus = np.concatenate([np.ones(1), np.diag(H, -1)])  # arnoldi.py:137
    alpha = np.diag(H)
    beta_plus = H - np.diag(np.diag(H)) - np.diag(np.diag(H, -1), -1)  # arnoldi.py:139

    P = Q.T.copy()
    pairs_v, pairs_cot = [], []
    for idx in range(k - 1, -1, -1):
        if reortho == "full":  # arnoldi.py:200-204 (cumulative row masking of P)
            P = ps_mask[idx][:, None] * P
            pvec = ps_mask[idx] * dH[:, idx]
            lam = lam - P.T @ (P @ lam) + P.T @ pvec
        q = Q[:, idx]
        z = op.apply_t(lam, *params)  # arnoldi.py:207-208: one vjp gives A^T lam and d/dtheta
        pairs_v.append(q)  # d/dtheta [lam^T A(theta) q], summed over idx (arnoldi.py:209) in one sweep below
        pairs_cot.append(lam)
        Gam[idx, :] = lower_mask[idx] * (Pi_gamma[idx] - z @ Q)  # arnoldi.py:212-213
        Lam[:, idx] = lam  # arnoldi.py:216
        xi = Pi_xi[idx] + (Gam + Gam.T)[idx, :] @ Q.T  # arnoldi.py:217
        lam = (xi - (alpha[idx] * lam - z) - beta_plus[idx] @ Lam.T) / beta_minus[idx]
    dparams = op.param_vjp(np.stack(pairs_v), np.stack(pairs_cot), *params)
    return lam * c, dparams


# --------------------------------------------------------------------------------------
# A3: full-reortho tridiagonalisation through Arnoldi (lanczos.py:152-169) and its VJP
# --------------------------------------------------------------------------------------


def tridiag_full(op, k, v, *params):
    """Returns ((basis (k,n), (diag (k,), offdiag (k-1,))), (q_rem (n,), beta_rem))."""
    Q, H, r, _c = arnoldi_forward(op, k, v, *params, reortho="full")
    T = 0.5 * (H + H.T)
    rn = np.linalg.norm(r)
    return (Q.T, (np.diag(T).copy(), np.diag(T, 1).copy())), (r / rn, rn)


def tridiag_full_vjp(op, k, v, params, cot):
    """cot mirrors the output pytree: ((dbasis (k,n), (ddiag, doff)), (dq_rem, dbeta_rem))."""
    (dbasis, (ddiag, doff)), (dq, db) = cot
    Q, H, r, c = arnoldi_forward(op, k, v, *params, reortho="full")
    rn = np.linalg.norm(r)
    u = r / rn
    dH = np.diag(ddiag) + 0.5 * (np.diag(doff, 1) + np.diag(doff, -1))
    dr = (dq - u * (u @ dq)) / rn + db * u
    return arnoldi_adjoint(
        op, params, Q=Q, H=H, r=r, c=c, dQ=dbasis.T, dH=dH, dr=dr, dc=0.0, reortho="full"
    )


# --------------------------------------------------------------------------------------
# A4: three-term Lanczos without re-orthogonalisation (lanczos.py:215-285)
# --------------------------------------------------------------------------------------


def tridiag_none(op, k, v, *params):
    """Same return pytree as tridiag_full.  vectors (k+1,n), diags (k,), offdiags (k,)."""
    n = v.shape[0]
    xs = np.zeros((k + 1, n), dtype=v.dtype)
    a = np.zeros(k, dtype=v.dtype)
    b = np.zeros(k, dtype=v.dtype)
    xs[0] = v / np.linalg.norm(v)
    prev = np.zeros_like(v)
    bprev = 0.0
    for i in range(k):
        w = op.apply(xs[i], *params)
        a[i] = xs[i] @ w
        rr = w - a[i] * xs[i] - bprev * prev
        b[i] = np.linalg.norm(rr)
        xs[i + 1] = rr / b[i]
        prev, bprev = xs[i], b[i]
    return (xs[:-1], (a, b[:-1])), (xs[-1], b[-1])


# A8: its adjoint (lanczos.py:288-335)
def tridiag_none_vjp(op, k, v, params, cot):
    (dxs_, (da, db_)), (dx_last, db_last) = cot
    (xs_, (a, b_)), (x_last, b_last) = tridiag_none(op, k, v, *params)
    xs = np.concatenate([xs_, x_last[None]])
    b = np.concatenate([b_, [b_last]])
    dxs = np.concatenate([dxs_, dx_last[None]])
    db = np.concatenate([db_, [db_last]])
    xi = -dxs[-1]
    lam_plus = np.zeros_like(xi)
    pairs_v, pairs_cot = [], []
    for j in range(k - 1, -1, -1):
        xplus, x = xs[j + 1], xs[j]
        xi = xi / b[j]
        mu = db[j] - lam_plus @ x + xplus @ xi
        nu = da[j] + x @ xi
        lam = -xi + mu * xplus + nu * x
        # Q4 (lanczos.py:328): A lam (not A^T lam), parameter-gradient of x^T A(theta) lam
        Alam = op.apply(lam, *params)
        pairs_v.append(lam)
        pairs_cot.append(x)
        xi = -dxs[j] - Alam + a[j] * lam + b[j] * lam_plus - b[j] * nu * xplus
        lam_plus = lam
    # Q3 (lanczos.py:305,311): the "lambda_1" used for the initial-vector gradient is the final xi
    dvec = ((xi @ xs[0]) * xs[0] - xi) / np.linalg.norm(v)
    dparams = op.param_vjp(np.stack(pairs_v), np.stack(pairs_cot), *params)  # lanczos.py:310 (sum over steps)
    return dvec, dparams


def tridiag(op, k, v, *params, reortho):
    """lanczos.py:142-149 dispatcher."""
    if reortho == "full":
        return tridiag_full(op, k, v, *params)
    if reortho == "none":
        return tridiag_none(op, k, v, *params)
    raise ValueError(f"reortho={reortho} unsupported. Choose eiter {'full', 'none'}.")


# --------------------------------------------------------------------------------------
# A1: SLQ integrand  |v|^2 e1^T f(T) e1  (lanczos.py:14-61) with value-and-gradient
# --------------------------------------------------------------------------------------

MATFUNS = {
    "log": (np.log, lambda x: 1.0 / x),
    "exp": (np.exp, np.exp),
    "inv": (lambda x: 1.0 / x, lambda x: -1.0 / x**2),
    "sqrt": (np.sqrt, lambda x: 0.5 / np.sqrt(x)),
    "identity": (lambda x: x, lambda x: np.ones_like(x)),
}


def dense_tridiag(diag, off):
    return np.diag(diag) + np.diag(off, 1) + np.diag(off, -1)


def quadform_from_tridiag(diag, off, matfun="log"):
    """e1^T f(T) e1 via eigh (lanczos.py:48-59) and its gradient w.r.t. (diag, off).

    Gradient = Frechet derivative through divided differences (what differentiating eigh yields
    for distinct eigenvalues): G = U (F o u0 u0^T) U^T, ddiag = diag(G), doff = 2 diag(G, 1).
    """
    f, df = MATFUNS[matfun] if isinstance(matfun, str) else matfun
    lam, U = np.linalg.eigh(dense_tridiag(diag, off))
    fl = f(lam)
    u0 = U[0]
    val = u0 @ (fl * u0)
    dl = lam[:, None] - lam[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        F = (fl[:, None] - fl[None, :]) / dl
    close = np.abs(dl) <= 1e-13 * np.maximum(np.abs(lam[:, None]), np.abs(lam[None, :]))
    Fd = 0.5 * (df(lam)[:, None] + df(lam)[None, :])
    F = np.where(close, Fd, F)
    G = U @ (F * np.outer(u0, u0)) @ U.T
    return val, (np.diag(G).copy(), 2.0 * np.diag(G, 1).copy()), (lam, U)


def integrand_spd_value_and_grad(op, k, v0, params, *, matfun="log", reortho="full"):
    """Value, d/dv0 and d/dparams of lanczos.integrand_spd(matfun, k, matvec)(v0, *params)."""
    scale = np.linalg.norm(v0)
    u = v0 / scale
    (basis, (diag, off)), (q, b) = tridiag(op, k, u, *params, reortho=reortho)
    g, (ddiag, doff), _ = quadform_from_tridiag(diag, off, matfun)
    value = scale**2 * g
    cot = ((np.zeros_like(basis), (ddiag, doff)), (np.zeros_like(q), 0.0))
    if reortho == "full":
        du, dparams = tridiag_full_vjp(op, k, u, params, cot)
    else:
        du, dparams = tridiag_none_vjp(op, k, u, params, cot)
    dv0 = 2.0 * scale * g * u + scale * (du - u * (u @ du))
    dparams = tuple(scale**2 * d for d in dparams)
    return value, dv0, dparams


# A9: integrand_spd_custom_vjp_reuse (lanczos.py:64-139): inexact, first-order-only gradient
def integrand_spd_reuse_value_and_grad(op, k, v0, params, *, matfun="log", reortho="full"):
    f, df = MATFUNS[matfun]
    scale = np.linalg.norm(v0)
    u = v0 / scale
    (basis, (diag, off)), _ = tridiag(op, k, u, *params, reortho=reortho)
    lam, U = np.linalg.eigh(dense_tridiag(diag, off))
    value = scale**2 * (U[0] @ (f(lam) * U[0]))
    sol = U @ (df(lam) * U[0])  # lanczos.py:116
    w1, w2 = scale**2 * (basis.T @ sol), u  # lanczos.py:117
    dparams = op.param_vjp(w2, w1, *params)  # lanczos.py:128: vjp of p -> (A(p) w2)^T w1
    return value, np.zeros_like(v0), dparams  # lanczos.py:131-134: zero gradient w.r.t. v0


# --------------------------------------------------------------------------------------
# A10: Hutchinson (hutchinson.py:8-65; matfree.hutchinson.hutchinson / sampler_rademacher)
# --------------------------------------------------------------------------------------

_M64 = (1 << 64) - 1


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M64)
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_M64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_M64)
    return z ^ (z >> np.uint64(31))


def rademacher(seed, num, n, first_probe=0, dtype=np.float64):
    """Counter-based +-1 stream shared bit-for-bit with the HIP sampler (csrc/mfx_sampler.hip).

    Element (b, i) depends only on (seed, first_probe + b, i), so any sharding of the probes over
    ranks yields the same global probe matrix.  (The reference's jax.random streams cannot be
    reproduced without JAX: parity is per explicit probe matrix.)
    """
    with np.errstate(over="ignore"):
        b = np.arange(first_probe, first_probe + num, dtype=np.uint64)[:, None]
        i = np.arange(n, dtype=np.uint64)[None, :]
        key = _splitmix64(np.uint64(seed) ^ (b * np.uint64(0xD1342543DE82EF95)))
        bits = _splitmix64(key + i)
    return np.where((bits >> np.uint64(63)) == 1, 1.0, -1.0).astype(dtype)


def hutchinson_value_and_grad(op, k, probes, params, **kw):
    """mean over probes of the integrand and of its parameter gradient (hutchinson.py:12-15)."""
    vals, grads = [], None
    for v in probes:
        val, _dv, dp = integrand_spd_value_and_grad(op, k, v, params, **kw)
        vals.append(val)
        grads = dp if grads is None else tuple(a + b for a, b in zip(grads, dp))
    num = len(probes)
    return np.mean(vals), tuple(g / num for g in grads), np.asarray(vals)


def hutchinson_batch(estimate, keys, *params):
    """hutchinson.py:57-65: mean over `num` sequential sub-estimates."""
    return np.mean([estimate(key, *params) for key in keys], axis=0)


def krylov_logdet_slq(op, k, probe_batches, params, **kw):
    """util/gp_util.py:552-576: mean/std over sequential probe batches."""
    values = [hutchinson_value_and_grad(op, k, pb, params, **kw)[0] for pb in probe_batches]
    if len(values) == 1:
        return values[0], {"std": 0.0, "std_rel": 0.0}
    mean, std = np.mean(values), np.std(values)
    return mean, {"std_abs": std, "std_rel": std / abs(mean)}


# --------------------------------------------------------------------------------------
# Test-matrix helpers (property-equivalent stand-ins for matfree.test_util, exp_util.hilbert)
# --------------------------------------------------------------------------------------


def symmetric_matrix_from_eigenvalues(eigvals, seed=0):
    """Any SPD matrix with the prescribed spectrum; the reference uses this helper only in
    property tests, so the particular orthogonal factor is immaterial."""
    n = len(eigvals)
    rng = np.random.default_rng(seed)
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (Qm * np.asarray(eigvals)) @ Qm.T


def hilbert(n):
    a = np.arange(n)
    return 1.0 / (1.0 + a[:, None] + a[None, :])  # util/exp_util.py:113-115


def spd_diag_plus_lowrank(n=512, rank=4, seed=0):
    """BASELINE config C1: diag(1 + 9 i/(n-1)) + U U^T, U ~ N(0,1)/sqrt(n)."""
    rng = np.random.default_rng(seed)
    U = rng.standard_normal((n, rank)) / np.sqrt(n)
    return np.diag(1.0 + 9.0 * np.arange(n) / (n - 1)) + U @ U.T


def laplacian_2d_plus_identity(m):
    """BASELINE config C3 throughput stand-in: 5-pt Laplacian on an m x m grid + I, as COO."""
    idx = np.arange(m * m).reshape(m, m)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [np.full(m * m, 5.0)]
    for a, b in ((idx[:-1], idx[1:]), (idx[:, :-1], idx[:, 1:])):
        rows += [a.ravel(), b.ravel()]
        cols += [b.ravel(), a.ravel()]
        vals += [np.full(a.size, -1.0)] * 2
    return np.concatenate(rows), np.concatenate(cols), np.concatenate(vals), m * m


# --------------------------------------------------------------------------------------
# "next" tier: conjugate gradients, partial Cholesky, preconditioner, logpdf
# --------------------------------------------------------------------------------------


def safe_divide(a, b):
    """cg.py:222-241: a / b where |b| > eps(dtype)^2, else a."""
    eps = np.finfo(np.asarray(a).dtype).eps ** 2
    return a / b if abs(b) > eps else a


def _pcg_body(A, P, state):
    """cg.py:44-57 (identical in the fixed-step and the adaptive solver, :112-127)."""
    x, p, r, z = state
    Ap = A(p)
    a = safe_divide(r @ z, p @ Ap)
    x = x + a * p
    rold, zold = r, z
    r = r - a * Ap
    z = P(r)
    b = safe_divide(r @ z, rold @ zold)
    p = z + b * p
    return x, p, r, z


def pcg_fixed_step(A, b, P=None, *, num_matvecs):
    """cg.py:19-60.  -> (x, info)"""
    P = (lambda v: v) if P is None else P
    x = np.zeros_like(b)
    r = b - A(x)
    z = P(r)
    state = (x, z, r, z)
    for _ in range(num_matvecs):
        state = _pcg_body(A, P, state)
    x, _p, r, _z = state
    with np.errstate(divide="ignore", invalid="ignore"):
        return x, {"residual_abs": r, "residual_rel": r / np.abs(x)}


def pcg_adaptive(A, b, P=None, *, atol, rtol, maxiter, miniter=0):
    """cg.py:74-137.  -> (x, info)"""
    P = (lambda v: v) if P is None else P
    x = np.zeros_like(b)
    r = b - A(x)
    z = P(r)
    state, nsteps = (x, z, r, z), 0
    while True:
        x, _p, r, _z = state
        error_rel = r / (atol + np.abs(x) * rtol)
        is_error_large = np.sqrt(np.mean(error_rel**2)) > 1.0
        if not ((is_error_large or nsteps < miniter) and nsteps < maxiter):
            break
        state = _pcg_body(A, P, state)
        nsteps += 1
    x, _p, r, _z = state
    with np.errstate(divide="ignore", invalid="ignore"):
        return x, {"residual_abs": r, "residual_rel": r / np.abs(x), "num_steps": nsteps}


def pcg_fixed_step_reortho(A, b, P=None, *, num_matvecs):
    """cg.py:151-219.  -> (x, {"residual_abs", "Q" (n, num_matvecs)})"""
    P = (lambda v: v) if P is None else P
    x = np.zeros_like(b)
    r = b - A(x)
    z = P(r)
    p = z
    Q = np.zeros((len(b), num_matvecs), dtype=b.dtype)
    rzdot = r @ z
    for i in range(num_matvecs):
        Ap = A(p)
        a = safe_divide(rzdot, p @ Ap)
        x = x + a * p
        r, rold = r - a * Ap, r
        z, zold = P(r), z
        den = np.sqrt(rzdot) if rzdot > 0.0 else 0.0  # _safe_sqrt, cg.py:244-246
        eps = np.finfo(b.dtype).eps ** 2
        Q[:, i] = rold / den if abs(den) > eps else rold
        r = r - Q @ (Q.T @ z)
        z = P(r)
        rzdot = r @ z
        bb = safe_divide(rzdot, rold @ zold)
        p = z + bb * p
    return x, {"residual_abs": r, "Q": Q}


def linear_solve_vjp(op, params, solver, x, dx):
    """The rule of jax.lax.custom_linear_solve(..., symmetric=True) (cg.py:23-25) for x = solver(A(theta), b):
    db = solver(A, dx) (the transpose solve IS the solve), dtheta = vjp of theta -> A(theta) x with cotangent -db."""
    lam, _ = solver(lambda v: op.apply(v, *params), dx)
    return lam, op.param_vjp(x, -lam, *params)


def cholesky_partial(element, n, rank):
    """low_rank.py:63-120.  element(i, j) -> K_ij.  -> L (n, rank)"""
    if rank > n:
        raise ValueError(f"Rank exceeds n: {rank} >= {n}.")
    if rank < 1:
        raise ValueError(f"Rank must be positive, but {rank} < {1}.")
    L = np.zeros((n, rank))
    for i in range(rank):
        l_ii = np.sqrt(element(i, i) - L[i] @ L[i])
        column = np.array([element(j, i) for j in range(n)])
        L[:, i] = (column - L @ L[i, :]) / l_ii
    return L, {}


def cholesky_partial_pivot(element, n, rank):
    """low_rank.py:123-228, with the reference's explicit row permutations (the HIP kernels work in the original
    row order instead; the tests check both give the same factor)."""
    if rank > n:
        raise ValueError(f"Rank exceeds n: {rank} >= {n}.")
    if rank < 1:
        raise ValueError(f"Rank must be positive, but {rank} < {1}.")
    L = np.zeros((n, rank))
    P = np.arange(n)
    Pm = np.arange(n)
    success = True
    with np.errstate(invalid="ignore"):
        for i in range(rank):
            diagonal = np.array([element(Pm[j], Pm[j]) for j in range(n)])
            res = np.abs(diagonal - np.einsum("jc,jc->j", L, L))
            k = int(np.argmax(res))
            Pm[[i, k]] = Pm[[k, i]]
            L[[i, k]] = L[[k, i]]
            P[[i, k]] = P[[k, i]]
            el = element(Pm[i], Pm[i])
            column = np.array([element(Pm[j], Pm[i]) for j in range(n)])
            l_ii_squared = el - L[i] @ L[i]
            l_ii = np.sqrt(l_ii_squared)
            l_ji = (column - L @ L[i, :]) / l_ii
            success = bool(success and l_ii_squared > 0.0)
            L[:, i] = l_ji
    return L[np.argsort(P)], {"success": success, "pivots": P[:rank].copy()}


def precondition_solve(chol, v, s):
    """low_rank.py:31-43: (s I + L L^T)^{-1} v via the capacitance matrix."""
    import scipy.linalg

    N, n = chol.shape
    assert n <= N, (N, n)
    U = chol / np.sqrt(s)
    V = chol.T / np.sqrt(s)
    v = v / s
    cap = scipy.linalg.cho_factor(np.eye(n) + V @ U)
    return v - U @ scipy.linalg.cho_solve(cap, V @ v)


def logpdf_cholesky(y, mean, cov):
    """util/gp_util.py:367-393."""
    import scipy.linalg

    chol = np.linalg.cholesky(cov)
    logdet = np.sum(np.log(np.diag(chol)))
    tmp = scipy.linalg.solve_triangular(chol, y - mean, lower=True)
    return -logdet - 0.5 * (tmp @ tmp) - len(mean) / 2 * np.log(2 * np.pi)


def logpdf_krylov(y, mean, *, logdet_value, solve):
    """util/gp_util.py:396-431: -logdet/2 - (y - m)^T solve(y - m)/2 - n/2 log(2 pi); `solve(b) -> (x, info)`."""
    tmp, info = solve(y - mean)
    return -logdet_value / 2 - 0.5 * ((y - mean) @ tmp) - len(mean) / 2 * np.log(2 * np.pi), info
