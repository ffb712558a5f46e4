"""Row-sharded restatement of the oracle's SLQ value-and-gradient (test infrastructure, CPU, fp64).

Same arithmetic as oracle/slq_oracle.py (arnoldi_forward :235-266, arnoldi_adjoint :274-310, the RBF parameter sweep
:198-227), but every n-vector is a ROW SHARD and the only cross-rank operations are the two the MI355X drivers use
(include/mfx.h, `mfx_comm`): an all-gather of the iterate before each operator application, and sum-all-reduces of
inner products / partial sums.  Run under gloo it shows that this decomposition -- and the host logic of
matfree_extensions.distributed around it (RowComm, make_grid, shard_probes, reduce_estimate) -- reproduces the
single-process oracle.
"""

import numpy as np
import torch

from oracle import slq_oracle as orc


class ShardedRbf:
    def __init__(self, X, comm, noise_minval=0.0):
        self.X, self.comm, self.noise_minval = np.asarray(X), comm, noise_minval
        self.rows = slice(comm.row0, comm.row0 + comm.nrows)

    def allsum(self, x):
        t = torch.as_tensor(np.atleast_1d(np.asarray(x, dtype=np.float64)).copy())
        return self.comm.all_reduce_(t).numpy().reshape(np.shape(x))

    def gather(self, x_local):
        return self.comm.gather_rows(torch.as_tensor(np.ascontiguousarray(x_local))).numpy()

    def constrained(self, raw_l, raw_s, raw_noise):
        return orc.softplus(raw_l), orc.softplus(raw_s), self.noise_minval + orc.softplus(raw_noise)

    def apply_rows(self, v_local, raw_l, raw_s, raw_noise):
        """this rank's rows of (K + noise I) v from the row shard of v"""
        ls, s, noise = self.constrained(raw_l, raw_s, raw_noise)
        K = orc.kernel_matrix("rbf", self.X[self.rows], self.X, ls, s, diag_offset=self.comm.row0)
        return K @ self.gather(v_local) + noise * v_local

    def param_vjp_rows(self, V_local, C_local, raw_l, raw_s, raw_noise):
        """partial sums over this rank's rows of d/d(raw) sum_b cot_b^T A v_b (scalar lengthscale); complete after allsum"""
        ls, s, noise = self.constrained(raw_l, raw_s, raw_noise)
        Xr = self.X[self.rows]
        K = orc.kernel_matrix("rbf", Xr, self.X, ls, s, diag_offset=self.comm.row0)
        V_full = np.stack([self.gather(v) for v in V_local])
        S = C_local.T @ V_full  # (nrows, n)
        g_s = (S * K).sum() / s
        W = S * (s * orc.kernel_lengthscale_weight("rbf", Xr, self.X, ls, diag_offset=self.comm.row0))
        sqn = (self.X * self.X).sum(-1)
        diff2 = np.maximum(0.0, sqn[self.rows, None] + sqn[None, :] - 2.0 * Xr @ self.X.T)
        diff2[np.arange(self.comm.nrows), np.arange(self.comm.row0, self.comm.row0 + self.comm.nrows)] = 0.0
        g_l = (W * diff2).sum() / ls**3
        g_n = float((C_local * V_local).sum())
        g = self.allsum(np.array([g_l, g_s, g_n]))
        return g[0] * orc.softplus_grad(raw_l), g[1] * orc.softplus_grad(raw_s), g[2] * orc.softplus_grad(raw_noise)


def arnoldi_forward(op, k, v, *params):
    """oracle arnoldi_forward (second pass on) on row shards: Q (nrows, k), H replicated, r (nrows,), c"""
    nrows = v.shape[0]
    Q = np.zeros((nrows, k))
    H = np.zeros((k, k))
    length0 = np.sqrt(op.allsum(v @ v))
    length, w = length0, v
    for i in range(k):
        q = w / length
        Q[:, i] = q
        w = op.apply_rows(q, *params)
        h = op.allsum(Q.T @ w)
        w = w - Q @ h
        w = w - Q @ op.allsum(Q.T @ w)
        length = np.sqrt(op.allsum(w @ w))
        if i + 1 < k:
            h[i + 1] = length
        H[:, i] = h
    return Q, H, w, 1.0 / length0


def arnoldi_adjoint(op, params, *, Q, H, r, c, dH):
    """oracle arnoldi_adjoint (reortho full) for dQ = 0, dr = 0, dc = 0 -- the SLQ cotangents -- on row shards"""
    nrows, k = Q.shape
    lower_mask = np.tril(np.ones((k, k))) - 0.5 * np.eye(k)
    ps_mask = np.tril(np.ones((k, k)), 1)
    eta = dH[:, -1].copy()
    lam = Q @ eta
    Lam = np.zeros_like(Q)
    Gam = np.zeros((k, k))
    Pi_xi = np.outer(eta, r)
    Pi_gamma = H @ dH.T
    beta_minus = np.concatenate([np.ones(1), np.diag(H, -1)])
    alpha = np.diag(H)
    beta_plus = H - np.diag(np.diag(H)) - np.diag(np.diag(H, -1), -1)
    P = Q.T.copy()
    pairs_v, pairs_cot = [], []
    for idx in range(k - 1, -1, -1):
        P = ps_mask[idx][:, None] * P
        lam = lam - P.T @ op.allsum(P @ lam) + P.T @ (ps_mask[idx] * dH[:, idx])
        z = op.apply_rows(lam, *params)  # symmetric operator
        pairs_v.append(Q[:, idx])
        pairs_cot.append(lam)
        Gam[idx, :] = lower_mask[idx] * (Pi_gamma[idx] - op.allsum(z @ Q))
        Lam[:, idx] = lam
        xi = Pi_xi[idx] + (Gam + Gam.T)[idx, :] @ Q.T
        lam = (xi - (alpha[idx] * lam - z) - beta_plus[idx] @ Lam.T) / beta_minus[idx]
    return lam * c, op.param_vjp_rows(np.stack(pairs_v), np.stack(pairs_cot), *params)


def integrand_value_and_grad(op, k, v0_local, params):
    """lanczos.integrand_spd(log, k, A)(v0) and d/dparams from the row shard of v0"""
    scale = np.sqrt(op.allsum(v0_local @ v0_local))
    u = v0_local / scale
    Q, H, r, c = arnoldi_forward(op, k, u, *params)
    T = 0.5 * (H + H.T)
    g, (ddiag, doff), _ = orc.quadform_from_tridiag(np.diag(T).copy(), np.diag(T, 1).copy(), "log")
    dH = np.diag(ddiag) + 0.5 * (np.diag(doff, 1) + np.diag(doff, -1))
    _, dparams = arnoldi_adjoint(op, params, Q=Q, H=H, r=r, c=c, dH=dH)
    return scale**2 * g, tuple(scale**2 * d for d in dparams)


# ---- the three-term recurrence and its adjoint on row shards (oracle tridiag_none :345-362, tridiag_none_vjp :365-390) ------------
def tridiag_none(op, k, v, *params):
    """xs (k + 1, nrows) row shards, a, b (k,) replicated -- the layout of mfx_lanczos_forward_sharded"""
    nrows = v.shape[0]
    xs = np.zeros((k + 1, nrows))
    a, b = np.zeros(k), np.zeros(k)
    vnorm = np.sqrt(op.allsum(v @ v))
    xs[0] = v / vnorm
    prev, bprev = np.zeros_like(v), 0.0
    for i in range(k):
        w = op.apply_rows(xs[i], *params)
        a[i] = op.allsum(xs[i] @ w)
        rr = w - a[i] * xs[i] - bprev * prev
        b[i] = np.sqrt(op.allsum(rr @ rr))
        xs[i + 1] = rr / b[i]
        prev, bprev = xs[i], b[i]
    return xs, a, b, vnorm


def tridiag_none_vjp(op, k, params, xs, a, b, vnorm, dxs, da, db):
    """cotangents dxs (k + 1, nrows) local, da, db (k,) replicated -> (dv local, parameter gradients complete)"""
    xi = -dxs[-1]
    lam_plus = np.zeros_like(xi)
    lams, cots = [], []
    for j in range(k - 1, -1, -1):
        xplus, x = xs[j + 1], xs[j]
        xi = xi / b[j]
        d = op.allsum(np.array([lam_plus @ x, xplus @ xi, x @ xi]))  # the three dots of a step: ONE all-reduce
        mu = db[j] - d[0] + d[1]
        nu = da[j] + d[2]
        lam = -xi + mu * xplus + nu * x
        Alam = op.apply_rows(lam, *params)
        lams.append(lam)
        cots.append(x)
        xi = -dxs[j] - Alam + a[j] * lam + b[j] * lam_plus - b[j] * nu * xplus
        lam_plus = lam
    dvec = (op.allsum(xi @ xs[0]) * xs[0] - xi) / vnorm
    return dvec, op.param_vjp_rows(np.stack(lams), np.stack(cots), *params)


# ---- conjugate gradients on row shards (oracle pcg_fixed_step :587-598, pcg_adaptive :601-618) ----------------------------------------
def pcg(op, b, params, *, Lt=None, maxiter, adaptive=False, atol=1.0, rtol=0.0, miniter=0):
    """b, x, r: row shards; Lt: this rank's columns of L^T (rank, nrows) of the Woodbury preconditioner, or None.  Every scalar is an
    all-reduced sum, so every rank takes the same steps -- the layout of mfx_pcg_solve_sharded."""
    _, _, noise = op.constrained(*params)
    if Lt is not None:
        r_ = Lt.shape[0]
        minv = np.linalg.inv(noise * np.eye(r_) + op.allsum(Lt @ Lt.T))

    def P(v):
        if Lt is None:
            return v
        return (v - Lt.T @ (minv @ op.allsum(Lt @ v))) / noise

    x = np.zeros_like(b)
    r = b.copy()
    z = P(r)
    p = z
    rz = op.allsum(r @ z)
    steps = 0
    for _ in range(maxiter):
        if adaptive:
            err = op.allsum(((r / (atol + np.abs(x) * rtol)) ** 2).sum())
            if not ((np.sqrt(err / op.comm.n) > 1.0 or steps < miniter) and steps < maxiter):
                break
        Ap = op.apply_rows(p, *params)
        alpha = orc.safe_divide(rz, op.allsum(p @ Ap))
        x = x + alpha * p
        r = r - alpha * Ap
        z = P(r)
        rz_new = op.allsum(r @ z)
        p = z + orc.safe_divide(rz_new, rz) * p
        rz = rz_new
        steps += 1
    return x, r, steps

