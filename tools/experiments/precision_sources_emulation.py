"""Which fp32 rounding of the SLQ value-and-gradient pipeline is the gradient's error made of?  A CPU emulation (NumPy, fp64 arithmetic
with selected quantities rounded to fp32 where the HIP pipeline stores fp32), small enough to run in the development container:
dense RBF Gram matrix, n points in d = 8 dimensions, lengthscale 2, noise scaled so that cond(K + noise I) matches BASELINE config 4
(2.6e5), k = 40 fully re-orthogonalised Lanczos steps through the Arnoldi recurrences of the reference (arnoldi.py:57-101 forward,
:104-220 adjoint, restated here batched over the probes), 64 +-1 probes.  One switch per stored quantity; output = relative error of
the log-det mean and of its gradient (lengthscale, outputscale, noise) against the all-fp64 run of the same probes.

    python tools/experiments/precision_sources_emulation.py [n] [seeds]
    python tools/experiments/precision_sources_emulation.py [n] [seeds] per-probe     (only the shipped arithmetic, error of every probe's own share)

Self-contained on purpose (no import of oracle/ or of the package): a measurement script, not a checker.  Result (round 5, n = 8192):
tools/experiments/precision_sources_emulation.log, DESIGN.md section 3.2.
"""
import sys
import time

import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
seeds = [int(s) for s in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3]
d, k, p = 8, 40, 64
ell, sigma = 2.0, 1.0
noise = 0.1 * n / 131072.0  # lambda_max ~ 0.2 n: the same condition number as config 4


def f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float64)


def ident(x):
    return x


def bits22(x):
    """every element rounded to 22 significant bits: what the hi + lo f16 split of the f16x3 kernels keeps of an operand"""
    m, e = np.frexp(np.asarray(x, dtype=np.float64))
    return np.ldexp(np.round(m * 4194304.0) / 4194304.0, e)


rng = np.random.default_rng(4)
X = rng.standard_normal((n, d))
sq = (X * X).sum(1)
D2 = np.maximum(sq[:, None] + sq[None, :] - 2.0 * X @ X.T, 0.0) / ell**2
K = sigma * np.exp(-0.5 * D2)
G_l = K * D2 / ell  # dK/d lengthscale
del D2
K22 = bits22(K)  # the Gram entries as the f16x3 matvec sees them (its products are exact, its sums fp32: modelled by the 'w' rounding)


def run(probes, R):
    """R: dict of rounding functions: 'w' matvec output, 'q' stored basis / orthogonalised vectors, 'h' Gram-Schmidt coefficients and H,
    'len' norms, 'lam' stored adjoint states and the adjoint's vectors, 'z' transposed-matvec output, 'g' small adjoint matrices"""
    rw, rq, rh, rl = R.get("w", ident), R.get("q", ident), R.get("h", ident), R.get("len", ident)
    rlam, rz, rg = R.get("lam", ident), R.get("z", ident), R.get("g", ident)
    Kf, vin_f = (K22, bits22) if R.get("opf") else (K, ident)   # operand quantisation of the forward matvec
    Ka, vin_a = (K22, bits22) if R.get("opa") else (K, ident)   # ... of the adjoint's matvec
    rgemm = bits22 if R.get("gemm") else ident                   # ... of the gradient GEMM's operands
    P = probes.shape[0]
    scale = np.linalg.norm(probes, axis=1)
    u = probes / scale[:, None]
    Q = np.zeros((P, k, n))
    H = np.zeros((P, k, k))  # H[b][:, i] = h of step i  -> stored as H[b, j, i]
    w = u.copy()
    length = np.ones(P)
    for i in range(k):
        q = rq(w / length[:, None])
        Q[:, i] = q
        w = rw(vin_f(q) @ Kf + noise * q)
        h = rh(np.einsum("pjn,pn->pj", Q, w))
        w = rq(w - np.einsum("pj,pjn->pn", h, Q))
        h2 = rh(np.einsum("pjn,pn->pj", Q, w))
        w = rq(w - np.einsum("pj,pjn->pn", h2, Q))
        length = rl(np.linalg.norm(w, axis=1))
        if i + 1 < k:
            h[:, i + 1] = length
        H[:, :, i] = h
    r = w
    c = np.ones(P)  # |u| = 1
    T = 0.5 * (H + H.transpose(0, 2, 1))
    diag = np.einsum("pii->pi", T).copy()
    off = np.stack([np.diagonal(T[b], 1) for b in range(P)])
    vals = np.zeros(P)
    dH = np.zeros((P, k, k))
    for b in range(P):
        Tm = np.diag(diag[b]) + np.diag(off[b], 1) + np.diag(off[b], -1)
        lam, U = np.linalg.eigh(Tm)
        fl = np.log(lam)
        u0 = U[0]
        vals[b] = u0 @ (fl * u0)
        dl = lam[:, None] - lam[None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            F = (fl[:, None] - fl[None, :]) / dl
        close = np.abs(dl) <= 1e-13 * np.maximum(np.abs(lam[:, None]), np.abs(lam[None, :]))
        F = np.where(close, 0.5 * (1 / lam[:, None] + 1 / lam[None, :]), F)
        Gm = U @ (F * np.outer(u0, u0)) @ U.T
        dd, do = np.diag(Gm), 2.0 * np.diag(Gm, 1)
        dH[b] = np.diag(dd) + 0.5 * (np.diag(do, 1) + np.diag(do, -1))
    dH = rg(dH)
    # ---- adjoint (dQ = 0, dr = 0, dc = 0) ----------------------------------------------------------------------------------------
    tril = np.tril(np.ones((k, k)))
    lower = tril - 0.5 * np.eye(k)
    ps_mask = np.tril(np.ones((k, k)), 1)
    eta = dH[:, :, -1].copy()
    lam_v = rlam(np.einsum("pj,pjn->pn", eta, Q))
    Lam = np.zeros((P, k, n))
    Gam = np.zeros((P, k, k))
    Pi_xi = eta[:, :, None] * r[:, None, :]  # (P, k, n)
    Pi_gamma = rg(np.einsum("pij,pkj->pik", H, dH))  # H @ dH^T
    beta_minus = np.concatenate([np.ones((P, 1)), np.stack([np.diagonal(H[b], -1) for b in range(P)])], axis=1)
    alpha = np.einsum("pii->pi", H)
    beta_plus = H.copy()
    for b in range(P):
        beta_plus[b] -= np.diag(np.diag(H[b])) + np.diag(np.diag(H[b], -1), -1)
    g_l = np.zeros(P)
    g_s = np.zeros(P)
    g_n = np.zeros(P)
    for idx in range(k - 1, -1, -1):
        m = ps_mask[idx]
        Pm = Q * m[None, :, None]
        pvec = m[None, :] * dH[:, :, idx]
        coef = rh(np.einsum("pjn,pn->pj", Pm, lam_v))
        lam_v = rlam(lam_v - np.einsum("pj,pjn->pn", coef - pvec, Pm))
        q = Q[:, idx]
        Kl = vin_a(lam_v) @ Ka
        z = rz(Kl + noise * lam_v)
        # parameter gradients of lam^T A(theta) q  (the deferred sweep): exact arithmetic on the STORED lam and q
        lg, qg = rgemm(lam_v), rgemm(q)
        g_s += np.einsum("pn,pn->p", lg @ K, qg) / sigma
        g_n += np.einsum("pn,pn->p", lam_v, q)
        g_l += np.einsum("pn,pn->p", lg @ G_l, qg)
        zq = rh(np.einsum("pn,pjn->pj", z, Q))
        Gam[:, idx, :] = rg(lower[idx][None, :] * (Pi_gamma[:, idx, :] - zq))
        Lam[:, idx] = lam_v
        gsym = (Gam + Gam.transpose(0, 2, 1))[:, idx, :]
        xi = Pi_xi[:, idx] + np.einsum("pj,pjn->pn", gsym, Q)
        lam_v = rlam((xi - (alpha[:, idx, None] * lam_v - z) - np.einsum("pj,pjn->pn", beta_plus[:, idx, :], Lam)) / beta_minus[:, idx, None])
    s2 = scale**2
    # (the scale^2 weights: all probes have the same norm sqrt(n))
    value = np.mean(s2 * vals)
    per_probe = np.stack([g_l, g_s, g_n]) * s2[0]  # (3, P): every probe's own gradient
    grad = per_probe.sum(1) / P
    if R.get("per_probe"):
        return value, grad, per_probe
    return value, grad


configs = [
    ("all fp32 (what the HIP pipeline stores)", dict(w=f32, q=f32, h=f32, len=f32, lam=f32, z=f32, g=f32)),
    ("only matvec outputs w, z", dict(w=f32, z=f32)),
    ("only stored basis / orthogonalised vectors q", dict(q=f32)),
    ("only Gram-Schmidt coefficients and H", dict(h=f32)),
    ("only norms", dict(len=f32)),
    ("only adjoint states lam", dict(lam=f32)),
    ("only small adjoint matrices (dH, Pi, Gamma)", dict(g=f32)),
    ("all but coefficients / H / norms / small matrices", dict(w=f32, q=f32, lam=f32, z=f32)),
    ("all but matvec outputs", dict(q=f32, h=f32, len=f32, lam=f32, g=f32)),
    ("only 22-bit operands, forward matvec", dict(opf=True)),
    ("only 22-bit operands, adjoint matvec", dict(opa=True)),
    ("only 22-bit operands, gradient GEMM", dict(gemm=True)),
    ("22-bit operands everywhere + fp32 outputs (operator only)", dict(opf=True, opa=True, gemm=True, w=f32, z=f32)),
    ("shipped: all fp32 + 22-bit operands", dict(w=f32, q=f32, h=f32, len=f32, lam=f32, z=f32, g=f32, opf=True, opa=True, gemm=True)),
    ("proposed: vectors fp32 + 22-bit operands, small quantities fp64", dict(w=f32, q=f32, lam=f32, z=f32, opf=True, opa=True, gemm=True)),
]
if len(sys.argv) > 3 and sys.argv[3] == "per-probe":
    shipped = dict(configs[-2][1], per_probe=True)
    print(f"n = {n}, cond ~ {0.2 * n / noise:.2e}, shipped arithmetic (all fp32 + 22-bit operands): error of each probe's OWN gradient, relative to the mean gradient")
    print(f"{'seed':>4} {'set: d l':>9} {'d s':>9} {'d noise':>9} | per-probe |d l| error: {'median':>9} {'90 %':>9} {'max':>9} {'(probe)':>7} | cond * eps(fp32) = {0.2 * n / noise * 6e-8:.1e}")
    for seed in seeds:
        probes = np.where(np.random.default_rng(1000 + seed).random((p, n)) < 0.5, -1.0, 1.0)
        v0, g0, pp0 = run(probes, dict(per_probe=True))
        v, g, pp = run(probes, shipped)
        e = np.abs(g - g0) / np.abs(g0)
        pe = np.abs(pp[0] - pp0[0]) / np.abs(g0[0])
        print(f"{seed:>4} {e[0]:9.2e} {e[1]:9.2e} {e[2]:9.2e} | {'':>24} {np.median(pe):9.2e} {np.quantile(pe, 0.9):9.2e} {pe.max():9.2e} {int(pe.argmax()):>7}", flush=True)
    sys.exit(0)
print(f"n = {n}, d = {d}, k = {k}, {p} probes, noise = {noise:.5f} (cond ~ {0.2 * n / noise:.2e}); relative errors against the all-fp64 run")
print(f"{'rounded to fp32':<52} {'seed':>4} {'value':>9} {'d l':>9} {'d s':>9} {'d noise':>9}")
for seed in seeds:
    probes = np.where(np.random.default_rng(1000 + seed).random((p, n)) < 0.5, -1.0, 1.0)
    t0 = time.time()
    v0, g0 = run(probes, {})
    for name, R in configs:
        v, g = run(probes, R)
        e = np.abs(g - g0) / np.abs(g0)
        print(f"{name:<52} {seed:>4} {abs(v - v0) / abs(v0):9.2e} {e[0]:9.2e} {e[1]:9.2e} {e[2]:9.2e}", flush=True)
    print(f"   (seed {seed}: {time.time() - t0:.0f} s; fp64 value {v0:.6f}, grad {g0})", flush=True)
