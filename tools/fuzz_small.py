"""The on-device tridiagonal eigen-solver and the quadrature's VJP (lanczos.py:48-59 of the reference) on random tridiagonals against NumPy:
k = 1 ... 400 (the LDS-resident kernels up to 120, the deep variants beyond), batches, graded / clustered / nearly-decoupled matrices, matfun log / exp / inverse, fp64 and fp32.
    python tools/fuzz_small.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))
from matfree_extensions.lanczos import _QuadformFn  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    k = int(rng.choice([1, 2, 3, rng.integers(4, 41), rng.integers(41, 121), rng.integers(121, 200), rng.integers(200, 401)]))
    p = int(rng.choice([1, 2, 7, 64, 200]))
    if k > 120:  # (the deep variants rotate in global memory, ~1 us per rotation, and the NumPy check is O(p k^3): small batches there)
        p = min(p, 7)
    kind = str(rng.choice(["lanczos_like", "graded", "clustered", "decoupled"]))
    dtype = torch.float64 if rng.integers(0, 3) else torch.float32
    fname = str(rng.choice(["log", "exp", "inv"]))
    f = {"log": torch.log, "exp": lambda x: torch.exp(-x), "inv": lambda x: 1.0 / x}[fname]
    fn = {"log": np.log, "exp": lambda x: np.exp(-x), "inv": lambda x: 1.0 / x}[fname]
    dfn = {"log": lambda x: 1.0 / x, "exp": lambda x: -np.exp(-x), "inv": lambda x: -1.0 / x**2}[fname]
    if kind == "lanczos_like":  # T of an SPD matrix: Q^T A Q of a random orthogonal Q and a positive spectrum
        diag = np.empty((p, k)); off = np.empty((p, max(k - 1, 0)))
        for b in range(p):
            lam = rng.uniform(0.1, 1.0, k) * 10.0 ** rng.uniform(0, 4)
            Qm, _ = np.linalg.qr(rng.standard_normal((k, k)))
            A = (Qm * lam) @ Qm.T
            # Lanczos on the small matrix gives a tridiagonal with the same spectrum
            from scipy.linalg import hessenberg
            Tm = hessenberg(A)
            diag[b], off[b] = np.diag(Tm), np.abs(np.diag(Tm, 1))
    else:
        diag = rng.uniform(1.0, 2.0, (p, k))
        off = rng.uniform(0.05, 0.4, (p, max(k - 1, 0)))
        if kind == "graded":
            diag *= 10.0 ** np.linspace(3, -2, k)[None, :]
            off *= 10.0 ** np.linspace(3, -2, k)[None, : max(k - 1, 0)] * 0.3
        elif kind == "clustered":
            diag = 1.0 + 1e-6 * diag
            off *= 1e-7
        elif kind == "decoupled" and k > 2:
            off[:, k // 2 - 1] = 1e-14
        # keep it positive definite (diagonal dominance)
        pad = np.zeros((p, k)); pad[:, :-1] += off; pad[:, 1:] += off
        diag = np.maximum(diag, pad * 1.1 + 1e-3 * diag)
    info = f"case {case}: k={k} p={p} {kind} {fname} {str(dtype).split('.')[-1]}"
    dt = torch.tensor(diag, dtype=dtype, device=dev, requires_grad=True)
    ot = torch.tensor(off, dtype=dtype, device=dev, requires_grad=True)
    try:
        value, evals, evecs = _QuadformFn.apply(f, dt, ot)
        gout = rng.standard_normal(p)
        gd, go = torch.autograd.grad(value, (dt, ot), torch.tensor(gout, dtype=dtype, device=dev), allow_unused=True)
        d64, o64 = dt.detach().double().cpu().numpy(), ot.detach().double().cpu().numpy()   # what the device saw (fp32 inputs are rounded)
        worst = [0.0, 0.0, 0.0]
        for b in range(p):
            Tm = np.diag(d64[b]) + np.diag(o64[b], 1) + np.diag(o64[b], -1)
            lam, U = np.linalg.eigh(Tm)
            val = U[0] ** 2 @ fn(lam)
            ev = np.sort(evals[b].double().cpu().numpy())
            dl = lam[:, None] - lam[None, :]
            with np.errstate(divide="ignore", invalid="ignore"):
                F = (fn(lam)[:, None] - fn(lam)[None, :]) / dl
            close = np.abs(dl) <= 1e-13 * np.maximum(np.abs(lam[:, None]), np.abs(lam[None, :]))
            F = np.where(close, 0.5 * (dfn(lam)[:, None] + dfn(lam)[None, :]), F)
            G = U @ (F * np.outer(U[0], U[0])) @ U.T
            worst[0] = max(worst[0], np.abs(ev - lam).max() / np.abs(lam).max())
            worst[1] = max(worst[1], abs(value[b].item() - val) / max(np.abs(fn(lam)).max(), 1e-300))  # (against the scale of f on the spectrum: the weighted sum may cancel)
            gref = np.concatenate([np.diag(G), 2.0 * np.diag(G, 1)]) * gout[b]
            ggot = np.concatenate([gd[b].double().cpu().numpy(), go[b].double().cpu().numpy() if k > 1 else np.zeros(0)])
            worst[2] = max(worst[2], np.abs(ggot - gref).max() / max(np.abs(F).max() * abs(gout[b]), 1e-300))
        eps = 2.3e-16 if dtype == torch.float64 else 1.2e-7
        # eigenvalues to a few ulps of the norm; the quadrature and its gradient amplify by the conditioning of f on the spectrum
        tols = (50 * eps * k, 200 * eps * k, 2000 * eps * k)
        # the quadrature and its gradient are compared where f is well conditioned on the spectrum as the device sees it: exp(-lambda) of a
        # spectrum reaching 1e3 and log of eigenvalues within 1e-6 of 1 turn an ulp of lambda into O(1) of f (first runs of this script)
        # (and clustered spectra make the divided differences (f(a) - f(b)) / (a - b) lose digits on BOTH sides of the comparison)
        benign = kind != "clustered" and (fname in ("inv", "log") or np.abs(d64).max() < 20.0)
        if not (worst[0] <= tols[0] and (not benign or (worst[1] <= tols[1] and worst[2] <= tols[2]))):
            bad += 1
            print(f"FAIL eigenvalues {worst[0]:.1e} (tol {tols[0]:.0e}), value {worst[1]:.1e} ({tols[1]:.0e}), gradient {worst[2]:.1e} ({tols[2]:.0e})   [{info}]", flush=True)
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print(f"EXCEPTION {type(exc).__name__}: {exc}   [{info}]", flush=True)
    if case % 25 == 0:
        print(info + " done", flush=True)
print(f"{cases} cases, {bad} failures")
