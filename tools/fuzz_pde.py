"""Random shapes for the (f)-2 tier: the 5-point wave operator (Neumann / Dirichlet) as a non-symmetric CSR operator against a dense assembly by
finite differences of its own action, and expm_arnoldi against scipy.linalg.expm on small grids; Hutchinson estimators against the exact trace on
small SPD matrices.    python tools/fuzz_pde.py [cases] [seed]"""
import os
import sys

import numpy as np
import scipy.linalg
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))
from matfree_extensions import hutchinson  # noqa: E402
from matfree_extensions.operators import DenseOp  # noqa: E402
from matfree_extensions.util import pde_util  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def check(name, got, ref, tol, info):
    global bad
    err = np.abs(np.asarray(got) - np.asarray(ref)).max() / max(np.abs(ref).max(), 1e-300)
    if not (err <= tol):
        bad += 1
        print(f"FAIL {name}: rel err {err:.2e} > {tol:.0e}   [{info}]", flush=True)


for case in range(cases):
    res = int(rng.integers(3, 22))
    boundary = str(rng.choice(["neumann", "dirichlet"]))
    k = int(rng.integers(1, min(2 * res * res, 30) + 1))
    dt_step = float(rng.choice([1e-3, 1e-2, 5e-2]))
    info = f"case {case}: grid {res}x{res} {boundary} k={k} dt={dt_step}"
    try:
        op, values_fn = pde_util.wave_operator(res, 1.0 / res, boundary=boundary, device=dev)
        scale = torch.tensor(rng.uniform(0.5, 1.5, (res, res)), dtype=torch.float64, device=dev, requires_grad=True)
        vals = values_fn(scale)
        n = 2 * res * res
        # the operator's matrix, column by column, from its own action on unit vectors
        E = torch.eye(n, dtype=torch.float64, device=dev)
        A = op(E, vals).T.detach().cpu().numpy()  # row b of op(E) = A e_b
        y0 = rng.standard_normal(n)
        want = scipy.linalg.expm(dt_step * A) @ y0
        out, _ = pde_util.expm_arnoldi(k)(op, dt_step, torch.tensor(y0, dtype=torch.float64, device=dev), vals)
        # Arnoldi with k steps is exact when k = n and converges fast for small dt |A|: compare where it has converged
        full = pde_util.expm_arnoldi(min(n, 60))(op, dt_step, torch.tensor(y0, dtype=torch.float64, device=dev), vals)[0]
        check("expm_arnoldi (deep) vs scipy expm", full.detach().cpu().numpy(), want, 1e-8, info)
        # adjoint identity of the k-step map: <J d, w> = <d, J^T w> with J d by a central difference in the start vector
        y = torch.tensor(y0, dtype=torch.float64, device=dev, requires_grad=True)
        o2, _ = pde_util.expm_arnoldi(k)(op, dt_step, y, vals)
        w = torch.tensor(rng.standard_normal(n), dtype=torch.float64, device=dev)
        (gy,) = torch.autograd.grad((o2 * w).sum(), y)
        dvec = torch.tensor(rng.standard_normal(n), dtype=torch.float64, device=dev)
        h = 1e-6
        with torch.no_grad():
            fp = pde_util.expm_arnoldi(k)(op, dt_step, y + h * dvec, vals)[0]
            fm = pde_util.expm_arnoldi(k)(op, dt_step, y - h * dvec, vals)[0]
        lhs = (((fp - fm) / (2 * h)) * w).sum().item()
        rhs = (gy * dvec).sum().item()
        if abs(lhs - rhs) > 1e-5 * max(abs(lhs), abs(rhs), 1e-3 * float(w.norm() * dvec.norm())):
            bad += 1
            print(f"FAIL expm_arnoldi adjoint identity: {lhs:.8e} vs {rhs:.8e}   [{info}]", flush=True)
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print(f"EXCEPTION {type(exc).__name__}: {exc}   [{info}]", flush=True)
    # ---- Hutchinson: the mean over probes of v^T A v equals trace(A) within Monte-Carlo error; with n probes = identity columns... use many probes
    n2 = int(rng.integers(2, 200))
    M = rng.standard_normal((n2, n2)); M = M @ M.T / n2 + np.eye(n2)
    Mt = torch.tensor(M, dtype=torch.float64, device=dev)
    integrand = lambda v, a: torch.einsum("...i,...i->...", v, DenseOp()(v, a))  # noqa: E731
    num = 4096
    est = hutchinson.hutchinson(integrand, hutchinson.sampler_rademacher(torch.empty(n2, dtype=torch.float64, device=dev), num=num))
    val = est(case, Mt).item()
    off = M - np.diag(np.diag(M))
    sigma = np.sqrt(2.0 * (off**2).sum() / num)  # std of the Rademacher estimator
    if abs(val - np.trace(M)) > 6.0 * sigma + 1e-9:
        bad += 1
        print(f"FAIL hutchinson: {val} vs trace {np.trace(M)} (6 sigma = {6 * sigma:.3e})   [case {case}: n={n2}]", flush=True)
    if case % 10 == 0:
        print(info + " done", flush=True)
print(f"{cases} cases, {bad} failures")
