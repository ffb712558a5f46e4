"""Random shapes of the Krylov drivers against the NumPy oracle, fp64 (tight tolerances: a wrong index shows, rounding does not) -- a wider
net than the parametrised cases of tests/test_gpu_parity.py; found nothing or something, either way the log says which shapes ran.
Not collected by pytest (no test_ prefix): a checker script that lives under tests/ because it calls the oracle.

    python tests/fuzz_krylov.py [cases] [seed]

Per case, at random: operator kind (dense symmetric, dense non-symmetric, CSR, RBF Gram, Python callable), n, depth k, probes p, reortho;
  * arnoldi.hessenberg forward (Q, H, r, c) and its adjoint under random cotangents        (arnoldi.py:57-101, :104-220)
  * lanczos.tridiag forward and adjoint, reortho full / none                                (lanczos.py:152-169, :215-335)
  * lanczos.integrand_spd(log) value and gradient, batched over p probes                    (lanczos.py:14-61)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in (ROOT, os.path.join(ROOT, "experiments-lanczos-adjoints_amd")):
    if q not in sys.path:
        sys.path.insert(0, q)
from matfree_extensions import arnoldi, lanczos  # noqa: E402
from matfree_extensions.operators import CsrOp, DenseOp, RbfGramOp  # noqa: E402
from oracle import slq_oracle as orc  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
T = lambda x, g=False: torch.tensor(np.asarray(x), dtype=torch.float64, device=dev, requires_grad=g)  # noqa: E731
N = lambda t: t.detach().cpu().numpy()  # noqa: E731
bad = 0


def check(name, got, ref, tol, info):
    global bad
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300) if ref.size else 0.0
    if not (err <= tol):
        bad += 1
        print(f"FAIL {name}: rel err {err:.2e} > {tol:.0e}   [{info}]", flush=True)


def make_operator(kind, n):
    """-> (oracle operator, native / callable matvec, oracle params, torch params)"""
    if kind == "dense_sym":
        A = orc.symmetric_matrix_from_eigenvalues(rng.uniform(0.5, 3.0, n), seed=int(rng.integers(1 << 30)))
        return orc.DenseOp(), DenseOp(), (A,), [T(A, True)]
    if kind == "dense_nonsym":
        A = np.eye(n) * 2.0 + rng.standard_normal((n, n)) / np.sqrt(n)
        return orc.DenseOp(), DenseOp(), (A,), [T(A, True)]
    if kind == "callable":
        A = orc.symmetric_matrix_from_eigenvalues(rng.uniform(0.5, 3.0, n), seed=int(rng.integers(1 << 30)))
        return orc.DenseOp(), (lambda v, a: v @ a.T if v.dim() == 2 else a @ v), (A,), [T(A, True)]
    if kind == "csr":
        nnz_row = int(rng.integers(1, 6))
        rows = np.repeat(np.arange(n), nnz_row)
        cols = rng.integers(0, n, size=rows.size)
        vals = rng.standard_normal(rows.size) * 0.3
        r = np.concatenate([rows, cols, np.arange(n)])
        c = np.concatenate([cols, rows, np.arange(n)])
        v = np.concatenate([vals, vals, np.full(n, 2.0 * nnz_row)])  # symmetric, diagonally dominant
        op, vt, order = CsrOp.from_coo(r, c, v, n, dev)
        o = orc.CooOp(r[order.numpy()], c[order.numpy()], n)
        return o, op, (v[order.numpy()],), [vt.double().requires_grad_(True)]
    d = int(rng.integers(1, 12))
    X = rng.standard_normal((n, d))
    ard = bool(rng.integers(0, 2))
    raw = (rng.standard_normal(d) * 0.2 + 0.6 if ard else np.array(0.6), np.array(0.3), np.array(-1.0))
    kernel = str(rng.choice(["rbf", "matern32"]))
    o = orc.RbfGramOp(X, noise_minval=1e-4, kernel=kernel, eps=float(np.finfo(np.float64).eps))
    return o, RbfGramOp(T(X), noise_minval=1e-4, kernel=kernel), raw, [T(r, True) for r in raw]


for case in range(cases):
    kind = str(rng.choice(["dense_sym", "dense_nonsym", "callable", "csr", "rbf"]))
    n = int(rng.choice([rng.integers(2, 40), rng.integers(40, 700), rng.integers(700, 2600)]))
    if kind in ("dense_sym", "dense_nonsym", "callable"):
        n = min(n, 900)
    k = int(rng.integers(1, min(n, 24) + 1))
    if os.environ.get("FUZZ_DEEP") and n > 200:  # the depths of the reference's SuiteSparse sweeps (benchmark.py:21,83: up to 50; its plots: 150)
        k = int(rng.integers(25, min(n, 160)))
    # without re-orthogonalisation the recurrences (and their adjoints) amplify rounding with the depth -- two correct implementations
    # then differ by far more than any tolerance a wrong index would exceed (first run of this script: every disagreement was there, or
    # in the remainder r at k = n, which is rounding noise by construction): shallow depths only, and r only where it is not ~ 0
    k_none = min(k, 4)
    info = f"case {case}: {kind} n={n} k={k}"
    o, mv, oparams, tparams = make_operator(kind, n)
    v = rng.standard_normal(n)
    try:
        # ---- arnoldi.hessenberg: forward + adjoint under random cotangents --------------------------------------------------------
        reortho = str(rng.choice(["full", "none"]))
        kk = k if reortho == "full" else k_none
        vt = T(v, True)
        Q, H, r, c = arnoldi.hessenberg(mv, kk, reortho=reortho)(vt, *tparams)
        Qo, Ho, ro, co = orc.arnoldi_forward(o, kk, v, *oparams, reortho=reortho)
        sub = np.abs(np.diag(Ho, -1))
        if sub.size and sub.min() < 1e-7 * np.abs(Ho).max():
            # (the Krylov space has found an invariant subspace -- a kernel matrix of low numerical rank: what follows a beta ~ 0 is normalised
            #  round-off in any implementation)
            print(info + ": near-breakdown (beta ~ 0), skipped", flush=True)
            continue
        check("hessenberg Q", N(Q), Qo, 5e-8, info + " " + reortho)
        check("hessenberg H", N(H), Ho, 1e-8, info + " " + reortho)
        if kk < n and np.abs(ro).max() > 1e-6 * np.abs(Ho).max():
            check("hessenberg r", N(r), ro, 1e-7, info + " " + reortho)
        check("hessenberg c", N(c), co, 1e-10, info + " " + reortho)
        cot = [rng.standard_normal(np.shape(x)) for x in (Qo, Ho, ro, co)]
        grads = torch.autograd.grad((Q, H, r, c), (vt, *tparams), [T(x) for x in cot], allow_unused=True)
        dv_o, dp_o = orc.arnoldi_adjoint(o, oparams, Q=Qo, H=Ho, r=ro, c=co, dQ=cot[0], dH=cot[1], dr=cot[2], dc=float(cot[3]), reortho=reortho)
        check("hessenberg adjoint dv", N(grads[0]), dv_o, 1e-6, info + " " + reortho)
        for g, go in zip(grads[1:], dp_o):
            check("hessenberg adjoint dparam", N(g).reshape(np.shape(go)), go, 1e-6, info + " " + reortho)
        # ---- the same forward pass on the fp32 kernels (other vector widths, other slice geometry): shallow depth, loose bar ----------------
        if kk <= 8 and kind in ("dense_sym", "dense_nonsym", "csr"):
            f32 = lambda t: t.detach().to(torch.float32)  # noqa: E731
            Q32, H32, r32, c32 = arnoldi.hessenberg(mv, kk, reortho=reortho)(f32(vt), *[f32(t) for t in tparams])
            check("hessenberg Q (fp32 kernels)", N(Q32.double()), Qo, 2e-4 * kk, info + " " + reortho)
            check("hessenberg H (fp32 kernels)", N(H32.double()), Ho, 2e-4 * kk, info + " " + reortho)
        # ---- lanczos.tridiag on the symmetric operators -----------------------------------------------------------------------------
        if kind != "dense_nonsym":
            reortho = str(rng.choice(["full", "none"]))
            kk = k if reortho == "full" else k_none
            vt = T(v / np.linalg.norm(v), True)
            (B, (a, b)), (q, br) = lanczos.tridiag(mv, kk, reortho=reortho)(vt, *tparams)
            (Bo, (ao, bo)), (qo, bro) = orc.tridiag(o, kk, v / np.linalg.norm(v), *oparams, reortho=reortho)
            tol = 1e-8 if reortho == "full" else 1e-6  # (the three-term recurrence amplifies rounding with the depth)
            check("tridiag diag", N(a), ao, tol, info + " " + reortho)
            check("tridiag offdiag", N(b), bo, tol, info + " " + reortho)
            check("tridiag basis", N(B), Bo, tol * 10, info + " " + reortho)
        # ---- integrand_spd(log), p probes at once (SPD operators only) ---------------------------------------------------------------
        # (k < n and n >= 4: at full depth a +-1 probe can be an exact eigenvector of a tiny symmetric matrix -- breakdown, beta = 0 --
        #  where the reference yields inf / NaN and every implementation its own garbage)
        if kind in ("dense_sym", "callable", "rbf", "csr") and n >= 4 and k < n:
            p = int(rng.choice([1, 2, 5, 8, 33]))
            probes = np.where(rng.random((p, n)) < 0.5, -1.0, 1.0)
            f = lanczos.integrand_spd(torch.log, k, mv)
            vals = f(T(probes), *tparams)
            gr = torch.autograd.grad(vals.sum(), tparams)
            vo = np.zeros(p)
            go = [np.zeros(np.shape(x)) for x in oparams]
            for bidx in range(p):
                val, _, dps = orc.integrand_spd_value_and_grad(o, k, probes[bidx], oparams)
                vo[bidx] = val
                for acc, dpx in zip(go, dps):
                    acc += np.asarray(dpx).reshape(acc.shape)
            check("integrand_spd values", N(vals), vo, 1e-8, info + f" p={p}")
            for g, gref in zip(gr, go):
                check("integrand_spd gradient", N(g).reshape(gref.shape), gref, 1e-5, info + f" p={p}")
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print(f"EXCEPTION {type(exc).__name__}: {exc}   [{info}]", flush=True)
    if case % 10 == 0:
        print(info + " done", flush=True)
print(f"{cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
