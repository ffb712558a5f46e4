"""Random shapes of the linear-solve tier (cg.py, low_rank.py of the reference) against the NumPy oracle, fp64: pivoted partial Cholesky of a kernel
Gram matrix, the Woodbury preconditioner built on it, (P)CG with a fixed number of steps, batched right-hand sides; and CG in the fp32 modes against
the fp64 HIP path at larger n.    python tests/fuzz_gp.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in (ROOT, os.path.join(ROOT, "experiments-lanczos-adjoints_amd")):
    if q not in sys.path:
        sys.path.insert(0, q)
from matfree_extensions import cg, low_rank  # noqa: E402
from matfree_extensions.operators import RbfGramOp  # noqa: E402
from oracle import slq_oracle as orc  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
T = lambda x, dt=torch.float64: torch.tensor(np.asarray(x), dtype=dt, device=dev)  # noqa: E731
N = lambda t: t.detach().double().cpu().numpy()  # noqa: E731
bad = 0


def check(name, got, ref, tol, info):
    global bad
    err = np.abs(np.asarray(got) - np.asarray(ref)).max() / max(np.abs(ref).max(), 1e-300)
    if not (err <= tol):
        bad += 1
        print(f"FAIL {name}: rel err {err:.2e} > {tol:.0e}   [{info}]", flush=True)


for case in range(cases):
    kernel = str(rng.choice(["rbf", "matern32", "matern12"]))
    d = int(rng.integers(1, 13))
    if os.environ.get("FUZZ_WIDE_D"):  # up to the wide kernels (d > 32); the inputs below are scaled so that distances stay O(1)
        d = int(rng.integers(13, 100))
    if case % 3 != 2:
        n = int(rng.choice([rng.integers(8, 100), rng.integers(100, 1500)]))
        rank = int(rng.integers(1, min(n, 48) + 1))
        nrhs = int(rng.choice([1, 2, 7, 16]))
        steps = int(rng.integers(1, 7))  # (CG amplifies a 1e-16 difference ~ 15x per step at these condition numbers: first run of this script)
        info = f"case {case}: {kernel} n={n} d={d} rank={rank} rhs={nrhs} steps={steps}"
        try:
            X = rng.uniform(-1, 1, (n, d)) * min(1.0, 3.5 / np.sqrt(d))
            raw = (np.float64(rng.uniform(-0.5, 0.8)), np.float64(0.4), np.float64(rng.uniform(-4.0, -1.0)))
            oop = orc.RbfGramOp(X, noise_minval=1e-4, kernel=kernel)
            ls, s, noise = oop.constrained(*raw)
            K = orc.kernel_matrix(kernel, X, X, ls, s, diag_offset=0)
            bound = RbfGramOp(T(X), noise_minval=1e-4, kernel=kernel).bind(*(T(q) for q in raw))
            want, winfo = orc.cholesky_partial_pivot(lambda i, j: K[i, j], n, rank)
            got, ginfo = low_rank.cholesky_partial_pivot(rank=rank)(low_rank.without_noise(bound), n)
            if not winfo["success"]:
                # (rank beyond the numerical rank of the Gram matrix: the remaining diagonal is round-off, both sides report success = False)
                if bool(ginfo["success"]):
                    check("partial Cholesky success flag", 1.0, 0.0, 0.0, info)
                continue
            gp = N(ginfo["pivots"]).astype(int)
            if not np.array_equal(gp, winfo["pivots"]):
                # Ties among equal diagonal entries (K_ii = outputscale for every i) are broken by round-off in the expanded distance form of
                # the reference / the oracle and by index here (the kernels take a point's distance to itself as exactly 0): another, equally
                # valid pivot order.  Check the defining property instead: L L^T reproduces K on the pivot rows, and the residual trace is no worse.
                LLt = N(got) @ N(got).T
                check("partial Cholesky (other pivot order): L L^T = K on the pivot rows", LLt[gp], K[gp], 1e-8, info)
                tr_g, tr_o = np.trace(K - LLt), np.trace(K - want @ want.T)
                if not (tr_g <= 2.0 * tr_o + 1e-9 * np.trace(K)):
                    check("partial Cholesky (other pivot order): residual trace", tr_g, tr_o, 1.0, info)
            else:
                check("partial Cholesky factor", N(got), want, 1e-7, info)
                pre, pinfo = low_rank.preconditioner(low_rank.cholesky_partial_pivot(rank=rank))(low_rank.without_noise(bound), n)
                V = rng.standard_normal((nrhs, n))
                wantz = np.stack([orc.precondition_solve(want, v, float(noise)) for v in V])
                check("preconditioner solve", N(pre(T(V), float(noise))), wantz, 1e-6, info)
                A = lambda v: K @ v + noise * v  # noqa: E731
                P = lambda v: orc.precondition_solve(want, v, noise)  # noqa: E731
                for b_i in range(min(nrhs, 2)):
                    wx, _ = orc.pcg_fixed_step(A, V[b_i], P, num_matvecs=steps)
                    x, _ = cg.pcg_fixed_step(steps)(bound, T(V[b_i]), pre.bind(float(noise)))
                    check("PCG fixed steps", N(x), wx, 1e-5, info)
                wx = np.stack([orc.pcg_fixed_step(A, v, None, num_matvecs=steps)[0] for v in V])
                x, _ = cg.cg_fixed_step(steps)(bound, T(V))
                check("CG fixed steps, batched", N(x), wx, 1e-6, info)
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print(f"EXCEPTION {type(exc).__name__}: {exc}   [{info}]", flush=True)
    else:
        n = int(rng.integers(2000, 25000))
        nrhs = int(rng.choice([1, 4, 8, 33, 64]))
        steps = int(rng.integers(1, 4))
        mode = str(rng.choice(["f16x3", "f16x3-matvec", "fp32"]))
        info = f"case {case}: {kernel} n={n} d={d} rhs={nrhs} steps={steps} {mode} vs fp64"
        try:
            X = rng.standard_normal((n, d)) * min(1.0, 3.5 / np.sqrt(d))
            raw = (0.7, 0.3, -1.0)
            V = rng.standard_normal((nrhs, n))
            sol = {}
            for dt, prec in ((torch.float64, "fp32"), (torch.float32, mode)):
                bound = RbfGramOp(T(X, dt), noise_minval=1e-4, kernel=kernel, precision=prec).bind(*(T(np.float64(q), dt) for q in raw))
                sol[dt], _ = cg.cg_fixed_step(steps)(bound, T(V, dt))
            check("CG in an fp32 mode", N(sol[torch.float32]), N(sol[torch.float64]), 2e-3 if kernel == "matern12" else 5e-4, info)
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print(f"EXCEPTION {type(exc).__name__}: {exc}   [{info}]", flush=True)
    if case % 10 == 0:
        print(info + " done", flush=True)
print(f"{cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
