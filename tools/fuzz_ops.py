"""Random sparsity patterns and shapes for the native CSR and dense operators against scipy / NumPy: apply, transposed apply (the adjoint's matvec),
the all-nnz parameter sweep (SDDMM) -- empty rows, duplicate entries, one very long row or column, p = 1 ... 70 vectors, fp64 and fp32.
    python tools/fuzz_ops.py [cases] [seed]"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))
from matfree_extensions.operators import CsrOp, DenseOp  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def check(name, got, ref, tol, info):
    global bad
    err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)
    if not (err <= tol):
        bad += 1
        print(f"FAIL {name}: rel err {err:.2e} > {tol:.0e}   [{info}]", flush=True)


for case in range(cases):
    dtype = torch.float64 if rng.integers(0, 2) else torch.float32
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    p = int(rng.choice([1, 2, 3, 8, 9, 33, 64, 70]))
    if case % 4 == 3:
        n = int(rng.integers(1, 1500))
        A = rng.standard_normal((n, n))
        info = f"case {case}: dense n={n} p={p} {dtype}"
        V, C = rng.standard_normal((p, n)), rng.standard_normal((p, n))
        At = torch.tensor(A, dtype=dtype, device=dev, requires_grad=True)
        Vt = torch.tensor(V, dtype=dtype, device=dev, requires_grad=True)
        y = DenseOp()(Vt, At)
        check("dense apply", y.detach().double().cpu().numpy(), V @ A.T, tol, info)
        gv, ga = torch.autograd.grad(y, (Vt, At), torch.tensor(C, dtype=dtype, device=dev))
        check("dense transposed apply", gv.double().cpu().numpy(), C @ A, tol, info)
        check("dense parameter sweep", ga.double().cpu().numpy(), C.T @ V, tol * 10, info)
    else:
        n = int(rng.choice([rng.integers(1, 60), rng.integers(60, 3000), rng.integers(3000, 60000)]))
        avg = float(rng.choice([0.3, 1.0, 4.0, 20.0]))
        nnz = max(1, int(avg * n))
        r = rng.integers(0, n, nnz)
        c = rng.integers(0, n, nnz)
        shape = str(rng.choice(["plain", "long_row", "long_col", "empty_tail", "dups"]))
        if shape == "long_row":
            m = min(n, int(rng.integers(100, 5000)))
            r = np.concatenate([r, np.full(m, rng.integers(0, n))]); c = np.concatenate([c, rng.integers(0, n, m)])
        elif shape == "long_col":
            m = min(n, int(rng.integers(100, 5000)))
            c = np.concatenate([c, np.full(m, rng.integers(0, n))]); r = np.concatenate([r, rng.integers(0, n, m)])
        elif shape == "empty_tail":
            r = r % max(1, n // 2)
        elif shape == "dups":
            r = np.concatenate([r, r[: nnz // 3]]); c = np.concatenate([c, c[: nnz // 3]])
        vals = rng.standard_normal(r.size)
        info = f"case {case}: csr n={n} nnz={r.size} {shape} p={p} {dtype}"
        try:
            op, vt, order = CsrOp.from_coo(r, c, vals, n, dev)
            vt = vt.to(dtype).requires_grad_(True)
            M = sp.coo_matrix((vals, (r, c)), shape=(n, n)).tocsr()
            V, C = rng.standard_normal((p, n)), rng.standard_normal((p, n))
            Vt = torch.tensor(V, dtype=dtype, device=dev, requires_grad=True)
            y = op(Vt, vt)
            check("csr apply", y.detach().double().cpu().numpy(), (M @ V.T).T, tol, info)
            gv, gvals = torch.autograd.grad(y, (Vt, vt), torch.tensor(C, dtype=dtype, device=dev))
            check("csr transposed apply", gv.double().cpu().numpy(), (M.T @ C.T).T, tol, info)
            ro, co = r[order.numpy()], c[order.numpy()]
            ref = np.einsum("bi,bi->i", C[:, ro], V[:, co])
            check("csr parameter sweep (all nnz)", gvals.double().cpu().numpy(), ref, tol * 10, info)
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print(f"EXCEPTION {type(exc).__name__}: {exc}   [{info}]", flush=True)
    if case % 20 == 0:
        print(info + " done", flush=True)
print(f"{cases} cases, {bad} failures")
