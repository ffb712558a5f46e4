"""The workloads behind the reference's own published timings (BASELINE.md section 1, V100 inferred), timed on this build:
  * RBF Gram matvec with ONE vector at N = 45 730 (protein shape, d = 9), 100 000 / 200 000 / 300 000 (d = 3)
    (experiments/benchmarks/gram_matvec_versus_keops/matvec/benchmark_size_toy.py:59-139, median of 5)
  * lanczos.tridiag / arnoldi.hessenberg forward and custom-VJP adjoint (cotangents on all outputs, gradient w.r.t. all stored
    values) on a sparse SPD matrix of bcsstk18's size (N = 11 948, 149 090 stored values: the file is not in the reference repo,
    a banded stand-in with the same n and nnz is used), reortho = none, k = 50 / 100 / 250
    (experiments/benchmarks/wall_times_vjp_through_lanczos_arnoldi/suite_sparse/benchmark.py:91-121)
Different hardware: orientation only."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))
from matfree_extensions import arnoldi, lanczos  # noqa: E402
from matfree_extensions.operators import CsrOp  # noqa: E402
from matfree_extensions.util import gp_util  # noqa: E402

dev = torch.device("cuda:0")


def median_ms(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def matvecs():
    for n, d in ((45730, 9), (100000, 3), (200000, 3), (300000, 3)):
        g = torch.Generator().manual_seed(n)
        X = torch.randn(n, d, generator=g).to(dev)
        v = torch.randn(1, n, generator=g).to(dev)
        params = [torch.tensor(0.5, device=dev), torch.tensor(0.5, device=dev), torch.tensor(-2.0, device=dev)]
        for precision in ("f16x3", "fp32"):
            op = gp_util.gram_operator(X, precision=precision)
            with torch.no_grad():
                ms = median_ms(lambda: op(v, *params))
            print(f"RBF Gram matvec, 1 vector, N={n} d={d} fp32 [{precision}]: {ms:.3f} ms")


def banded_spd(n, nnz_target, rng):
    """symmetric banded matrix with ~nnz_target stored values and a dominant diagonal"""
    half = max(1, (nnz_target // n - 1) // 2)
    rows, cols, vals = [], [], []
    for off in range(-half, half + 1):
        i = np.arange(max(0, -off), min(n, n - off))
        rows.append(i)
        cols.append(i + off)
        vals.append(np.full(i.shape, 2.0 * half + 1.0) if off == 0 else -np.ones(i.shape) * (0.5 + 0.5 * rng.random()))
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    # symmetrise the off-diagonal values
    key = {}
    for a, b, x in zip(r, c, v):
        key[(min(a, b), max(a, b))] = x
    v = np.array([key[(min(a, b), max(a, b))] for a, b in zip(r, c)])
    return r, c, v


def krylov():
    n = 11948
    rng = np.random.default_rng(0)
    r, c, vals = banded_spd(n, 149090, rng)
    op, vt, _ = CsrOp.from_coo(r, c, vals, n, dev)
    vt = vt.float()
    x0 = torch.randn(n, device=dev)
    print(f"sparse SPD stand-in for bcsstk18: N={n}, stored values {len(vals)}, fp32, reortho=none")
    for name, make in (("lanczos.tridiag", lambda k: lanczos.tridiag(op, k, reortho="none")),
                       ("arnoldi.hessenberg", lambda k: arnoldi.hessenberg(op, k, reortho="none"))):
        for k in (50, 100, 250):
            alg = make(k)
            with torch.no_grad():
                fwd = median_ms(lambda: alg(x0, vt))
            xg, vg = x0.clone().requires_grad_(True), vt.clone().requires_grad_(True)
            outs = alg(xg, vg)
            flat = [t for t in torch.utils._pytree.tree_leaves(outs)]
            cot = [torch.randn_like(t) for t in flat]

            def both():
                o = torch.utils._pytree.tree_leaves(alg(xg, vg))
                return torch.autograd.grad(o, (xg, vg), cot)

            tot = median_ms(both)
            print(f"  {name} k={k}: forward {fwd:.3f} ms, forward + custom adjoint {tot:.3f} ms (adjoint alone ~{tot - fwd:.3f} ms)")


if __name__ == "__main__":
    matvecs()
    krylov()
