"""Timings of the other BASELINE configs (parity cases of bench.py's headline config) on one MI355X.

    python tools/bench_configs.py            # C1 dense 512, C2-like RBF 45730x9, C3 CSR Laplacian 102400 (fp64)

Each line: forward (tridiag / SLQ value) and forward+adjoint wall time, median of 5 after 2 warm-ups.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))
from matfree_extensions import lanczos  # noqa: E402
from matfree_extensions.operators import CsrOp, DenseOp  # noqa: E402
from matfree_extensions.util import gp_util  # noqa: E402
from oracle import slq_oracle as orc  # noqa: E402  (input generators only)

dev = torch.device("cuda:0")


def timeit(fn, reps=5, warm=2):
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


def c1():
    A = torch.tensor(orc.spd_diag_plus_lowrank(512, 4, seed=0), device=dev)
    probe = torch.tensor(orc.rademacher(1, 1, 512)[0], device=dev)
    for dt in (torch.float64, torch.float32):
        At = A.to(dt).requires_grad_(True)
        f = lanczos.integrand_spd(torch.log, 20, DenseOp())
        fwd = timeit(lambda: f(probe.to(dt), At))
        both = timeit(lambda: torch.autograd.grad(f(probe.to(dt), At), At))
        print(f"C1 dense 512x512 k=20 p=1 {dt}: value {fwd:.3f} ms, value+grad {both:.3f} ms")


def c2():
    # BASELINE config 2 on its own data: all 45 730 rows of the UCI protein set, z-scored (tests/golden/uci_protein_X.npz)
    k, p = 30, 8
    X = torch.tensor(np.load(os.path.join(ROOT, "tests", "golden", "uci_protein_X.npz"))["X"], dtype=torch.float32, device=dev)
    n, d = X.shape
    probes = torch.tensor(orc.rademacher(2, p, n), dtype=torch.float32, device=dev)
    for tag, shape in (("ARD", (d,)), ("scalar lengthscale", ())):
        params = [torch.zeros(shape if i == 0 else (), dtype=torch.float32, device=dev, requires_grad=True) for i in range(3)]
        f = lanczos.integrand_spd(torch.log, k, gp_util.gram_operator(X, noise_minval=1e-4))
        fwd = timeit(lambda: f(probes, *params), reps=3, warm=1)
        both = timeit(lambda: torch.autograd.grad(f(probes, *params).sum(), params), reps=3, warm=1)
        print(f"C2 UCI protein RBF N={n} d={d} {tag} k={k} p={p} fp32: value {fwd:.2f} ms, value+grad {both:.2f} ms")


def c3():
    m = 320
    r, c, vals, n = orc.laplacian_2d_plus_identity(m)
    op, v, _ = CsrOp.from_coo(r, c, vals, n, dev)
    vec = torch.randn(n, dtype=torch.float64, device=dev)
    for reortho in ("full", "none"):
        vt = v.clone().requires_grad_(True)
        x0 = vec.clone().requires_grad_(True)
        alg = lanczos.tridiag(op, 50, reortho=reortho)

        def fwd_only():
            return alg(x0, vt)

        (Q, (a, b)), (q, br) = alg(x0, vt)
        cot = [torch.randn_like(t) for t in (Q, a, b, q, br)]

        def both():
            (Q, (a, b)), (q, br) = alg(x0, vt)
            return torch.autograd.grad((Q, a, b, q, br), (x0, vt), cot)

        print(f"C3 CSR 5-pt Laplacian+I N={n} nnz={op.nnz} k=50 fp64 reortho={reortho}: forward {timeit(fwd_only):.3f} ms, "
              f"forward+adjoint (all outputs, grad wrt all nnz) {timeit(both):.3f} ms")


def c5():
    """BASELINE config 5: 5-pt Laplacian wave system on a 1000 x 1000 grid (state 2e6), exp(t A) y0 by Arnoldi, fp64."""
    from matfree_extensions.util import pde_util

    res = 1000
    op, values_fn = pde_util.wave_operator(res, 1.0 / res, boundary="neumann", device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    scale = (0.01 * torch.randn((res, res), dtype=torch.float64, device=dev, generator=g)) ** 2 + 1e-6
    y0 = torch.randn(2 * res * res, dtype=torch.float64, device=dev, generator=g)
    for k in (10, 20, 30):
        expm = pde_util.expm_arnoldi(k)
        sc = scale.clone().requires_grad_(True)

        def fwd_only():
            with torch.no_grad():
                return expm(op, 1e-3, y0, values_fn(sc))[0]

        def both():
            out, _ = expm(op, 1e-3, y0, values_fn(sc))
            return torch.autograd.grad(out.sum(), sc)

        print(f"C5 wave system {res}x{res} (state {2 * res * res}, nnz {op.nnz}) expm_arnoldi k={k} fp64: forward {timeit(fwd_only):.3f} ms, "
              f"forward+gradient wrt the coefficient field {timeit(both):.3f} ms")


if __name__ == "__main__":
    import sys

    for name in (sys.argv[1:] or ["c1", "c2", "c3", "c5"]):
        globals()[name]()
