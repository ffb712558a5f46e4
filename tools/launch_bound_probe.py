"""Is a small config bound by the host's launch rate or by the GPU?  Enqueue N calls without synchronising and compare the
host time to enqueue with the time until the GPU has finished: python tools/launch_bound_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))
sys.path.insert(0, ROOT)
from matfree_extensions import lanczos  # noqa: E402
from matfree_extensions.operators import CsrOp, DenseOp  # noqa: E402
from oracle import slq_oracle as orc  # noqa: E402  (input generators only)

dev = torch.device("cuda:0")


def probe(name, fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e3 * (t1 - t0) / reps:.3f} ms/call, until GPU idle {1e3 * (t2 - t0) / reps:.3f} ms/call")


A = torch.tensor(orc.spd_diag_plus_lowrank(512, 4, seed=0), device=dev)
v = torch.tensor(orc.rademacher(1, 1, 512)[0], device=dev)
alg = lanczos.tridiag(DenseOp(), 20, reortho="full")
with torch.no_grad():
    probe("C1 dense 512 k=20 tridiag forward", lambda: alg(v, A))
f = lanczos.integrand_spd(torch.log, 20, DenseOp())
with torch.no_grad():
    probe("C1 dense 512 k=20 integrand value", lambda: f(v, A))

r, c, vals, n = orc.laplacian_2d_plus_identity(320)
op, vv, _ = CsrOp.from_coo(r, c, vals, n, dev)
x0 = torch.randn(n, dtype=torch.float64, device=dev)
alg = lanczos.tridiag(op, 50, reortho="full")
with torch.no_grad():
    probe("C3 csr 102400 k=50 tridiag forward", lambda: alg(x0, vv))
