"""BASELINE config 3 (CSR 5-pt Laplacian + I, n = 102 400, k = 50, fp64, one vector, full re-orthogonalisation), forward + adjoint
R times and nothing else: the process tools/prof_c3_traffic.sh puts under the PMC counters (R = 1 and R = 6: the difference is five runs
without the set-up).     python tools/c3_run.py R"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))
from matfree_extensions import lanczos  # noqa: E402
from matfree_extensions.operators import CsrOp  # noqa: E402
from oracle import slq_oracle as orc  # noqa: E402  (input generator only)

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
r, c, vals, n = orc.laplacian_2d_plus_identity(320)
op, v, _ = CsrOp.from_coo(r, c, vals, n, dev)
g = torch.Generator(device=dev).manual_seed(0)
x0 = torch.randn(n, dtype=torch.float64, device=dev, generator=g).requires_grad_(True)
vt = v.clone().requires_grad_(True)
alg = lanczos.tridiag(op, 50, reortho="full")
shapes = [(50, n), (50,), (49,), (n,), ()]
cot = [torch.randn(s, dtype=torch.float64, device=dev, generator=g) for s in shapes]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    (Q, (a, b)), (q, br) = alg(x0, vt)
    torch.autograd.grad((Q, a, b, q, br), (x0, vt), cot)
torch.cuda.synchronize()
print(f"C3 forward+adjoint x {R}: {(time.perf_counter() - t0) * 1e3 / R:.3f} ms per run (first run included)")
