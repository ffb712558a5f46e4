"""One Gram matvec configuration (C4 shape by default), repeated: for profiling."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "experiments-lanczos-adjoints_amd"))
import torch
from matfree_extensions.operators import RbfGramOp

dev = torch.device("cuda:0")
n, d, p, reps = 131072, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 10
kernel = sys.argv[3] if len(sys.argv) > 3 else "rbf"
X = torch.randn(n, d, device=dev)
op = RbfGramOp(X, noise_minval=1e-4, kernel=kernel)
inv = lambda v: float(torch.log(torch.expm1(torch.tensor(v))))
params = [torch.tensor(inv(v), device=dev) for v in (2.0, 1.0, 0.1)]
v = torch.randn(p, n, device=dev)
with torch.no_grad():
    op(v, *params)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        op(v, *params)
    torch.cuda.synchronize()
print(f"{kernel} p={p}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per matvec")
