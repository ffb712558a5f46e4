"""fp64 Gram matvec (VALU kernel) at a few sizes, for the record."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "experiments-lanczos-adjoints_amd"))
import torch
from matfree_extensions.operators import RbfGramOp
dev = torch.device("cuda:0")
for n, p in ((16384, 8), (36584, 10), (131072, 8), (131072, 64)):
    X = torch.randn(n, 8, device=dev, dtype=torch.float64)
    op = RbfGramOp(X, noise_minval=1e-4)
    params = [torch.zeros((), device=dev, dtype=torch.float64) for _ in range(3)]
    v = torch.randn(p, n, device=dev, dtype=torch.float64)
    with torch.no_grad():
        op(v, *params); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): op(v, *params)
        torch.cuda.synchronize()
    print(f"fp64 n={n} p={p}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per matvec", flush=True)
