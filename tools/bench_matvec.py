"""Gram matvec time vs number of right-hand sides (fp32, n = 131072)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "experiments-lanczos-adjoints_amd"))
import torch
from matfree_extensions.operators import RbfGramOp

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
for d in (3, 8):
    X = torch.rand(n, d, device=dev) * 2 - 1
    for kernel in ("rbf", "matern32"):
        op = RbfGramOp(X, noise_minval=1e-4, kernel=kernel)
        params = [torch.zeros((), device=dev) for _ in range(3)]
        for p in (1, 2, 4, 8, 16, 32, 64):
            v = torch.randn(p, n, device=dev)
            with torch.no_grad():
                op(v, *params)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    op(v, *params)
                torch.cuda.synchronize()
            print(f"d={d} {kernel:9s} p={p:3d}  {(time.perf_counter() - t0) / 5 * 1e3:8.2f} ms", flush=True)
