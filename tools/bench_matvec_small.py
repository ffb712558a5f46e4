"""Gram matvec at GP-training sizes (n = 36584, d = 9, Matern-3/2 / RBF): ms per matvec vs right-hand sides."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "experiments-lanczos-adjoints_amd"))
import torch
from matfree_extensions.operators import RbfGramOp
dev = torch.device("cuda:0")
for n in (9000, 16384, 36584, 65536):
    X = torch.randn(n, 9, device=dev)
    for kernel in ("matern32",):
        op = RbfGramOp(X, noise_minval=1e-4, kernel=kernel)
        params = [torch.zeros(9, device=dev), torch.zeros((), device=dev), torch.zeros((), device=dev)]
        for p in (1, 10, 64):
            v = torch.randn(p, n, device=dev)
            with torch.no_grad():
                op(v, *params); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): op(v, *params)
                torch.cuda.synchronize()
            print(f"n={n} {kernel} p={p:3d} {(time.perf_counter() - t0) / 10 * 1e3:7.3f} ms", flush=True)
