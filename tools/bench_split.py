"""Column-split sweep (MFX_RBF_SPLIT) for one matvec shape: n p [kernel]."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "experiments-lanczos-adjoints_amd"))
import torch
from matfree_extensions.operators import RbfGramOp
dev = torch.device("cuda:0")
n, p = int(sys.argv[1]), int(sys.argv[2])
X = torch.randn(n, 9, device=dev)
op = RbfGramOp(X, noise_minval=1e-4, kernel=sys.argv[3] if len(sys.argv) > 3 else "matern32")
params = [torch.zeros(9, device=dev), torch.zeros((), device=dev), torch.zeros((), device=dev)]
v = torch.randn(p, n, device=dev)
with torch.no_grad():
    op(v, *params); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): op(v, *params)
    torch.cuda.synchronize()
print(f"n={n} p={p} split={os.environ.get('MFX_RBF_SPLIT', 'auto')}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
