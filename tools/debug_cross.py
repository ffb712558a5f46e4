import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "experiments-lanczos-adjoints_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from oracle import slq_oracle as orc
from matfree_extensions.operators import RbfGramOp
from matfree_extensions import cg
DEV = torch.device("cuda:0")
T = lambda x: torch.tensor(np.asarray(x), dtype=torch.float64, device=DEV)
rng = np.random.default_rng(6)
n, m, d = 700, 333, 3
X, Xs = rng.uniform(-1, 1, (n, d)), rng.uniform(-1, 1, (m, d))
raw = (np.array([0.1, 0.3, -0.2]), np.float64(0.3), np.float64(-2.0))
oop = orc.RbfGramOp(X, noise_minval=1e-2)
ls, s, noise = oop.constrained(*raw)
Kx = orc.kernel_matrix("rbf", Xs, X, ls, s)
op = RbfGramOp(T(X), noise_minval=1e-2)
w = rng.standard_normal(n)
got = op.cross_apply(T(Xs), T(w), *(T(q) for q in raw)).cpu().numpy()
print("cross err", np.abs(got - Kx @ w).max(), np.abs(Kx @ w).max())
y = np.sin(X.sum(-1)) - 0.3
for steps in (5, 20, 40):
    wx, _ = orc.pcg_fixed_step(lambda v: oop.apply(v, *raw), y, num_matvecs=steps)
    x, info = cg.cg_fixed_step(steps)(op.bind(*(T(q) for q in raw)), T(y))
    print(steps, "cg err", np.abs(x.cpu().numpy() - wx).max(), np.abs(wx).max(), float(info["residual_abs"].norm()))
