"""Where does k_rbf_fat8_apply differ from the three-product fat kernel?  (debugging aid)
Probe b has V[0] = 1 and V[j2(b)] = 0.3: after the row scaling lo(V) is non-zero ONLY at column j2, so y8 - y3 isolates entry (i, j2)
of the FP8 product."""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "experiments-lanczos-adjoints_amd"))
n, d, p = 1024, 8, 64
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from matfree_extensions.operators import RbfGramOp
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    X = (0.3 * torch.randn(n, d, generator=g)).to(dev)
    op = RbfGramOp(X, noise_minval=1e-4)
    params = [torch.tensor(v, device=dev) for v in (0.9, 0.4, -1.0)]
    V = torch.zeros(p, n, device=dev)
    V[:, 0] = 1.0
    for b in range(p):
        V[b, 64 + b] = 0.3          # probe b: column b of tile 1
    y = op(V, *params)
    np.save(sys.argv[2], y.cpu().numpy())
    sys.exit(0)
out = {}
for fat in ("1", "2"):
    f = f"/tmp/dbg_{fat}.npy"
    subprocess.run([sys.executable, __file__, "child", f], env=dict(os.environ, MFX_RBF_FAT=fat), check=True)
    out[fat] = np.load(f)
a, b = out["1"], out["2"]
rel = np.abs(a - b).max(axis=1) / np.abs(a).max(axis=1)
np.set_printoptions(linewidth=200, precision=1)
print("probe b <-> column b of a tile; relative difference of the two kernels per probe (x 1e6):")
print((rel * 1e6).reshape(8, 8))
