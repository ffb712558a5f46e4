"""Error of ONE Gram matvec W = (K + noise I) V against the fp64 HIP path, per kind of input vector (run on the GPU box).

Question behind it: the f16 MFMA rounds its internal sum towards -infinity.  For sign-mixed inputs (random probes) that is
harmless, but Krylov vectors are dominated by the smooth leading eigenvectors of an all-positive kernel matrix: the
accumulator then grows monotonically over the 8192 k-steps of a row and the floor bias may add up coherently.

    python tools/diag_matvec_bias.py --n 131072 --mode f16x3-matvec      (MFX_RBF_SPLIT=s forces s column splits)
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))
from matfree_extensions.util import gp_util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=131072)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--mode", default="f16x3-matvec")
args = ap.parse_args()
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(4)
X64 = torch.randn((args.n, args.d), generator=gen, dtype=torch.float32).double().to(dev)
inv = lambda x: float(np.log(np.expm1(x)))  # noqa: E731
raw64 = [torch.tensor(v, dtype=torch.float64, device=dev) for v in (inv(2.0), inv(1.0), inv(0.1))]
op64 = gp_util.gram_operator(X64, precision="fp32")
n = args.n
with torch.no_grad():
    rows, names = [], []
    g2 = torch.Generator().manual_seed(7)
    for i in range(4):
        rows.append((torch.randint(0, 2, (n,), generator=g2).double() * 2 - 1).to(dev)); names.append(f"rademacher{i}")
    rows.append(torch.randn(n, generator=g2, dtype=torch.float64).to(dev)); names.append("gaussian")
    v = torch.ones(n, dtype=torch.float64, device=dev) / np.sqrt(n)
    rows.append(v.clone()); names.append("ones/sqrt(n)")
    for it in range(3):  # power iteration: towards the Perron vector
        v = op64(v, *raw64)
        v = v / v.norm()
        rows.append(v.clone()); names.append(f"power{it + 1}")
    # orthogonalised against the Perron-like vector: what later Lanczos vectors look like
    w = torch.randn(n, generator=g2, dtype=torch.float64).to(dev)
    w = op64(w, *raw64)
    w = w - (w @ v) * v
    rows.append(w / w.norm()); names.append("K*gauss minus perron")
    while len(rows) < 16:
        rows.append(torch.zeros(n, dtype=torch.float64, device=dev)); names.append("zero")
    V64 = torch.stack(rows)
    W64 = op64(V64, *raw64)
    V32 = V64.float()
    W64r = op64(V32.double(), *raw64)  # reference on the fp32-rounded inputs
    op32 = gp_util.gram_operator(X64.float(), precision=args.mode)
    W32 = op32(V32, *[r.float() for r in raw64]).double()
    torch.cuda.synchronize()
    print(f"mode {args.mode} split {os.environ.get('MFX_RBF_SPLIT', 'auto')} n {n}")
    for b, name in enumerate(names):
        if name == "zero":
            continue
        e = W32[b] - W64r[b]
        rel = (e.norm() / W64r[b].norm()).item()
        bias = (e.sum() / W64r[b].abs().sum()).item()
        print(f"  {name:<22} rel_l2 {rel:9.2e}  signed mean err / mean |w| {bias:+9.2e}   |w|_2 {W64r[b].norm().item():.3e}")
