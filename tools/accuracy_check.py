"""Accuracy of the fp32 paths against the fp64 HIP path on a C4-like problem (run on the GPU box).

    python tools/accuracy_check.py --n 16384 --k 40 --p 64

Prints SLQ log-det value and gradient in fp64 (VALU kernels), fp32 (exact fp32 MFMA) and, when
MFX_RBF_SPLIT_F16=1 is set for the process, the 3 x f16 split path; relative errors vs fp64.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))
from matfree_extensions import hutchinson, lanczos  # noqa: E402
from matfree_extensions.util import gp_util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--k", type=int, default=40)
ap.add_argument("--p", type=int, default=64)
ap.add_argument("--skip64", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(4)
X64 = torch.randn((args.n, args.d), generator=gen, dtype=torch.float32).double().to(dev)
inv = lambda x: float(np.log(np.expm1(x)))  # noqa: E731
raw = (inv(2.0), inv(1.0), inv(0.1))


def run(dtype):
    X = X64.to(dtype)
    params = [torch.tensor(v, dtype=dtype, device=dev, requires_grad=True) for v in raw]
    integrand = lanczos.integrand_spd(torch.log, args.k, gp_util.gram_operator(X))
    probes = hutchinson.sampler_rademacher(X[:, 0], num=args.p)(0)
    vals = integrand(probes, *params)
    g = torch.autograd.grad(vals.sum(), params)
    return vals.double().mean().item(), np.array([t.item() for t in g]) / args.p


res = {}
if not args.skip64:
    res["f64"] = run(torch.float64)
res["f32"] = run(torch.float32)
ref = res.get("f64", res["f32"])
for kname, (v, g) in res.items():
    print(kname, "split_f16=" + os.environ.get("MFX_RBF_SPLIT_F16", "0"), "value", v, "grad", g,
          "rel_err value", abs(v - ref[0]) / abs(ref[0]), "rel_err grad", np.abs(g - ref[1]) / np.abs(ref[1]))
