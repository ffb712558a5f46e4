"""Accuracy of each arithmetic mode of the Gram kernels against the fp64 HIP path at the C4 size (run on the GPU box).

    python tools/accuracy_gate.py --mode f64 --n 131072 --p 64 --out gpurun_out/acc/n131072_f64.json
    python tools/accuracy_gate.py --mode f16x3 ...            (one process per mode: libmfx reads its A/B switches once)
    python tools/accuracy_gate.py --table gpurun_out/acc      (relative errors of every mode against the f64 file of its n)

Workload = BASELINE config 4: X ~ N(0,1) (n, 8), lengthscale 2, outputscale 1, noise 0.1, k = 40 fully re-orthogonalised
Lanczos steps, p explicit +-1 probes (seed 0).  Reported: SLQ log-det mean over the probes and its gradient w.r.t. the three
raw hyper-parameters.  north_star tolerance: rtol 1e-4 (value and gradient).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="f64", choices=["f64", "fp32", "f16x3-matvec", "f16x3"])
ap.add_argument("--n", type=int, default=131072)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--k", type=int, default=40)
ap.add_argument("--p", type=int, default=64)
ap.add_argument("--tag", default="")
ap.add_argument("--seed", type=int, default=0, help="key of the +-1 probes (0 = the committed tables)")
ap.add_argument("--out", default="")
ap.add_argument("--table", default="")
args = ap.parse_args()


def table(folder):
    rows = [json.load(open(f)) for f in sorted(glob.glob(os.path.join(folder, "*.json")))]
    refs = {(r["n"], r["p"]): r for r in rows if r["mode"] == "f64"}
    print(f"{'n':>7} {'p':>3} {'mode':<22} {'value':>10} {'d raw_l':>10} {'d raw_s':>10} {'d raw_noise':>11} {'max grad':>9}  seconds")
    for r in rows:
        ref = refs.get((r["n"], r["p"]))
        if ref is None or r["mode"] == "f64":
            continue
        ev = abs(r["value"] - ref["value"]) / abs(ref["value"])
        eg = np.abs(np.array(r["grad"]) - np.array(ref["grad"])) / np.abs(np.array(ref["grad"]))
        name = r["mode"] + (" " + r["tag"] if r["tag"] else "")
        print(f"{r['n']:>7} {r['p']:>3} {name:<22} {ev:10.2e} {eg[0]:10.2e} {eg[1]:10.2e} {eg[2]:11.2e} {eg.max():9.2e}  {r['seconds']:.1f}")
    for (n, p), ref in sorted(refs.items()):
        print(f"f64 reference n={n} p={p}: value {ref['value']:.9g} grad {ref['grad']} ({ref['seconds']:.1f} s)")


if args.table:
    table(args.table)
    sys.exit(0)

import torch  # noqa: E402

from matfree_extensions import hutchinson, lanczos  # noqa: E402
from matfree_extensions.util import gp_util  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(4)
X64 = torch.randn((args.n, args.d), generator=gen, dtype=torch.float32).double().to(dev)
inv = lambda x: float(np.log(np.expm1(x)))  # noqa: E731
raw = (inv(2.0), inv(1.0), inv(0.1))
dtype = torch.float64 if args.mode == "f64" else torch.float32
X = X64.to(dtype)
params = [torch.tensor(v, dtype=dtype, device=dev, requires_grad=True) for v in raw]
op = gp_util.gram_operator(X, precision="fp32" if args.mode == "f64" else args.mode)
integrand = lanczos.integrand_spd(torch.log, args.k, op)
probes = hutchinson.sampler_rademacher(X[:, 0], num=args.p)(args.seed)
torch.cuda.synchronize()
t0 = time.perf_counter()
vals = integrand(probes, *params)
g = torch.autograd.grad(vals.sum(), params)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
res = {"mode": args.mode, "tag": args.tag, "n": args.n, "d": args.d, "k": args.k, "p": args.p,
       "value": vals.double().mean().item(), "grad": [t.double().item() / args.p for t in g], "seconds": dt}
print(json.dumps(res))
if args.out:
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"))
