"""Which arithmetic is the C4 accuracy gate's error made of?  Two crossed runs per probe set at the C4 size, against the fp64 HIP path:

  krylov32_op64 : the shipped fp32 Krylov kernels (fp32-stored basis and adjoint states, fp64 dot products) around an EXACT operator --
                  the fp64 Gram kernels behind a Python callable that casts (v.double() -> K v -> .float()), gradient through autograd;
  krylov64_op32 : fp64 Krylov kernels around the shipped f16x3 operator (v.float() -> f16x3 Gram matvec -> .double(); its parameter
                  gradient is the f16x3 gradient GEMM, per Krylov step instead of one deferred sweep).

    python tools/accuracy_sources.py --which krylov32_op64 --seed 8 --out gpurun_out/acc_src/seed8_krylov32_op64.json
    python tools/accuracy_sources.py --table gpurun_out/acc_src        (needs the f64 references: profiles/r05b_accuracy_16_seeds/f64_refs/)
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
ap.add_argument("--which", default="krylov32_op64", choices=["krylov32_op64", "krylov64_op32"])
ap.add_argument("--n", type=int, default=131072)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--k", type=int, default=40)
ap.add_argument("--p", type=int, default=64)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--out", default="")
ap.add_argument("--table", default="")
args = ap.parse_args()

if args.table:
    print(f"{'seed':>4} {'run':<16} {'value':>10} {'d raw_l':>10} {'d raw_s':>10} {'d raw_noise':>11} {'max grad':>9}  seconds")
    for f in sorted(glob.glob(os.path.join(args.table, "*.json"))):
        r = json.load(open(f))
        ref = json.load(open(os.path.join(ROOT, "profiles", "r05b_accuracy_16_seeds", "f64_refs", f"seed{r['seed']}_f64.json")))
        ev = abs(r["value"] - ref["value"]) / abs(ref["value"])
        eg = np.abs(np.array(r["grad"]) - np.array(ref["grad"])) / np.abs(np.array(ref["grad"]))
        print(f"{r['seed']:>4} {r['which']:<16} {ev:10.2e} {eg[0]:10.2e} {eg[1]:10.2e} {eg[2]:11.2e} {eg.max():9.2e}  {r['seconds']:.1f}")
    sys.exit(0)

import torch  # noqa: E402

from matfree_extensions import hutchinson, lanczos  # noqa: E402
from matfree_extensions.util import gp_util  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(4)
X64 = torch.randn((args.n, args.d), generator=gen, dtype=torch.float32).double().to(dev)
inv = lambda x: float(np.log(np.expm1(x)))  # noqa: E731
raw = (inv(2.0), inv(1.0), inv(0.1))
kry = torch.float32 if args.which == "krylov32_op64" else torch.float64
opd = torch.float64 if args.which == "krylov32_op64" else torch.float32
params = [torch.tensor(v, dtype=kry, device=dev, requires_grad=True) for v in raw]
inner = gp_util.gram_operator(X64.to(opd), precision="fp32" if opd == torch.float64 else "f16x3")


def matvec(v, *ps):  # a plain Python callable: libmfx drives the Krylov loop in the dtype of v and calls back for every matvec
    return inner(v.to(opd), *[q.to(opd) for q in ps]).to(kry)


integrand = lanczos.integrand_spd(torch.log, args.k, matvec)
probes = hutchinson.sampler_rademacher(X64[:, 0].to(kry), num=args.p)(args.seed)
torch.cuda.synchronize()
t0 = time.perf_counter()
vals = integrand(probes, *params)
g = torch.autograd.grad(vals.sum(), params)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
res = {"which": args.which, "seed": args.seed, "n": args.n, "d": args.d, "k": args.k, "p": args.p,
       "value": vals.double().mean().item(), "grad": [t.double().item() / args.p for t in g], "seconds": dt}
print(json.dumps(res))
if args.out:
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"))
