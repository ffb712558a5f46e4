"""Where in a probe set does the default mode's gradient error sit?  One C4-size probe set (BASELINE config 4, as tools/accuracy_gate.py),
the SAME batched forward pass in both arithmetics (fp64 HIP path / shipped f16x3 default), and one backward pass per GROUP of probes:
the cotangent of the other probes is zero, so every kernel runs at the shipped shapes and a group's parameter gradient is exactly its
share of the batch gradient.  Then the worst group (on d raw_lengthscale) once more, probe by probe.

    python tools/accuracy_per_probe.py --seed 16 [--groups 8] --out gpurun_out/acc_probe/seed16.json

Why: 29 of 32 probe sets land within 1.5e-4 of the fp64 gradient, three (keys 16, 17, 18) at 2.8e-4 .. 4.0e-4
(profiles/r05b_accuracy_16_seeds/) -- is that one probe with an ill-conditioned recurrence, or all 64 leaning the same way?
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=131072)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--k", type=int, default=40)
ap.add_argument("--p", type=int, default=64)
ap.add_argument("--seed", type=int, default=16)
ap.add_argument("--groups", type=int, default=8)
ap.add_argument("--out", default="")
args = ap.parse_args()

import torch  # noqa: E402

from matfree_extensions import hutchinson, lanczos  # noqa: E402
from matfree_extensions.util import gp_util  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(4)
X64 = torch.randn((args.n, args.d), generator=gen, dtype=torch.float32).double().to(dev)
inv = lambda x: float(np.log(np.expm1(x)))  # noqa: E731
raw = (inv(2.0), inv(1.0), inv(0.1))


def passes(mode, index_sets):
    """-> per-probe values (p,), one gradient (3,) per index set (each already divided by p: shares of the batch-mean gradient)"""
    dtype = torch.float64 if mode == "f64" else torch.float32
    X = X64.to(dtype)
    op = gp_util.gram_operator(X, precision="fp32" if mode == "f64" else mode)
    integrand = lanczos.integrand_spd(torch.log, args.k, op)
    probes = hutchinson.sampler_rademacher(X[:, 0], num=args.p)(args.seed)
    out, vals = [], None
    for idx in index_sets:
        params = [torch.tensor(v, dtype=dtype, device=dev, requires_grad=True) for v in raw]
        vals = integrand(probes, *params)
        g = torch.autograd.grad(vals[torch.as_tensor(idx, device=dev)].sum(), params)
        out.append([t.double().item() / args.p for t in g])
        print(f"  {mode} probes {idx[0]}..{idx[-1]}: {out[-1]}", flush=True)
    return vals.detach().double().cpu().numpy(), np.array(out)


def compare(title, index_sets):
    t0 = time.perf_counter()
    v64, g64 = passes("f64", index_sets)
    v32, g32 = passes("f16x3", index_sets)
    total = np.abs(np.array(REF))
    print(f"== {title}   ({time.perf_counter() - t0:.0f} s)   error of each share, in units of the WHOLE batch gradient's component")
    print(f"{'probes':>9} {'d raw_l':>10} {'d raw_s':>10} {'d raw_noise':>11}   value (rel. to the probe's own)")
    rows = []
    for idx, a, b in zip(index_sets, g64, g32):
        e = (b - a) / total
        ev = np.abs(v32[idx] - v64[idx]).max() / np.abs(v64[idx]).max()
        rows.append({"probes": [int(idx[0]), int(idx[-1])], "err": e.tolist(), "f64": a.tolist(), "f16x3": b.tolist()})
        print(f"{idx[0]:>4}-{idx[-1]:<4} {e[0]:10.2e} {e[1]:10.2e} {e[2]:11.2e}   {ev:.2e}")
    s = (g32.sum(0) - g64.sum(0)) / total
    print(f"{'sum':>9} {s[0]:10.2e} {s[1]:10.2e} {s[2]:11.2e}")
    return rows


ref_file = os.path.join(ROOT, "profiles", "r05b_accuracy_16_seeds", "f64_refs", f"seed{args.seed}_f64.json")
REF = json.load(open(ref_file))["grad"]
per = args.p // args.groups
groups = [list(range(g * per, (g + 1) * per)) for g in range(args.groups)]
rows_g = compare(f"probe key {args.seed}: {args.groups} groups of {per}", groups)
worst = int(np.argmax([abs(r["err"][0]) for r in rows_g]))
rows_p = compare(f"group {worst}, probe by probe", [[i] for i in groups[worst]])
if args.out:
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({"seed": args.seed, "groups": rows_g, "worst_group": worst, "single": rows_p}, open(args.out, "w"), indent=1)
