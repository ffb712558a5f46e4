"""Random shapes of the whole SLQ value-and-gradient path in the fp32 modes against the fp64 HIP path: n, d, p, k, kernel family, ARD or not --
the deferred gradient sweep at batches p k up to a few thousand rows, probe chunks beyond 64, column splits at small n, d <= 4 ...
    python tools/fuzz_slq.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))
from matfree_extensions import hutchinson, lanczos  # noqa: E402
from matfree_extensions.util import gp_util  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    n = int(rng.choice([rng.integers(300, 2500), rng.integers(2500, 12000), rng.integers(12000, 40000)]))
    if os.environ.get("FUZZ_LARGE"):  # the headline's size class (the fp64 leg takes ~10-40 s per case)
        n = int(rng.integers(60000, 140000))
    d = int(rng.integers(1, 17))
    if os.environ.get("FUZZ_WIDE_D"):  # up to the wide kernels (d > 32)
        d = int(rng.integers(17, 130))
        n = min(n, 12000)
    p = int(rng.choice([1, 3, 8, 16, 31, 33, 64, 65, 100, 130]))
    k = int(rng.integers(3, 25))
    kernel = str(rng.choice(["rbf", "rbf", "matern32"]))
    ard = bool(rng.integers(0, 2))
    mode = str(rng.choice(["f16x3", "f16x3", "f16x3-matvec", "fp32"]))
    X = torch.tensor(rng.standard_normal((n, d)) * min(1.0, 4.0 / np.sqrt(d)), dtype=torch.float64, device=dev)
    raw = [np.full(d, 0.9) + 0.2 * rng.standard_normal(d) if ard else np.array(0.9), np.array(0.3), np.array(-0.5)]
    res = {}
    for dt, prec in ((torch.float64, "fp32"), (torch.float32, mode)):
        params = [torch.tensor(r, dtype=dt, device=dev, requires_grad=True) for r in raw]
        op = gp_util.gram_operator(X.to(dt), precision=prec, kernel=kernel, noise_minval=1e-4)
        probes = hutchinson.sampler_rademacher(X[:, 0].to(dt), num=p)(case)
        vals = lanczos.integrand_spd(torch.log, k, op)(probes, *params)
        g = torch.autograd.grad(vals.sum(), params)
        res[dt] = (vals.detach().double().cpu().numpy(), [x.detach().double().cpu().numpy() / p for x in g])
    v64, g64 = res[torch.float64]
    v32, g32 = res[torch.float32]
    ev = np.abs(v32 - v64).max() / np.abs(v64).max()
    gs = max(np.abs(x).max() for x in g64)   # gradient components against the largest one (the small ones cancel)
    eg = max(np.abs(a - b).max() for a, b in zip(g32, g64)) / gs
    vt, gt = (2e-3, 2e-2) if mode == "fp32" else (3e-4, 5e-3)
    flag = "" if (ev < vt and eg < gt) else "   <-- FAIL"
    bad += bool(flag)
    if flag or case % 10 == 0:
        print(f"case {case}: n={n} d={d} p={p} k={k} {kernel} ard={ard} {mode}: value err {ev:.1e}, gradient err {eg:.1e}{flag}", flush=True)
print(f"{cases} cases, {bad} failures")
