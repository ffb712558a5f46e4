"""Random row-sharded layouts against the single-rank run: R logical ranks as threads of one process (tests/_local_world.py), random n (ragged
last shard), probes, depth, RBF / Matern operators in the fp32 modes and in fp64 -- every row-block launch of the Gram kernels (row0 a multiple of
64, arbitrary nrows, column splits) and the sharded drivers' bookkeeping.   python tests/fuzz_sharded.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in (ROOT, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"), os.path.join(ROOT, "tests")):
    if q not in sys.path:
        sys.path.insert(0, q)
from _local_world import LocalWorld  # noqa: E402
from matfree_extensions.distributed import rows_per_rank, slq_value_and_grad  # noqa: E402
from matfree_extensions.util import gp_util  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    ranks = int(rng.choice([2, 3, 4, 5, 8]))
    n = int(rng.integers(ranks * 64 + 1, 9000))
    # (equal 64-aligned shards: some n leave the last rank without rows -- e.g. 858 rows on 8 ranks: shards of 128, 7 * 128 > 858 -- and are
    #  refused with a ValueError, checked here; the sweep itself runs on a layout that exists)
    try:
        rows_per_rank(n, ranks)
    except ValueError:
        if (ranks - 1) * (-(-(-(-n // ranks)) // 64) * 64) < n:
            bad += 1
            print(f"FAIL rows_per_rank refused a layout that exists: n={n} ranks={ranks}", flush=True)
        n = ranks * (-(-n // (64 * ranks)) * 64) - int(rng.integers(0, 64))  # the next n whose last shard is not empty
    d = int(rng.integers(1, 17))
    if os.environ.get("FUZZ_WIDE_D"):  # up to the wide kernels (d > 32)
        d = int(rng.integers(17, 130))
    p = int(rng.choice([1, 4, 8, 16, 33, 64, 70]))
    k = int(rng.integers(2, 14))
    kernel = str(rng.choice(["rbf", "rbf", "matern32"]))
    dtype, mode = [(torch.float64, "fp32"), (torch.float32, "f16x3"), (torch.float32, "f16x3-matvec"), (torch.float32, "fp32")][int(rng.integers(0, 4))]
    ard = bool(rng.integers(0, 2))
    X = torch.tensor(rng.standard_normal((n, d)) * min(1.0, 4.0 / np.sqrt(d)), dtype=dtype, device=dev)
    raw = [np.full(d, 0.8) if ard else np.array(0.8), np.array(0.3), np.array(-0.7)]
    info = f"case {case}: ranks={ranks} n={n} d={d} p={p} k={k} {kernel} ard={ard} {str(dtype).split('.')[-1]} {mode}"
    try:
        op = gp_util.gram_operator(X, precision=mode, kernel=kernel, noise_minval=1e-4)
        ps = [torch.tensor(r, dtype=dtype, device=dev, requires_grad=True) for r in raw]
        mean, std, grads = slq_value_and_grad(op, torch.log, k, ps, n=n, seed=case, num_probes=p, dtype=dtype, device=dev)

        def body(handle):
            mine = [torch.tensor(r, dtype=dtype, device=dev, requires_grad=True) for r in raw]
            m, s, g = slq_value_and_grad(op, torch.log, k, mine, n=n, seed=case, num_probes=p, row_group_size=handle.world, group=handle,
                                         dtype=dtype, device=dev)
            return m.item(), [t.detach().double().cpu().numpy() for t in g]

        res = LocalWorld(ranks).run(body)
        vt, gt = (1e-10, 1e-7) if dtype == torch.float64 else (3e-5, 2e-3)
        for m, g in res:
            if m != res[0][0]:
                raise AssertionError("ranks disagree")
            ev = abs(m - mean.item()) / max(abs(mean.item()), 0.05 * n)  # (a log-determinant near 0 is a sum of n logs that cancel)
            gs = max(np.abs(x.detach().cpu().numpy()).max() for x in grads)
            eg = max(np.abs(a - b.detach().double().cpu().numpy()).max() for a, b in zip(g, grads)) / gs
            if not (ev < vt and eg < gt):
                raise AssertionError(f"value err {ev:.1e}, gradient err {eg:.1e}")
        if case % 5 == 0:
            print(info + f": ok (value err {ev:.1e}, gradient err {eg:.1e})", flush=True)
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print(f"FAIL {type(exc).__name__}: {exc}   [{info}]", flush=True)
print(f"{cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
