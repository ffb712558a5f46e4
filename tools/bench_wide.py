"""Gram matvec and parameter sweep at input dimensions beyond 16, for the record: the wide VALU kernels of csrc/mfx_ops.hip (d > 64; the
shapes of the reference's widest UCI loaders, util/uci_util.py: song d = 90, slice d = 385), the exact-fp32 matrix-core kernels
(16 < d <= 64; sweep <= 32) and, for scale, d = 16 on the split kernels.    python tools/bench_wide.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "experiments-lanczos-adjoints_amd"))
import torch
from matfree_extensions.operators import RbfGramOp
dev = torch.device("cuda:0")
for name, n, d, p, ard in (("slice", 53500, 385, 8, False), ("slice ARD", 53500, 385, 8, True), ("song (100k rows)", 100000, 90, 8, False),
                           ("song (100k rows) ARD", 100000, 90, 8, True), ("song (100k rows), 64 vectors", 100000, 90, 64, False), ("d = 128, exact-fp32 matrix cores", 53500, 128, 8, False), ("d = 64, exact-fp32 matrix cores", 53500, 64, 8, False), ("d = 32, pre-packed split kernel h3 (default mode)", 53500, 32, 8, False),
                           ("d = 20 (kegg_directed), the same", 53500, 20, 8, False), ("d = 20 ARD", 53500, 20, 8, True),
                           ("d = 16, split kernels (f16x3)", 53500, 16, 8, False)):
    X = torch.randn(n, d, device=dev) * min(1.0, 4.0 / d ** 0.5)
    op = RbfGramOp(X, noise_minval=1e-4)
    params = [torch.zeros(d if ard else (), device=dev, requires_grad=True)] + [torch.zeros((), device=dev, requires_grad=True) for _ in range(2)]
    v, c = torch.randn(p, n, device=dev), torch.randn(p, n, device=dev)
    with torch.no_grad():
        op(v, *params); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2): op(v, *params)
        torch.cuda.synchronize(); tm = (time.perf_counter() - t0) / 2
    y = op(v, *params); torch.cuda.synchronize(); t0 = time.perf_counter()
    torch.autograd.grad(y, params, c); torch.cuda.synchronize(); tg = time.perf_counter() - t0
    print(f"{name:<58} n={n} d={d} p={p}: matvec {tm * 1e3:8.1f} ms ({2 * n * n * (d + p) / tm * 1e-12:5.1f} TFLOP/s), parameter sweep (batch {p}) {tg * 1e3:8.1f} ms", flush=True)
