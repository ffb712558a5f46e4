"""(Round 5: this found the wrong lengthscale / outputscale gradients of the split gradient GEMM for non-ARD RBF operators with d = 2 .. 4.)
Random shapes of the matrix-core Gram matvec (and of the parameter-gradient sweep) against the fp64 HIP kernels: n, d, p, kernel family,
ARD or not, row blocks -- a wider net than the fixed cases of tests/test_gpu_matvec_kernels.py.   python tools/fuzz_matvec.py [cases] [seed]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))
from matfree_extensions import _lib  # noqa: E402
from matfree_extensions.operators import RbfGramOp  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def apply_block(op, cparams, V, row0, nrows):
    p, n = V.shape
    desc = op.descriptor(cparams, V.dtype, n)
    desc.row0, desc.nrows = row0, nrows
    ws = _lib.workspace(desc, n, 1, p, V.device)
    y = torch.empty((p, nrows), dtype=V.dtype, device=V.device)
    _lib.check(_lib.get().mfx_op_apply(C.byref(desc), _lib.ptr(V), n, _lib.ptr(y), nrows, p, 0, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(V.device)))
    return y


worst = 0.0
for case in range(cases):
    n = int(rng.choice([rng.integers(65, 700), rng.integers(700, 6000), rng.integers(6000, 30000)]))
    if os.environ.get("FUZZ_LARGE"):  # a few cases at the headline's size class: unsplit sweeps with many chain folds, 2048 tiles
        n = int(rng.integers(60000, 140000))
    d = int(rng.integers(1, 17))
    wide = os.environ.get("FUZZ_WIDE_D", "")
    if wide == "1":  # d = 17 .. 32: the widest padded row of the register kernels (DPAD = 32), which no BASELINE config uses
        d = int(rng.integers(17, 33))
    elif wide:       # d = 33 .. 200: the wide kernels (the d axis in chunks through LDS); the sweeps cost d / 32 times more with ARD
        d = int(rng.integers(33, 201))
        n = min(n, 5000)
    p = int(rng.choice([1, 2, 3, 7, 8, 16, 31, 32, 33, 47, 64, 65, 96, 100, 128, 130]))
    kernel = str(rng.choice(["rbf", "rbf", "matern32", "matern12"]))
    ard = bool(rng.integers(0, 2))
    X = rng.standard_normal((n, d)) * rng.choice([0.3, 1.0, 3.0]) * min(1.0, 4.0 / np.sqrt(d))  # (distances stay O(1) .. O(100) at any d)
    raw = (rng.standard_normal(d) * 0.3 + 0.6 if ard else np.array(0.6 + 0.3 * rng.standard_normal()), np.array(0.3), np.array(-1.0))
    V = rng.standard_normal((p, n)) * np.exp(rng.standard_normal((p, 1)) * 2.0)
    mode = str(rng.choice(["f16x3", "f16x3-matvec", "fp32"]))
    op32 = RbfGramOp(torch.tensor(X, dtype=torch.float32, device=dev), noise_minval=1e-4, kernel=kernel, precision=mode)
    op64 = RbfGramOp(torch.tensor(X, dtype=torch.float64, device=dev), noise_minval=1e-4, kernel=kernel)
    p32 = [torch.tensor(np.asarray(r), dtype=torch.float32, device=dev) for r in raw]
    p64 = [torch.tensor(np.asarray(r), dtype=torch.float64, device=dev) for r in raw]
    V32, V64 = torch.tensor(V, dtype=torch.float32, device=dev), torch.tensor(V, dtype=torch.float64, device=dev)
    ref = op64(V64, *p64)
    if rng.integers(0, 3) == 0 and n > 200:  # a row block: start on a multiple of 64, ragged end
        row0 = int(rng.integers(0, n // 64)) * 64
        nrows = int(rng.integers(1, n - row0 + 1))
        y = apply_block(op32, op32.constrain(*p32), V32, row0, nrows).double()
        r = ref[:, row0:row0 + nrows]
        what = f"rows [{row0}, {row0 + nrows})"
    else:
        y, r, what = op32(V32, *p32).double(), ref, "all rows"
    err = ((y - r).abs().amax(dim=1) / ref.abs().amax(dim=1)).max().item()
    # (Matern: the fp64 operator adds eps of ITS dtype under the square root, util/gp_util.py:99,140 -- K_ii differs by sqrt(eps_fp32) by definition)
    tol = (1e-3 if kernel == "matern12" else 2e-4 if kernel == "matern32" else 0.0) + (2e-4 if mode == "fp32" else 1e-4)
    flag = "" if err < tol else "   <-- FAIL"
    worst = max(worst, err / tol)
    if flag or case % 20 == 0:
        print(f"case {case}: n={n} d={d} p={p} {kernel} ard={ard} {mode} {what}: err {err:.2e}{flag}", flush=True)
    # the parameter-gradient sweep on the same operator (batch = p rows)
    if rng.integers(0, 2) == 0:
        L = torch.tensor(rng.standard_normal((p, n)), dtype=torch.float32, device=dev)
        pg32 = [q.clone().requires_grad_(True) for q in p32]
        pg64 = [q.clone().requires_grad_(True) for q in p64]
        g32 = torch.autograd.grad((L * op32(V32, *pg32)).sum(), pg32)
        g64 = torch.autograd.grad((L.double() * op64(V64, *pg64)).sum(), pg64)
        # error relative to the gradient OR to the scale of its terms (|L| |V| n, times the squared scaled distances for the lengthscale:
        # a gradient that cancels to nothing has no relative error -- e.g. far-apart points, where dK/dl lives on round-off of the diagonal)
        xmax2 = float((X / 0.5).__pow__(2).sum(1).max())
        scale = 1e-7 * float(L.abs().max() * V32.abs().max()) * n
        names = ("l", "s", "noise")
        for idx, (a, b) in enumerate(zip(g32, g64)):
            if kernel != "rbf" and idx < 2:
                continue  # Matern: the fp64 operator adds eps of ITS dtype under the square root -- another K_ii by definition (see above)
            e = ((a.double() - b).abs().max() / (b.abs().max() + scale * (1.0 + xmax2 if idx == 0 else 1.0))).item()
            gt = 5e-3 if mode == "fp32" else 2e-3
            if e > gt:
                print(f"case {case}: n={n} d={d} p={p} {kernel} ard={ard} {mode}: gradient d{names[idx]} err {e:.2e}   <-- FAIL", flush=True)
                worst = max(worst, e / gt)
print(f"{cases} cases, worst error / tolerance = {worst:.2f}")
