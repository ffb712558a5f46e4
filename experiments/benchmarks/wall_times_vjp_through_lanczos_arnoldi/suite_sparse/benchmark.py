"""Wall time of the forward pass and of the adjoint VJP through Lanczos/Arnoldi on a sparse matrix.

Same flags and output files as the reference's harness of the same path (BASELINE config 3); the gradient is taken
w.r.t. the start vector AND every stored matrix entry.  Differences: timings are of the libmfx HIP kernels (there is
nothing to pre-compile, ``--precompile`` only adds a warm-up call); the "backprop through the loop" column
(``custom_vjp=False``, benchmark.py:125-139 of the reference) is torch.autograd through the recurrence written in torch ops
around the HIP operator (``matfree_extensions/_autodiff.py``), for depths up to ``--backprop_until``.
``--synthetic`` builds a matrix of the named one's size when ./data/matrices/<name>/<name>.mtx is absent.
"""

import argparse
import os
import sys
import time

import numpy as np
import torch

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "../../../.."))
sys.path.insert(0, os.path.join(_ROOT, "experiments-lanczos-adjoints_amd"))

from matfree_extensions import arnoldi, lanczos  # noqa: E402
from matfree_extensions.util import exp_util  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument("--lanczos_or_arnoldi", type=str, required=True)
parser.add_argument("--reortho", type=str, required=True)
parser.add_argument("--which_matrix", type=str, default="1138_bus")
parser.add_argument("--num_runs", type=int, default=3)
parser.add_argument("--max_krylov_depth", type=int, default=50)
parser.add_argument("--backprop_until", type=int, default=50)
parser.add_argument("--precompile", action="store_true")
parser.add_argument("--synthetic", action="store_true")
parser.add_argument("--dtype", type=str, default="float32")
args = parser.parse_args()
print(args)

LABEL = f"{args.lanczos_or_arnoldi}_{args.which_matrix}_reortho_{args.reortho}_precompile_{args.precompile}"
print("Label:", LABEL)

device = torch.device("cuda:0")
dtype = getattr(torch, args.dtype)
path = "./data/matrices/"
if args.synthetic or not os.path.exists(f"{path}{args.which_matrix}/{args.which_matrix}.mtx"):
    print("(synthetic stand-in matrix)")
    op, params = exp_util.suite_sparse_synthetic(args.which_matrix, device=device, dtype=dtype)
else:
    op, params = exp_util.suite_sparse_load(args.which_matrix, path=path, device=device, dtype=dtype)
n = op.n
vector = torch.randn(n, dtype=dtype, device=device, generator=torch.Generator(device=device).manual_seed(1))


def flatten(tree):
    if torch.is_tensor(tree):
        return [tree]
    return [t for x in tree for t in flatten(x)]


def make(kdepth, custom_vjp=True):
    make_alg = {"arnoldi": arnoldi.hessenberg, "lanczos": lanczos.tridiag}[args.lanczos_or_arnoldi]
    algorithm = make_alg(op, kdepth, custom_vjp=custom_vjp, reortho=args.reortho)

    def decompose(v, p):
        return flatten(algorithm(v, p))

    return decompose


def timed(fun):
    if args.precompile:
        fun()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.num_runs):
        fun()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.num_runs


times_fwdpass, times_custom, times_autodiff = [], [], []
step = args.backprop_until // 10
krylov_depths = np.arange(step, args.max_krylov_depth + step, step, dtype=int)
for krylov_depth in krylov_depths:
    krylov_depth = int(krylov_depth)
    print("Krylov-depth:", krylov_depth)
    implementation = make(krylov_depth)

    with torch.no_grad():
        time_fwdpass = timed(lambda: implementation(vector, params))
    times_fwdpass.append(time_fwdpass)
    print("Time (forward pass):\n\t", time_fwdpass)

    v, p = vector.clone().requires_grad_(True), params.clone().requires_grad_(True)
    outs = implementation(v, p)
    gen = torch.Generator(device=device).manual_seed(krylov_depth)
    dnu = [torch.randn(o.shape, dtype=dtype, device=device, generator=gen) for o in outs]
    time_custom = timed(lambda: torch.autograd.grad(outs, (v, p), dnu, retain_graph=True))
    times_custom.append(time_custom)
    print("Time (adjoint):\n\t", time_custom)
    if krylov_depth <= args.backprop_until:  # backprop through the loop: forward + backward each run (the graph is the baseline's cost)
        baseline = make(krylov_depth, custom_vjp=False)

        def through_the_loop():
            v2, p2 = vector.clone().requires_grad_(True), params.clone().requires_grad_(True)
            return torch.autograd.grad(baseline(v2, p2), (v2, p2), dnu)

        time_autodiff = timed(through_the_loop)
        times_autodiff.append(time_autodiff)
        print("Time (forward + backprop through the loop):\n\t", time_autodiff)
    print()

print("Saving to a file")
directory = exp_util.matching_directory(os.path.abspath(__file__), "results/")
os.makedirs(directory, exist_ok=True)
np.save(f"{directory}/{LABEL}_krylov_depths.npy", krylov_depths)
np.save(f"{directory}/{LABEL}_times_fwdpass.npy", np.asarray(times_fwdpass))
np.save(f"{directory}/{LABEL}_times_custom.npy", np.asarray(times_custom))
np.save(f"{directory}/{LABEL}_times_autodiff.npy", np.asarray(times_autodiff))
