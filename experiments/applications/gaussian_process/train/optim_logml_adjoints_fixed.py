"""GP hyper-parameter training on the Krylov log-marginal likelihood: SLQ log-determinant + preconditioned CG.

Same structure and flags as the reference's training script of the same name (Matern-3/2 kernel, constant mean, pivoted
partial-Cholesky preconditioner, fixed-step PCG, Adam).  There is no network here, so instead of the '3droad' download a
synthetic 3-d regression set of the requested size is generated (``--data file.npy`` loads an (N, 4) array instead).
``--num_partitions`` is accepted for parity: the native Gram operator never materialises kernel tiles.
"""

import argparse
import os
import sys
import time

import numpy as np
import torch

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "../../../.."))
sys.path.insert(0, os.path.join(_ROOT, "experiments-lanczos-adjoints_amd"))

from matfree_extensions import cg, hutchinson, low_rank  # noqa: E402
from matfree_extensions.util import gp_util  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument("--name", type=str, default="run")
parser.add_argument("--seed", type=int, default=1)
parser.add_argument("--num_data", type=int, required=True)
parser.add_argument("--rank_precon", type=int, default=100)
parser.add_argument("--num_partitions", type=int, default=1)
parser.add_argument("--num_matvecs", type=int, default=10)
parser.add_argument("--num_samples", type=int, default=8)
parser.add_argument("--num_epochs", type=int, default=5)
parser.add_argument("--data", type=str, default="")
parser.add_argument("--precision", type=str, default="f16x3")
args = parser.parse_args()
print(f"\nRUNNING: {args.name}\n\n{args}")

dev, dt = torch.device("cuda:0"), torch.float32
if args.data:
    data = torch.as_tensor(np.load(args.data), dtype=dt)[: args.num_data]
else:
    g = torch.Generator().manual_seed(args.seed)
    xyz = torch.rand((args.num_data, 3), generator=g) * 4 - 2
    alt = torch.sin(2 * xyz[:, 0]) * torch.cos(xyz[:, 1]) + 0.3 * xyz[:, 2] + 0.1 * torch.randn(args.num_data, generator=g)
    data = torch.cat([xyz, alt[:, None]], 1).to(dt)
ntrain = int(0.9 * len(data))
train_x, train_y, test_x, test_y = data[:ntrain, :-1], data[:ntrain, -1], data[ntrain:, :-1], data[ntrain:, -1]
mean, std = train_x.mean(0, keepdim=True), train_x.std(0, keepdim=True) + 1e-6
train_x, test_x = ((train_x - mean) / std).to(dev), ((test_x - mean) / std).to(dev)
mean, std = train_y.mean(), train_y.std()
train_y, test_y = ((train_y - mean) / std).to(dev), ((test_y - mean) / std).to(dev)
print("Train:", tuple(train_x.shape), "Test:", tuple(test_x.shape))

num_matvecs_train_cg = args.num_samples * args.num_matvecs
noise_bd = 1e-4
constrain = gp_util.constraint_greater_than(noise_bd)
solve_p = cg.pcg_fixed_step(num_matvecs_train_cg)
sample = hutchinson.sampler_rademacher(torch.ones(ntrain, dtype=dt, device=dev), num=args.num_samples)
# the reference draws num_samples sequential single-probe batches; here they are ONE batch of probes (one Gram sweep
# serves them all): same estimator, std over probes is reported by hutchinson_batch-style info when num_batches > 1
logdet = gp_util.krylov_logdet_slq(args.num_matvecs, sample=sample, num_batches=1)
precondition = low_rank.preconditioner(low_rank.cholesky_partial_pivot(rank=args.rank_precon))
logpdf_p = gp_util.logpdf_krylov_p(solve_p=solve_p, logdet=logdet)
gram_matvec = gp_util.gram_matvec_partitioned(args.num_partitions, checkpoint=False, precision=args.precision)
likelihood, _ = gp_util.likelihood_pdf_p(gram_matvec, logpdf_p, precondition, constrain=constrain)
m, _ = gp_util.mean_constant(shape_out=())
k, _ = gp_util.kernel_scaled_matern_32(shape_in=(3,), shape_out=())
prior = gp_util.model_gp(m, k)
loss = gp_util.target_logml(prior, likelihood)

params = {q: torch.zeros((), dtype=dt, device=dev, requires_grad=True)
          for q in ("constant_value", "raw_lengthscale", "raw_outputscale", "raw_noise")}


def mll_lanczos(key):
    val, info = loss(train_x, train_y, key, params_mean={"constant_value": params["constant_value"]},
                     params_kernel={"raw_lengthscale": params["raw_lengthscale"], "raw_outputscale": params["raw_outputscale"]},
                     params_likelihood={"raw_noise": params["raw_noise"]})
    return -val / ntrain, info


def predict_mean():
    solve = cg.pcg_adaptive(atol=1e-2, rtol=0.0, maxiter=1_000)
    lik, _ = gp_util.likelihood_condition_p(gram_matvec, solve, precondition=precondition, constrain=constrain)
    post, _ = gp_util.target_posterior(prior, lik)(
        train_x, train_y, {"constant_value": params["constant_value"].detach()},
        {"raw_lengthscale": params["raw_lengthscale"].detach(), "raw_outputscale": params["raw_outputscale"].detach()},
        {"raw_noise": params["raw_noise"].detach()})
    return post(test_x)


def timed(fun):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fun()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


with torch.no_grad():
    mll_lanczos(args.seed)  # warm-up
    (value, aux), t = timed(lambda: mll_lanczos(args.seed))
print("Runtime (value):", t)
residual = aux["logpdf"]["solve"]["residual_abs"]
print("CG error:", float(torch.linalg.vector_norm(residual) / np.sqrt(residual.numel())))


def value_and_grad(key):
    for q in params.values():
        q.grad = None
    value, info = mll_lanczos(key)
    value.backward()
    return value, info


value_and_grad(args.seed)
_, t = timed(lambda: value_and_grad(args.seed))
print("Runtime (value-and-gradient):", t)

optimizer = torch.optim.Adam(list(params.values()), lr=0.1)
for epoch in range(args.num_epochs):
    (value, info), t = timed(lambda: value_and_grad(args.seed + 1 + epoch))
    optimizer.step()
    (pred, pinfo), tp = timed(predict_mean)
    rmse = float(torch.sqrt(torch.mean((pred - test_y) ** 2)))
    print(f"epoch {epoch}: loss {float(value):.5f}  rmse {rmse:.4f}  step {t:.3f}s  predict {tp:.3f}s "
          f"(cg steps {int(pinfo['solve']['num_steps'])})", flush=True)
