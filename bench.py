"""Headline benchmark: SLQ log-det value-and-gradient for a matrix-free RBF GP kernel (BASELINE config 4).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one SLQ log-determinant value AND gradient w.r.t. (raw_lengthscale, raw_outputscale,
raw_noise) over this rank's batch of Hutchinson probes:  40 fully re-orthogonalised Lanczos steps
(forward), the k x k eigen-quadrature, the Arnoldi adjoint scan (40 more Gram matvecs) and the deferred
parameter-gradient sweep, followed by the single fused all-reduce of [sum q, sum q^2, sum dq/dtheta].
Operator: X ~ N(0,1) of shape (131072, 8), lengthscale 2, outputscale 1, noise 0.1, fp32, 64 probes per
GPU (weak scaling: probes are independent units, the operator is replicated, no data-path collective).

Prints ONE JSON line on rank 0 (see the README / DESIGN.md for the field meanings).
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "experiments-lanczos-adjoints_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--k", type=int, default=40)
    ap.add_argument("--probes-per-gpu", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-n", type=int, default=4608)
    ap.add_argument("--kernel", default="rbf", choices=["rbf", "matern32", "matern12"],
                    help="kernel family (BASELINE config 4 is the RBF kernel; the reference's UCI runs use matern32)")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f16x3-matvec", "fp32"],
                    help="arithmetic of the fp32 Gram kernels: 3 x f16 split on the f16 matrix pipe (default; -matvec keeps the gradient GEMM in exact fp32) or exact fp32 MFMA")
    return ap.parse_args()


def inv_softplus(x):
    return float(np.log(np.expm1(x)))


def cpu_baseline(args):
    """The oracle (NumPy port of the reference algorithm) on a bounded sample of the same workload:
    identical d, k and hyper-parameters, ONE probe, n reduced to --cpu-sample-n; the Gram work scales
    as n^2, so the full-size figure is the measured one times (n_sample / n)^2 (labelled estimate)."""
    from threadpoolctl import threadpool_limits

    from oracle import slq_oracle as orc

    ns = args.cpu_sample_n
    rng = np.random.default_rng(4)
    X = rng.standard_normal((ns, args.d)).astype(np.float32)
    raw = tuple(np.float32(v) for v in (inv_softplus(2.0), inv_softplus(1.0), inv_softplus(0.1)))
    probes = orc.rademacher(5, 1, ns, dtype=np.float32)
    threads = min(16, len(os.sched_getaffinity(0)))  # the 1-GPU box's CPU share is 16 cores
    with threadpool_limits(limits=threads):
        t0 = time.perf_counter()
        orc.hutchinson_value_and_grad(orc.RbfGramOp(X, cache_limit=0), args.k, probes, raw)
        dt = time.perf_counter() - t0
    measured = 1.0 / dt
    est_full = measured * (ns / args.n) ** 2
    return {
        "value": est_full,
        "unit": "probes/s",
        "cores": threads,
        "kind": "port",
        "sample": f"NumPy oracle, 1 probe, k={args.k}, d={args.d}, n={ns} (of {args.n}): {dt:.2f} s measured = "
                  f"{measured:.4f} probes/s at n={ns}; value = that x (n_sample/n)^2 (Gram work ~ n^2), BLAS threads={threads}",
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (there is no CPU product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm; one rank per GPU

    from matfree_extensions import _lib, hutchinson, lanczos
    from matfree_extensions.distributed import reduce_estimate, shard_probes
    from matfree_extensions.util import gp_util

    n, d, k, p = args.n, args.d, args.k, args.probes_per_gpu
    p_total = p * world
    gen = torch.Generator().manual_seed(4)
    X = torch.randn((n, d), generator=gen, dtype=torch.float32).to(dev)
    params = [torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True)
              for v in (inv_softplus(2.0), inv_softplus(1.0), inv_softplus(0.1))]
    op = gp_util.gram_operator(X, precision=args.precision, kernel=args.kernel)
    integrand = lanczos.integrand_spd(torch.log, k, op)
    first, count = shard_probes(p_total, rank, world)
    sampler = hutchinson.sampler_rademacher(X[:, 0], num=count)

    def step(seed):
        probes = sampler((seed, first))  # this rank's slice of ONE global +-1 probe matrix
        values = integrand(probes, *params)
        grads = torch.autograd.grad(values.sum(), params)
        return reduce_estimate(values.detach(), grads, p_total)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        out = step(100 + w)
    fence()
    _lib.timing_reset()
    _lib.timing_enable(True)
    t0 = time.perf_counter()
    for s in range(args.steps):
        out = step(s)
    fence()
    elapsed = time.perf_counter() - t0
    _lib.timing_enable(False)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()

    apply_ms, apply_cnt = _lib.timing_read(0)
    grad_ms, grad_cnt = _lib.timing_read(1)
    vec_ms, vec_cnt = _lib.timing_read(2)
    _lib.timing_reset()

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = p_total * args.steps / elapsed
        # dominant kernel: the RBF Gram matvec.  Algorithmic flops per launch: the contraction 2 n^2 p
        # (what the matrix cores must deliver); distance + exp are extra work priced in DESIGN.md.
        flops_launch = 2.0 * n * n * p
        avg_ms = apply_ms / max(apply_cnt, 1)
        achieved = flops_launch / (avg_ms * 1e-3) / 1e12 if apply_cnt else 0.0
        split = args.precision.startswith("f16x3")
        peak = MFMA_F16_PEAK_TFLOPS if split else MFMA_F32_PEAK_TFLOPS
        kernel = ("k_rbf_mfma_apply_h3 (Gram matvec, fp32 emulated by 3 f16 MFMA products, distances included)"
                  if split else "k_rbf_mfma_apply (Gram matvec, exact fp32 MFMA)")
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            traffic = json.load(open(tfile)).get("k_rbf_mfma_apply_h3_hbm_bytes_per_launch" if split
                                                 else "k_rbf_mfma_apply_hbm_bytes_per_launch")
        mean, std, grads = out
        line = {
            "metric": "slq_logdet_value_and_grad_throughput",
            "value": value,
            "unit": "probes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if not split else "f32 (Gram contraction emulated with 3 f16 MFMA products, fp32 accumulate)",
            "data": "synthetic",
            "config": {
                "workload": f"matrix-free RBF GP kernel N={n} d={d}, SLQ log-det value+grad, {k} Lanczos steps "
                            f"(full reortho) x {p} probes per GPU, fp32 (BASELINE config 4)",
                "N": n, "d": d, "krylov_depth": k, "probes_per_gpu": p, "probes_total": p_total,
                "parallelism": f"probe-sharded x{world}, operator replicated, one all-reduce per step",
                "gram_precision": args.precision, "kernel": args.kernel,
            },
            "roofline": {
                "kernel": kernel + f", {apply_cnt // max(args.steps, 1)} launches per step",
                "bound": "mfma",
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                "traffic": traffic,
                "avg_launch_ms": avg_ms,
                "launches": apply_cnt,
                "algorithmic_flops_per_launch": flops_launch,
                # the 3-product emulation executes 3x the algorithmic MFMA flops: its ceiling is peak / 3
                "executed_mfma_flops_factor": 3.0 if split else 1.0,
                "frac_of_fp32_mfma_peak": achieved / MFMA_F32_PEAK_TFLOPS,
            },
            "breakdown_ms_per_step": {
                "gram_matvec": apply_ms / args.steps,
                "param_grad_sweep": grad_ms / args.steps,
                "krylov_vector_kernels": vec_ms / args.steps,
            },
            "param_grad_gemm": {
                # S = L^T R over all (probe, step) pairs, batch = p (k + 1): algorithmic flops 2 n^2 batch, once per step
                "algorithmic_flops": 2.0 * n * n * p * (k + 1),
                "achieved_TFLOPs": 2.0 * n * n * p * (k + 1) / max(grad_ms / max(grad_cnt, 1) * 1e-3, 1e-12) / 1e12,
                "peak_TFLOPs": MFMA_F16_PEAK_TFLOPS if args.precision == "f16x3" else MFMA_F32_PEAK_TFLOPS,
                "executed_mfma_flops_factor": 3.0 if args.precision == "f16x3" else 1.0,
            },
            "krylov_vector_hbm": {
                # B_fwd + B_bwd = p n s [2k(k+1)+3k] + p n s [3k^2+9k] algorithmic bytes
                "algorithmic_GB_per_step": p * n * 4 * (2 * k * (k + 1) + 3 * k + 3 * k * k + 9 * k) / 1e9,
                "achieved_GBps": (p * n * 4 * (2 * k * (k + 1) + 3 * k + 3 * k * k + 9 * k) / 1e9)
                                 / max(vec_ms / args.steps * 1e-3, 1e-12),
                "peak_GBps": HBM_PEAK_GBS,
            },
            "result": {"logdet_mean": float(mean), "probe_std": float(std), "grad": [float(g) for g in grads]},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
