"""world_size-2 gloo test of the N > 1 path: probe sharding + the single fused all-reduce.

The per-rank integrand here is the CPU oracle (tests may use it as the checker); what is under test is
the host logic of matfree_extensions.distributed -- shard bounds, packing, one collective, mean/std and
gradient scaling -- which is the same code the RCCL path runs on GPUs.
"""

import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, K, P_TOTAL = 20, 5, 6


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "experiments-lanczos-adjoints_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _oracle_integrand(A_np):
    from oracle import slq_oracle as orc

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, probes, A):
            vals, grads = [], []
            for v in probes.numpy():
                val, _, (dA,) = orc.integrand_spd_value_and_grad(orc.DenseOp(), K, v, (A.numpy(),))
                vals.append(val)
                grads.append(dA)
            ctx.grads = torch.tensor(np.stack(grads))
            return torch.tensor(vals)

        @staticmethod
        def backward(ctx, g):
            return None, (g[:, None, None] * ctx.grads).sum(0)

    return lambda probes, A: _Fn.apply(probes, A)


def _worker(rank, world, port, out):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from matfree_extensions.distributed import shard_probes, value_and_grad_sharded
    from oracle import slq_oracle as orc

    A = torch.tensor(orc.spd_diag_plus_lowrank(N, 2, seed=0), requires_grad=True)
    first, count = shard_probes(P_TOTAL, rank, world)
    sample_local = lambda: torch.tensor(orc.rademacher(3, count, N, first_probe=first))  # noqa: E731
    mean, std, (gA,) = value_and_grad_sharded(_oracle_integrand(None), sample_local, (A,), num_total=P_TOTAL)
    if rank == 0:
        torch.save({"mean": mean, "std": std, "gA": gA}, out)
    dist.destroy_process_group()


def test_sharded_estimate_equals_single_process(tmp_path):
    _setup_paths()
    from oracle import slq_oracle as orc

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    A = orc.spd_diag_plus_lowrank(N, 2, seed=0)
    probes = orc.rademacher(3, P_TOTAL, N)
    val, (gA,), vals = orc.hutchinson_value_and_grad(orc.DenseOp(), K, probes, (A,))
    assert np.isclose(got["mean"].item(), val, rtol=1e-12)
    assert np.isclose(got["std"].item(), np.std(vals), rtol=1e-9)
    assert np.allclose(got["gA"].numpy(), gA, rtol=1e-10, atol=1e-12)
