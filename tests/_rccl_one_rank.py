"""Child process of tests/test_gpu_sharded.py::test_rccl_one_rank_group_runs_the_collective_callbacks: a ONE-rank "nccl" (= RCCL)
process group on the one GPU of the box, the row-sharded SLQ value-and-gradient with the collectives forced through it, against
the single-device drivers.  Exit code 0 = equal."""
import datetime
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "experiments-lanczos-adjoints_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
    from matfree_extensions import hutchinson, lanczos
    from matfree_extensions.distributed import RowComm
    from matfree_extensions.operators import RowShardedOp
    from matfree_extensions.util import gp_util

    n, d, k, p = 2304, 8, 12, 8
    g = torch.Generator().manual_seed(7)
    X = torch.randn((n, d), generator=g, dtype=torch.float32).to(dev)
    op = gp_util.gram_operator(X, precision="f16x3")
    params = [torch.tensor(v, dtype=torch.float32, device=dev) for v in (0.5, 0.2, -1.0)]
    probes = hutchinson.sampler_rademacher(torch.empty(n, dtype=torch.float32, device=dev), num=p)(3)

    def run(matvec):
        ps = [q.clone().requires_grad_(True) for q in params]
        vals = lanczos.integrand_spd(torch.log, k, matvec)(probes, *ps)
        return vals.detach(), torch.autograd.grad(vals.sum(), ps)

    v0, g0 = run(op)
    comm = RowComm(n)
    comm.force_collectives = True
    assert comm.world == 1
    v1, g1 = run(RowShardedOp(op, comm))
    torch.cuda.synchronize()
    ok = torch.allclose(v0, v1, rtol=2e-5)
    for a, b in zip(g0, g1):
        ok = ok and torch.allclose(a, b, rtol=5e-4, atol=5e-4 * a.abs().max().item())
    print("values", v0[:3].tolist(), v1[:3].tolist(), "ok", ok)
    # the NATIVE path: libmfx's own one-rank RCCL communicator (ncclCommInitRank / ncclAllGather / ncclAllReduce issued from C, the
    # grouped per-vector all-gather straight into the operator input); same arithmetic -> the same numbers as the callback path
    from matfree_extensions.distributed import NativeRowComm

    ncomm = NativeRowComm(n, gather="grouped")
    v2, g2 = run(RowShardedOp(op, ncomm))
    torch.cuda.synchronize()
    same = torch.equal(v1, v2) and all(torch.equal(a, b) for a, b in zip(g1, g2))
    print("native communicator: values", v2[:3].tolist(), "bit-identical to the callback path:", same)
    ok = ok and same
    # the other sharded drivers through the native communicator: three-term recurrence + adjoint, (P)CG -- against the callback path
    from matfree_extensions import cg

    def others(comm_):
        sop = RowShardedOp(op, comm_)
        ps = [q.clone().requires_grad_(True) for q in params]
        (xs, (al, be)), (xl, bl) = lanczos.tridiag(sop, 6, reortho="none")(probes[:3], *ps)
        ((al * al).sum() + (be * be).sum() + (bl * bl).sum() + xs[:, :, ::5].sum()).backward()
        with torch.no_grad():
            x, info = cg.cg_adaptive(atol=1e-5, rtol=0.0, maxiter=100, miniter=1)(sop.bind(*params), probes[:3])
        return [al.detach(), be.detach(), x, info["num_steps"]] + [q.grad for q in ps]

    same2 = all(torch.equal(a, b) for a, b in zip(others(comm), others(ncomm)))
    print("native communicator, three-term + CG: bit-identical to the callback path:", same2)
    ok = ok and same2
    ncomm.close()
    # the other gather leg (the default): pack, ONE ncclAllGather of the whole (p, nloc) shard, unpack -- same numbers again
    pcomm = NativeRowComm(n)
    assert pcomm.gather == "packed"
    v3, g3 = run(RowShardedOp(op, pcomm))
    torch.cuda.synchronize()
    same3 = torch.equal(v1, v3) and all(torch.equal(a, b) for a, b in zip(g1, g3)) and pcomm.self_test()
    print("native communicator, packed gather: bit-identical to the callback path:", same3)
    ok = ok and same3
    # what RCCL itself reports for libmfx's communicator (ncclCommCount / ncclCommUserRank: the figure bench.py prints next to
    # torch.distributed's world size), and the switch between the gather legs on a live communicator (bench.py times both legs of one group)
    counted = pcomm.rccl_count() == (1, 0)
    pcomm.set_gather("grouped")
    v4, g4 = run(RowShardedOp(op, pcomm))
    pcomm.set_gather("packed")
    v5, g5 = run(RowShardedOp(op, pcomm))
    torch.cuda.synchronize()
    switched = torch.equal(v1, v4) and torch.equal(v1, v5) and all(torch.equal(a, b) for a, b in zip(g1, g4)) and pcomm.gather == "packed"
    print("native communicator: ranks / rank as RCCL reports them:", pcomm.rccl_count(), "; gather legs switched on the live communicator:", switched)
    ok = ok and counted and switched
    pcomm.close()
    # what Layout does by default on an RCCL group: native communicator after an agreed availability check and self-test; when the
    # self-test (here: made to) fails on a rank, ALL ranks take the torch.distributed callbacks, with a warning
    import warnings

    from matfree_extensions import distributed as D

    c1 = D._native_or_callbacks(n, None)
    agreed = isinstance(c1, NativeRowComm) and c1.self_test()
    c1.close()
    real = NativeRowComm.self_test
    NativeRowComm.self_test = lambda self: False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        c2 = D._native_or_callbacks(n, None)
    NativeRowComm.self_test = real
    fell_back = type(c2) is RowComm and any("native RCCL" in str(x.message) for x in w)
    print("default path: native after self-test:", agreed, "; fallback when a self-test fails:", fell_back)
    ok = ok and agreed and fell_back
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
