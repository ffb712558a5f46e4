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


# ---- rows x probes grids (the strong-scaling layout): host logic + decomposition under gloo --------------------------
N2, D2, K2, P2 = 200, 3, 6, 4  # 2 row shards of 128 and 72 rows (ragged)


def _row_worker(rank, world, port, rows, out):
    _setup_paths()
    import datetime

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    import _sharded_oracle as so
    from matfree_extensions.distributed import RowComm, make_grid, reduce_estimate, shard_probes
    from oracle import slq_oracle as orc

    X = np.random.default_rng(0).standard_normal((N2, D2))
    raw = (np.array(0.4), np.array(0.2), np.array(-1.0))
    row_group, probe_index, probe_groups = make_grid(rows) if world > rows else (None, 0, 1)
    comm = RowComm(N2, row_group)
    assert (comm.world, comm.nloc) == (rows, 128) and comm.nrows == (128 if comm.rank == 0 else 72)
    first, count = shard_probes(P2, probe_index, probe_groups)
    probes = orc.rademacher(3, count, N2, first_probe=first)
    # gather_rows is the inverse of rows() on every rank
    assert np.array_equal(comm.gather_rows(comm.rows(torch.as_tensor(probes))).numpy(), probes)
    op = so.ShardedRbf(X, comm, noise_minval=1e-4)
    vals, grads = [], np.zeros(3)
    for v in probes:
        val, g = so.integrand_value_and_grad(op, K2, v[comm.row0 : comm.row0 + comm.nrows], raw)
        vals.append(val)
        grads += np.array([float(x) for x in g])
    mean, std, (g,) = reduce_estimate(torch.tensor(vals), (torch.tensor(grads),), P2, replicas=rows)
    if rank == world - 1:
        torch.save({"mean": mean, "std": std, "g": g}, out)
    dist.barrier()
    dist.destroy_process_group()


def _single_process_reference():
    from oracle import slq_oracle as orc

    X = np.random.default_rng(0).standard_normal((N2, D2))
    raw = (np.array(0.4), np.array(0.2), np.array(-1.0))
    probes = orc.rademacher(3, P2, N2)
    return orc.hutchinson_value_and_grad(orc.RbfGramOp(X, noise_minval=1e-4), K2, probes, raw)


def _run_grid(tmp_path, world, rows):
    _setup_paths()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "last.pt")
    mp.spawn(_row_worker, args=(world, port, rows, out), nprocs=world, join=True)
    got = torch.load(out)
    val, grads, vals = _single_process_reference()
    assert np.isclose(got["mean"].item(), val, rtol=1e-11)
    assert np.isclose(got["std"].item(), np.std(vals), rtol=1e-8)
    assert np.allclose(got["g"].numpy(), np.array([float(x) for x in grads]), rtol=1e-9, atol=1e-12)


def test_row_sharded_estimate_equals_single_process_world2(tmp_path):
    _run_grid(tmp_path, 2, 2)


def test_rows_x_probes_grid_2x2_equals_single_process(tmp_path):
    _run_grid(tmp_path, 4, 2)


def test_eight_logical_ranks_in_one_process_equal_single_process():
    """The 8-way row shard that `bench.py --gpus 8` runs, with the ranks as eight THREADS (`tests/_local_world.LocalWorld`: the rehearsal
    transport of tests/test_gpu_sharded.py and tools/rehearse_eight_ranks.py -- a GPU box admits at most six processes on its card):
    RowComm / reduce_estimate on a LocalWorld handle and the sharded restatement of the oracle reproduce the single-process oracle.
    n = 8 x 64 - 30: seven full shards and a ragged last one."""
    _setup_paths()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _sharded_oracle as so
    from _local_world import LocalWorld
    from matfree_extensions.distributed import Layout, RowComm, reduce_estimate
    from oracle import slq_oracle as orc

    n, d, k, p, world = 8 * 64 - 30, 3, 6, 3, 8
    X = np.random.default_rng(0).standard_normal((n, d))
    raw = (np.array(0.4), np.array(0.2), np.array(-1.0))
    probes = orc.rademacher(3, p, n)

    def rank_body(handle):
        comm = RowComm(n, handle)
        assert (comm.world, comm.nloc, comm.rank) == (world, 64, handle.rank)
        lay = Layout(n, world, handle)  # pure row sharding over the logical ranks
        assert lay.comm.nrows == comm.nrows and lay.probe_groups == 1 and not lay.native
        assert np.array_equal(comm.gather_rows(comm.rows(torch.as_tensor(probes))).numpy(), probes)
        op = so.ShardedRbf(X, comm, noise_minval=1e-4)
        vals, grads = [], np.zeros(3)
        for v in probes:
            val, g = so.integrand_value_and_grad(op, k, v[comm.row0 : comm.row0 + comm.nrows], raw)
            vals.append(val)
            grads += np.array([float(x) for x in g])
        mean, std, (g,) = reduce_estimate(torch.tensor(vals), (torch.tensor(grads),), p, group=handle, replicas=world)
        return mean.item(), std.item(), g.numpy()

    results = LocalWorld(world).run(rank_body)
    val, grads, vals = orc.hutchinson_value_and_grad(orc.RbfGramOp(X, noise_minval=1e-4), k, probes, raw)
    for mean, std, g in results:
        assert np.isclose(mean, val, rtol=1e-11) and np.isclose(std, np.std(vals), rtol=1e-8)
        assert np.allclose(g, np.array([float(x) for x in grads]), rtol=1e-9, atol=1e-12)


# ---- the other row-sharded drivers: three-term recurrence + adjoint, (P)CG -- decomposition under gloo -----------------------------
def _lz_cg_worker(rank, world, port, out):
    _setup_paths()
    import datetime

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    import _sharded_oracle as so
    from matfree_extensions.distributed import RowComm

    X, raw, v, cot, b, L = _lz_cg_problem()
    comm = RowComm(N2)
    rows = slice(comm.row0, comm.row0 + comm.nrows)
    op = so.ShardedRbf(X, comm, noise_minval=1e-4)
    xs, a, bb, vnorm = so.tridiag_none(op, K2, v[rows], *raw)
    (dxs, da, db) = cot
    dv, dp = so.tridiag_none_vjp(op, K2, raw, xs, a, bb, vnorm, dxs[:, rows], da, db)
    x_fix, r_fix, _ = so.pcg(op, b[rows], raw, Lt=L.T[:, rows], maxiter=5)
    x_ad, _r, steps = so.pcg(op, b[rows], raw, maxiter=200, adaptive=True, atol=1e-8, rtol=0.0, miniter=2)
    gather = lambda t: comm.gather_rows(torch.as_tensor(np.ascontiguousarray(t))).numpy()
    got = {"xs": gather(xs), "a": a, "b": bb, "dv": gather(dv), "dp": [float(g) for g in dp], "x_fix": gather(x_fix), "r_fix": gather(r_fix),
           "x_ad": gather(x_ad), "steps": steps}
    if rank == world - 1:
        torch.save(got, out)
    dist.barrier()
    dist.destroy_process_group()


def _lz_cg_problem():
    rng = np.random.default_rng(4)
    X = rng.standard_normal((N2, D2))
    raw = (np.array(0.4), np.array(0.2), np.array(-1.0))
    v = rng.standard_normal(N2)
    cot = (rng.standard_normal((K2 + 1, N2)), rng.standard_normal(K2), rng.standard_normal(K2))
    b = rng.standard_normal(N2)
    L = 0.3 * rng.standard_normal((N2, 5))  # any low-rank factor makes a valid Woodbury preconditioner
    return X, raw, v, cot, b, L


def test_row_sharded_three_term_recurrence_and_pcg_equal_single_process(tmp_path):
    """tridiag(reortho="none") + its adjoint and (P)CG, restated on row shards with the collectives of mfx_lanczos_*_sharded /
    mfx_pcg_solve_sharded (one gather per operator application, all-reduced scalars), under gloo with 2 ranks (128 + 72 rows)."""
    _setup_paths()
    from oracle import slq_oracle as orc

    X, raw, v, (dxs, da, db), b, L = _lz_cg_problem()
    o = orc.RbfGramOp(X, noise_minval=1e-4)
    (xs_, (a, b_)), (xl, bl) = orc.tridiag_none(o, K2, v, *raw)
    cot = ((dxs[:-1], (da, db[:-1])), (dxs[-1], db[-1]))
    dv, dp = orc.tridiag_none_vjp(o, K2, v, raw, cot)
    _, _, noise = o.constrained(*raw)
    A = lambda u: o.apply(u, *raw)
    x_fix, info = orc.pcg_fixed_step(A, b, lambda u: orc.precondition_solve(L, u, noise), num_matvecs=5)
    x_ad, ainfo = orc.pcg_adaptive(A, b, None, atol=1e-8, rtol=0.0, maxiter=200, miniter=2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "lzcg.pt")
    mp.spawn(_lz_cg_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=False)
    assert np.allclose(got["xs"], np.concatenate([xs_, xl[None]]), rtol=1e-10, atol=1e-12)
    assert np.allclose(got["a"], a, rtol=1e-11) and np.allclose(got["b"], np.concatenate([b_, [bl]]), rtol=1e-11)
    assert np.allclose(got["dv"], dv, rtol=1e-9, atol=1e-11 * np.abs(dv).max())
    assert np.allclose(got["dp"], [float(g) for g in dp], rtol=1e-9)
    assert np.allclose(got["x_fix"], x_fix, rtol=1e-10, atol=1e-12) and np.allclose(got["r_fix"], info["residual_abs"], rtol=1e-8, atol=1e-12)
    assert got["steps"] == ainfo["num_steps"] and np.allclose(got["x_ad"], x_ad, rtol=1e-7, atol=1e-8 * np.abs(x_ad).max())  # 42 CG steps amplify the different summation order


# ---- neighbour exchange plan of a sparse operator (the halo of a stencil) ------------------------------------------------------
def _halo_worker(rank, world, port, out):
    _setup_paths()
    import datetime

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from matfree_extensions.distributed import RowComm

    n = 300  # 3 ranks x 128 rows (last: 44); tridiagonal matrix + one far coupling 0 <-> 299
    rows = np.repeat(np.arange(n), 3)
    cols = np.clip(rows + np.tile([-1, 0, 1], n), 0, n - 1)
    rows, cols = np.concatenate([rows, [0, n - 1]]), np.concatenate([cols, [n - 1, 0]])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    crow = torch.tensor(np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]), dtype=torch.int32)
    col = torch.tensor(cols, dtype=torch.int32)
    comm = RowComm(n)
    plan = comm.plan_exchange(crow, col)
    x = torch.arange(2 * n, dtype=torch.float64).reshape(2, n)  # two vectors
    full = torch.full((2, n), -1.0, dtype=torch.float64)
    full[:, comm.row0 : comm.row0 + comm.nrows] = comm.rows(x)
    comm._exchange(comm.rows(x), full, plan)
    needed = np.unique(cols[(rows >= comm.row0) & (rows < comm.row0 + comm.nrows)])
    ok = bool(torch.equal(full[:, needed], x[:, needed]))
    gathered = [None] * world
    dist.all_gather_object(gathered, (plan, ok))
    if rank == 0:
        torch.save(gathered, out)
    dist.destroy_process_group()


def test_neighbour_exchange_plan_and_transfer(tmp_path):
    _setup_paths()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "halo.pt")
    mp.spawn(_halo_worker, args=(3, port, out), nprocs=3, join=True)
    got = torch.load(out)
    assert all(ok for _plan, ok in got)  # every entry a rank's rows read arrived
    (recv0, send0), (recv1, send1), (recv2, send2) = (p for p, _ in got)
    assert sorted(recv0) == [(1, 128, 129), (2, 299, 300)] and sorted(send0) == [(1, 127, 128), (2, 0, 1)]
    assert sorted(recv1) == [(0, 127, 128), (2, 256, 257)] and sorted(recv2) == [(0, 0, 1), (1, 255, 256)]
