"""Row-sharded Krylov drivers on the GPU (SURVEY.md §8(e), strong-scaling layout of BASELINE config 4).

* row blocks of every native operator (`mfx_operator.row0 / nrows`): apply, transpose-apply and the parameter sweep of a
  block equal the corresponding rows / the partial sums of the whole operator;
* `mfx_arnoldi_*_sharded` with a one-rank communicator (all-gather = copy, all-reduce = identity) reproduce the
  single-device drivers;
* 2 and 4 PROCESSES sharing this one GPU (gloo moves the collectives through the host; RCCL needs one GPU per rank, which
  the 1-GPU test box does not have): rows x probes grids 2x1, 4x1 and 2x2 reproduce the single-process estimate.
"""

import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rbf(n, d, dtype, precision="fp32", kernel="rbf", ard=True, seed=0):
    from matfree_extensions.util import gp_util

    g = torch.Generator().manual_seed(seed)
    X = torch.randn((n, d), generator=g, dtype=torch.float64).to(dtype).to(_dev())
    raw_l = torch.full((d,), 0.9, dtype=dtype, device=_dev()) if ard else torch.tensor(0.9, dtype=dtype, device=_dev())
    params = [raw_l, torch.tensor(0.3, dtype=dtype, device=_dev()), torch.tensor(-1.0, dtype=dtype, device=_dev())]
    return gp_util.gram_operator(X, noise_minval=1e-4, precision=precision, kernel=kernel), params


def _apply_block(op, cparams, V, row0, nrows, transpose=0):
    from matfree_extensions import _lib

    p, n = V.shape
    desc = op.descriptor(cparams, V.dtype, n)
    desc.row0, desc.nrows = row0, nrows  # nrows = 0: the whole operator
    nrows = nrows or n
    ws = _lib.workspace(desc, n, 1, p, V.device)
    y = torch.empty((p, nrows), dtype=V.dtype, device=V.device)
    _lib.check(_lib.get().mfx_op_apply(C.byref(desc), _lib.ptr(V), n, _lib.ptr(y), nrows, p, transpose, _lib.ptr(ws), ws.numel(),
                                       _lib.stream_ptr(V.device)))
    return y


def _grad_block(op, cparams, L, R, row0, nrows):
    from matfree_extensions import _lib

    batch, n = R.shape
    desc = op.descriptor(cparams, R.dtype, n)
    desc.row0, desc.nrows = row0, nrows
    ws = _lib.workspace(desc, n, 1, batch, R.device)
    gs, gt = op.new_grads(*cparams)
    Lb = L[:, row0 : row0 + nrows].contiguous()
    _lib.check(_lib.get().mfx_op_vjp_params(C.byref(desc), _lib.ptr(Lb), nrows, _lib.ptr(R), n, batch, C.byref(gs), _lib.ptr(ws),
                                            ws.numel(), _lib.stream_ptr(R.device)))
    return gt


@pytest.mark.parametrize("dtype,precision,tol", [(torch.float64, "fp32", 1e-12), (torch.float32, "fp32", 2e-5),
                                                 (torch.float32, "f16x3-matvec", 2e-5), (torch.float32, "f16x3", 2e-5)])
@pytest.mark.parametrize("kernel", ["rbf", "matern32"])
def test_gram_row_blocks_equal_the_rows_of_the_whole_operator(dtype, precision, tol, kernel):
    n, d, p = 2600, 8, 40  # not a multiple of the 512-row workgroups; blocks start on multiples of 64
    op, params = _rbf(n, d, dtype, precision, kernel)
    cparams = op.constrain(*params)
    g = torch.Generator().manual_seed(1)
    V = torch.randn((p, n), generator=g, dtype=torch.float64).to(dtype).to(_dev())
    full = _apply_block(op, cparams, V, 0, 0)
    for row0, nrows in [(0, 1344), (1344, 1256), (640, 64), (2560, 40)]:
        blk = _apply_block(op, cparams, V, row0, nrows)
        ref = full[:, row0 : row0 + nrows]
        assert torch.allclose(blk, ref, rtol=tol, atol=tol * ref.abs().max().item()), (row0, nrows, (blk - ref).abs().max())
    # parameter sweep: the partial sums of the row blocks add up to the whole sweep
    L = torch.randn((48, n), generator=g, dtype=torch.float64).to(dtype).to(_dev())
    R = torch.randn((48, n), generator=g, dtype=torch.float64).to(dtype).to(_dev())
    whole = _grad_block(op, cparams, L, R, 0, n)
    parts = [_grad_block(op, cparams, L, R, r0, nr) for r0, nr in [(0, 1344), (1344, 1256)]]
    gtol = 1e-10 if dtype == torch.float64 else 2e-4
    for w, a, b in zip(whole, *parts):
        assert torch.allclose(w, a + b, rtol=gtol, atol=gtol * w.abs().max().item()), (w, a + b)


def test_dense_and_csr_row_blocks():
    from matfree_extensions.operators import CsrOp, DenseOp

    n, p = 300, 5
    g = torch.Generator().manual_seed(2)
    A = torch.randn((n, n), generator=g, dtype=torch.float64).to(_dev())
    V = torch.randn((p, n), generator=g, dtype=torch.float64).to(_dev())
    L = torch.randn((7, n), generator=g, dtype=torch.float64).to(_dev())
    R = torch.randn((7, n), generator=g, dtype=torch.float64).to(_dev())
    dop = DenseOp()
    mask = torch.rand((n, n), generator=g) < 0.05
    rows, cols = mask.nonzero(as_tuple=True)
    cop, vals, _ = CsrOp.from_coo(rows, cols, A.cpu()[rows, cols], n, _dev())
    Asp = torch.zeros_like(A)
    Asp[rows.to(_dev()), cols.to(_dev())] = A[rows.to(_dev()), cols.to(_dev())]
    for op, cparams, M in [(dop, (A,), A), (cop, (vals,), Asp)]:
        for transpose in (0, 1):
            ref = V @ (M if transpose else M.T)
            for row0, nrows in [(0, 128), (128, 172), (37, 5)]:
                blk = _apply_block(op, cparams, V, row0, nrows, transpose)
                assert torch.allclose(blk, ref[:, row0 : row0 + nrows], rtol=1e-12, atol=1e-12)
        whole = _grad_block(op, cparams, L, R, 0, n)[0]
        parts = sum(_grad_block(op, cparams, L, R, r0, nr)[0] for r0, nr in [(0, 128), (128, 172)])
        assert torch.allclose(whole, parts, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dtype,precision,vtol,gtol", [(torch.float64, "fp32", 1e-10, 1e-8), (torch.float32, "f16x3", 2e-5, 5e-4)])
def test_sharded_drivers_with_one_rank_equal_the_single_device_drivers(dtype, precision, vtol, gtol):
    from matfree_extensions import hutchinson, lanczos
    from matfree_extensions.distributed import RowComm
    from matfree_extensions.operators import RowShardedOp

    n, d, k, p = 2304, 8, 12, 8
    op, params = _rbf(n, d, dtype, precision, ard=False)
    probes = hutchinson.sampler_rademacher(torch.empty(n, dtype=dtype, device=_dev()), num=p)(3)

    def run(matvec):
        ps = [q.clone().requires_grad_(True) for q in params]
        vals = lanczos.integrand_spd(torch.log, k, matvec)(probes, *ps)
        return vals.detach(), torch.autograd.grad(vals.sum(), ps)

    v0, g0 = run(op)
    v1, g1 = run(RowShardedOp(op, RowComm(n)))
    assert torch.allclose(v0, v1, rtol=vtol)
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b, rtol=gtol, atol=gtol * a.abs().max().item()), (a, b)


# ---- several processes on this one GPU -----------------------------------------------------------------------------
_MP_SHAPE = (2290, 8, 12, 8)  # n = 2290: 4 ranks own 576, 576, 576, 562 rows (ragged last shard, no 16-byte alignment)


def _worker(rank, world, port, rows, out, dtype_name, precision):
    for p in (ROOT, os.path.join(ROOT, "experiments-lanczos-adjoints_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    import datetime

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from matfree_extensions.distributed import slq_value_and_grad

    dtype = getattr(torch, dtype_name)
    n, d, k, p = _MP_SHAPE
    op, params = _rbf(n, d, dtype, precision, ard=False)
    params = [q.clone().requires_grad_(True) for q in params]
    mean, std, grads = slq_value_and_grad(op, torch.log, k, params, n=n, seed=3, num_probes=p, row_group_size=rows,
                                          dtype=dtype, device=_dev())
    torch.cuda.synchronize()
    if rank == world - 1:  # the LAST rank reports: its shard is the ragged one
        torch.save({"mean": mean.cpu(), "std": std.cpu(), "grads": [g.cpu() for g in grads]}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rows", [(2, 2), (4, 4), (4, 2)])
@pytest.mark.parametrize("dtype_name,precision,vtol,gtol", [("float64", "fp32", 1e-10, 1e-8), ("float32", "f16x3", 2e-5, 5e-4)])
def test_row_sharded_processes_reproduce_the_single_process_estimate(tmp_path, world, rows, dtype_name, precision, vtol, gtol):
    import torch.multiprocessing as mp

    from matfree_extensions.distributed import slq_value_and_grad

    dtype = getattr(torch, dtype_name)
    n, d, k, p = _MP_SHAPE
    op, params = _rbf(n, d, dtype, precision, ard=False)
    params = [q.clone().requires_grad_(True) for q in params]
    mean, std, grads = slq_value_and_grad(op, torch.log, k, params, n=n, seed=3, num_probes=p, dtype=dtype, device=_dev())
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "last.pt")
    mp.spawn(_worker, args=(world, port, rows, out, dtype_name, precision), nprocs=world, join=True)
    got = torch.load(out)
    assert np.isclose(got["mean"].item(), mean.item(), rtol=vtol)
    assert np.isclose(got["std"].item(), std.item(), rtol=1e-4, atol=1e-6 * abs(mean.item()))
    for a, b in zip(got["grads"], grads):
        assert torch.allclose(a, b.cpu(), rtol=gtol, atol=gtol * b.abs().max().item()), (a, b)


def test_nested_native_operator_inside_a_callback_keeps_the_outer_workspace():
    """a CallbackOp whose Python matvec calls a native RbfGramOp re-enters the workspace cache while the outer
    driver keeps its iterate / partial sums / adjoint state at offset 0 of its buffer.  tridiag + adjoint through the callback
    must equal the native path."""
    from matfree_extensions import lanczos
    from matfree_extensions.operators import CallbackOp

    n, d, k, p = 700, 3, 9, 3
    op, params = _rbf(n, d, torch.float64, ard=False)
    g = torch.Generator().manual_seed(5)
    V = torch.randn((p, n), generator=g, dtype=torch.float64).to(_dev())

    def run(matvec):
        ps = [q.clone().requires_grad_(True) for q in params]
        (Q, (dg, off)), (q, b) = lanczos.tridiag(matvec, k, reortho="full")(V, *ps)
        loss = (Q * torch.linspace(0, 1, n, dtype=torch.float64, device=_dev())).sum() + dg.sum() + 2 * off.sum() + q.sum() + b.sum()
        return dg.detach(), off.detach(), torch.autograd.grad(loss, ps)

    d0, e0, g0 = run(op)
    d1, e1, g1 = run(CallbackOp(lambda v, *q: op(v, *q)))
    assert torch.allclose(d0, d1, rtol=1e-11) and torch.allclose(e0, e1, rtol=1e-9, atol=1e-12)
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b, rtol=1e-8, atol=1e-10 * a.abs().max().item()), (a, b)


# ---- BASELINE config 5 layout: the non-symmetric wave operator, rows sharded ------------------------------------------------
_WAVE = (24, 14, 0.05)  # grid 24 x 24 -> state 1152 = 2 x 576 rows; Arnoldi depth; time step


def _wave_problem(dtype=torch.float64):
    from matfree_extensions.util import pde_util

    res = _WAVE[0]
    g = torch.Generator().manual_seed(9)
    scale = (0.5 + torch.rand((res, res), generator=g, dtype=torch.float64)).to(_dev())
    y0 = torch.randn((2 * res * res,), generator=g, dtype=torch.float64).to(_dev())
    w = torch.randn((2 * res * res,), generator=g, dtype=torch.float64).to(_dev())
    op, values_fn = pde_util.wave_operator(res, 1.0 / (res - 1), boundary="neumann", device=_dev(), dtype=dtype)
    return op, values_fn, scale, y0, w


def _wave_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "experiments-lanczos-adjoints_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import datetime

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from matfree_extensions.distributed import RowComm
    from matfree_extensions.operators import RowShardedOp
    from matfree_extensions.util import pde_util

    op, values_fn, scale, y0, w = _wave_problem()
    comm = RowComm(op.n)
    sc = scale.clone().requires_grad_(True)
    y1, _ = pde_util.expm_arnoldi(_WAVE[1])(RowShardedOp(op, comm), _WAVE[2], comm.rows(y0), values_fn(sc))
    (y1 * comm.rows(w)).sum().backward()  # this rank's part of the loss sum_i w_i y1_i; the scale gradient comes out complete
    full = comm.gather_rows(y1.detach())
    torch.cuda.synchronize()
    if rank == world - 1:
        torch.save({"y1": full.cpu(), "dscale": sc.grad.cpu()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_wave_expm_and_its_gradient_equal_the_single_process_result(tmp_path):
    """C5's layout on 2 processes: expm_arnoldi on the row-sharded non-symmetric CSR operator (A and A^T row blocks, SDDMM
    gradient of the rows each rank owns) and the gradient w.r.t. the coefficient field."""
    import torch.multiprocessing as mp

    from matfree_extensions.util import pde_util

    op, values_fn, scale, y0, w = _wave_problem()
    sc = scale.clone().requires_grad_(True)
    y1, _ = pde_util.expm_arnoldi(_WAVE[1])(op, _WAVE[2], y0, values_fn(sc))
    (y1 * w).sum().backward()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "wave.pt")
    mp.spawn(_wave_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    assert torch.allclose(got["y1"], y1.detach().cpu(), rtol=1e-10, atol=1e-12)
    assert torch.allclose(got["dscale"], sc.grad.cpu(), rtol=1e-8, atol=1e-10 * sc.grad.abs().max().item())


# ---- the C4 shape on two row shards (both ranks on this one GPU) -----------------------------------------------------------------
def _c4_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "experiments-lanczos-adjoints_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import datetime

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    from matfree_extensions.distributed import slq_value_and_grad

    op, params, n, k, p = _c4_problem()
    mean, std, grads = slq_value_and_grad(op, torch.log, k, params, n=n, seed=5, num_probes=p, row_group_size=world,
                                          dtype=torch.float32, device=_dev())
    torch.cuda.synchronize()
    if rank == world - 1:
        torch.save({"mean": mean.cpu(), "grads": [g.cpu() for g in grads]}, out)
    dist.barrier()
    dist.destroy_process_group()


def _c4_problem():
    from matfree_extensions.util import gp_util

    n, d, k, p = 131072, 8, 10, 8
    gen = torch.Generator().manual_seed(4)
    X = torch.randn((n, d), generator=gen, dtype=torch.float32).to(_dev())
    inv = lambda v: float(np.log(np.expm1(v)))  # noqa: E731
    params = [torch.tensor(inv(v), dtype=torch.float32, device=_dev(), requires_grad=True) for v in (2.0, 1.0, 0.1)]
    return gp_util.gram_operator(X, precision="f16x3"), params, n, k, p


def test_c4_shape_on_two_row_shards_equals_the_single_process_estimate(tmp_path):
    """n = 131072, d = 8 (config 4's operator; k = 10, 8 probes to keep it short): the matrix-core Gram kernels on a 65536-row
    block with its own column splits, the 256 x 256 gradient tile on a row block, 32-bit staging offsets at the full column count."""
    import torch.multiprocessing as mp

    from matfree_extensions.distributed import slq_value_and_grad

    op, params, n, k, p = _c4_problem()
    mean, _std, grads = slq_value_and_grad(op, torch.log, k, params, n=n, seed=5, num_probes=p, dtype=torch.float32, device=_dev())
    del op
    torch.cuda.empty_cache()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "c4.pt")
    mp.spawn(_c4_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    assert np.isclose(got["mean"].item(), mean.item(), rtol=2e-5)
    for a, b in zip(got["grads"], grads):
        assert torch.allclose(a, b.cpu(), rtol=5e-4, atol=5e-4 * b.abs().max().item()), (a, b)


# ---- the three-term recurrence (tridiag(reortho="none"), lanczos.py:231-335) on row shards ---------------------------------
_LZ = (1900, 7, 3)  # n (3 ranks own 640, 640, 620 rows), depth, start vectors


def _lz_problem(kind):
    g = torch.Generator().manual_seed(21)
    n, k, p = _LZ
    if kind == "rbf":
        op, params = _rbf(n, 5, torch.float64, "fp32", ard=True, seed=4)
    else:  # a banded SPD matrix as CSR: the neighbour-exchange path of a row-sharded sparse operator
        from matfree_extensions.operators import CsrOp

        rows, cols, vals = [], [], []
        for off, val in ((0, 4.0), (1, -1.0), (-1, -1.0), (17, -0.5), (-17, -0.5)):
            i = torch.arange(max(0, -off), min(n, n - off))
            rows.append(i), cols.append(i + off), vals.append(torch.full((len(i),), val, dtype=torch.float64))
        rows, cols, vals = torch.cat(rows), torch.cat(cols), torch.cat(vals)
        vals = vals * (1.0 + 0.1 * torch.rand(len(vals), generator=g, dtype=torch.float64))  # non-symmetric on purpose (Q4: A, not A^T)
        op, stored, _ = CsrOp.from_coo(rows, cols, vals, n, _dev())
        params = [stored]
    V = torch.randn((p, n), generator=g, dtype=torch.float64).to(_dev())
    wa = torch.randn((p, k), generator=g, dtype=torch.float64).to(_dev())
    wb = torch.randn((p, k - 1), generator=g, dtype=torch.float64).to(_dev())
    wx = torch.randn((p, k, n), generator=g, dtype=torch.float64).to(_dev())
    wl = torch.randn((p, n), generator=g, dtype=torch.float64).to(_dev())
    return op, params, V, (wa, wb, wx, wl)


def _lz_loss(out, weights, rows=slice(None)):
    (xs, (alpha, beta)), (xlast, blast) = out
    wa, wb, wx, wl = weights
    local = (xs * wx[:, :, rows]).sum() + (xlast * wl[:, rows]).sum()
    return local, (alpha * wa).sum() + (beta * wb).sum() + (blast ** 2).sum()


def _lz_worker(rank, world, port, out, kind):
    for p in (ROOT, os.path.join(ROOT, "experiments-lanczos-adjoints_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import datetime

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from matfree_extensions import lanczos
    from matfree_extensions.distributed import RowComm
    from matfree_extensions.operators import RowShardedOp

    op, params, V, weights = _lz_problem(kind)
    comm = RowComm(_LZ[0])
    sl = slice(comm.row0, comm.row0 + comm.nrows)
    ps = [q.clone().requires_grad_(True) for q in params]
    v = V[:, sl].clone().requires_grad_(True)
    res = lanczos.tridiag(RowShardedOp(op, comm), _LZ[1], reortho="none")(v, *ps)
    local, replicated = _lz_loss(res, weights, sl)
    # this rank's part of the row-wise loss + the loss on the replicated coefficients, which every rank evaluates in full (the
    # convention of the sharded drivers: cotangents of replicated outputs are complete on every rank)
    (local + replicated).backward()
    (xs, (alpha, beta)), _ = res
    torch.cuda.synchronize()
    got = {"alpha": alpha.detach().cpu(), "beta": beta.detach().cpu(), "xs": comm.gather_rows(xs.detach().reshape(-1, comm.nrows)).cpu(),
           "dv": comm.gather_rows(v.grad).cpu(), "grads": [q.grad.cpu() for q in ps]}
    if rank == world - 1:
        torch.save(got, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["rbf", "csr"])
def test_row_sharded_three_term_recurrence_and_adjoint(tmp_path, kind):
    """tridiag(reortho="none") on 3 processes (ragged last shard; CSR: neighbour exchange) against the single-process drivers:
    alpha, beta, the basis, dv and the parameter gradients, fp64."""
    import torch.multiprocessing as mp

    from matfree_extensions import lanczos

    op, params, V, weights = _lz_problem(kind)
    ps = [q.clone().requires_grad_(True) for q in params]
    v = V.clone().requires_grad_(True)
    res = lanczos.tridiag(op, _LZ[1], reortho="none")(v, *ps)
    local, replicated = _lz_loss(res, weights)
    (local + replicated).backward()
    (xs, (alpha, beta)), _ = res
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "lz.pt")
    mp.spawn(_lz_worker, args=(3, port, out, kind), nprocs=3, join=True)
    got = torch.load(out)
    n, k, p = _LZ
    assert torch.allclose(got["alpha"], alpha.detach().cpu(), rtol=1e-10, atol=1e-12)
    assert torch.allclose(got["beta"], beta.detach().cpu(), rtol=1e-10, atol=1e-12)
    assert torch.allclose(got["xs"].reshape(p, k, n), xs.detach().cpu(), rtol=1e-9, atol=1e-11)
    assert torch.allclose(got["dv"], v.grad.cpu(), rtol=1e-8, atol=1e-10 * v.grad.abs().max().item())
    for a, b in zip(got["grads"], ps):
        assert torch.allclose(a, b.grad.cpu(), rtol=1e-8, atol=1e-10 * b.grad.abs().max().item()), (a, b.grad)


@pytest.mark.parametrize("k", [1, 2, 9])
def test_three_term_and_pcg_with_a_one_rank_communicator_equal_the_single_device_drivers(k):
    """The sharded drivers with world = 1 (all-gather = copy, all-reduce = identity): depth 1 and 2 (the three dots of an adjoint step need
    more staging rows than the depth has), adaptive PCG with the preconditioner."""
    from matfree_extensions import cg, lanczos, low_rank
    from matfree_extensions.distributed import RowComm
    from matfree_extensions.operators import RowShardedOp

    n, d, p = 1216, 6, 4
    op, params = _rbf(n, d, torch.float64, "fp32", ard=True, seed=9)
    g = torch.Generator().manual_seed(2)
    V = torch.randn((p, n), generator=g, dtype=torch.float64).to(_dev())
    sop = RowShardedOp(op, RowComm(n))

    def run(mv):
        ps = [q.clone().requires_grad_(True) for q in params]
        v = V.clone().requires_grad_(True)
        (xs, (al, be)), (xl, bl) = lanczos.tridiag(mv, k, reortho="none")(v, *ps)
        ((xs ** 2).sum() * 0.5 + (al * be.sum(-1, keepdim=True)).sum() + (xl[:, ::7]).sum() + (bl ** 2).sum()).backward()
        return [al.detach(), be.detach(), xs.detach(), v.grad] + [q.grad for q in ps]

    for a, b in zip(run(op), run(sop)):
        assert a.shape == b.shape
        if a.numel():  # (depth 1 has no off-diagonal)
            assert torch.allclose(a, b, rtol=1e-10, atol=1e-12 * max(1.0, a.abs().max().item()))
    with torch.no_grad():
        pre, _ = low_rank.preconditioner(low_rank.cholesky_partial_pivot(rank=8))(low_rank.without_noise(op.bind(*params)), n)
        noise = op.constrain(*params)[2]
        solve = cg.pcg_adaptive(atol=1e-9, rtol=0.0, maxiter=400, miniter=1)
        x0, i0 = solve(op.bind(*params), V, pre.bind(noise))
        x1, i1 = solve(sop.bind(*params), V, pre.bind(noise))
    assert torch.equal(i0["num_steps"], i1["num_steps"]) and torch.allclose(x0, x1, rtol=1e-9, atol=1e-11)


# ---- (preconditioned) conjugate gradients on row shards (cg.py:19-137) ---------------------------------------------------
_CG = (1900, 5, 3, 12)  # n, d, right-hand sides, preconditioner rank


def _cg_problem():
    n, d, p, rank = _CG
    g = torch.Generator().manual_seed(31)
    op, params = _rbf(n, d, torch.float64, "fp32", ard=True, seed=6)
    params[2] = torch.tensor(0.5, dtype=torch.float64, device=_dev())  # a noise level CG converges on in tens of steps
    B = torch.randn((p, n), generator=g, dtype=torch.float64).to(_dev())
    W = torch.randn((p, n), generator=g, dtype=torch.float64).to(_dev())
    return op, params, B, W


def _cg_solve(matvec_op, params, B, W, rows=slice(None)):
    """-> {fixed-step PCG solution + its gradients, adaptive CG solution + step counts}; B, W: the rows this caller holds."""
    from matfree_extensions import cg, low_rank

    n, _, _, rank = _CG
    ps = [q.clone().requires_grad_(True) for q in params]
    b = B.clone().requires_grad_(True)
    native = getattr(matvec_op, "op", matvec_op)
    with torch.no_grad():  # the preconditioner: replicated, from the whole operator (every rank builds the same one)
        pre, _ = low_rank.preconditioner(low_rank.cholesky_partial_pivot(rank=rank))(low_rank.without_noise(native.bind(*params)), n)
        noise = native.constrain(*params)[2]
    x, info = cg.pcg_fixed_step(8)(matvec_op.bind(*ps), b, pre.bind(noise))
    (x * W).sum().backward()
    with torch.no_grad():
        xa, ainfo = cg.cg_adaptive(atol=1e-8, rtol=0.0, maxiter=300, miniter=2)(matvec_op.bind(*params), B)
    return {"x": x.detach(), "db": b.grad, "grads": [q.grad for q in ps], "xa": xa, "steps": ainfo["num_steps"],
            "res": info["residual_abs"].detach()}


def _cg_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "experiments-lanczos-adjoints_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import datetime

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from matfree_extensions.distributed import RowComm
    from matfree_extensions.operators import RowShardedOp

    op, params, B, W = _cg_problem()
    comm = RowComm(_CG[0])
    got = _cg_solve(RowShardedOp(op, comm), params, comm.rows(B), comm.rows(W))
    torch.cuda.synchronize()
    full = {k: comm.gather_rows(got[k]).cpu() for k in ("x", "db", "xa", "res")}
    full["grads"] = [g.cpu() for g in got["grads"]]
    full["steps"] = got["steps"].cpu()
    if rank == world - 1:
        torch.save(full, out)
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_pcg_and_its_gradient_rule(tmp_path):
    """pcg_fixed_step with the Woodbury preconditioner (solution, residual, d/db, parameter gradients) and cg_adaptive (solution,
    step counts) on 3 processes against the single-process solver, fp64."""
    import torch.multiprocessing as mp

    op, params, B, W = _cg_problem()
    want = _cg_solve(op, params, B, W)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "cg.pt")
    mp.spawn(_cg_worker, args=(3, port, out), nprocs=3, join=True)
    got = torch.load(out)
    for key, tol in (("x", 1e-9), ("res", 1e-7), ("db", 1e-8), ("xa", 1e-7)):
        w = want[key].cpu()
        assert torch.allclose(got[key], w, rtol=tol, atol=tol * w.abs().max().item()), key
    assert torch.equal(got["steps"], want["steps"].cpu())
    for a, b in zip(got["grads"], want["grads"]):
        assert torch.allclose(a, b.cpu(), rtol=1e-7, atol=1e-9 * b.abs().max().item()), (a, b)


def test_rccl_one_rank_group_runs_the_collective_callbacks():
    """RCCL needs one GPU per rank, so the multi-rank runs of this file go through gloo and the host.  What a one-GPU box CAN
    check of the RCCL path: a one-rank "nccl" group with the collectives forced through it -- all_reduce / all_gather_into_tensor on
    views of the libmfx workspace, issued from the C callbacks, ordered with the kernels on the current stream."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k_, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_one_rank.py"), str(port)], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]


# ---- the 8-way row shard of BASELINE config 4, rehearsed on ONE GPU in ONE process ---------------------------------------------
@pytest.mark.parametrize("dtype_name,precision,vtol,gtol", [("float64", "fp32", 1e-10, 1e-8), ("float32", "f16x3", 2e-5, 5e-4)])
@pytest.mark.parametrize("n", [8 * 2048, 8 * 2048 - 700])
def test_eight_logical_ranks_in_one_process_reproduce_the_single_rank_estimate(dtype_name, precision, vtol, gtol, n):
    """`bench.py --gpus 8` shards the rows of config 4 eight ways.  A GPU box admits at most six processes on its card, so the
    eight ranks are eight THREADS here (`tests/_local_world.LocalWorld`: host-rendezvous collectives, everything else -- the sharded
    drivers, their workspaces and callbacks, the row-block matvec with its column splits, the row-block gradient sweep, the fused
    reduction of the estimate -- is the code the eight processes run).  Even shards (8 x 2048 rows) and a ragged last shard."""
    from _local_world import LocalWorld
    from matfree_extensions.distributed import slq_value_and_grad

    dtype = getattr(torch, dtype_name)
    d, k, p = 8, 10, 16
    op, params = _rbf(n, d, dtype, precision, ard=False)
    ps = [q.clone().requires_grad_(True) for q in params]
    mean, std, grads = slq_value_and_grad(op, torch.log, k, ps, n=n, seed=3, num_probes=p, dtype=dtype, device=_dev())

    def rank_body(handle):
        mine = [q.detach().clone().requires_grad_(True) for q in params]
        m, s, g = slq_value_and_grad(op, torch.log, k, mine, n=n, seed=3, num_probes=p, row_group_size=handle.world, group=handle,
                                     dtype=dtype, device=_dev())
        return m.item(), s.item(), [t.detach().cpu() for t in g]

    results = LocalWorld(8).run(rank_body)
    for m, s, g in results:  # identical on every rank, equal to the single-rank estimate
        assert m == results[0][0] and s == results[0][1]
        assert np.isclose(m, mean.item(), rtol=vtol)
        assert np.isclose(s, std.item(), rtol=1e-4, atol=1e-6 * abs(mean.item()))
        for a, b in zip(g, grads):
            assert torch.allclose(a, b.cpu(), rtol=gtol, atol=gtol * b.abs().max().item()), (a, b)
