"""What ONE rank of the 8-way row-sharded C4 run computes, timed on one GPU (DESIGN.md section 5 budget).

    python tools/bench_row_block.py [--world 8] [--reps 10]

Config 4 (N = 131072, d = 8, k = 40, 64 probes, f16x3) sharded by rows over `world` ranks (the reference's own row partition of
the Gram matvec, util/gp_util.py:496-509): rank r owns rows [r nloc, (r + 1) nloc) of K and of every Krylov vector.  Measured:
  1. the Gram matvec of EACH of the `world` row blocks against all columns, 64 vectors (`mfx_op_apply` with row0 / nrows), next
     to the whole operator: does `world` x per-rank reach the full-operator time, or does the short block lose efficiency?
  2. the parameter-gradient GEMM on one row block at batch 2560 (`mfx_op_vjp_params` with row0 / nrows);
  3. one rank's whole value-and-gradient step through the row-sharded drivers (`mfx_arnoldi_*_sharded`), with a communicator
     that claims `world` ranks and whose collectives are local stand-ins (all-reduce: identity; all-gather: this rank's block
     into its slot -- the other slots keep stale data, so the NUMBERS of the step are meaningless, its kernel launches are the
     real ones): per kernel class from libmfx's own hipEvent timers (0 operator, 1 gradient sweep, 2 Krylov vector kernels,
     3 all-gather incl. pack / unpack -- here a device-to-device copy).
No collective runs and no second GPU is needed: this is the per-rank COMPUTE; the communication terms of the budget are modelled.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiments-lanczos-adjoints_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from matfree_extensions import _lib, hutchinson, lanczos  # noqa: E402
from matfree_extensions.distributed import RowComm  # noqa: E402
from matfree_extensions.operators import RowShardedOp  # noqa: E402
from matfree_extensions.util import gp_util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--n", type=int, default=131072)
ap.add_argument("--k", type=int, default=40)
ap.add_argument("--p", type=int, default=64)
ap.add_argument("--out", default="")
args = ap.parse_args()

dev = torch.device("cuda:0")
n, d, k, p, world = args.n, 8, args.k, args.p, args.world
gen = torch.Generator().manual_seed(4)
X = torch.randn((n, d), generator=gen, dtype=torch.float32).to(dev)
inv = lambda v: float(np.log(np.expm1(v)))  # noqa: E731
params = [torch.tensor(inv(v), dtype=torch.float32, device=dev, requires_grad=True) for v in (2.0, 1.0, 0.1)]
op = gp_util.gram_operator(X, precision="f16x3")
cparams = op.constrain(*[q.detach() for q in params])
lib = _lib.get()
nloc = n // world
assert nloc * world == n and nloc % 64 == 0
V = torch.randn((p, n), generator=torch.Generator().manual_seed(1)).to(dev)
res = {"n": n, "k": k, "p": p, "world": world, "nloc": nloc}


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


# ---- 1. Gram matvec: whole operator and each row block -------------------------------------------------------------------
def matvec(row0, nrows):
    desc = op.descriptor(cparams, V.dtype, n)
    desc.row0, desc.nrows = row0, nrows
    ws = _lib.workspace(desc, n, 1, p, dev)
    y = torch.empty((p, nrows or n), dtype=V.dtype, device=dev)

    def call():
        _lib.check(lib.mfx_op_apply(C.byref(desc), _lib.ptr(V), n, _lib.ptr(y), nrows or n, p, 0, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))

    return timed(call, args.reps)


full_ms = matvec(0, 0)
blocks = [matvec(r * nloc, nloc) for r in range(world)]
res["matvec_ms"] = {"whole_operator": full_ms, "row_blocks": blocks, "sum_of_blocks": sum(blocks),
                    "blocks_over_whole": sum(blocks) / full_ms}
print(f"Gram matvec, {p} vectors: whole operator {full_ms:.3f} ms; {world} row blocks of {nloc}: "
      + " ".join(f"{b:.3f}" for b in blocks) + f" ms; sum {sum(blocks):.3f} ms = {sum(blocks) / full_ms:.3f} x the whole operator", flush=True)

# ---- 2. gradient GEMM on one row block -------------------------------------------------------------------------------------
batch = p * k
g = torch.Generator(device=dev).manual_seed(0)
L = torch.randn(batch, nloc, device=dev, generator=g)
R = torch.randn(batch, n, device=dev, generator=g)


def grad_block(row0, nrows, Lb):
    desc = op.descriptor(cparams, torch.float32, n)
    desc.row0, desc.nrows = row0, nrows
    ws = _lib.scratch(int(lib.mfx_workspace_bytes(C.byref(desc), n, batch - 1, 1)), dev)

    def call():
        gs, grads = op.new_grads(*cparams)
        _lib.check(lib.mfx_op_vjp_params(C.byref(desc), _lib.ptr(Lb), nrows or n, _lib.ptr(R), n, batch, C.byref(gs), _lib.ptr(ws), ws.numel(),
                                         _lib.stream_ptr(dev)))

    return timed(call, max(2, args.reps // 3))


gb = grad_block(3 * nloc, nloc, L)
res["grad_gemm_ms"] = {"row_block": gb, "batch": batch}
print(f"gradient GEMM, batch {batch}, one row block of {nloc}: {gb:.2f} ms", flush=True)
del L, R
torch.cuda.empty_cache()


# ---- 3. one rank's whole step through the sharded drivers, collectives replaced by local stand-ins ----------------------------
class StandInComm(RowComm):
    """claims `world` ranks; all-reduce = identity, all-gather = own block into its slot (the rest of the buffer is stale)"""

    def __init__(self, n, world, rank):
        super().__init__(n)
        self.world, self.rank = world, rank
        self.nloc = n // world
        self.row0 = rank * self.nloc
        self.nrows = self.nloc

    def all_reduce_(self, t):
        return t

    def struct(self, ws, tensors=(), plans=None):
        base, nbytes = ws.data_ptr(), ws.numel()
        rank, world = self.rank, self.world

        def view(ptr, count, code):
            dt, es = (torch.float32, 4) if code == _lib.MFX_F32 else (torch.float64, 8)
            off = ptr - base
            assert 0 <= off and off + count * es <= nbytes
            return ws[off : off + count * es].view(dt)

        def allreduce(_ctx, buf, count, code, _stream):
            return 0

        def allgather(_ctx, inp, out, count, code, _stream):
            view(out, count * world, code)[rank * count : (rank + 1) * count].copy_(view(inp, count, code))
            return 0

        cb_r, cb_g, cb_x = _lib.ALLREDUCE_T(allreduce), _lib.ALLGATHER_T(allgather), _lib.EXCHANGE_T()
        cm = _lib.Comm()
        cm.rank, cm.world, cm.nloc = rank, world, self.nloc
        cm.allreduce_sum, cm.allgather, cm.exchange = cb_r, cb_g, cb_x
        return cm, (cb_r, cb_g, [], cb_x)


comm = StandInComm(n, world, 3)
integrand = lanczos.integrand_spd(torch.log, k, RowShardedOp(op, comm))
probes = comm.rows(hutchinson.sampler_rademacher(X[:, 0], num=p)(0))


def step():
    values = integrand(probes, *params)
    return torch.autograd.grad(values.sum(), params)


step()
torch.cuda.synchronize()
_lib.timing_reset()
_lib.timing_enable(True)
t0 = time.perf_counter()
reps = max(2, args.reps // 3)
for _ in range(reps):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps * 1e3
_lib.timing_enable(False)
cls = {name: _lib.timing_read(i) for i, name in enumerate(("gram_matvec", "param_grad_sweep", "krylov_vector_kernels", "allgather_stand_in"))}
res["one_rank_step_ms"] = {"wall_with_timers": wall, **{name: ms / reps for name, (ms, cnt) in cls.items()},
                           "launches_per_step": {name: cnt // reps for name, (ms, cnt) in cls.items()}}
print(f"one rank of {world} (rows {comm.row0}..{comm.row0 + comm.nrows}), whole value-and-gradient step, collectives replaced by local "
      f"stand-ins: {wall:.1f} ms wall (with per-launch timers): "
      + ", ".join(f"{name} {ms / reps:.2f} ms / {cnt // reps} launches" for name, (ms, cnt) in cls.items()), flush=True)
_lib.timing_reset()
t0 = time.perf_counter()
for _ in range(reps):
    step()
torch.cuda.synchronize()
res["one_rank_step_ms"]["wall_without_timers"] = (time.perf_counter() - t0) / reps * 1e3
print(f"the same without the per-launch timers: {res['one_rank_step_ms']['wall_without_timers']:.1f} ms wall", flush=True)
print(json.dumps(res))
if args.out:
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
