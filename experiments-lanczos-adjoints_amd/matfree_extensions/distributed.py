"""Multi-GPU stochastic Lanczos quadrature: one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-device (no collective anywhere).  Two independent axes shard its SLQ estimate:

* **probes** -- the estimator is a mean over independent probes (hutchinson.py:14-15): every rank runs the full forward +
  adjoint for its own probes, the operator is replicated, and ONE fused all-reduce of [sum q, sum q^2, sum dq/dtheta]
  finishes the estimate (`shard_probes`, `reduce_estimate`).  Kernel evaluations are replicated per rank, so for a FIXED
  probe count this axis scales badly (64 probes on 8 GPUs: 2.1x).
* **rows** -- the rows of the kernel matrix are independent (the reference's own row partition of the Gram matvec,
  util/gp_util.py:496-509): rank r owns rows [r nloc, (r+1) nloc) of the operator AND of every Krylov vector, all probes on
  every rank.  Per Krylov step libmfx needs one all-gather of the (p, nloc) iterate and the sum-all-reduce of the (p, k+1)
  Gram-Schmidt coefficients / norms (`Q.T @ v`, arnoldi.py:87-92, becomes a partial sum per rank); it calls back into
  `RowComm` for both, on its stream (`mfx_comm`, include/mfx.h).  This is the strong-scaling layout of BASELINE config 4.

The axes compose: `make_grid(world, rows)` splits the world into row groups of `rows` ranks; different row groups take
different probes.
"""

from __future__ import annotations

import contextlib
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib


class TorchGroup:
    """A torch.distributed process group (None = the default group; no process group at all = one rank) behind the four calls the
    helpers of this module make on a group: ``world``, ``rank``, ``all_reduce_sum``, ``all_gather``.  Anything else that offers these
    four -- a transport of the caller's own -- can be passed as ``group=`` in its place (`as_group`); the helpers below have ONE path,
    through this interface.  (tests/_local_world.py plays eight ranks as threads of one process that way.)"""

    torch_backed = True

    def __init__(self, pg=None):
        self.pg = pg

    @property
    def _on(self):
        return dist.is_available() and dist.is_initialized()

    @property
    def world(self):
        return dist.get_world_size(self.pg) if self._on else 1

    @property
    def rank(self):
        return dist.get_rank(self.pg) if self._on else 0

    def all_reduce_sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return t

    def all_gather(self, out, inp):
        dist.all_gather_into_tensor(out, inp, group=self.pg)
        return out


def as_group(group):
    """``group=`` of this module's classes and functions -> an object with world / rank / all_reduce_sum / all_gather: a
    torch.distributed process group (or None) is wrapped, an object that already offers the four calls is taken as it is."""
    if all(hasattr(group, a) for a in ("world", "rank", "all_reduce_sum", "all_gather")):
        return group
    return TorchGroup(group)


def shard_probes(num_total: int, rank: int, world_size: int):
    """Contiguous split of ``num_total`` probes -> (first_probe, count) of this rank."""
    base, rem = divmod(num_total, world_size)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def rows_per_rank(n: int, world_size: int) -> int:
    """Rows per rank of the row-sharded layout: equal on every rank, a multiple of 64 (the matrix-core Gram kernels
    start their row blocks on multiples of 64); the last rank owns what is left."""
    nloc = -(-n // world_size)
    nloc = -(-nloc // 64) * 64
    if (world_size - 1) * nloc >= n:
        raise ValueError(f"n = {n} cannot be laid out on {world_size} ranks: equal 64-aligned shards are {nloc} rows, which leaves the last "
                         f"{world_size - -(-n // nloc)} rank(s) without rows -- {-(-n // nloc)} ranks have one")
    return nloc


def _active(group=None):
    return as_group(group).world > 1


def reduce_estimate(local_values, local_grads, num_total: int, group=None, replicas: int = 1):
    """All-reduce per-probe values and summed parameter gradients in ONE collective.

    local_values: (p_local,) integrand values of this rank's probes.
    local_grads : tuple of tensors = d/dtheta of SUM_b value_b over this rank's probes.
    replicas    : ranks of ``group`` holding the SAME probes (the row group size of a row-sharded run): their
                  contributions are identical, so the sums are divided by it.
    -> (mean, std over probes, tuple of gradients of the mean)
    """
    lv = local_values.double()  # mean^2 ~ 1e10 at n = 1e5: the second moment needs fp64
    flat = [lv.sum().reshape(1), (lv**2).sum().reshape(1)]
    flat += [g.reshape(-1).double() for g in local_grads]
    buf = torch.cat(flat).contiguous()
    if _active(group):
        as_group(group).all_reduce_sum(buf)
    buf = buf / replicas
    mean = buf[0] / num_total
    var = torch.clamp_min(buf[1] / num_total - mean**2, 0.0)
    grads, off = [], 2
    for g in local_grads:
        grads.append((buf[off : off + g.numel()] / num_total).reshape(g.shape).to(g.dtype))
        off += g.numel()
    return mean, var.sqrt(), tuple(grads)


def value_and_grad_sharded(integrand, sample_local, params, *, num_total: int, group=None, replicas: int = 1):
    """SLQ value-and-gradient with probes sharded over the ranks of ``group``.

    integrand   : batched integrand (e.g. lanczos.integrand_spd(...)), called once on this rank's probes
    sample_local: zero-argument callable returning this rank's (p_local, n) probes
    params      : tuple of parameter tensors (requires_grad leaves) passed to the integrand
    """
    probes = sample_local()
    values = integrand(probes, *params)
    diff = [q for q in params if torch.is_tensor(q) and q.requires_grad]
    grads = torch.autograd.grad(values.sum(), diff, allow_unused=True)
    grads = tuple(torch.zeros_like(q) if g is None else g for q, g in zip(diff, grads))
    return reduce_estimate(values.detach(), grads, num_total, group=group, replicas=replicas)


# ------------------------------------------------------------------------------------------------
# row sharding
# ------------------------------------------------------------------------------------------------
class RowComm:
    """The ranks of ``group`` as row shards of an n x n operator, and the collectives libmfx asks for.

    rank r owns rows [r nloc, min(n, (r + 1) nloc)).  ``struct(ws)`` builds the ``mfx_comm`` the sharded drivers take:
    its two callbacks receive raw device pointers into the workspace tensor ``ws`` of the call in progress, turn them
    back into views and run ``all_reduce`` / ``all_gather_into_tensor`` on them (stream-ordered on the current stream
    for "nccl"; blocking for "gloo").
    """

    def __init__(self, n: int, group=None):
        self.transport = as_group(group)
        self.group = getattr(self.transport, "pg", None)  # the torch.distributed process group, where there is one
        self.world = self.transport.world
        self.rank = self.transport.rank
        self.n = int(n)
        self.nloc = rows_per_rank(self.n, self.world)
        self.row0 = self.rank * self.nloc
        self.nrows = min(self.nloc, self.n - self.row0)
        # run the collectives even in a one-rank group (they are no-ops then): lets ONE GPU exercise the RCCL code path of the
        # callbacks -- tensor views of the workspace, stream ordering -- which gloo-through-the-host tests cannot
        self.force_collectives = False

    # ---- host-side helpers ---------------------------------------------------------------------
    def rows(self, t):
        """this rank's slice of a tensor whose LAST axis has length n"""
        return t[..., self.row0 : self.row0 + self.nrows].contiguous()

    def all_reduce_(self, t):
        if self.world > 1 or (self.force_collectives and dist.is_available() and dist.is_initialized()):
            self.transport.all_reduce_sum(t)
        return t

    def gather_rows(self, t_local):
        """(..., nrows) row shards -> (..., n) on every rank (tests / small host-side needs; not on the hot path)"""
        lead = t_local.shape[:-1]
        flat = t_local.reshape(-1, self.nrows)
        send = torch.zeros((flat.shape[0], self.nloc), dtype=t_local.dtype, device=t_local.device)
        send[:, : self.nrows] = flat
        out = torch.empty((self.world * flat.shape[0], self.nloc), dtype=t_local.dtype, device=t_local.device)
        if self.world > 1:
            self.transport.all_gather(out, send)  # rank-major blocks along axis 0
        else:
            out.copy_(send)
        out = out.reshape(self.world, flat.shape[0], self.nloc).movedim(0, 1).reshape(flat.shape[0], self.world * self.nloc)
        return out[:, : self.n].reshape(*lead, self.n).contiguous()

    # ---- neighbour exchange for sparse operators ------------------------------------------------------
    def plan_exchange(self, crow, col):
        """Which entries of the iterate do this rank's rows of a CSR matrix (crow, col: device int32) read from other ranks?
        -> (recv, send): lists of (peer, lo, hi) column ranges -- one bounding range per owner, which is tight for banded / block
        structures (stencils) -- to receive from / send to each peer.  Collective over the row group (one all_gather_object)."""
        if not getattr(self.transport, "torch_backed", False):
            raise NotImplementedError("the neighbour exchange is point-to-point over a torch.distributed process group; a transport of "
                                      "the caller's own offers the all-gather layout only")
        c = col[int(crow[self.row0]) : int(crow[self.row0 + self.nrows])].to(torch.int64)
        owner = torch.div(c, self.nloc, rounding_mode="floor")
        recv = []
        for q in range(self.world):
            if q == self.rank:
                continue
            cq = c[owner == q]
            if cq.numel():
                recv.append((q, int(cq.min()), int(cq.max()) + 1))
        everyone = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(everyone, recv, group=self.group)
        else:
            everyone[0] = recv
        send = [(q, lo, hi) for q in range(self.world) if q != self.rank for (peer, lo, hi) in everyone[q] if peer == self.rank]
        return recv, send

    def _exchange(self, local, full, plan):
        """local (p, nrows) view, full (p, n) view: receive the planned column ranges into ``full``, send ours."""
        recv, send = plan
        ranks = dist.get_process_group_ranks(self.group) if self.group is not None else list(range(self.world))
        staged = dist.get_backend(self.group) != "nccl"  # gloo moves device tensors through the host
        ops, landing = [], []
        for q, lo, hi in send:
            t = local[:, lo - self.row0 : hi - self.row0].contiguous()
            ops.append(dist.P2POp(dist.isend, t.cpu() if staged else t, ranks[q], group=self.group))
        for q, lo, hi in recv:
            buf = torch.empty((full.shape[0], hi - lo), dtype=full.dtype, device="cpu" if staged else full.device)
            ops.append(dist.P2POp(dist.irecv, buf, ranks[q], group=self.group))
            landing.append((lo, hi, buf))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for lo, hi, buf in landing:
            full[:, lo:hi] = buf.to(full.device, non_blocking=False)

    # ---- mfx_comm ---------------------------------------------------------------------------------
    def struct(self, ws: torch.Tensor, tensors=(), plans=None):
        """-> (mfx_comm, keepalive).  ``ws`` is the uint8 workspace tensor the collective callbacks' pointers lie in;
        ``tensors``: further tensors the exchange callback's pointers may lie in (basis, gathered basis, adjoint states);
        ``plans``: (forward, transpose) exchange plans of a sparse operator, or None for the all-gather."""
        base, nbytes = ws.data_ptr(), ws.numel()
        failure = []
        world, transport = self.world, self.transport
        collective = world > 1 or (self.force_collectives and dist.is_available() and dist.is_initialized())
        regs = [(ws.data_ptr(), ws.numel(), ws)] + [(t.data_ptr(), t.numel() * t.element_size(), t.view(-1).view(torch.uint8))
                                                   for t in tensors if t is not None]

        def on(stream):
            """the collectives are enqueued on the stream libmfx hands over (it is torch's current stream unless the caller passed
            another one to the driver)"""
            if ws.is_cuda and stream and stream != torch.cuda.current_stream(ws.device).cuda_stream:
                return torch.cuda.stream(torch.cuda.ExternalStream(stream, device=ws.device))
            return contextlib.nullcontext()

        def strided(ptr, ld, p, ncols, dtype_code):
            dt = torch.float32 if dtype_code == _lib.MFX_F32 else torch.float64
            es = 4 if dtype_code == _lib.MFX_F32 else 8
            for b0, nb, flat in regs:
                if b0 <= ptr < b0 + nb:
                    return flat[ptr - b0 :].view(dt).as_strided((p, ncols), (ld, 1))
            raise RuntimeError("libmfx exchange callback received an unknown pointer")

        def exchange(_ctx, local, ldlocal, full, ldfull, p, dtype_code, transpose, stream):
            try:
                with on(stream):
                    self._exchange(strided(local, ldlocal, p, self.nrows, dtype_code), strided(full, ldfull, p, self.n, dtype_code),
                                   plans[1 if transpose else 0])
                return 0
            except Exception as exc:
                failure.append(exc)
                return 1

        def view(ptr, count, dtype_code):
            dt = torch.float32 if dtype_code == _lib.MFX_F32 else torch.float64
            es = 4 if dtype_code == _lib.MFX_F32 else 8
            off = ptr - base
            if off < 0 or off + count * es > nbytes:
                raise RuntimeError("libmfx comm callback received a pointer outside the workspace")
            return ws[off : off + count * es].view(dt)

        def allreduce(_ctx, buf, count, dtype_code, stream):
            try:
                if collective:
                    with on(stream):
                        transport.all_reduce_sum(view(buf, count, dtype_code))
                return 0
            except Exception as exc:  # never let an exception cross the C boundary
                failure.append(exc)
                return 1

        def allgather(_ctx, inp, out, count, dtype_code, stream):
            try:
                tin, tout = view(inp, count, dtype_code), view(out, count * world, dtype_code)
                with on(stream):
                    if collective:
                        transport.all_gather(tout, tin)
                    else:
                        tout.copy_(tin)
                return 0
            except Exception as exc:
                failure.append(exc)
                return 1

        cb_r, cb_g = _lib.ALLREDUCE_T(allreduce), _lib.ALLGATHER_T(allgather)
        cb_x = _lib.EXCHANGE_T(exchange) if plans is not None else _lib.EXCHANGE_T()
        cm = _lib.Comm()
        cm.rank, cm.world, cm.nloc = self.rank, self.world, self.nloc
        cm.allreduce_sum, cm.allgather, cm.exchange = cb_r, cb_g, cb_x
        return cm, (cb_r, cb_g, failure, cb_x)


class NativeRowComm(RowComm):
    """RowComm whose ``mfx_comm`` is libmfx's own RCCL communicator (``mfx_comm_create_rccl``, csrc/mfx_comm.hip): the
    all-gather of the iterate and the small all-reduces of a Krylov step are issued by the library itself on the stream of
    the driver call -- no Python between two kernels.  Creation is collective over the row group: rank 0 makes the RCCL
    unique id, torch.distributed (any backend) carries it to the others, every rank calls ``ncclCommInitRank`` on its
    CURRENT device (``torch.cuda.set_device(local_rank)`` first).  The host-side helpers (``all_reduce_`` of the parameter
    gradients, ``gather_rows``) and a sparse operator's neighbour exchange stay on torch.distributed."""

    GATHER = {"grouped": 0, "packed": 1}

    def __init__(self, n: int, group=None, gather: str | None = None):
        """gather: "packed" (pack, ONE all-gather of p nloc elements, unpack; the default: one large message is the pattern RCCL's
        bandwidth figures are quoted for) or "grouped" (p all-gathers of nloc elements in one group launch, straight into the operator
        input: no copies, but p small operations) -- `include/mfx.h`, MFX_GATHER_*; default from $MFX_GATHER."""
        super().__init__(n, group)
        if not getattr(self.transport, "torch_backed", False):
            raise TypeError("the native RCCL communicator is created over a torch.distributed process group (its unique id travels through it)")
        lib = _lib.get()
        gather = gather or os.environ.get("MFX_GATHER", "packed")
        if gather not in self.GATHER:
            raise ValueError(f"gather mode {gather!r}: expected one of {sorted(self.GATHER)}")
        self.gather = gather
        uid = torch.zeros(128, dtype=torch.uint8)
        # rank 0 ALWAYS takes part in the broadcast, also when it could not make the id: (ok, message, id) travels, and every rank
        # raises -- or goes on -- after the same collective (a rank 0 that raised before the broadcast would leave its peers inside it)
        ok, why = True, ""
        if self.rank == 0:
            try:
                _lib.check(lib.mfx_rccl_unique_id(uid.data_ptr(), uid.numel()))
            except Exception as exc:  # noqa: BLE001
                ok, why = False, str(exc)
        if self.world > 1:
            box = [(ok, why, bytes(uid.tolist()))]
            pg = self.group  # (the torch.distributed process group behind `group`, also when a TorchGroup wrapper was passed)
            src = dist.get_process_group_ranks(pg)[0] if pg is not None else 0
            dist.broadcast_object_list(box, src=src, group=pg)
            ok, why, raw = box[0]
            uid = torch.tensor(list(raw), dtype=torch.uint8)
        if not ok:
            raise RuntimeError(f"the first rank of the row group could not make an RCCL unique id: {why}")
        self._cm = _lib.Comm()
        _lib.check(lib.mfx_comm_create_rccl(uid.data_ptr(), uid.numel(), self.rank, self.world, self.nloc, C.byref(self._cm)))
        _lib.check(lib.mfx_comm_rccl_gather_mode(C.byref(self._cm), self.GATHER[gather]))

    def struct(self, ws, tensors=(), plans=None):
        if plans is not None:  # neighbour exchange: callbacks (the exchange plan lives on the host side)
            return super().struct(ws, tensors, plans)
        return self._cm, (None, None, [], None)

    def self_test(self) -> bool:
        """One all-reduce and one grouped row all-gather through the native function pointers, checked on this rank."""
        dev = torch.device("cuda", torch.cuda.current_device())
        stream = _lib.stream_ptr(dev)
        ones = torch.ones(8, dtype=torch.float64, device=dev)
        rc = self._cm.allreduce_sum(self._cm.ctx, ones.data_ptr(), ones.numel(), _lib.MFX_F64, stream)
        local = torch.full((2, 4), float(self.rank + 1), dtype=torch.float32, device=dev)
        full = torch.zeros((2, 4 * self.world), dtype=torch.float32, device=dev)
        if self.gather == "grouped":
            rc |= self._cm.allgather_rows(self._cm.ctx, local.data_ptr(), 4, full.data_ptr(), 4 * self.world, 2, 4, _lib.MFX_F32, stream)
            want = torch.arange(1, self.world + 1, dtype=torch.float32, device=dev).repeat_interleave(4).expand(2, -1)
        else:  # the packed leg: one all-gather of the whole (2, 4) shard, rank-major blocks
            rc |= self._cm.allgather(self._cm.ctx, local.data_ptr(), full.data_ptr(), 8, _lib.MFX_F32, stream)
            want = torch.arange(1, self.world + 1, dtype=torch.float32, device=dev).repeat_interleave(8).reshape(2, -1)
        return rc == 0 and bool((ones == self.world).all()) and bool((full == want).all())

    def rccl_count(self):
        """(ranks, rank) as RCCL itself reports them for this communicator (``ncclCommCount`` / ``ncclCommUserRank``) -- libmfx's own
        view of the row group, as opposed to torch.distributed's world size."""
        ranks, rank = C.c_int32(-1), C.c_int32(-1)
        _lib.check(_lib.get().mfx_comm_rccl_count(C.byref(self._cm), C.byref(ranks), C.byref(rank)))
        return int(ranks.value), int(rank.value)

    def set_gather(self, gather: str):
        """switch the gather leg of this communicator ("packed" / "grouped"): same results, two message patterns to time"""
        if gather not in self.GATHER:
            raise ValueError(f"gather mode {gather!r}: expected one of {sorted(self.GATHER)}")
        _lib.check(_lib.get().mfx_comm_rccl_gather_mode(C.byref(self._cm), self.GATHER[gather]))
        self.gather = gather

    def close(self):
        if getattr(self, "_cm", None) is not None:
            _lib.get().mfx_comm_destroy_rccl(C.byref(self._cm))
            self._cm = None

    def __del__(self):  # the communicator must go before the process group does: call close() explicitly in long-lived programs
        try:
            self.close()
        except Exception:
            pass


def _all_ranks(ok: bool, group) -> bool:
    """True when ``ok`` holds on every rank of ``group`` (the ranks must take the same branch afterwards)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item())


def _native_or_callbacks(n: int, group):
    """The default on an RCCL process group: libmfx's native communicator, and if ANY rank of the group cannot bind librccl,
    fails to create the communicator or gets a wrong answer from its self-test, every rank falls back to the
    torch.distributed callbacks (the same RCCL underneath, the same results) and says so on stderr.  Each decision is taken
    by all ranks together, so the group never splits between the two paths."""
    import warnings

    why = None
    if not _all_ranks(bool(_lib.get().mfx_rccl_available()), group):
        why = "librccl could not be bound on every rank"
    else:
        comm, err = None, None
        try:
            comm = NativeRowComm(n, group)
        except Exception as e:  # noqa: BLE001 -- any failure of the collective creation takes the fallback
            err = e
        if not _all_ranks(comm is not None, group):
            why = f"communicator creation failed ({err})"
        else:
            try:
                ok = comm.self_test()
            except Exception as e:  # noqa: BLE001
                ok, err = False, e
            if _all_ranks(ok, group):
                return comm
            why = f"self-test of the native collectives failed ({err})"
        if comm is not None:
            comm.close()
    warnings.warn(f"matfree_extensions.distributed: native RCCL row-group collectives unavailable -- {why}; "
                  "using the torch.distributed callbacks", RuntimeWarning, stacklevel=3)
    return RowComm(n, group)


class _ShardedSumSq(torch.autograd.Function):
    """sum over ALL rows of v^2 from row shards (p, nrows) -> (p,), identical on every rank.  Backward: every rank holds
    the same downstream cotangent, and its rows enter the sum only through its own partial sum."""

    @staticmethod
    def forward(ctx, comm, V):
        ctx.save_for_backward(V)
        return comm.all_reduce_((V.double() ** 2).sum(-1)).to(V.dtype)

    @staticmethod
    def backward(ctx, g):
        (V,) = ctx.saved_tensors
        return None, 2.0 * V * g[:, None]


class _SumGrad(torch.autograd.Function):
    """Identity on a REPLICATED tensor (H, c of a row-sharded factorisation) whose consumers are row-sharded: each rank's
    backward then holds only the part of the cotangent that flows through ITS rows, and the drivers need the complete one."""

    @staticmethod
    def forward(ctx, comm, t):
        ctx.comm = comm
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return None, ctx.comm.all_reduce_(g.contiguous().clone())


def sum_grad(comm: RowComm, t):
    """Use a replicated output of a row-sharded driver inside a computation on row shards (e.g. Q_local @ f(H)): the
    cotangent of ``t`` is summed over the row group on the way back.  NOT needed when every rank evaluates the same function
    of ``t`` (the SLQ quadrature): there each rank already holds the complete cotangent."""
    return _SumGrad.apply(comm, t)


def sharded_norm(comm: RowComm, V):
    """Euclidean norm of row-sharded vectors (p, nrows) -> (p,)"""
    return torch.sqrt(_ShardedSumSq.apply(comm, V))


def make_grid(rows: int, group=None):
    """Split the ranks of ``group`` into row groups of ``rows`` consecutive ranks.

    -> (row_group, probe_index, num_probe_groups): ``row_group`` is the process group this rank shards rows over,
    ``probe_index`` says which slice of the probes its row group takes.  Every rank must call this (collective).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world % rows != 0:
        raise ValueError(f"row group size {rows} does not divide the world size {world}")
    ngroups = world // rows
    mine = None
    members = dist.get_process_group_ranks(group) if group is not None else list(range(world))
    for g in range(ngroups):
        pg = dist.new_group(ranks=[members[g * rows + r] for r in range(rows)])
        if rank // rows == g:
            mine = pg
    return mine, rank // rows, ngroups


class Layout:
    """A (row groups x probe groups) grid over the ranks of ``group``, built ONCE (process groups are collective to create).

    row_group_size = 1: pure probe sharding; = world size: pure row sharding (all probes on every rank).
    """

    def __init__(self, n: int, row_group_size: int = 1, group=None, native: bool | None = None):
        """native: the row group's collectives as libmfx's own RCCL calls (NativeRowComm); default: whenever the process
        group's backend is "nccl" (= RCCL), i.e. one GPU per rank -- gloo rehearsals keep the callback path."""
        on = dist.is_available() and dist.is_initialized()
        transport = as_group(group)
        self.group = group
        self.world = transport.world
        rank = transport.rank
        self.n = int(n)
        self.replicas = int(row_group_size)
        if self.replicas > 1:
            if not getattr(transport, "torch_backed", False):  # a transport of the caller's own: its ranks are ONE row group
                if self.world != self.replicas:
                    raise ValueError("a group that is not a torch.distributed process group cannot be split: row_group_size must equal its number of ranks")
                self.probe_index, self.probe_groups = 0, 1
                self.comm = RowComm(n, group)
                self.native = False
                return
            pg = transport.pg
            if self.world > self.replicas:
                row_group, self.probe_index, self.probe_groups = make_grid(self.replicas, pg)
            elif self.world == self.replicas:
                row_group, self.probe_index, self.probe_groups = pg, 0, 1
            else:
                raise ValueError(f"row group size {row_group_size} exceeds the world size {self.world}")
            if native is None:
                # MFX_NATIVE_COMM=0: operational escape hatch back to the torch.distributed callbacks (same results)
                native = on and dist.get_backend(row_group) == "nccl" and os.environ.get("MFX_NATIVE_COMM", "1") != "0"
                self.comm = _native_or_callbacks(n, row_group) if native else RowComm(n, row_group)
            else:  # asked for explicitly: no way back
                self.comm = NativeRowComm(n, row_group) if native else RowComm(n, row_group)
            self.native = isinstance(self.comm, NativeRowComm)
        else:
            self.comm, self.probe_index, self.probe_groups = None, rank, self.world
            self.native = False

    def describe(self):
        return f"{self.replicas} row shard(s) x {self.probe_groups} probe group(s)"


def slq_value_and_grad(op, matfun, krylov_depth, params, *, n: int, seed, num_probes: int, row_group_size: int = 1,
                       group=None, dtype=None, device=None, layout: Layout | None = None):
    """SLQ value-and-gradient on a (row groups x probe groups) grid of ranks.

    op             : native operator (e.g. gp_util.gram_operator(X)) -- replicated, X is a few MB
    row_group_size : ranks per row group (1 = pure probe sharding, world size = pure row sharding); or pass a ``Layout``
                     built once when the estimate is evaluated repeatedly
    -> (mean, std over probes, gradients of the mean), identical on every rank.
    """
    from . import hutchinson, lanczos
    from .operators import RowShardedOp

    lay = layout if layout is not None else Layout(n, row_group_size, group)
    first, count = shard_probes(num_probes, lay.probe_index, lay.probe_groups)
    like = torch.empty(n, dtype=dtype, device=device)
    probes = hutchinson.sampler_rademacher(like, num=count)((seed, first))  # this group's slice of ONE global probe matrix
    if lay.comm is not None:
        probes = lay.comm.rows(probes)
        matvec = RowShardedOp(op, lay.comm)
    else:
        matvec = op
    integrand = lanczos.integrand_spd(matfun, krylov_depth, matvec)
    return value_and_grad_sharded(integrand, lambda: probes, params, num_total=num_probes, group=lay.group,
                                  replicas=lay.replicas)
