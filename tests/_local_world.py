"""R logical ranks as R THREADS of one process sharing one GPU: the rehearsal stand-in for a process group on a box with fewer GPUs
than ranks (a GPU box admits few processes on its card; eight threads are one process).  Test / rehearsal infrastructure -- it lives
here, not in the shipped package: `matfree_extensions.distributed` takes any object with ``world``, ``rank``, ``all_reduce_sum`` and
``all_gather`` as ``group=`` (`distributed.as_group`), and a `LocalRank` is such an object.

``LocalWorld(8).ranks()`` gives the eight group handles; pass one as ``group=`` to `RowComm`, `Layout`, `reduce_estimate`,
`slq_value_and_grad` in the thread that plays that rank.  Collectives are host rendezvous (a barrier, device copies in rank order --
sums are deterministic): every kernel launch, every workspace layout and every callback of the row-sharded drivers is the real one;
only the transport is not RCCL.  Timings mean nothing.  ``run(fn)`` starts the threads, passes each its handle and returns the
results in rank order (first exception re-raised)."""

import contextlib
import threading

import torch


class LocalWorld:
    def __init__(self, world: int, timeout: float = 120.0):
        self.world = int(world)
        self._barrier = threading.Barrier(self.world, timeout=timeout)
        self._slots = [None] * self.world

    def ranks(self):
        return [LocalRank(self, r) for r in range(self.world)]

    def run(self, fn):
        out, err = [None] * self.world, [None] * self.world

        def body(handle):
            try:
                # a stream of its own per logical rank: the scratch cache of the host layer is keyed by (device, stream), so the ranks
                # get separate workspaces exactly as separate processes would
                own = torch.cuda.stream(torch.cuda.Stream()) if torch.cuda.is_available() else contextlib.nullcontext()
                # backward passes in THIS thread: the autograd engine otherwise runs every rank's backward nodes on the one worker
                # thread of the device, one after the other -- the first rank's adjoint driver then waits in its first collective
                # for ranks whose backward is queued behind it
                with own, torch.autograd.set_multithreading_enabled(False):
                    out[handle.rank] = fn(handle)
                    if torch.cuda.is_available():
                        torch.cuda.synchronize()
            except BaseException as exc:  # noqa: BLE001 -- a rank that dies must not leave the others in a rendezvous
                err[handle.rank] = exc
                self._barrier.abort()

        threads = [threading.Thread(target=body, args=(h,), name=f"mfx-rank-{h.rank}") for h in self.ranks()]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        first = next((e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
        if first is None:
            first = next((e for e in err if e is not None), None)
        if first is not None:
            raise first
        return out


class LocalRank:
    """One logical rank of a `LocalWorld`: the ``group=`` argument of `matfree_extensions.distributed` in the thread that plays it."""

    torch_backed = False

    def __init__(self, shared: LocalWorld, rank: int):
        self.shared, self.rank, self.world = shared, int(rank), shared.world

    def _publish(self, t):
        if t.is_cuda:
            torch.cuda.synchronize(t.device)
        self.shared._slots[self.rank] = t
        self.shared._barrier.wait()

    def all_reduce_sum(self, t):
        self._publish(t)
        acc = self.shared._slots[0].clone()
        for r in range(1, self.world):
            acc = acc + self.shared._slots[r]
        if t.is_cuda:
            torch.cuda.synchronize(t.device)
        self.shared._barrier.wait()  # everybody has read everybody's input
        t.copy_(acc)
        return t

    def all_gather(self, out, inp):
        self._publish(inp)
        blocks = out.view(self.world, -1)
        for r in range(self.world):
            blocks[r].copy_(self.shared._slots[r].reshape(-1))
        if out.is_cuda:
            torch.cuda.synchronize(out.device)
        self.shared._barrier.wait()
        return out

    def barrier(self):
        self.shared._barrier.wait()
