"""ctypes binding of libmfx.so (the C-ABI declared in include/mfx.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every hot-path operation goes
through the hand-written HIP kernels behind ``mfx_*``.  There is deliberately NO CPU or eager
fallback: if the library is missing, or a tensor is not on a ROCm device, calls raise.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MFX_LIBRARY_PATH") or os.path.join(_HERE, "libmfx.so")  # override: A/B builds

MFX_F32, MFX_F64 = 0, 1
OP_DENSE, OP_CSR, OP_RBF, OP_CALLBACK = 0, 1, 2, 3
REORTHO_NONE, REORTHO_FULL = 0, 1
RBF_FP32, RBF_F16X3_MATVEC, RBF_F16X3 = 0, 1, 2
KERNEL_RBF, KERNEL_MATERN12, KERNEL_MATERN32 = 0, 1, 2

CALLBACK_T = C.CFUNCTYPE(
    C.c_int,  # return
    C.c_void_p,  # ctx
    C.c_int,  # mode
    C.c_void_p,  # x
    C.c_int64,  # ldx
    C.c_void_p,  # aux
    C.c_int64,  # ldaux
    C.c_void_p,  # y
    C.c_int64,  # ldy
    C.c_int64,  # p
    C.c_int64,  # n
    C.c_void_p,  # stream
)


class Operator(C.Structure):
    """mirror of ``struct mfx_operator`` (include/mfx.h)."""

    _fields_ = [
        ("kind", C.c_int32),
        ("dtype", C.c_int32),
        ("n", C.c_int64),
        ("dense_a", C.c_void_p),
        ("lda", C.c_int64),
        ("crow", C.c_void_p),
        ("col", C.c_void_p),
        ("row", C.c_void_p),
        ("val", C.c_void_p),
        ("nnz", C.c_int64),
        ("max_row_nnz", C.c_int64),
        ("t_crow", C.c_void_p),
        ("t_col", C.c_void_p),
        ("t_perm", C.c_void_p),
        ("x", C.c_void_p),
        ("d", C.c_int32),
        ("ard", C.c_int32),
        ("rbf_mode", C.c_int32),
        ("kernel_fn", C.c_int32),
        ("lengthscale", C.c_void_p),
        ("outputscale", C.c_void_p),
        ("noise", C.c_void_p),
        ("callback", CALLBACK_T),
        ("ctx", C.c_void_p),
        ("row0", C.c_int64),
        ("nrows", C.c_int64),
    ]


ALLREDUCE_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)  # ctx, buf, count, dtype, stream
ALLGATHER_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)  # ctx, in, out, count, dtype, stream
# ctx, local, ldlocal, full, ldfull, p, dtype, transpose, stream
EXCHANGE_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p)


# ctx, local, ldlocal, full, ldfull, p, count, dtype, stream
ALLGATHER_ROWS_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p)


class Comm(C.Structure):
    """mirror of ``struct mfx_comm`` (row-sharded drivers)."""

    _fields_ = [
        ("rank", C.c_int32),
        ("world", C.c_int32),
        ("nloc", C.c_int64),
        ("allreduce_sum", ALLREDUCE_T),
        ("allgather", ALLGATHER_T),
        ("ctx", C.c_void_p),
        ("exchange", EXCHANGE_T),
        ("allgather_rows", ALLGATHER_ROWS_T),
    ]


class OpGrads(C.Structure):
    """mirror of ``struct mfx_op_grads``."""

    _fields_ = [
        ("dense_a", C.c_void_p),
        ("val", C.c_void_p),
        ("lengthscale", C.c_void_p),
        ("outputscale", C.c_void_p),
        ("noise", C.c_void_p),
    ]


# every symbol include/mfx.h declares: (name, restype, argtypes)
_P, _I64, _I = C.c_void_p, C.c_int64, C.c_int
_OPP, _GRP, _CMP = C.POINTER(Operator), C.POINTER(OpGrads), C.POINTER(Comm)
SYMBOLS = {
    "mfx_last_error": (C.c_char_p, []),
    "mfx_version": (_I, []),
    "mfx_workspace_bytes": (_I64, [_OPP, _I64, _I64, _I64]),
    "mfx_op_apply": (_I, [_OPP, _P, _I64, _P, _I64, _I64, _I, _P, _I64, _P]),
    "mfx_op_vjp_params": (_I, [_OPP, _P, _I64, _P, _I64, _I64, _GRP, _P, _I64, _P]),
    "mfx_arnoldi_forward": (_I, [_OPP, _P, _I64, _I64, _I64, _I, _P, _P, _P, _P, _P, _I64, _P]),
    "mfx_complex_workspace_bytes": (_I64, [_OPP, _I64, _I64, _I64]),
    "mfx_arnoldi_forward_complex": (_I, [_OPP, _P, _I64, _I64, _I64, _I, _P, _P, _P, _P, _P, _I64, _P]),
    "mfx_arnoldi_adjoint": (
        _I,
        [_OPP, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _GRP, _P, _I64, _P],
    ),
    "mfx_lanczos_forward_sharded": (_I, [_OPP, _CMP, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _I64, _P]),
    "mfx_lanczos_adjoint_sharded": (_I, [_OPP, _CMP, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _GRP, _P, _I64, _P]),
    "mfx_rccl_available": (_I, []),
    "mfx_rccl_unique_id": (_I, [_P, _I64]),
    "mfx_comm_create_rccl": (_I, [_P, _I64, C.c_int32, C.c_int32, _I64, _CMP]),
    "mfx_comm_destroy_rccl": (_I, [_CMP]),
    "mfx_comm_rccl_gather_mode": (_I, [_CMP, _I]),
    "mfx_comm_rccl_count": (_I, [_CMP, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mfx_sharded_workspace_bytes": (_I64, [_OPP, _CMP, _I64, _I64, _I64]),
    "mfx_arnoldi_forward_sharded": (_I, [_OPP, _CMP, _P, _I64, _I64, _I64, _I, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "mfx_arnoldi_adjoint_sharded": (
        _I,
        [_OPP, _CMP, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _GRP, _P, _I64, _P],
    ),
    "mfx_lanczos_forward": (_I, [_OPP, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _I64, _P]),
    "mfx_lanczos_adjoint": (
        _I,
        [_OPP, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _GRP, _P, _I64, _P],
    ),
    "mfx_tridiag_eigh": (_I, [_P, _P, _I64, _I64, _I64, _I, _P, _P, _P]),
    "mfx_slq_quadform_bwd": (_I, [_P, _P, _P, _P, _P, _I64, _I64, _I, _P, _P, _I64, _P]),
    "mfx_rademacher": (_I, [C.c_uint64, _I64, _I64, _I64, _I, _P, _P]),
    "mfx_pcg_workspace_bytes": (_I64, [_OPP, _I64, _I64, _I64]),
    "mfx_pcg_sharded_workspace_bytes": (_I64, [_OPP, _CMP, _I64, _I64, _I64]),
    "mfx_pcg_solve_sharded": (
        _I,
        [_OPP, _CMP, _P, _I64, _I64, _I64, _P, _I64, _P, _P, _I64, _I64, C.c_double, C.c_double, _I, _P, _P, _P, _P, _I64, _P],
    ),
    "mfx_pcg_solve": (
        _I,
        [_OPP, _P, _I64, _I64, _I64, _P, _I64, _P, _P, _I64, _I64, C.c_double, C.c_double, _I, _P, _P, _P, _P, _I64, _P],
    ),
    "mfx_pcg_solve_reortho": (_I, [_OPP, _P, _I64, _I64, _I64, _P, _I64, _P, _P, _I64, _P, _P, _P, _P, _I64, _P]),
    "mfx_precond_apply": (_I, [_I, _I64, _I64, _P, _P, _P, _P, _I64, _P, _I64, _I64, _P, _I64, _P]),
    "mfx_partial_cholesky": (_I, [_OPP, _I64, _I, _I, _P, _P, _P, _P, _I64, _P]),
    "mfx_gram_cross_workspace_bytes": (_I64, [_OPP, _I64]),
    "mfx_gram_cross_apply": (_I, [_OPP, _P, _I64, _P, _I64, _P, _I64, _I64, _P, _I64, _P]),
    "mfx_timing_enable": (_I, [_I]),
    "mfx_timing_reset": (_I, []),
    "mfx_timing_read": (_I, [_I, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mfx_graph_stats": (_I, [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
}

_lib = None
_lock = threading.Lock()


class MfxError(RuntimeError):
    pass


def get():
    """Load libmfx.so once; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise MfxError(
                    f"libmfx.so not found at {LIB_PATH}: build it with "
                    "`python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C experiments-lanczos-adjoints_amd/csrc`). There is no CPU fallback."
                )
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(lib, name)  # AttributeError if the library does not export it
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(rc: int):
    if rc != 0:
        msg = get().mfx_last_error().decode()
        if rc == -1:
            raise ValueError(msg)  # MFX_ERR_INVALID mirrors the reference's ValueError conventions
        raise MfxError(f"libmfx error {rc}: {msg}")


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return MFX_F32
    if dtype == torch.float64:
        return MFX_F64
    raise TypeError(f"libmfx supports float32/float64 (got {dtype}); complex Arnoldi is out of scope")


def require_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise MfxError(
                "matfree_extensions (MI355X build): tensors must live on a ROCm device "
                f"(got device={t.device}); there is no CPU fallback path."
            )


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_ws_cache = {}  # (device index, stream) -> list of scratch buffers
_ws_busy = set()  # id() of buffers a driver call in progress is using


def _take(need: int, device) -> torch.Tensor:
    """A scratch buffer of at least ``need`` bytes, cached per (device, stream) so repeated calls do not re-allocate.
    A buffer that a driver call in progress holds (``busy``) is never handed out again: a CallbackOp whose Python matvec
    calls a native operator re-enters here while the outer Krylov driver still keeps its iterate, partial sums and adjoint
    state at the start of ITS buffer."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    bufs = _ws_cache.setdefault(key, [])
    for i, buf in enumerate(bufs):
        if id(buf) in _ws_busy:
            continue
        if buf.numel() < need:
            buf = torch.empty(int(need) + 256, dtype=torch.uint8, device=device)
            bufs[i] = buf
        return buf
    buf = torch.empty(int(need) + 256, dtype=torch.uint8, device=device)
    bufs.append(buf)
    return buf


class busy:
    """``with _lib.busy(ws): lib.mfx_...(ws)`` -- marks the buffer as in use for the duration of a call that may call back
    into Python (callback operators, row-sharded collectives), AND makes the buffer's device the current one for the call: libmfx
    creates its graph-capture stream on the current device and replays a captured driver call only on a stream of that device
    (a single process with tensors on cuda:1 while cuda:0 is current would otherwise never get its launch-bound calls replayed)."""

    def __init__(self, buf):
        self.key = id(buf)
        self.guard = torch.cuda.device(buf.device) if buf.is_cuda else None

    def __enter__(self):
        _ws_busy.add(self.key)
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            self.guard.__exit__(*exc)
        _ws_busy.discard(self.key)
        return False


def workspace(desc: Operator, n: int, k: int, p: int, device) -> torch.Tensor:
    """Caller-owned scratch for the Krylov drivers / operator calls of an (n, k, p) problem."""
    need = int(get().mfx_workspace_bytes(C.byref(desc), n, k, p))
    if need < 0:
        raise MfxError("mfx_workspace_bytes failed")
    return _take(need, device)


def scratch(need: int, device) -> torch.Tensor:
    """Caller-owned scratch of at least ``need`` bytes from the same per-(device, stream) cache."""
    if need < 0:
        raise MfxError("workspace size query failed")
    return _take(int(need), device)


def workspace_pcg(desc: Operator, n: int, p: int, rank: int, device) -> torch.Tensor:
    return scratch(int(get().mfx_pcg_workspace_bytes(C.byref(desc), n, p, rank)), device)


def timing_enable(flag: bool):
    check(get().mfx_timing_enable(int(flag)))


def timing_reset():
    check(get().mfx_timing_reset())


def timing_read(cls: int):
    tot, cnt = C.c_double(0.0), C.c_int64(0)
    check(get().mfx_timing_read(cls, C.byref(tot), C.byref(cnt)))
    return tot.value, cnt.value


def graph_stats():
    """(driver calls captured into a hipGraph, driver calls replayed from one) since the library was loaded."""
    cap, rep = C.c_int64(0), C.c_int64(0)
    check(get().mfx_graph_stats(C.byref(cap), C.byref(rep)))
    return cap.value, rep.value
