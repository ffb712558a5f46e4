"""The RBF parameter-gradient sweep alone (C4 shape: n = 131072, d = 8, batch = 2560): ms per call, for profiling."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "experiments-lanczos-adjoints_amd"))
import torch
from matfree_extensions import _lib
from matfree_extensions.operators import RbfGramOp

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2560
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(n, 8, device=dev, generator=g)
op = RbfGramOp(X, noise_minval=1e-4)
inv = lambda v: float(torch.log(torch.expm1(torch.tensor(v))))
params = [torch.tensor(inv(v), device=dev) for v in (2.0, 1.0, 0.1)]
cparams = op.constrain(*params)
desc = op.descriptor(cparams, torch.float32, n)
L = torch.randn(batch, n, device=dev, generator=g)
R = torch.randn(batch, n, device=dev, generator=g)
DECAY = float(os.environ.get("DECAY", "0"))  # > 0: row b scaled by 10^(-DECAY b / batch): the decay of the adjoint states over the Krylov steps
if DECAY > 0:      # (the public entry packs the rows in the order given, 32 consecutive rows = one stage)
    L.mul_(torch.pow(10.0, -DECAY * torch.arange(batch, device=dev) / batch)[:, None])
lib = _lib.get()
ws = _lib.scratch(int(lib.mfx_workspace_bytes(C.byref(desc), n, batch - 1, 1)), dev)


def call():
    gs, grads = op.new_grads(*cparams)
    _lib.check(lib.mfx_op_vjp_params(C.byref(desc), _lib.ptr(L), n, _lib.ptr(R), n, batch, C.byref(gs), _lib.ptr(ws),
                                     ws.numel(), _lib.stream_ptr(dev)))
    return grads


call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    grads = call()
torch.cuda.synchronize()
print(f"n={n} batch={batch}: {(time.perf_counter() - t0) / reps * 1e3:.1f} ms per sweep; grads", [float(t.flatten()[0]) for t in grads])
