"""Back-to-back launches of one kernel class for `seconds` (tools/power_sample.sh samples rocm-smi meanwhile):
matvec = Gram matvec at the C4 shape (n = 131072, d = 8, 64 vectors, f16x3); grad = the parameter-gradient GEMM, batch 2560."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "experiments-lanczos-adjoints_amd"))
import torch

from matfree_extensions import _lib
from matfree_extensions.util import gp_util

what, seconds = sys.argv[1], float(sys.argv[2])
dev = torch.device("cuda:0")
n, d, p, k = 131072, 8, 64, 40
X = torch.rand(n, d, device=dev) * 2 - 1
op = gp_util.gram_operator(X, noise_minval=1e-4, precision="f16x3")
params = [torch.zeros((), device=dev) for _ in range(3)]
cparams = op.constrain(*params)
desc = op.descriptor(cparams, torch.float32, n)
lib = _lib.get()
if what == "matvec":
    v = torch.randn(p, n, device=dev)
    y = torch.empty_like(v)
    ws = _lib.workspace(desc, n, 1, p, dev)
    call = lambda: _lib.check(lib.mfx_op_apply(C.byref(desc), _lib.ptr(v), n, _lib.ptr(y), n, p, 0, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
else:
    batch = p * k
    L = torch.randn(batch, n, device=dev)
    R = torch.randn(batch, n, device=dev)
    gstruct, grads = op.new_grads(*cparams)
    ws = _lib.workspace(desc, n, k, p, dev)
    call = lambda: _lib.check(lib.mfx_op_vjp_params(C.byref(desc), _lib.ptr(L), n, _lib.ptr(R), n, batch, C.byref(gstruct), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
call()
torch.cuda.synchronize()
t0 = time.perf_counter()
calls = 0
while time.perf_counter() - t0 < seconds:
    for _ in range(20 if what == "matvec" else 1):
        call()
        calls += 1
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{what}: {calls} launches in {dt:.2f} s = {dt / calls * 1e3:.3f} ms per launch")
