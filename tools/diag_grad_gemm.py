"""Isolate the parameter-gradient GEMM: exact fp32 vs 3xf16 split vs fp64, random operands (run on the GPU box)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "experiments-lanczos-adjoints_amd"))
from matfree_extensions import _lib
from matfree_extensions.operators import RbfGramOp

dev = torch.device("cuda:0")
n, d, batch = int(os.environ.get("N", 4096)), 8, int(os.environ.get("B", 512))
g = torch.Generator().manual_seed(0)
X = torch.randn(n, d, generator=g, dtype=torch.float64).to(dev)
L = torch.randn(batch, n, generator=g, dtype=torch.float64).to(dev)
R = torch.randn(batch, n, generator=g, dtype=torch.float64).to(dev)
if os.environ.get("DECAY"):
    L = L * torch.logspace(3, -3, batch, dtype=torch.float64, device=dev)[:, None]
raw = [torch.tensor(v, dtype=torch.float64, device=dev) for v in (float(os.environ.get("RAWL", 1.8545)), 0.5413, -2.2522)]

def grads(dtype, precision):
    op = RbfGramOp(X.to(dtype), precision=precision)
    cp = op.constrain(*[r.to(dtype) for r in raw])
    desc = op.descriptor(cp, dtype, n)
    ws = _lib.workspace(desc, n, 1, batch, dev)
    gs, gt = op.new_grads(*cp)
    Ld, Rd = L.to(dtype).contiguous(), R.to(dtype).contiguous()
    _lib.check(_lib.get().mfx_op_vjp_params(C.byref(desc), _lib.ptr(Ld), n, _lib.ptr(Rd), n, batch, C.byref(gs), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
    torch.cuda.synchronize()
    return np.array([t.double().sum().item() for t in gt])

ref = grads(torch.float64, "fp32")
for prec in ("fp32", "f16x3"):
    got = grads(torch.float32, prec)
    print(prec, got, "rel err", np.abs(got - ref) / np.abs(ref))
print("f64", ref)
