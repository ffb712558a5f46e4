import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "experiments-lanczos-adjoints_amd"))
from matfree_extensions import hutchinson, lanczos, arnoldi
from matfree_extensions.util import gp_util
import matfree_extensions.arnoldi as A
n,d,k,p = int(os.environ.get("DIAG_N", "32768")),8,40,64
dev=torch.device("cuda:0")
gen=torch.Generator().manual_seed(4)
X=torch.randn((n,d),generator=gen,dtype=torch.float32).to(dev)
inv=lambda x: float(np.log(np.expm1(x)))
params=[torch.tensor(v,dtype=torch.float32,device=dev,requires_grad=True) for v in (inv(2.0),inv(1.0),inv(0.1))]
orig_empty=torch.empty
stash={}
def hook_empty(*a, **kw):
    t=orig_empty(*a, **kw)
    if len(a)==1 and isinstance(a[0], tuple) and len(a[0])==3 and a[0]==(p,k,n): stash.setdefault('t',[]).append(t)
    return t
torch.empty=hook_empty
integrand=lanczos.integrand_spd(torch.log,k,gp_util.gram_operator(X))
probes=hutchinson.sampler_rademacher(X[:,0],num=p)(0)
vals=integrand(probes,*params)
g=torch.autograd.grad(vals.sum(),params)
torch.cuda.synchronize()
print("grads", [t.item()/p for t in g])
Q, Lam = stash['t'][0], stash['t'][1]
am = Lam.abs().amax(dim=-1)   # (p,k)
print("Lam row amax: min %.3e max %.3e  ; per-step median:" % (am.min().item(), am.max().item()))
print(am.median(dim=0).values.cpu().numpy())
print("Q row amax min/max", Q.abs().amax(dim=-1).min().item(), Q.abs().amax(dim=-1).max().item())
