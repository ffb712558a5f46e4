"""Multi-GPU Hutchinson: probes shard across ranks, ONE all-reduce per estimate.

The reference is single-device (no collective anywhere).  The estimator is a mean over
independent probes (hutchinson.py:14-15), so the natural MI355X decomposition is one process per GPU,
the operator replicated (X is a few MB), each rank running the full forward + adjoint for its own
probes, and a single fused RCCL all-reduce of [sum q, sum q^2, sum d/dtheta q] over xGMI -- tens of
bytes to a few MB, i.e. latency-bound, hence exactly one collective per value-and-grad.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def shard_probes(num_total: int, rank: int, world_size: int):
    """Contiguous split of ``num_total`` probes -> (first_probe, count) of this rank."""
    base, rem = divmod(num_total, world_size)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def reduce_estimate(local_values, local_grads, num_total: int, group=None):
    """All-reduce per-probe values and summed parameter gradients in ONE collective.

    local_values: (p_local,) integrand values of this rank's probes.
    local_grads : tuple of tensors = d/dtheta of SUM_b value_b over this rank's probes.
    -> (mean, std over probes, tuple of gradients of the mean)
    """
    lv = local_values.double()  # mean^2 ~ 1e10 at n = 1e5: the second moment needs fp64
    flat = [lv.sum().reshape(1), (lv**2).sum().reshape(1)]
    flat += [g.reshape(-1).double() for g in local_grads]
    buf = torch.cat(flat).contiguous()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    mean = buf[0] / num_total
    var = torch.clamp_min(buf[1] / num_total - mean**2, 0.0)
    grads, off = [], 2
    for g in local_grads:
        grads.append((buf[off : off + g.numel()] / num_total).reshape(g.shape).to(g.dtype))
        off += g.numel()
    return mean, var.sqrt(), tuple(grads)


def value_and_grad_sharded(integrand, sample_local, params, *, num_total: int, group=None):
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
    return reduce_estimate(values.detach(), grads, num_total, group=group)
