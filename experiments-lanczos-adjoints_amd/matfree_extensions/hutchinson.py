"""Hutchinson trace estimation -- MI355X build.

Mirrors the reference's ``matfree_extensions/hutchinson.py`` (``hutchinson_nograd``,
``hutchinson_custom_vjp``, ``hutchinson_batch``) and the two functions it borrows from third-party
``matfree`` at its call sites (``hutchinson``, ``sampler_rademacher`` / ``sampler_normal``;
util/gp_util.py:557, optim_logml_adjoints_adaptive.py:109).

Keys.  ``jax.random`` streams cannot be reproduced without JAX, so a "key" here is either
  * an ``int`` seed (or ``(seed, first_probe)``): probes come from libmfx's counter-based sampler
    (``mfx_rademacher``), identical for any sharding of the probes over GPUs, or
  * an explicit probe tensor ``(num, n)`` which samplers return unchanged (the parity interface).
Probes are processed as ONE batch (p, n) by the integrand when it supports batching (ours do), which
replaces ``jax.vmap`` (hutchinson.py:14,53).
"""

from __future__ import annotations

import torch

from . import _lib

_MASK = (1 << 64) - 1


def _splitmix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & _MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


def split(key, num: int = 2):
    """Deterministic stand-in for jax.random.split on integer keys (hutchinson.py:33,61)."""
    if torch.is_tensor(key):
        if key.shape[0] % num != 0:
            raise ValueError("explicit probe tensors split along axis 0 need a divisible length")
        return list(key.reshape(num, key.shape[0] // num, *key.shape[1:]))
    if isinstance(key, tuple):
        seed, first = key
        return [(_splitmix64((int(seed) << 1) ^ i), first) for i in range(num)]
    return [_splitmix64((int(key) << 1) ^ i) for i in range(num)]


def sampler_rademacher(x_like, /, *, num):
    """matfree.hutchinson.sampler_rademacher(x_like, num=): key -> (num, n) tensor of +-1."""
    n = x_like.numel()
    dtype, device = x_like.dtype, x_like.device

    def sample(key):
        if torch.is_tensor(key):
            return key
        seed, first = key if isinstance(key, tuple) else (key, 0)
        _lib.require_device(x_like)
        out = torch.empty((num, n), dtype=dtype, device=device)
        _lib.check(_lib.get().mfx_rademacher(int(seed) & _MASK, int(first), num, n, _lib.dtype_code(dtype),
                                             _lib.ptr(out), _lib.stream_ptr(device)))
        return out

    return sample


def sampler_normal(x_like, /, *, num):
    """matfree.hutchinson.sampler_normal: key -> (num, n) standard-normal tensor (torch generator)."""
    n = x_like.numel()
    dtype, device = x_like.dtype, x_like.device

    def sample(key):
        if torch.is_tensor(key):
            return key
        seed = key[0] ^ (key[1] * 0x9E3779B1) if isinstance(key, tuple) else key
        gen = torch.Generator(device=device)
        gen.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
        return torch.randn((num, n), dtype=dtype, device=device, generator=gen)

    return sample


def _tree_mean(tree):
    if torch.is_tensor(tree):
        return tree.mean(dim=0)
    if isinstance(tree, dict):
        return {k: _tree_mean(v) for k, v in tree.items()}
    if isinstance(tree, (tuple, list)):
        return type(tree)(_tree_mean(v) for v in tree)
    raise TypeError(f"unsupported output {type(tree)}")


def _tree_stack(items):
    first = items[0]
    if torch.is_tensor(first):
        return torch.stack(items)
    if isinstance(first, dict):
        return {k: _tree_stack([it[k] for it in items]) for k in first}
    return type(first)(_tree_stack([it[i] for it in items]) for i in range(len(first)))


def _map_over_samples(integrand_fun, samples, parameters):
    if getattr(integrand_fun, "batched", False):
        return integrand_fun(samples, *parameters)
    return _tree_stack([integrand_fun(vec, *parameters) for vec in samples])


def hutchinson(integrand_fun, /, sample_fun):
    """matfree.hutchinson.hutchinson: estimate(key, *params) = mean_i integrand(v_i, *params)."""

    def sample(key, *parameters):
        samples = sample_fun(key)
        return _tree_mean(_map_over_samples(integrand_fun, samples, parameters))

    return sample


def hutchinson_nograd(integrand_fun, /, sample_fun):
    """Hutchinson's estimator with gradients through the samples stopped (hutchinson.py:8-17)."""

    def sample(key, *parameters):
        samples = sample_fun(key).detach()
        return _tree_mean(_map_over_samples(integrand_fun, samples, parameters))

    return sample


def hutchinson_custom_vjp(integrand_fun, /, sample_fun):
    """Different probes in the backward pass (hutchinson.py:20-48).

    Forward: estimate with the ORIGINAL key (Q6, hutchinson.py:33-34).  Backward: fresh probes from
    the second half of ``split(key)`` and the mean of the integrand's VJP over them.
    """

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, key_holder, *parameters):
            key = key_holder[0]
            with torch.no_grad():
                value = _tree_mean(_map_over_samples(integrand_fun, sample_fun(key), parameters))
            if not torch.is_tensor(value):
                raise TypeError("hutchinson_custom_vjp supports tensor-valued integrands")
            ctx.key_bwd = split(key, 2)[1]
            ctx.save_for_backward(*[q for q in parameters if torch.is_tensor(q)])
            ctx.nontensor = [None if torch.is_tensor(q) else q for q in parameters]
            return value

        @staticmethod
        def backward(ctx, vjp_incoming):
            it = iter(ctx.saved_tensors)
            params = [next(it) if q is None else q for q in ctx.nontensor]
            with torch.enable_grad():
                live = [q.detach().requires_grad_(True) if torch.is_tensor(q) and q.is_floating_point() else q
                        for q in params]
                diff = [q for q in live if torch.is_tensor(q) and q.requires_grad]
                samples = sample_fun(ctx.key_bwd).detach()
                value = _tree_mean(_map_over_samples(integrand_fun, samples, live))
                grads = torch.autograd.grad(value, diff, vjp_incoming, allow_unused=True)
            it = iter(grads)
            out = [next(it) if (torch.is_tensor(q) and q.requires_grad) else None for q in live]
            return (None, *out)

    def sample(key, *parameters):
        return _Fn.apply((key,), *parameters)

    return sample


def hutchinson_batch(estimate_fun, /, num):
    """Mean over ``num`` sequential sub-estimates with split keys (hutchinson.py:57-65)."""

    def estimate_b(key, *parameters):
        keys = split(key, num)
        estimates = _tree_stack([estimate_fun(k, *parameters) for k in keys])
        return _tree_mean(estimates)

    return estimate_b
