"""Gaussian-process helpers on the SLQ log-determinant path -- MI355X build.

Covers the pieces of the reference's ``util/gp_util.py`` that sit on the hot path: the scaled-RBF
kernel parametrisation (:151-184), the softplus constraint (:187-201), the Gram matvec (:434-549,
here a native matrix-free operator instead of materialised row partitions) and the SLQ log-determinant
estimators (:552-621).  The GP model plumbing / CG / preconditioner side (:15-66,216-431) is the
"next" tier and is not part of this build.
"""

from __future__ import annotations

import torch

from .. import hutchinson, lanczos
from ..operators import RbfGramOp, softplus


def constraint_greater_than(minval, /):
    """util/gp_util.py:187-201: s -> minval + softplus(s) (torch-style threshold 20)."""
    return lambda s: minval + softplus(s)


def kernel_scaled_rbf(*, shape_in, shape_out):
    """(parametrize, params_like) with k(x, y) = softplus(raw_s) exp(-|x/l - y/l|^2 / 2), l = softplus(raw_l).

    Same parametrisation as GPyTorch's ScaleKernel(RBFKernel) (util/gp_util.py:151-184).  The returned
    scalar kernel is plain torch (used for small dense checks); the hot path uses ``gram_operator``.
    """
    constrain = constraint_greater_than(0.0)

    def parametrize(*, raw_lengthscale, raw_outputscale):
        def k(x, y):
            _assert_shapes(x, y, shape_in)
            lengthscale = constrain(raw_lengthscale)
            outputscale = constrain(raw_outputscale)
            xs, ys = x / lengthscale, y / lengthscale
            log_k = (xs * xs).sum() + (ys * ys).sum() - 2 * (xs * ys).sum()
            log_k = torch.clamp_min(log_k, 0.0)  # util/gp_util.py:173
            return outputscale * torch.exp(-log_k / 2)

        return k

    params_like = {"raw_lengthscale": torch.empty(shape_in), "raw_outputscale": torch.empty(shape_out)}
    return parametrize, params_like


def _scaled_kernel(shape_in, shape_out, radial):
    constrain = constraint_greater_than(0.0)

    def parametrize(*, raw_lengthscale, raw_outputscale):
        def k(x, y):
            _assert_shapes(x, y, shape_in)
            lengthscale = constrain(raw_lengthscale)
            outputscale = constrain(raw_outputscale)
            xs, ys = x / lengthscale, y / lengthscale
            scaled = torch.clamp_min((xs * xs).sum() + (ys * ys).sum() - 2 * (xs * ys).sum(), 0.0)
            return outputscale * radial(scaled)

        return k

    params_like = {"raw_lengthscale": torch.empty(shape_in), "raw_outputscale": torch.empty(shape_out)}
    return parametrize, params_like


def kernel_scaled_matern_32(*, shape_in, shape_out):
    """util/gp_util.py:69-107: (1 + r) exp(-r), r = sqrt(3 |x/l - y/l|^2 + eps).  Hot path: gram_operator(kernel="matern32")."""

    def radial(s):
        r = torch.sqrt(3.0 * s + torch.finfo(s.dtype).eps)
        return (1 + r) * torch.exp(-r)

    return _scaled_kernel(shape_in, shape_out, radial)


def kernel_scaled_matern_12(*, shape_in, shape_out):
    """util/gp_util.py:110-148: exp(-r), r = sqrt(|x/l - y/l|^2 + eps).  Hot path: gram_operator(kernel="matern12")."""

    def radial(s):
        return torch.exp(-torch.sqrt(s + torch.finfo(s.dtype).eps))

    return _scaled_kernel(shape_in, shape_out, radial)


def _assert_shapes(x, y, shape_in):
    if tuple(x.shape) != tuple(y.shape):
        error = "The arguments have different shapes: "
        error += f"{tuple(x.shape)} != {tuple(y.shape)})"
        raise ValueError(error)
    if tuple(x.shape) != tuple(shape_in):
        error = f"The shape {tuple(x.shape)} of the first argument "
        error += f"does not match 'shape_in'={shape_in}"
        raise ValueError(error)


def gram_matrix(fun, /):
    """util/gp_util.py:546-549: kernel function -> dense Gram matrix function (small inputs only)."""

    def gram(xs, ys):
        return torch.stack([torch.stack([fun(x, y) for y in ys]) for x in xs])

    return gram


def gram_operator(inputs, *, noise_minval=0.0, precision="f16x3", kernel="rbf"):
    """Matrix-free (K(X, X) + noise I) operator: the native replacement for
    gram_matvec / gram_matvec_partitioned / gram_matvec_sequential applied to the lazy RBF kernel
    with the noise on its diagonal (util/gp_util.py:225-226, 434-543).  No partition count is
    needed: tiles of K live only in registers.

    Use as  ``A = gram_operator(X).bind(raw_lengthscale, raw_outputscale, raw_noise)``.
    """
    return RbfGramOp(inputs, noise_minval=noise_minval, precision=precision, kernel=kernel)


def krylov_logdet_slq(krylov_depth, /, *, sample, num_batches: int, checkpoint: bool = False):
    """util/gp_util.py:552-576: logdet(A, key) -> (value, info) by stochastic Lanczos quadrature.

    ``checkpoint`` is accepted for signature parity; rematerialisation is a JAX memory device, the
    HIP path stores the (p, k, n) basis once and never materialises kernel tiles.
    """

    def logdet(A, /, key):
        integrand = lanczos.integrand_spd(torch.log, krylov_depth, A)
        estimate = hutchinson.hutchinson(integrand, sample)
        if num_batches == 1:
            value = estimate(key)
            return value, {"std": 0.0, "std_rel": 0.0}
        keys = hutchinson.split(key, num_batches)
        values = torch.stack([estimate(k) for k in keys])
        mean = values.mean(dim=0)
        std = values.std(dim=0, unbiased=False)
        return mean, {"std_abs": std, "std_rel": std / mean.abs()}

    return logdet


def krylov_logdet_slq_vjp_reuse(krylov_depth, /, *, sample, num_batches: int, checkpoint: bool = False):
    """util/gp_util.py:579-621: same value, cheap inexact gradient (re-used Lanczos basis)."""

    def logdet(A, /, key):
        integrand = lanczos.integrand_spd_custom_vjp_reuse(torch.log, krylov_depth, A)
        estimate = hutchinson.hutchinson(integrand, sample)
        keys = hutchinson.split(key, num_batches)
        values = torch.stack([estimate(k) for k in keys])
        return values.mean(dim=0), {"std": values.std(dim=0, unbiased=False)}

    return logdet
