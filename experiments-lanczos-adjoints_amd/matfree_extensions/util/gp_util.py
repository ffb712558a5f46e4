"""Gaussian-process helpers on the SLQ log-determinant path -- MI355X build.

Covers the pieces of the reference's ``util/gp_util.py`` that sit on the hot path: the scaled-RBF
kernel parametrisation (:151-184), the softplus constraint (:187-201), the Gram matvec (:434-549,
here a native matrix-free operator instead of materialised row partitions) and the SLQ log-determinant
estimators (:552-621); plus the "next" tier: the model / likelihood / logpdf plumbing of the
log-marginal likelihood and of the posterior mean (:15-66, 216-351, 367-431) on top of ``cg`` and ``low_rank``.
"""

from __future__ import annotations

import torch

import math

from .. import hutchinson, lanczos, low_rank
from ..operators import RbfGramOp, softplus


def constraint_greater_than(minval, /):
    """util/gp_util.py:187-201: s -> minval + softplus(s) (torch-style threshold 20)."""

    def constrain(s):
        return minval + softplus(s)

    constrain.minval = minval  # lets the native Gram operator apply the constraint itself
    return constrain


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

        k.native = ("rbf", raw_lengthscale, raw_outputscale)
        return k

    params_like = {"raw_lengthscale": torch.empty(shape_in), "raw_outputscale": torch.empty(shape_out)}
    return parametrize, params_like


def _scaled_kernel(shape_in, shape_out, radial, kind):
    constrain = constraint_greater_than(0.0)

    def parametrize(*, raw_lengthscale, raw_outputscale):
        def k(x, y):
            _assert_shapes(x, y, shape_in)
            lengthscale = constrain(raw_lengthscale)
            outputscale = constrain(raw_outputscale)
            xs, ys = x / lengthscale, y / lengthscale
            scaled = torch.clamp_min((xs * xs).sum() + (ys * ys).sum() - 2 * (xs * ys).sum(), 0.0)
            return outputscale * radial(scaled)

        k.native = (kind, raw_lengthscale, raw_outputscale)
        return k

    params_like = {"raw_lengthscale": torch.empty(shape_in), "raw_outputscale": torch.empty(shape_out)}
    return parametrize, params_like


def kernel_scaled_matern_32(*, shape_in, shape_out):
    """util/gp_util.py:69-107: (1 + r) exp(-r), r = sqrt(3 |x/l - y/l|^2 + eps).  Hot path: gram_operator(kernel="matern32")."""

    def radial(s):
        r = torch.sqrt(3.0 * s + torch.finfo(s.dtype).eps)
        return (1 + r) * torch.exp(-r)

    return _scaled_kernel(shape_in, shape_out, radial, "matern32")


def kernel_scaled_matern_12(*, shape_in, shape_out):
    """util/gp_util.py:110-148: exp(-r), r = sqrt(|x/l - y/l|^2 + eps).  Hot path: gram_operator(kernel="matern12")."""

    def radial(s):
        return torch.exp(-torch.sqrt(s + torch.finfo(s.dtype).eps))

    return _scaled_kernel(shape_in, shape_out, radial, "matern12")


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


# ------------------------------------------------------------------------------------------------
# "next" tier: the log-marginal likelihood around the SLQ log-determinant and the PCG solve
# ------------------------------------------------------------------------------------------------
def target_logml(model, likelihood, /):
    """util/gp_util.py:15-32."""

    def mll(inputs, targets, *p_logpdf, params_mean: dict, params_kernel: dict, params_likelihood: dict):
        mean, kernel = model(params_mean=params_mean, params_kernel=params_kernel)
        loss = likelihood(inputs, mean=mean, kernel=kernel, params=params_likelihood)
        value, info_pdf = loss(targets, *p_logpdf)
        return value, info_pdf

    return mll


def model_gp(mean_fun, kernel_fun):
    """util/gp_util.py:48-56."""

    def prior(params_mean: dict, params_kernel: dict):
        return mean_fun(**params_mean), kernel_fun(**params_kernel)

    return prior


def mean_constant(*, shape_out):
    """util/gp_util.py:59-66."""

    def parametrize(*, constant_value):
        def mean(_x):
            return constant_value

        mean.batched = lambda xs: constant_value.expand((xs.shape[0], *constant_value.shape))
        return mean

    return parametrize, {"constant_value": torch.empty(shape_out)}


def _mean_array(mean, inputs):
    if hasattr(mean, "batched"):
        return mean.batched(inputs)
    return torch.stack([mean(x) for x in inputs])


class _GramStrategy:
    """What gram_matvec* return here: the matvec strategy is always the native matrix-free operator."""

    def __init__(self, precision="f16x3"):
        self.precision = precision


def gram_matvec(*, precision="f16x3"):
    """util/gp_util.py:525-543."""
    return _GramStrategy(precision)


def gram_matvec_partitioned(num: int = 1, *, checkpoint: bool = False, precision="f16x3"):
    """util/gp_util.py:470-522; ``num`` / ``checkpoint`` are accepted for signature parity (no Gram tile ever reaches HBM)."""
    return _GramStrategy(precision)


def gram_matvec_sequential(*, checkpoint: bool = False, precision="f16x3"):
    """util/gp_util.py:434-467."""
    return _GramStrategy(precision)


def _native_cov(matvec, inputs, kernel, constrain, raw_noise):
    if not hasattr(kernel, "native"):
        raise TypeError("the kernel must come from kernel_scaled_rbf / kernel_scaled_matern_32 / kernel_scaled_matern_12 "
                        "(the native Gram operator evaluates it on the device)")
    if not hasattr(constrain, "minval"):
        raise TypeError("constrain must come from constraint_greater_than")
    kind, raw_lengthscale, raw_outputscale = kernel.native
    precision = matvec.precision if isinstance(matvec, _GramStrategy) else "f16x3"
    op = RbfGramOp(inputs, noise_minval=constrain.minval, precision=precision, kernel=kind)
    return op.bind(raw_lengthscale, raw_outputscale, raw_noise)


def likelihood_pdf(matvec, logpdf, *, constrain):
    """Gaussian likelihood, noise inside the lazy kernel (util/gp_util.py:216-240)."""

    def likelihood(inputs, mean, kernel, params: dict):
        cov_matvec = _native_cov(matvec, inputs, kernel, constrain, params["raw_noise"])

        def logpdf_partial(targets, *p_logpdf):
            return logpdf(targets, *p_logpdf, mean=_mean_array(mean, inputs), cov_matvec=cov_matvec)

        return logpdf_partial

    return likelihood, {"raw_noise": torch.empty(())}


def likelihood_pdf_p(matvec, logpdf_p, precondition, *, constrain):
    """Gaussian likelihood with a preconditioner built from the NOISE-FREE kernel (util/gp_util.py:243-276)."""

    def likelihood(inputs, mean, kernel, params: dict):
        cov_matvec = _native_cov(matvec, inputs, kernel, constrain, params["raw_noise"])
        noise = constrain(params["raw_noise"])
        pre, info = precondition(low_rank.without_noise(cov_matvec), len(inputs))

        def logpdf_partial(targets, *p_logpdf):
            val, aux = logpdf_p(targets, *p_logpdf, mean=_mean_array(mean, inputs), cov_matvec=cov_matvec,
                                P=pre.bind(noise))
            return val, {"precondition": info, "logpdf": aux}

        return logpdf_partial

    return likelihood, {"raw_noise": torch.empty(())}


def logpdf_scipy_stats():
    """Dense reference (util/gp_util.py:354-364: jax.scipy.stats.multivariate_normal.logpdf of the materialised covariance) -- here
    torch.distributions.MultivariateNormal on the same materialised matrix; used by the reference's gpytorch comparison test and its
    fixed-step training script (optim_logml_adjoints_fixed.py:136) as the exact baseline."""

    def logpdf(y, /, *, mean, cov_matvec):
        n = mean.shape[0]
        cov_matrix = cov_matvec(torch.eye(n, dtype=mean.dtype, device=mean.device)).t()
        cov_matrix = 0.5 * (cov_matrix + cov_matrix.t())  # (the distribution insists on exact symmetry)
        return torch.distributions.MultivariateNormal(mean, covariance_matrix=cov_matrix).log_prob(y), {}

    return logpdf


def logpdf_cholesky():
    """Dense baseline (util/gp_util.py:367-393): materialise the covariance through the operator, then Cholesky."""

    def logpdf(y, /, *, mean, cov_matvec):
        n = mean.shape[0]
        cov_matrix = cov_matvec(torch.eye(n, dtype=mean.dtype, device=mean.device)).t()
        cholesky = torch.linalg.cholesky(cov_matrix)
        logdet = torch.log(torch.diagonal(cholesky)).sum()
        tmp = torch.linalg.solve_triangular(cholesky, (y - mean)[:, None], upper=False)[:, 0]
        mahalanobis = tmp @ tmp
        return -logdet - 0.5 * mahalanobis - n / 2 * math.log(2 * math.pi), {}

    return logpdf


def logpdf_krylov(solve, logdet):
    """util/gp_util.py:396-411."""

    def logpdf(y, *params_logdet, mean, cov_matvec):
        logdet_, info_logdet = logdet(cov_matvec, *params_logdet)
        logdet_ = logdet_ / 2
        tmp, info_solve = solve(cov_matvec, y - mean)
        mahalanobis = (y - mean) @ tmp
        info = {"logdet": info_logdet, "solve": info_solve}
        (n,) = mean.shape
        return -logdet_ - 0.5 * mahalanobis - n / 2 * math.log(2 * math.pi), info

    return logpdf


def logpdf_krylov_p(solve_p, logdet):
    """util/gp_util.py:414-431."""

    def logpdf(y, *params_logdet, mean, cov_matvec, P):
        logdet_, info_logdet = logdet(cov_matvec, *params_logdet)
        logdet_ = logdet_ / 2
        tmp, info_solve = solve_p(cov_matvec, y - mean, P=P)
        mahalanobis = (y - mean) @ tmp
        info = {"logdet": info_logdet, "solve": info_solve}
        (n,) = mean.shape
        return -logdet_ - 0.5 * mahalanobis - n / 2 * math.log(2 * math.pi), info

    return logpdf


def target_posterior(model, likelihood, /):
    """util/gp_util.py:35-45: -> posterior(inputs, targets, params_mean, params_kernel, params_likelihood) -> (predict, {})
    with predict(xs) -> (posterior mean at xs, info)."""

    def posterior(inputs, targets, params_mean: dict, params_kernel: dict, params_likelihood: dict):
        mean, kernel = model(params_mean, params_kernel)
        condition = likelihood(inputs, mean, kernel, params=params_likelihood)
        return (lambda xs: condition(xs, targets=targets)), {}

    return posterior


def likelihood_condition(matvec, solve, *, constrain):
    """util/gp_util.py:279-310: weights = solve(K + noise I, y - m);  prediction = m(xs) + K(xs, X) weights."""

    def likelihood(inputs, mean, kernel, params: dict):
        cov_matvec = _native_cov(matvec, inputs, kernel, constrain, params["raw_noise"])

        def condition_partial(xs, targets):
            weights, info = solve(cov_matvec, targets - _mean_array(mean, inputs))
            return _mean_array(mean, xs) + cov_matvec.op.cross_apply(xs, weights, *cov_matvec.params), {"solve": info}

        return condition_partial

    return likelihood, {"raw_noise": torch.empty(())}


def likelihood_condition_p(matvec, solve_p, *, precondition, constrain):
    """util/gp_util.py:313-351: the same with a preconditioned solve."""

    def likelihood(inputs, mean, kernel, params: dict):
        cov_matvec = _native_cov(matvec, inputs, kernel, constrain, params["raw_noise"])
        noise = constrain(params["raw_noise"])
        pre, _info = precondition(low_rank.without_noise(cov_matvec), len(inputs))

        def condition_partial(xs, targets):
            weights, info = solve_p(cov_matvec, targets - _mean_array(mean, inputs), P=pre.bind(noise))
            return _mean_array(mean, xs) + cov_matvec.op.cross_apply(xs, weights, *cov_matvec.params), {"solve": info}

        return condition_partial

    return likelihood, {"raw_noise": torch.empty(())}
