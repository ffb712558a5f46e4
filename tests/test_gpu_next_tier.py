"""GPU parity for the "next" tier: compositions on top of the hot path.

expm_arnoldi / sampler_lanczos / wave operator (util/pde_util.py) against the oracle's Arnoldi + scipy's dense expm.
fp64 build vs fp64 oracle rtol 1e-9 (values) / 1e-6 (gradients vs central differences); fp32 rtol 2e-4.
"""

import numpy as np
import pytest
import scipy.linalg
import torch

from oracle import slq_oracle as orc

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from matfree_extensions.operators import DenseOp
    from matfree_extensions.util import pde_util

DEV = torch.device("cuda:0")


def T(x, dtype=torch.float64, grad=False):
    t = torch.tensor(np.asarray(x), dtype=dtype, device=DEV)
    return t.requires_grad_(True) if grad else t


def N(t):
    return t.detach().cpu().numpy().astype(np.float64)


def _oracle_expm(A, k, dt, y0):
    Q, H, _r, c = orc.arnoldi_forward(orc.DenseOp(), k, y0, A, reortho="full")
    return Q @ scipy.linalg.expm(dt * H)[:, 0] / c


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-4)])
@pytest.mark.parametrize("k", [6, 24])
def test_expm_arnoldi_dense_nonsymmetric(dtype, tol, k):
    rng = np.random.default_rng(3)
    n = 24
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    y0 = rng.standard_normal(n)
    dt = 0.7
    expm = pde_util.expm_arnoldi(k)
    out, info = expm(DenseOp(), dt, T(y0, dtype), T(A, dtype))
    assert info["num_matvecs"] == k
    want = _oracle_expm(A, k, dt, y0)
    assert np.allclose(N(out), want, rtol=tol, atol=tol * np.abs(want).max())
    if k == n:  # full depth: the Krylov approximation is exact
        assert np.allclose(N(out), scipy.linalg.expm(dt * A) @ y0, rtol=max(tol, 1e-8), atol=max(tol, 1e-8))


def test_expm_arnoldi_gradient_matches_central_differences():
    rng = np.random.default_rng(4)
    n, k, dt = 16, 7, 0.5
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    y0 = rng.standard_normal(n)
    w = rng.standard_normal(n)
    expm = pde_util.expm_arnoldi(k)
    At, yt = T(A, grad=True), T(y0, grad=True)
    out, _ = expm(DenseOp(), dt, yt, At)
    (out * T(w)).sum().backward()
    dA, dy = rng.standard_normal((n, n)), rng.standard_normal(n)
    h = 1e-6
    f = lambda s: w @ _oracle_expm(A + s * dA, k, dt, y0 + s * dy)
    fd = (f(h) - f(-h)) / (2 * h)
    got = (N(At.grad) * dA).sum() + N(yt.grad) @ dy
    assert abs(got - fd) <= 1e-6 * max(1.0, abs(fd))


@pytest.mark.parametrize("boundary", ["neumann", "dirichlet"])
def test_wave_operator_matches_padded_convolution_and_solver_expm(boundary):
    import scipy.signal

    rng = np.random.default_rng(5)
    res, dx = 8, 0.25
    op, values_fn = pde_util.wave_operator(res, dx, boundary=boundary, device=DEV)
    scale = rng.uniform(0.5, 1.5, (res, res))
    x = rng.standard_normal((2, res, res))
    # util/pde_util.py:126-157 restated with numpy/scipy
    st = np.array([[0.0, 1.0, 0.0], [1.0, -2.0, 1.0], [0.0, 1.0, 0.0]]) / dx**2
    pad = np.pad(x[0], 1, mode="edge") if boundary == "neumann" else np.pad(x[0], 1)
    want = np.stack([x[1], scipy.signal.convolve2d(st, pad, mode="valid") * scale])
    vals = values_fn(T(scale))
    got = op(T(x).reshape(-1), vals)
    assert np.allclose(N(got).reshape(2, res, res), want, rtol=1e-12, atol=1e-12)

    # dense A for the oracle, then exp(t A) y0 via Arnoldi on the CSR kernels (forward + gradient wrt the scale field)
    n = 2 * res * res
    A = N(op(torch.eye(n, dtype=torch.float64, device=DEV), vals)).T
    k, t1 = 20, 0.05
    sc = T(scale, grad=True)
    solve = pde_util.solver_expm(0.0, t1, lambda v, vv: op(v.reshape(-1), vv).reshape(v.shape), pde_util.expm_arnoldi(k))
    # the vector field is a python callable here: exercises the callback operator with a non-symmetric matvec
    y1, _ = solve(T(x), values_fn(sc))
    want = _oracle_expm(A, k, t1, x.reshape(-1))
    assert np.allclose(N(y1).reshape(-1), want, rtol=1e-8, atol=1e-8 * np.abs(want).max())
    w = rng.standard_normal(n)
    (y1.reshape(-1) * T(w)).sum().backward()
    ds = rng.standard_normal((res, res))
    h = 1e-6

    def f(s):
        v = N(values_fn(T(scale + s * ds)))
        Ad = N(op(torch.eye(n, dtype=torch.float64, device=DEV), T(v))).T
        return w @ _oracle_expm(Ad, k, t1, x.reshape(-1))

    fd = (f(h) - f(-h)) / (2 * h)
    got = (N(sc.grad) * ds).sum()
    assert abs(got - fd) <= 1e-5 * max(1.0, abs(fd))


def test_sampler_lanczos_full_rank_is_matrix_square_root():
    rng = np.random.default_rng(6)
    n, num = 12, 5
    eig = rng.uniform(0.5, 2.0, n)
    A = orc.symmetric_matrix_from_eigenvalues(eig, seed=1)
    eps = rng.standard_normal((num, n))
    mean = rng.standard_normal(n)
    sample = pde_util.sampler_lanczos(mean=T(mean), cov_matvec=DenseOp().bind(T(A)), num=num, lanczos_rank=n)
    got = N(sample(T(eps)))
    w, v = np.linalg.eigh(A)
    want = eps @ ((v * np.sqrt(w)) @ v.T) + mean
    assert np.allclose(got, want, rtol=1e-8, atol=1e-8)
    # low rank: matches the oracle's Lanczos restatement of util/pde_util.py:335-356
    k = 5
    sample = pde_util.sampler_lanczos(mean=T(mean), cov_matvec=DenseOp().bind(T(A)), num=num, lanczos_rank=k)
    got = N(sample(T(eps)))
    for b in range(num):
        nrm = np.linalg.norm(eps[b])
        (Q, (dg, off)), _ = orc.tridiag_full(orc.DenseOp(), k, eps[b] / nrm, A)
        K = np.diag(dg) + np.diag(off, 1) + np.diag(off, -1)
        ww, vv = np.linalg.eigh(K)
        fac = (vv * np.sqrt(np.maximum(ww, 0))) @ vv.T
        want = nrm * Q.T @ fac @ (Q @ (eps[b] / nrm)) + mean
        assert np.allclose(got[b], want, rtol=1e-8, atol=1e-8)
