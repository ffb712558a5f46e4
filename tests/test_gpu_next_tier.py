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


def test_expm_pade_baseline_and_small_pde_helpers():
    """util/pde_util.py:271-280 (dense baseline of expm_arnoldi), :14-15 (mesh), :160-173 (losses): against scipy / NumPy."""
    rng = np.random.default_rng(4)
    n = 24
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    y0 = rng.standard_normal(n)
    want = scipy.linalg.expm(0.7 * A) @ y0
    out = pde_util.expm_pade()(DenseOp(), 0.7, T(y0), T(A))
    assert np.allclose(N(out), want, rtol=1e-10, atol=1e-12)
    out_cb = pde_util.expm_pade()(lambda v, a: v @ a.T if v.dim() == 2 else a @ v, 0.7, T(y0), T(A))  # a plain callable
    assert np.allclose(N(out_cb), want, rtol=1e-10, atol=1e-12)
    full, _ = pde_util.expm_arnoldi(n)(DenseOp(), 0.7, T(y0), T(A))  # full depth: the Krylov form is exact
    assert np.allclose(N(full), N(out), rtol=1e-8, atol=1e-10)
    x, y = np.linspace(0, 1, 5), np.linspace(-1, 1, 3)
    assert np.array_equal(N(pde_util.mesh_tensorproduct(T(x), T(y))), np.stack(np.meshgrid(x, y)))
    sol, tg = rng.standard_normal((3, 4)), rng.standard_normal((3, 4))
    assert np.isclose(float(pde_util.loss_mse()(T(sol), targets=T(tg))), np.mean((sol - tg) ** 2))
    rel = pde_util.loss_mse_relative(nugget=0.1)(T(sol), targets=T(tg))
    assert np.isclose(float(rel), np.mean((sol - tg) ** 2 / (0.1 + np.abs(tg))))


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


# ------------------------------------------------------------------------------------------------
# §8f-1: conjugate gradients, partial Cholesky, preconditioner, log-marginal likelihood
# ------------------------------------------------------------------------------------------------
if torch.cuda.is_available():
    from matfree_extensions import cg, hutchinson, low_rank
    from matfree_extensions.operators import RbfGramOp
    from matfree_extensions.util import gp_util

F64 = (torch.float64, 1e-9)
F32 = (torch.float32, 2e-3)


def _spd(n, lo=1.0, hi=10.0, seed=0):
    return orc.symmetric_matrix_from_eigenvalues(np.linspace(lo, hi, n), seed=seed)


@pytest.mark.parametrize("dtype,tol", [F64, F32])
@pytest.mark.parametrize("n,steps", [(9, 9), (300, 25), (2500, 12)])
def test_cg_fixed_step_matches_oracle(dtype, tol, n, steps):
    A = _spd(n)
    b = np.arange(1.0, n + 1.0) / n
    want, winfo = orc.pcg_fixed_step(lambda v: A @ v, b, num_matvecs=steps)
    x, info = cg.cg_fixed_step(steps)(DenseOp().bind(T(A, dtype)), T(b, dtype))
    assert np.allclose(N(x), want, rtol=tol, atol=tol * np.abs(want).max())
    assert np.allclose(N(info["residual_abs"]), winfo["residual_abs"], atol=10 * tol * np.abs(b).max())
    assert set(info) == {"residual_abs", "residual_rel"}
    if n == 9:  # test_cg.py:10-18, with a plain callable like the reference's test
        At = T(A, dtype)
        x, _ = cg.cg_fixed_step(n)(lambda v: At @ v, T(b, dtype))
        assert np.allclose(N(x), np.linalg.solve(A, b), rtol=max(tol, 1e-8), atol=max(tol, 1e-8))


def test_cg_runs_past_convergence_and_zero_steps():
    A = _spd(5, 1.0, 5.0)
    b = np.arange(1.0, 6.0)
    x, _ = cg.cg_fixed_step(60)(DenseOp().bind(T(A)), T(b))  # 0/0 steps are absorbed by _safe_divide (cg.py:222-241)
    assert np.all(np.isfinite(N(x))) and np.allclose(N(x), np.linalg.solve(A, b))
    x, info = cg.cg_fixed_step(0)(DenseOp().bind(T(A)), T(b))
    assert np.all(N(x) == 0) and np.allclose(N(info["residual_abs"]), b)


@pytest.mark.parametrize("dtype,tol", [F64, F32])
def test_cg_adaptive_matches_oracle_per_right_hand_side(dtype, tol):
    n = 200
    A = _spd(n, 1.0, 200.0)
    rng = np.random.default_rng(0)
    # right-hand sides of very different difficulty: an eigenvector converges in one step
    w, V = np.linalg.eigh(A)
    B = np.stack([rng.standard_normal(n), V[:, 3] * 2.0, 1e-3 * rng.standard_normal(n)])
    kw = dict(atol=1e-4, rtol=1e-4, maxiter=500, miniter=2)
    solve = cg.cg_adaptive(**kw)
    X, info = solve(DenseOp().bind(T(A, dtype)), T(B, dtype))
    steps = N(info["num_steps"]).astype(int)
    for i in range(3):
        want, winfo = orc.pcg_adaptive(lambda v: A @ v, B[i], **kw)
        if dtype == torch.float64:
            assert steps[i] == winfo["num_steps"], (i, steps[i], winfo["num_steps"])
        else:
            assert abs(steps[i] - winfo["num_steps"]) <= 2
        assert np.allclose(N(X[i]), want, rtol=max(tol, 1e-3), atol=2e-4)
        xi, ii = solve(DenseOp().bind(T(A, dtype)), T(B[i], dtype))  # a batch equals independent solves
        assert int(ii["num_steps"]) == steps[i]
        assert torch.equal(xi, X[i])
    assert steps[1] == 2 and steps[0] > steps[2] >= 2
    # maxiter caps the iteration (cg.py:106)
    _x, info = cg.cg_adaptive(atol=1e-12, rtol=0.0, maxiter=7, miniter=0)(DenseOp().bind(T(A, dtype)), T(B[0], dtype))
    assert int(info["num_steps"]) == 7


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 2e-4)])
@pytest.mark.parametrize("pivot", [False, True])
def test_partial_cholesky_dense_matches_oracle(dtype, tol, pivot):
    n, rank = 40, 12
    cov = orc.symmetric_matrix_from_eigenvalues(0.1 + np.random.default_rng(1).uniform(size=n))
    el = lambda i, j: cov[i, j]
    make = low_rank.cholesky_partial_pivot if pivot else low_rank.cholesky_partial
    want, winfo = (orc.cholesky_partial_pivot if pivot else orc.cholesky_partial)(el, n, rank)
    got, info = make(rank=rank)(T(cov, dtype), n)
    assert got.shape == (n, rank)
    assert np.allclose(N(got), want, rtol=tol, atol=tol)
    if pivot:
        assert bool(info["success"]) and winfo["success"]
        assert np.array_equal(N(info["pivots"]).astype(int), winfo["pivots"])
    # full rank reconstructs the matrix (test_low_rank.py:12-25); without pivoting it IS the Cholesky factor (:28-41)
    full, _ = make(rank=n)(T(cov, dtype), n)
    assert np.allclose(N(full @ full.T), cov, atol=max(tol, 1e-12) * 10)
    if not pivot:
        assert np.allclose(N(full), np.linalg.cholesky(cov), atol=1e-6 if dtype == torch.float64 else 2e-3)
    with pytest.raises(ValueError, match="Rank exceeds n"):
        make(rank=n + 1)(T(cov, dtype), n)
    with pytest.raises(ValueError, match="Rank must be positive"):
        make(rank=0)(T(cov, dtype), n)


@pytest.mark.parametrize("kernel", ["rbf", "matern32", "matern12"])
def test_partial_cholesky_pivot_kernel_gram_and_preconditioner(kernel):
    rng = np.random.default_rng(2)
    n, d, rank = 500, 3, 24
    X = rng.uniform(-1, 1, (n, d))
    raw = (np.float64(0.2), np.float64(0.4), np.float64(-3.0))
    oop = orc.RbfGramOp(X, noise_minval=1e-4, kernel=kernel)
    ls, s, noise = oop.constrained(*raw)
    K = orc.kernel_matrix(kernel, X, X, ls, s, diag_offset=0)
    bound = RbfGramOp(T(X), noise_minval=1e-4, kernel=kernel).bind(*(T(q) for q in raw))
    want, winfo = orc.cholesky_partial_pivot(lambda i, j: K[i, j], n, rank)
    got, info = low_rank.cholesky_partial_pivot(rank=rank)(low_rank.without_noise(bound), n)
    assert np.array_equal(N(info["pivots"]).astype(int), winfo["pivots"])
    assert np.allclose(N(got), want, rtol=1e-8, atol=1e-9)
    # with the noise on the diagonal (likelihood_pdf's lazy kernel, util/gp_util.py:225-226)
    Kn = K + noise * np.eye(n)
    want_n, _ = orc.cholesky_partial_pivot(lambda i, j: Kn[i, j], n, rank)
    got_n, _ = low_rank.cholesky_partial_pivot(rank=rank)(bound, n)
    assert np.allclose(N(got_n), want_n, rtol=1e-8, atol=1e-9)

    # preconditioner solve (low_rank.py:31-43), single vector and batch
    pre, pinfo = low_rank.preconditioner(low_rank.cholesky_partial_pivot(rank=rank))(low_rank.without_noise(bound), n)
    assert bool(pinfo["success"])
    V = rng.standard_normal((3, n))
    for sval in (float(noise), 0.5):
        wantz = np.stack([orc.precondition_solve(want, v, sval) for v in V])
        assert np.allclose(N(pre(T(V), sval)), wantz, rtol=1e-7, atol=1e-7 * np.abs(wantz).max())
        assert np.allclose(N(pre(T(V[0]), T(sval))), wantz[0], rtol=1e-7, atol=1e-7 * np.abs(wantz).max())
    v = T(V[0], grad=True)
    with pytest.raises(RuntimeError):  # low_rank.py:47-55
        pre(v, 0.5).sum().backward()

    # PCG with it, fixed and adaptive, against the oracle
    A = lambda v: K @ v + noise * v
    P = lambda v: orc.precondition_solve(want, v, noise)
    b = rng.standard_normal(n)
    wx, winfo = orc.pcg_fixed_step(A, b, P, num_matvecs=10)
    x, info = cg.pcg_fixed_step(10)(bound, T(b), pre.bind(noise))
    assert np.allclose(N(x), wx, rtol=1e-6, atol=1e-6 * np.abs(wx).max())
    _wx0, plain = orc.pcg_fixed_step(A, b, None, num_matvecs=10)
    assert np.linalg.norm(N(info["residual_abs"])) < 0.5 * np.linalg.norm(plain["residual_abs"])
    kw = dict(atol=1e-6, rtol=0.0, maxiter=400, miniter=3)
    wx, winfo = orc.pcg_adaptive(A, b, P, **kw)
    x, info = cg.pcg_adaptive(**kw)(bound, T(b), pre.bind(T(noise)))
    assert abs(int(info["num_steps"]) - winfo["num_steps"]) <= 1
    assert np.allclose(N(x), wx, rtol=1e-5, atol=1e-5 * np.abs(wx).max())
    with pytest.raises(TypeError):
        cg.pcg_fixed_step(3)(bound, T(b), lambda v: v)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-7), (torch.float32, 5e-3)])
def test_linear_solve_gradient_rule(dtype, tol):
    rng = np.random.default_rng(3)
    n, d, steps = 300, 2, 30
    X = rng.uniform(-1, 1, (n, d))
    raw = (np.array([0.3, -0.1]), np.float64(0.2), np.float64(-1.5))
    oop = orc.RbfGramOp(X, noise_minval=1e-3)
    B = rng.standard_normal((2, n))
    W = rng.standard_normal((2, n))
    params = [T(q, dtype, grad=True) for q in raw]
    Bt = T(B, dtype, grad=True)
    x, _ = cg.cg_fixed_step(steps)(RbfGramOp(T(X, dtype), noise_minval=1e-3).bind(*params), Bt)
    (x * T(W, dtype)).sum().backward()
    solver = lambda A, rhs: orc.pcg_fixed_step(A, rhs, num_matvecs=steps)
    gl = [0.0, 0.0, 0.0]
    for i in range(2):
        wx, _ = solver(lambda v: oop.apply(v, *raw), B[i])
        assert np.allclose(N(x[i]), wx, rtol=tol, atol=tol * np.abs(wx).max())
        lam, dp = orc.linear_solve_vjp(oop, raw, solver, wx, W[i])
        assert np.allclose(N(Bt.grad[i]), lam, rtol=tol, atol=tol * np.abs(lam).max())
        gl = [a + b for a, b in zip(gl, dp)]
    for got, want in zip(params, gl):
        # CG amplifies round-off (loss of conjugacy), hence the looser bound on the parameter sweep
        assert np.allclose(N(got.grad), want, rtol=50 * tol, atol=50 * tol * np.abs(want).max()), (N(got.grad), want)
    # same rule through a Python callable (the parameter VJP then goes through torch.autograd)
    A0 = _spd(20)
    At = T(A0, grad=True)
    bt = T(B[0, :20], grad=True)
    x, _ = cg.cg_fixed_step(20)(lambda v, M: (M + M.T) @ v / 2, bt) if False else cg.cg_fixed_step(20)(DenseOp().bind(At), bt)
    x.sum().backward()
    lam = np.linalg.solve(A0, np.ones(20))
    assert np.allclose(N(bt.grad), lam, rtol=1e-8)
    assert np.allclose(N(At.grad), -np.outer(lam, np.linalg.solve(A0, B[0, :20])), rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("dtype,vtol,gtol", [(torch.float64, 1e-6, 2e-5), (torch.float32, 1e-4, 5e-3)])
@pytest.mark.parametrize("kernel_name", ["rbf", "matern32"])
def test_target_logml_krylov_preconditioned_value_and_grad(dtype, vtol, gtol, kernel_name):
    """The composition the GP experiments train with (optim_logml_adjoints_fixed.py:84-113):
    SLQ log-determinant + preconditioned CG behind target_logml, value and gradient w.r.t. every parameter."""
    rng = np.random.default_rng(4)
    n, d, k, nprobes, rank, steps = 384, 3, 12, 8, 16, 20
    X = rng.uniform(-1, 1, (n, d))
    y = np.sin(X.sum(-1)) + 0.1 * rng.standard_normal(n)
    raw = {"raw_lengthscale": np.float64(0.1), "raw_outputscale": np.float64(0.3), "raw_noise": np.float64(-2.0)}
    cval = 0.25
    minval = 1e-3

    make_kernel = {"rbf": gp_util.kernel_scaled_rbf, "matern32": gp_util.kernel_scaled_matern_32}[kernel_name]
    k_fun, _ = make_kernel(shape_in=(d,), shape_out=())
    m_fun, _ = gp_util.mean_constant(shape_out=())
    sample = hutchinson.sampler_rademacher(torch.empty(n, dtype=dtype, device=DEV), num=nprobes)
    logdet = gp_util.krylov_logdet_slq(k, sample=sample, num_batches=1)
    logpdf_p = gp_util.logpdf_krylov_p(solve_p=cg.pcg_fixed_step(steps), logdet=logdet)
    precondition = low_rank.preconditioner(low_rank.cholesky_partial_pivot(rank=rank))
    likelihood, _ = gp_util.likelihood_pdf_p(gp_util.gram_matvec(precision="f16x3-matvec"), logpdf_p, precondition,
                                             constrain=gp_util.constraint_greater_than(minval))
    loss = gp_util.target_logml(gp_util.model_gp(m_fun, k_fun), likelihood)
    tl, ts, tn = (T(raw[q], dtype, grad=True) for q in ("raw_lengthscale", "raw_outputscale", "raw_noise"))
    tc = T(cval, dtype, grad=True)
    ty = T(y, dtype, grad=True)
    seed = 11
    value, info = loss(T(X, dtype), ty, seed, params_mean={"constant_value": tc},
                       params_kernel={"raw_lengthscale": tl, "raw_outputscale": ts},
                       params_likelihood={"raw_noise": tn})
    value.backward()
    assert bool(info["precondition"]["success"])

    # oracle composition
    eps = float(torch.finfo(dtype).eps)
    oop = orc.RbfGramOp(X, noise_minval=minval, kernel=kernel_name, eps=eps)
    params = (raw["raw_lengthscale"], raw["raw_outputscale"], raw["raw_noise"])
    ls, s, noise = oop.constrained(*params)
    K = orc.kernel_matrix(kernel_name, X, X, ls, s, diag_offset=0, eps=eps)
    L, _ = orc.cholesky_partial_pivot(lambda i, j: K[i, j], n, rank)
    P = lambda v: orc.precondition_solve(L, v, noise)
    probes = orc.rademacher(seed, nprobes, n)
    ld, ld_grads, _ = orc.hutchinson_value_and_grad(oop, k, probes, params)
    b = y - cval
    solver = lambda A, rhs: orc.pcg_fixed_step(A, rhs, P, num_matvecs=steps)
    wvalue, _ = orc.logpdf_krylov(y, np.full(n, cval), logdet_value=ld, solve=lambda rhs: solver(lambda v: oop.apply(v, *params), rhs))
    assert abs(float(value.detach()) - wvalue) <= vtol * abs(wvalue), (float(value.detach()), wvalue)  # CG amplifies round-off
    x, _ = solver(lambda v: oop.apply(v, *params), b)
    lam, dp = orc.linear_solve_vjp(oop, params, solver, x, -0.5 * b)  # cotangent of x in -1/2 b^T x
    db = -0.5 * x + lam
    want = [-0.5 * g + q for g, q in zip(ld_grads, dp)]
    for got, w in zip((tl, ts, tn), want):
        assert np.allclose(N(got.grad), w, rtol=gtol, atol=gtol * max(abs(np.asarray(w)).max(), 1.0)), (N(got.grad), w)
    assert np.allclose(N(ty.grad), db, rtol=gtol, atol=gtol * np.abs(db).max())
    assert np.allclose(float(tc.grad), -db.sum(), rtol=gtol, atol=gtol * np.abs(db).sum())


def test_logpdf_cholesky_and_unpreconditioned_likelihood():
    rng = np.random.default_rng(5)
    n, d = 64, 2
    X = rng.uniform(-1, 1, (n, d))
    y = rng.standard_normal(n)
    raw = (np.float64(0.5), np.float64(0.1), np.float64(-1.0))
    k_fun, _ = gp_util.kernel_scaled_rbf(shape_in=(d,), shape_out=())
    m_fun, _ = gp_util.mean_constant(shape_out=())
    constrain = gp_util.constraint_greater_than(1e-2)
    oop = orc.RbfGramOp(X, noise_minval=1e-2)
    cov = np.stack([oop.apply(e, *raw) for e in np.eye(n)]).T
    want = orc.logpdf_cholesky(y, np.full(n, 0.1), cov)
    kw = dict(params_mean={"constant_value": T(0.1)}, params_kernel={"raw_lengthscale": T(raw[0]), "raw_outputscale": T(raw[1])},
              params_likelihood={"raw_noise": T(raw[2])})
    lik, _ = gp_util.likelihood_pdf(gp_util.gram_matvec(), gp_util.logpdf_cholesky(), constrain=constrain)
    value, _ = gp_util.target_logml(gp_util.model_gp(m_fun, k_fun), lik)(T(X), T(y), **kw)
    assert abs(float(value) - want) <= 1e-9 * abs(want)
    # the reference's other dense baseline (util/gp_util.py:354-364, multivariate_normal.logpdf of the materialised covariance)
    lik_s, _ = gp_util.likelihood_pdf(gp_util.gram_matvec(), gp_util.logpdf_scipy_stats(), constrain=constrain)
    value_s, _ = gp_util.target_logml(gp_util.model_gp(m_fun, k_fun), lik_s)(T(X), T(y), **kw)
    assert abs(float(value_s) - want) <= 1e-9 * abs(want)
    # Krylov logpdf with k = n Lanczos steps and many probes is the same number up to Monte-Carlo error; with an
    # exact solve the Mahalanobis term is exact, so compare that part tightly via the info dict
    sample = hutchinson.sampler_rademacher(torch.empty(n, dtype=torch.float64, device=DEV), num=64)
    logdet = gp_util.krylov_logdet_slq(n, sample=sample, num_batches=1)
    lik, _ = gp_util.likelihood_pdf(gp_util.gram_matvec(), gp_util.logpdf_krylov(cg.cg_fixed_step(3 * n), logdet), constrain=constrain)
    value_k, info = gp_util.target_logml(gp_util.model_gp(m_fun, k_fun), lik)(T(X), T(y), 3, **kw)
    assert abs(float(value_k) - want) <= 0.05 * abs(want)
    assert float(torch.linalg.vector_norm(info["solve"]["residual_abs"])) < 1e-8


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-8), (torch.float32, 2e-3)])
@pytest.mark.parametrize("kernel_name", ["rbf", "matern32", "matern12"])
def test_target_posterior_mean(dtype, tol, kernel_name):
    """util/gp_util.py:35-45,279-351: posterior mean = m(xs) + K(xs, X) (K + noise I)^{-1} (y - m)."""
    rng = np.random.default_rng(6)
    n, m, d, rank = 700, 333, 3, 20
    X, Xs = rng.uniform(-1, 1, (n, d)), rng.uniform(-1, 1, (m, d))
    y = np.sin(X.sum(-1))
    raw = (np.array([0.1, 0.3, -0.2]), np.float64(0.3), np.float64(-2.0))
    make = {"rbf": gp_util.kernel_scaled_rbf, "matern32": gp_util.kernel_scaled_matern_32,
            "matern12": gp_util.kernel_scaled_matern_12}[kernel_name]
    k_fun, _ = make(shape_in=(d,), shape_out=())
    m_fun, _ = gp_util.mean_constant(shape_out=())
    constrain = gp_util.constraint_greater_than(1e-2)
    kw = dict(params_mean={"constant_value": T(0.3, dtype)},
              params_kernel={"raw_lengthscale": T(raw[0], dtype), "raw_outputscale": T(raw[1], dtype)},
              params_likelihood={"raw_noise": T(raw[2], dtype)})
    eps = float(torch.finfo(dtype).eps)
    oop = orc.RbfGramOp(X, noise_minval=1e-2, kernel=kernel_name, eps=eps)
    ls, s, noise = oop.constrained(*raw)
    A = lambda v: oop.apply(v, *raw)
    Kx = orc.kernel_matrix(kernel_name, Xs, X, ls, s, eps=eps)

    # few steps: on these fast-decaying spectra CG loses conjugacy within ~10 steps and two correct implementations then
    # differ at 1e-3 of the remaining error (the oracle does so against itself under 1e-16 perturbations)
    steps = 6
    lik, _ = gp_util.likelihood_condition(gp_util.gram_matvec(), cg.cg_fixed_step(steps), constrain=constrain)
    post, _ = gp_util.target_posterior(gp_util.model_gp(m_fun, k_fun), lik)(T(X, dtype), T(y, dtype), **kw)
    got, info = post(T(Xs, dtype))
    w, _ = orc.pcg_fixed_step(A, y - 0.3, num_matvecs=steps)
    want = 0.3 + Kx @ w
    assert got.shape == (m,) and "solve" in info
    assert np.allclose(N(got), want, rtol=tol, atol=tol * np.abs(want).max())

    K = orc.kernel_matrix(kernel_name, X, X, ls, s, diag_offset=0, eps=eps)
    L, _ = orc.cholesky_partial_pivot(lambda i, j: K[i, j], n, rank)
    P = lambda v: orc.precondition_solve(L, v, noise)
    lik, _ = gp_util.likelihood_condition_p(gp_util.gram_matvec(), cg.pcg_fixed_step(steps),
                                            precondition=low_rank.preconditioner(low_rank.cholesky_partial_pivot(rank=rank)),
                                            constrain=constrain)
    post, _ = gp_util.target_posterior(gp_util.model_gp(m_fun, k_fun), lik)(T(X, dtype), T(y, dtype), **kw)
    got, _ = post(T(Xs, dtype))
    w, _ = orc.pcg_fixed_step(A, y - 0.3, P, num_matvecs=steps)
    want = 0.3 + Kx @ w
    assert np.allclose(N(got), want, rtol=tol, atol=tol * np.abs(want).max())
    # converged adaptive solve against the dense posterior mean
    lik, _ = gp_util.likelihood_condition_p(gp_util.gram_matvec(precision="fp32"),
                                            cg.pcg_adaptive(atol=1e-6, rtol=0.0, maxiter=2000, miniter=1),
                                            precondition=low_rank.preconditioner(low_rank.cholesky_partial_pivot(rank=rank)),
                                            constrain=constrain)
    post, _ = gp_util.target_posterior(gp_util.model_gp(m_fun, k_fun), lik)(T(X, dtype), T(y, dtype), **kw)
    got, info = post(T(Xs, dtype))
    dense = 0.3 + Kx @ np.linalg.solve(K + noise * np.eye(n), y - 0.3)
    assert int(info["solve"]["num_steps"]) < 2000
    assert np.allclose(N(got), dense, atol=1e-4 if dtype == torch.float64 else 5e-3)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-3)])
def test_cg_fixed_step_reortho_matches_oracle(dtype, tol):
    A = orc.symmetric_matrix_from_eigenvalues(np.arange(1.0, 10.0))
    b = np.arange(1.0, 10.0)
    x, info = cg.cg_fixed_step_reortho(len(A))(DenseOp().bind(T(A, dtype)), T(b, dtype))  # test_cg.py:32-40
    assert np.allclose(N(x), np.linalg.solve(A, b), rtol=max(tol, 1e-8), atol=max(tol, 1e-8))
    m = len(A) // 2
    want, winfo = orc.pcg_fixed_step_reortho(lambda v: A @ v, b, num_matvecs=m)
    x, info = cg.cg_fixed_step_reortho(m)(DenseOp().bind(T(A, dtype)), T(b, dtype))
    assert info["Q"].shape == (len(A), m)
    assert np.allclose(N(x), want, rtol=tol, atol=tol * np.abs(want).max())
    assert np.allclose(N(info["Q"]), winfo["Q"], atol=10 * tol)
    assert np.allclose(N(info["Q"].T @ info["Q"]), np.eye(m), atol=max(10 * tol, 1e-8))
    # re-orthogonalisation helps on an ill-conditioned matrix (test_cg.py:102-116)
    A = orc.symmetric_matrix_from_eigenvalues(1.5 ** np.arange(-20.0, 20.0))
    b = np.arange(1.0, 1.0 + len(A))
    if dtype == torch.float64:
        _x, i0 = cg.cg_fixed_step(len(A) // 2)(DenseOp().bind(T(A)), T(b))
        _x, i1 = cg.cg_fixed_step_reortho(len(A) // 2)(DenseOp().bind(T(A)), T(b))
        assert float(torch.linalg.vector_norm(i1["residual_abs"])) < 0.9 * float(torch.linalg.vector_norm(i0["residual_abs"]))
    # with the low-rank preconditioner, against the oracle (short run: see the note on CG parity in DESIGN.md)
    rng = np.random.default_rng(8)
    n, rank = 300, 10
    X = rng.uniform(-1, 1, (n, 2))
    raw = (np.float64(0.3), np.float64(0.2), np.float64(-2.0))
    oop = orc.RbfGramOp(X, noise_minval=1e-3)
    ls, s, noise = oop.constrained(*raw)
    K = orc.kernel_matrix("rbf", X, X, ls, s, diag_offset=0)
    L, _ = orc.cholesky_partial_pivot(lambda i, j: K[i, j], n, rank)
    bound = RbfGramOp(T(X, dtype), noise_minval=1e-3).bind(*(T(q, dtype) for q in raw))
    pre, _ = low_rank.preconditioner(low_rank.cholesky_partial_pivot(rank=rank))(low_rank.without_noise(bound), n)
    bb = rng.standard_normal((2, n))
    x, info = cg.pcg_fixed_step_reortho(5)(bound, T(bb, dtype), pre.bind(noise))
    for i in range(2):
        want, winfo = orc.pcg_fixed_step_reortho(lambda v: K @ v + noise * v, bb[i], lambda v: orc.precondition_solve(L, v, noise),
                                                 num_matvecs=5)
        assert np.allclose(N(x[i]), want, rtol=max(tol, 1e-7), atol=max(tol, 1e-7) * np.abs(want).max())
        assert np.allclose(N(info["Q"][i]), winfo["Q"], atol=max(10 * tol, 1e-7) * np.abs(winfo["Q"]).max())


def test_bnn_lanczos_samplers():
    """util/bnn_util.py:372-409 against a numpy restatement on the oracle's Lanczos."""
    from matfree_extensions.util import bnn_util

    rng = np.random.default_rng(9)
    n, num, k = 30, 4, 8
    ggn = orc.symmetric_matrix_from_eigenvalues(rng.uniform(0.5, 3.0, n), seed=3)
    variables = rng.standard_normal(n)
    eps = rng.standard_normal((num, n))
    sample = bnn_util.sampler_lanczos(ggn_fun=lambda *_a: T(ggn), num=num, lanczos_rank=k)
    got = N(sample(T(eps), None, T(variables), None, None))
    got2 = N(bnn_util.lanczos_sampler(ggn_vp=DenseOp().bind(T(ggn)), num_samples=num, lanczos_rank=k, key=T(eps), params_vec=T(variables)))
    for b in range(num):
        (Q, (dg, off)), _ = orc.tridiag_full(orc.DenseOp(), k, eps[b], ggn)
        Tm = np.diag(dg) + np.diag(off, 1) + np.diag(off, -1)
        want = Q.T @ (np.linalg.cholesky(np.linalg.inv(Tm)) @ (Q @ eps[b])) + variables
        assert np.allclose(got[b], want, rtol=1e-8, atol=1e-8)
        w, v = np.linalg.eigh(Tm)
        want2 = variables + (Q.T @ v) @ (np.sqrt(1 / w) * eps[b][:k])
        # eigenvector signs are arbitrary: compare through the sign-invariant form
        got_sign = N(T(got2[b] - variables))
        proj = (Q.T @ v).T @ got_sign  # coefficients in the Ritz basis, up to sign
        assert np.allclose(np.abs(proj), np.abs(np.sqrt(1 / w) * eps[b][:k]), rtol=1e-7, atol=1e-9)
    full = bnn_util.sampler_cholesky(ggn_fun=lambda *_a: T(ggn), num=num)
    assert N(full(T(eps), None, T(variables), None, None)).shape == (num, n)


@pytest.mark.parametrize("kind", ["rbf", "matern32"])
def test_gp_logml_golden_fixture(kind):
    """fp64 build against the committed fixture tests/golden/gp_logml_n256.npz: partial Cholesky, preconditioner, PCG (fixed and
    adaptive) and value + gradient of the composed log-marginal likelihood."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gp_logml_n256.npz"))
    X, y = g["X"], g["y"]
    n, rank, steps, k, nprobes, seed = len(y), int(g["rank"]), int(g["steps"]), int(g["k"]), int(g["nprobes"]), int(g["seed"])
    minval, cval = float(g["minval"]), float(g["cval"])
    tl, ts, tn = T(g["raw_l"], grad=True), T(g["raw_s"], grad=True), T(g["raw_n"], grad=True)
    tc, ty = T(cval, grad=True), T(y, grad=True)
    bound = RbfGramOp(T(X), noise_minval=minval, kernel=kind).bind(tl.detach(), ts.detach(), tn.detach())
    noise = minval + float(torch.nn.functional.softplus(tn.detach()))
    chol, info = low_rank.cholesky_partial_pivot(rank=rank)(low_rank.without_noise(bound), n)
    assert np.array_equal(N(info["pivots"]).astype(int), g[kind + "_pivots"])
    assert np.allclose(N(chol), g[kind + "_L"], rtol=1e-8, atol=1e-10)
    pre, _ = low_rank.preconditioner(low_rank.cholesky_partial_pivot(rank=rank))(low_rank.without_noise(bound), n)
    b = T(y - cval)
    assert np.allclose(N(pre(b, noise)), g[kind + "_precond_b"], rtol=1e-8, atol=1e-10)
    x, sinfo = cg.pcg_fixed_step(steps)(bound, b, pre.bind(noise))
    assert np.allclose(N(x), g[kind + "_x_pcg"], rtol=1e-7, atol=1e-9)
    assert np.allclose(N(sinfo["residual_abs"]), g[kind + "_r_pcg"], atol=1e-8)
    x, _ = cg.cg_fixed_step(steps)(bound, b)
    assert np.allclose(N(x), g[kind + "_x_cg"], rtol=1e-7, atol=1e-9)
    x, ainfo = cg.pcg_adaptive(atol=1e-3, rtol=0.0, maxiter=100, miniter=2)(bound, b, pre.bind(noise))
    assert int(ainfo["num_steps"]) == int(g[kind + "_steps_adaptive"])
    assert np.allclose(N(x), g[kind + "_x_adaptive"], rtol=1e-6, atol=1e-8)

    make_kernel = {"rbf": gp_util.kernel_scaled_rbf, "matern32": gp_util.kernel_scaled_matern_32}[kind]
    k_fun, _ = make_kernel(shape_in=(X.shape[1],), shape_out=())
    m_fun, _ = gp_util.mean_constant(shape_out=())
    sample = hutchinson.sampler_rademacher(torch.empty(n, dtype=torch.float64, device=DEV), num=nprobes)
    logpdf_p = gp_util.logpdf_krylov_p(solve_p=cg.pcg_fixed_step(steps), logdet=gp_util.krylov_logdet_slq(k, sample=sample, num_batches=1))
    likelihood, _ = gp_util.likelihood_pdf_p(gp_util.gram_matvec(), logpdf_p,
                                             low_rank.preconditioner(low_rank.cholesky_partial_pivot(rank=rank)),
                                             constrain=gp_util.constraint_greater_than(minval))
    value, _ = gp_util.target_logml(gp_util.model_gp(m_fun, k_fun), likelihood)(
        T(X), ty, seed, params_mean={"constant_value": tc}, params_kernel={"raw_lengthscale": tl, "raw_outputscale": ts},
        params_likelihood={"raw_noise": tn})
    value.backward()
    assert abs(float(value.detach()) - float(g[kind + "_logml"])) <= 1e-8 * abs(float(g[kind + "_logml"]))
    for got, key in ((tl, "_g_l"), (ts, "_g_s"), (tn, "_g_n"), (ty, "_g_y"), (tc, "_g_c")):
        want = g[kind + key]
        assert np.allclose(N(got.grad), want, rtol=1e-6, atol=1e-7 * max(1.0, float(np.abs(want).max()))), key
