"""GPU parity: the HIP path (through the C-ABI of libmfx.so) against the CPU oracle and the committed
golden fixtures.  Tolerances: fp64 build vs fp64 oracle rtol 1e-9 (forward) / 1e-7 (gradients);
fp32 build vs fp64 oracle value rtol 1e-4, gradients rtol 2e-3 of the gradient's max-norm
(the reference's own fp32 tolerance is sqrt(eps) ~ 3.5e-4, test_integrand_spd_value_and_grad.py:36-38).
"""

import os

import numpy as np
import pytest
import torch

from oracle import slq_oracle as orc

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from matfree_extensions import _lib, arnoldi, hutchinson, lanczos
    from matfree_extensions.operators import CsrOp, DenseOp, RbfGramOp
    from matfree_extensions.util import gp_util

DEV = torch.device("cuda:0")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def T(x, dtype=torch.float64, grad=False):
    t = torch.tensor(np.asarray(x), dtype=dtype, device=DEV)
    return t.requires_grad_(True) if grad else t


def N(t):
    return t.detach().cpu().numpy().astype(np.float64)


def close(a, b, rtol, atol_rel=None):
    a, b = N(a) if torch.is_tensor(a) else np.asarray(a), np.asarray(b)
    atol = (atol_rel if atol_rel is not None else rtol) * max(np.abs(b).max(), 1e-300)
    ok = np.allclose(a, b, rtol=rtol, atol=atol)
    if not ok:
        print("max abs err", np.abs(a - b).max(), "scale", np.abs(b).max())
    return ok


# ------------------------------------------------------------------------------------------------
# operators: apply / transpose / parameter sweep
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
@pytest.mark.parametrize("p", [1, 3, 9])
def test_dense_op(dtype, tol, p):
    rng = np.random.default_rng(0)
    n = 77
    A, V, Cc = rng.standard_normal((n, n)), rng.standard_normal((p, n)), rng.standard_normal((p, n))
    At, Vt = T(A, dtype, True), T(V, dtype, True)
    y = DenseOp()(Vt, At)
    assert close(y, V @ A.T, tol)
    gV, gA = torch.autograd.grad(y, (Vt, At), T(Cc, dtype))
    assert close(gV, Cc @ A, tol)
    assert close(gA, Cc.T @ V, tol)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
def test_csr_op(dtype, tol):
    r, c, vals, n = orc.laplacian_2d_plus_identity(13)
    rng = np.random.default_rng(1)
    vals = vals + 0.1 * rng.standard_normal(vals.shape)  # non-symmetric values: exercises A^T
    op, v_dev, order = CsrOp.from_coo(r, c, vals, n, DEV)
    o = orc.CooOp(r, c, n)
    V, Cc = rng.standard_normal((3, n)), rng.standard_normal((3, n))
    vt, Vt = v_dev.to(dtype).requires_grad_(True), T(V, dtype, True)
    y = op(Vt, vt)
    assert close(y, np.stack([o.apply(v, vals) for v in V]), tol)
    gV, gvals = torch.autograd.grad(y, (Vt, vt), T(Cc, dtype))
    assert close(gV, np.stack([o.apply_t(cc, vals) for cc in Cc]), tol)
    ref = sum(o.param_vjp(v, cc, vals)[0] for v, cc in zip(V, Cc))
    assert close(gvals, ref[order.numpy()], tol)


@pytest.mark.parametrize("skew", ["dense_row_and_column", "short_rows"])
def test_csr_skewed_rows_pick_the_step_kernel_by_the_longest_row(skew):
    """One dense row and one dense column over a banded matrix (mean row length 5, longest 1500): libmfx decides fused-step vs
    8-lanes-per-row kernels on `mfx_operator.max_row_nnz` (mean-based, this matrix would serialise a whole row in one lane);
    the banded matrix alone stays on the fused step.  Arnoldi forward + adjoint and the three-term pair against the oracle, fp64."""
    n, k = 1500, 9
    rng = np.random.default_rng(8)
    rows, cols = [np.arange(n)] * 3, [np.arange(n), (np.arange(n) + 1) % n, (np.arange(n) - 1) % n]
    if skew == "dense_row_and_column":
        rows += [np.full(n, 700), np.arange(n)]
        cols += [np.arange(n), np.full(n, 31)]
    r, c = np.concatenate(rows), np.concatenate(cols)
    key = np.unique(r * n + c)
    r, c = key // n, key % n
    vals = rng.standard_normal(len(r)) * 0.2 + np.where(r == c, 4.0, 0.0)
    op, v_dev, order = CsrOp.from_coo(r, c, vals, n, DEV)
    assert (op.max_row_nnz > 64) == (skew == "dense_row_and_column")
    o = orc.CooOp(r, c, n)
    v = rng.standard_normal(n)
    wQ, wH = rng.standard_normal((n, k)), rng.standard_normal((k, k))
    vt, pt = T(v, grad=True), v_dev.to(torch.float64).requires_grad_(True)
    Q, H, rr, cc = arnoldi.hessenberg(op, k, reortho="full")(vt, pt)
    ((Q * T(wQ)).sum() + (H * T(wH)).sum()).backward()
    Qo, Ho, ro, co = orc.arnoldi_forward(o, k, v, vals, reortho="full")
    assert close(Q, Qo, 1e-10) and close(H, Ho, 1e-10)
    dv, dp = orc.arnoldi_adjoint(o, (vals,), Q=Qo, H=Ho, r=ro, c=co, dQ=wQ, dH=wH, dr=np.zeros(n), dc=0.0, reortho="full")
    assert close(vt.grad, dv, 1e-8) and close(pt.grad, dp[0][order.numpy()], 1e-8)
    # three-term recurrence on the symmetrised values (same structure: the pattern is symmetric by construction of the band;
    # the dense row / column make it non-symmetric, which the forward does not mind)
    (xs, (al, be)), _ = lanczos.tridiag(op, k, reortho="none")(T(v), v_dev.to(torch.float64))
    (xo, (ao, bo)), _ = orc.tridiag(o, k, v, vals, reortho="none")
    assert close(al, ao, 1e-9) and close(be, bo, 1e-9) and close(xs, xo, 1e-8)


@pytest.mark.parametrize("dtype,tol,precision", [(torch.float64, 1e-11, "fp32"), (torch.float32, 5e-5, "fp32"),
                                                 (torch.float32, 5e-5, "f16x3-matvec"), (torch.float32, 5e-5, "f16x3")])
@pytest.mark.parametrize("kernel", ["rbf", "matern32", "matern12"])
@pytest.mark.parametrize("ard", [False, True])
@pytest.mark.parametrize("n,d,p", [(300, 3, 1), (515, 8, 5), (700, 9, 8), (640, 8, 64), (333, 5, 17), (1000, 8, 40),
                                   # d <= 4 ON the matrix-core gradient kernels (batch > 32 or n >= 2048): the register epilogue of the split
                                   # GEMM with its 2-MFMA distance chain -- wrong by O(1) in rounds 2-4 for non-ARD RBF (an inline-asm v_exp
                                   # read the distance block before the MFMA had written it), never reached by the small d = 3 case above
                                   (2100, 2, 40), (2304, 4, 3), (700, 3, 33)])
def test_rbf_op_apply_and_param_sweep(dtype, tol, precision, ard, n, d, p, kernel):
    """p >= 4 in fp32 takes the MFMA kernels (exact fp32 or the 3 x f16 split), everything else the VALU kernel;
    kernels: util/gp_util.py:69-184 (scaled RBF, Matern-3/2, Matern-1/2)."""
    rng = np.random.default_rng(2)
    X = rng.standard_normal((n, d))
    raw = (rng.standard_normal(d) * 0.3 + 0.5 if ard else np.array(0.7), np.array(0.4), np.array(-1.0))
    V, Cc = rng.standard_normal((p, n)), rng.standard_normal((p, n))
    # Matern kernels add eps(compute dtype) under the square root (util/gp_util.py:99,140): give the fp64 oracle the
    # eps of the dtype under test, otherwise K_ii differs by sqrt(eps_fp32) = 3.5e-4 by definition
    o = orc.RbfGramOp(X, noise_minval=1e-4, kernel=kernel, eps=float(torch.finfo(dtype).eps))
    op = RbfGramOp(T(X, dtype), noise_minval=1e-4, precision=precision, kernel=kernel)
    params = [T(r, dtype, True) for r in raw]
    Vt = T(V, dtype, True)
    y = op(Vt, *params)
    assert close(y, o.apply(V, *raw), tol)
    grads = torch.autograd.grad(y, (Vt, *params), T(Cc, dtype))
    assert close(grads[0], o.apply(Cc, *raw), tol)
    ref = o.param_vjp(V, Cc, *raw)
    gtol = tol * (50 if dtype == torch.float32 else 10)
    for g, rr in zip(grads[1:], ref):
        assert close(g.reshape(np.shape(rr)), rr, gtol, atol_rel=gtol * np.sqrt(n))


@pytest.mark.parametrize("dtype,tol,precision", [(torch.float64, 1e-11, "fp32"), (torch.float32, 5e-5, "fp32"), (torch.float32, 5e-5, "f16x3")])
@pytest.mark.parametrize("kernel", ["rbf", "matern32", "matern12"])
@pytest.mark.parametrize("ard", [False, True])
@pytest.mark.parametrize("n,d,p", [(300, 33, 3), (515, 90, 8), (260, 385, 5), (700, 64, 1), (257, 100, 11),
                                   # 16 < d <= 128 in fp32 with >= 4 vectors (or n >= 2048): the exact-fp32 matrix-core matvec in EVERY mode;
                                   # d <= 32 and batch >= 16 (or n >= 2048): the exact-fp32 matrix-core sweep as well
                                   (600, 20, 8), (2304, 32, 3), (900, 27, 40), (520, 50, 8), (2100, 64, 5), (1000, 40, 70), (777, 17, 33),
                                   (2100, 128, 5), (640, 97, 70), (300, 129, 8)])
def test_rbf_wide_inputs_apply_and_param_sweep(dtype, tol, precision, ard, n, d, p, kernel):
    """d > 32: the reference's kernels take any input dimension (util/gp_util.py:151-184) and its UCI loaders reach d = 90 (song) and
    385 (slice) (util/uci_util.py:85-99,303-310).  The wide kernels of csrc/mfx_ops.hip (distance as a small GEMM over chunks of the d
    axis; with ARD the sweep's grid selects 32 lengthscale derivatives at a time) against the NumPy oracle: matvec, its transpose
    through autograd, all parameter gradients (d of them with ARD), the cross-covariance matvec of the posterior mean.  Every mode
    runs the same arithmetic here: VALU for d > 128 (and for fp64, few vectors at small n, the cross-covariance), the exact-fp32 matrix-core
    kernels for 16 < d <= 128 (sweep: <= 32) -- the split kernels stop at d = 16.  Inputs scaled by 1 / sqrt(d): distances O(1)."""
    rng = np.random.default_rng(d)
    X = rng.standard_normal((n, d)) / np.sqrt(d) * 1.5
    raw = (rng.standard_normal(d) * 0.3 + 0.5 if ard else np.array(0.7), np.array(0.4), np.array(-1.0))
    V, Cc = rng.standard_normal((p, n)), rng.standard_normal((p, n))
    o = orc.RbfGramOp(X, noise_minval=1e-4, kernel=kernel, eps=float(torch.finfo(dtype).eps))
    op = RbfGramOp(T(X, dtype), noise_minval=1e-4, precision=precision, kernel=kernel)
    params = [T(r, dtype, True) for r in raw]
    Vt = T(V, dtype, True)
    y = op(Vt, *params)
    assert close(y, o.apply(V, *raw), tol)
    grads = torch.autograd.grad(y, (Vt, *params), T(Cc, dtype))
    assert close(grads[0], o.apply(Cc, *raw), tol)
    ref = o.param_vjp(V, Cc, *raw)
    gtol = tol * (50 if dtype == torch.float32 else 10)
    for g, rr in zip(grads[1:], ref):
        assert close(g.reshape(np.shape(rr)), rr, gtol, atol_rel=gtol * np.sqrt(n))
    # K(X_new, X) v without the noise term (util/gp_util.py:299-305)
    m = 70
    Xn = rng.standard_normal((m, d)) / np.sqrt(d) * 1.5
    ls, s = orc.softplus(np.asarray(raw[0], dtype=np.float64)), orc.softplus(np.float64(raw[1]))
    Kx = orc.kernel_matrix(kernel, Xn, X, ls, s, eps=float(torch.finfo(dtype).eps))
    with torch.no_grad():
        got = op.cross_apply(T(Xn, dtype), T(V, dtype), *[q.detach() for q in params])
    assert close(got, V @ Kx.T, tol)


@pytest.mark.parametrize("decay,batch", [(0.0, 96), (9.0, 96), (9.0, 200), (30.0, 77)])
def test_split_param_sweep_with_rows_of_very_different_size(decay, batch):
    """The split gradient GEMM packs the batch rows by decreasing size and multiplies the tail -- the smallest rows whose bounds add
    up to <= 2^-10 of the sum of all bounds -- hi hi only (csrc/mfx_rbf_mfma.hip, k_order_rows).  Rows decaying over 9 and 30 orders of
    magnitude (the adjoint states of a long Krylov recursion), in scrambled order, ragged batch sizes, against the fp64 oracle: the
    parameter gradients must be as accurate as with rows of equal size (decay 0: no tail, only the ordering).  The sweep being summed:
    arnoldi.py:207-209 over all (probe, step) pairs."""
    n, d = 2304, 8
    rng = np.random.default_rng(11)
    X = rng.standard_normal((n, d))
    raw = (np.array(0.7), np.array(0.4), np.array(-1.0))
    scale = 10.0 ** (-decay * rng.permutation(batch) / batch)
    L = rng.standard_normal((batch, n)) * scale[:, None]
    R = rng.standard_normal((batch, n))
    o = orc.RbfGramOp(X, noise_minval=1e-4)
    ref = o.param_vjp(R, L, *raw)  # d/dtheta sum_b L_b^T A(theta) R_b

    def sweep(precision):
        op = RbfGramOp(T(X, torch.float32), noise_minval=1e-4, precision=precision)
        params = [T(r, torch.float32, True) for r in raw]
        y = op(T(R, torch.float32), *params)
        return [float(g) for g in torch.autograd.grad(y, params, T(L, torch.float32))]

    split, exact = sweep("f16x3"), sweep("f16x3-matvec")  # the latter: the exact-fp32 MFMA gradient GEMM, no ordering, no tail
    for gs, ge, rr in zip(split, exact, ref):
        rr = float(rr)
        # as accurate as the exact-fp32 GEMM up to a small factor, with a floor at 2e-5 of the value (both sums cancel heavily)
        assert abs(gs - rr) <= 3.0 * abs(ge - rr) + 2e-5 * abs(rr), (decay, batch, gs, ge, rr)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 5e-5)])
@pytest.mark.parametrize("kernel", ["rbf", "matern32"])
@pytest.mark.parametrize("n,d,p", [(2100, 14, 2), (2050, 16, 33), (600, 20, 6), (4097, 3, 1), (2304, 12, 65)])
def test_rbf_op_dispatch_corners(dtype, tol, kernel, n, d, p):
    """Corners of the Gram-kernel dispatch: d = 13..16 (four f16 distance MFMAs per block), 1-3 right-hand sides on the matrix-core
    kernel (n >= 2048), d > 16 (VALU kernel), two probe chunks (p > 64), ragged n."""
    rng = np.random.default_rng(7)
    X = rng.standard_normal((n, d))
    raw = (rng.standard_normal(d) * 0.2 + 1.0, np.array(0.4), np.array(-1.0))
    V, Cc = rng.standard_normal((p, n)), rng.standard_normal((p, n))
    o = orc.RbfGramOp(X, noise_minval=1e-4, kernel=kernel, eps=float(torch.finfo(dtype).eps))
    op = RbfGramOp(T(X, dtype), noise_minval=1e-4, kernel=kernel)
    params = [T(r, dtype, True) for r in raw]
    Vt = T(V, dtype, True)
    y = op(Vt, *params)
    assert close(y, o.apply(V, *raw), tol)
    grads = torch.autograd.grad(y, (Vt, *params), T(Cc, dtype))
    assert close(grads[0], o.apply(Cc, *raw), tol)
    ref = o.param_vjp(V, Cc, *raw)
    gtol = tol * (50 if dtype == torch.float32 else 10)
    for g, rr in zip(grads[1:], ref):
        assert close(g.reshape(np.shape(rr)), rr, gtol, atol_rel=gtol * np.sqrt(n))


# ------------------------------------------------------------------------------------------------
# small dense pieces
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 3e-5)])
@pytest.mark.parametrize("k", [1, 2, 7, 40, 90])
def test_tridiag_eigh_and_quadform_backward(dtype, tol, k):
    rng = np.random.default_rng(k)
    p = 5
    d = rng.uniform(1.0, 3.0, size=(p, k))
    e = rng.uniform(0.1, 0.5, size=(p, max(k - 1, 0)))
    dt, et = T(d, dtype, True), T(e, dtype, True)
    val, evals, evecs = lanczos._QuadformFn.apply(torch.log, dt, et)
    gd, ge = torch.autograd.grad(val.sum(), (dt, et), allow_unused=True)
    for b in range(p):
        ref, (rd, re), (lam, U) = orc.quadform_from_tridiag(d[b], e[b], "log")
        assert close(val[b], ref, tol)
        assert close(np.sort(N(evals[b])), lam, tol)
        Tm = orc.dense_tridiag(d[b], e[b])
        Ub = N(evecs[b])
        assert np.allclose(Ub.T @ Ub, np.eye(k), atol=tol * 10)
        assert np.allclose(Tm @ Ub, Ub * N(evals[b])[None, :], atol=tol * 30)
        assert close(gd[b], rd, tol * 20, atol_rel=tol * 20)
        if k > 1:
            assert close(ge[b], re, tol * 20, atol_rel=tol * 20)


def test_rademacher_bit_exact_and_shardable():
    x_like = torch.empty(1000, dtype=torch.float32, device=DEV)
    full = hutchinson.sampler_rademacher(x_like, num=6)(11)
    assert np.array_equal(N(full), orc.rademacher(11, 6, 1000))
    part = hutchinson.sampler_rademacher(x_like, num=2)((11, 4))
    assert np.array_equal(N(part), orc.rademacher(11, 2, 1000, first_probe=4))
    explicit = torch.ones(3, 1000, device=DEV)
    assert hutchinson.sampler_rademacher(x_like, num=3)(explicit) is explicit


# ------------------------------------------------------------------------------------------------
# Arnoldi forward / adjoint (reference: tests/test_arnoldi/*)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-4)])
@pytest.mark.parametrize("k", [1, 5, 10])
@pytest.mark.parametrize("reortho", ["none", "full"])
def test_hessenberg_forward_identities_and_oracle(dtype, tol, k, reortho):
    n = 10
    rng = np.random.default_rng(1)
    A, v = rng.standard_normal((n, n)), rng.standard_normal(n)
    Q, H, r, c = arnoldi.hessenberg(DenseOp(), k, reortho=reortho)(T(v, dtype), T(A, dtype))
    assert Q.shape == (n, k) and H.shape == (k, k) and r.shape == (n,) and c.shape == ()
    Qn, Hn, rn, cn = N(Q), N(H), N(r), N(c)
    eK = np.eye(k)[-1]
    assert np.allclose(A @ Qn - Qn @ Hn - np.outer(rn, eK), 0.0, atol=tol * 10)
    assert np.allclose(Qn.T @ Qn, np.eye(k), atol=tol * 10)
    assert np.allclose(Qn[:, 0], cn * v, atol=tol)
    Qo, Ho, ro, co = orc.arnoldi_forward(orc.DenseOp(), k, v, A, reortho=reortho)
    assert close(Q, Qo, tol * 10) and close(H, Ho, tol * 10)
    assert np.allclose(rn, ro, atol=tol * 100)  # at k = n the remainder is round-off only
    assert close(c, co, tol)


@pytest.mark.parametrize("dtype,tol", [(torch.complex128, 1e-10), (torch.complex64, 1e-4)])
@pytest.mark.parametrize("k", [1, 5, 10])
@pytest.mark.parametrize("reortho", ["none", "full"])
@pytest.mark.parametrize("kind", ["native", "callable"])
def test_hessenberg_forward_complex(dtype, tol, k, reortho, kind):
    """tests/test_arnoldi/test_hessenberg_forward.py:10-37 with dtype=complex: the decomposition identities and the oracle; the
    dense operator on the native kernel (real form of the complex matrix) and the reference's own `lambda s, p: p @ s`."""
    n = 10
    rng = np.random.default_rng(1)
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    matvec = DenseOp() if kind == "native" else (lambda s, p: p @ s)
    At, vt = torch.tensor(A, dtype=dtype, device=DEV), torch.tensor(v, dtype=dtype, device=DEV)
    Q, H, r, c = arnoldi.hessenberg(matvec, k, reortho=reortho)(vt, At)
    assert Q.shape == (n, k) and H.shape == (k, k) and r.shape == (n,) and c.shape == ()
    assert Q.dtype == H.dtype == r.dtype == c.dtype == dtype
    Qn, Hn, rn, cn = (t.cpu().numpy().astype(np.complex128) for t in (Q, H, r, c))
    eK = np.eye(k)[-1]
    assert np.allclose(A @ Qn - Qn @ Hn - np.outer(rn, eK), 0.0, atol=tol * 10)
    assert np.allclose(Qn.T.conj() @ Qn, np.eye(k), atol=tol * 10)
    assert np.allclose(Qn[:, 0], cn * v, atol=tol)
    Qo, Ho, ro, co = orc.arnoldi_forward(orc.DenseOp(), k, v, A, reortho=reortho)
    assert np.allclose(Qn, Qo, atol=tol * 10) and np.allclose(Hn, Ho, atol=tol * 10 * np.abs(Ho).max())
    assert np.allclose(rn, ro, atol=tol * 100) and np.allclose(cn, co, atol=tol)


def test_hessenberg_complex_batched_larger_and_limits():
    """A batch of complex start vectors at a size where the vector kernels run several slices; forward only."""
    n, k, p = 3000, 12, 3
    rng = np.random.default_rng(2)
    A = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n)
    V = rng.standard_normal((p, n)) + 1j * rng.standard_normal((p, n))
    At, Vt = torch.tensor(A, dtype=torch.complex128, device=DEV), torch.tensor(V, dtype=torch.complex128, device=DEV)
    Q, H, r, c = arnoldi.hessenberg(DenseOp(), k, reortho="full")(Vt, At)
    assert Q.shape == (p, n, k) and H.shape == (p, k, k) and r.shape == (p, n) and c.shape == (p,)
    for b in range(p):
        Qo, Ho, ro, co = orc.arnoldi_forward(orc.DenseOp(), k, V[b], A, reortho="full")
        assert np.allclose(Q[b].cpu().numpy(), Qo, atol=1e-10) and np.allclose(H[b].cpu().numpy(), Ho, atol=1e-10)
        assert np.allclose(r[b].cpu().numpy(), ro, atol=1e-10) and np.allclose(c[b].cpu().numpy(), co, atol=1e-12)
    with pytest.raises(NotImplementedError, match="forward only"):
        arnoldi.hessenberg(DenseOp(), k, reortho="full")(Vt, At.clone().requires_grad_(True))
    with pytest.raises(ValueError, match="depth"):
        arnoldi.hessenberg(DenseOp(), n + 1, reortho="full")(Vt[0], At)


def test_hessenberg_batched_callable_and_errors():
    n, k = 12, 4
    rng = np.random.default_rng(5)
    A, V = rng.standard_normal((n, n)), rng.standard_normal((3, n))
    At = T(A)
    Qb, Hb, rb, cb = arnoldi.hessenberg(DenseOp(), k, reortho="full")(T(V), At)
    Qc, Hc, rc, cc = arnoldi.hessenberg(lambda s, p: p @ s, k, reortho="full")(T(V), At)  # Python callable path
    assert Qb.shape == (3, n, k) and Hb.shape == (3, k, k) and rb.shape == (3, n) and cb.shape == (3,)
    assert close(Qc, N(Qb), 1e-12) and close(Hc, N(Hb), 1e-12) and close(rc, N(rb), 1e-10, atol_rel=1e-10)
    for b in range(3):
        Qo, Ho, ro, co = orc.arnoldi_forward(orc.DenseOp(), k, V[b], A, reortho="full")
        assert close(Qb[b], Qo, 1e-10) and close(Hb[b], Ho, 1e-10)
    # single pass only when reortho_vjp="none" (quirk Q1; experiments/benchmarks/loss_of_orthogonality/measure.py:47-49)
    Q1, H1, _, _ = arnoldi.hessenberg(DenseOp(), k, reortho="none", reortho_vjp="none")(T(V[0]), At)
    Qo, Ho, _, _ = orc.arnoldi_forward(orc.DenseOp(), k, V[0], A, reortho="none", reortho_vjp="none")
    assert close(H1, Ho, 1e-10)
    for bad in (0, n + 1):
        with pytest.raises(ValueError, match="depth"):
            arnoldi.hessenberg(DenseOp(), bad, reortho="full")(T(V[0]), At)
    for bad in (True, "full_with_sparsity", "None"):
        with pytest.raises(TypeError, match="Unexpected input"):
            arnoldi.hessenberg(lambda s: s, 1, reortho=bad)
    with pytest.raises(ValueError, match="unsupported"):
        lanczos.tridiag(DenseOp(), 1, reortho="half")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("tag,reortho", [("rand3", "full"), ("rand3", "none"), ("rand10", "full"), ("rand10", "none"),
                                         ("hilbert15", "full")])
@pytest.mark.parametrize("kind", ["native", "callable"])
def test_arnoldi_adjoint_golden(dtype, tag, reortho, kind):
    """tests/test_arnoldi/test_hessenberg_adjoint.py: dense random cotangents over (Q, H, r, c)."""
    g = np.load(os.path.join(GOLD, "arnoldi_adjoint.npz"))
    pre = f"{tag}_{reortho}_"
    A, v = T(g[tag + "_A"], dtype, True), T(g[tag + "_v"], dtype, True)
    k = g[pre + "H"].shape[0]
    mv = DenseOp() if kind == "native" else (lambda s, p: p @ s)
    Q, H, r, c = arnoldi.hessenberg(mv, k, reortho=reortho)(v, A)
    ftol = 1e-9 if dtype == torch.float64 else 2e-4
    if tag != "hilbert15":
        assert close(Q, g[pre + "Q"], ftol) and close(H, g[pre + "H"], ftol)
    cot = [T(g[pre + s], dtype) for s in ("dQ", "dH", "dr", "dc")]
    dv, dA = torch.autograd.grad((Q, H, r, c), (v, A), cot)
    if dtype == torch.float64:
        gtol = 1e-7
    else:
        gtol = 5e-3 if tag != "hilbert15" else None  # fp32 on cond ~1e17 is meaningless (reference runs it in x64 only)
    if gtol is not None:
        assert close(dv, g[pre + "dv"], gtol, atol_rel=gtol)
        assert close(dA, g[pre + "dA"], gtol, atol_rel=gtol)


# ------------------------------------------------------------------------------------------------
# Lanczos tridiag forward / adjoint (reference: tests/test_lanczos/*)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 1e-4)])
@pytest.mark.parametrize("reortho", ["full", "none"])
@pytest.mark.parametrize("k", [1, 5, 11, 12])
def test_tridiag_forward_golden(dtype, tol, reortho, k):
    g = np.load(os.path.join(GOLD, "tridiag_forward_n12.npz"))
    A, v = g["A"], g["v"]
    (Q, (d, e)), (q, b) = lanczos.tridiag(DenseOp(), k, reortho=reortho)(T(v, dtype), T(A, dtype))
    Qn, Tm = N(Q), orc.dense_tridiag(N(d), N(e))
    if k == 12:
        rt = 1e-5 if reortho == "full" else 1e-1
        assert np.allclose(Qn.T @ Tm @ Qn, A, atol=rt, rtol=rt)
        assert np.allclose(Qn @ Qn.T, np.eye(12), atol=rt, rtol=rt)
    else:
        eK = np.eye(k)[-1]
        assert np.allclose(A @ Qn.T, Qn.T @ Tm + np.outer(eK, N(q) * N(b)).T, atol=1e-5, rtol=1e-5)
        pre = f"{reortho}_{k}_"
        loose = tol * (1 if reortho == "full" else 30)
        assert close(d, g[pre + "d"], loose) and close(e, g[pre + "e"], loose, atol_rel=loose) if k > 1 else True
        assert close(Q, g[pre + "Q"], loose * 10, atol_rel=loose * 10)
        assert close(b, g[pre + "b"], loose * 10)


@pytest.mark.parametrize("dtype,gtol", [(torch.float64, 1e-8), (torch.float32, 2e-3)])
@pytest.mark.parametrize("reortho", ["full", "none"])
@pytest.mark.parametrize("kind", ["native", "callable"])
def test_tridiag_adjoint_golden(dtype, gtol, reortho, kind):
    """tests/test_lanczos/test_tridiag_adjoint.py:12-50: random cotangents over ALL outputs."""
    g = np.load(os.path.join(GOLD, "tridiag_adjoint_n10.npz"))
    A, v = T(g["A"], dtype, True), T(g["v"], dtype, True)
    mv = DenseOp() if kind == "native" else (lambda s, p: p @ s)
    (Q, (d, e)), (q, b) = lanczos.tridiag(mv, 4, reortho=reortho)(v, A)
    pre = reortho + "_"
    cot = [T(g[pre + s], dtype) for s in ("dQ", "dd", "de", "dq", "db")]
    dv, dA = torch.autograd.grad((Q, d, e, q, b), (v, A), cot)
    assert close(dv, g[pre + "dv"], gtol, atol_rel=gtol)
    # Q4: the no-reortho adjoint returns sum x lambda^T, exact only on the symmetric subspace
    ref = g[pre + "dA"]
    if reortho == "none":
        dA, ref = 0.5 * (dA + dA.T), 0.5 * (ref + ref.T)
    assert close(dA, ref, gtol, atol_rel=gtol)


@pytest.mark.parametrize("dtype,gtol", [(torch.float64, 1e-8), (torch.float32, 5e-3)])
@pytest.mark.parametrize("reortho", ["full", "none"])
def test_csr_1138_bus_tridiag_and_adjoint(dtype, gtol, reortho):
    """BASELINE config 3 parity case: SuiteSparse 1138_bus, gradient w.r.t. ALL stored values
    (experiments/benchmarks/wall_times_vjp_through_lanczos_arnoldi/suite_sparse/benchmark.py:57-121)."""
    g = np.load(os.path.join(GOLD, "csr_1138_bus.npz"))
    n, k = g["v"].shape[0], int(g["k"])
    op, vals, order = CsrOp.from_coo(g["row"], g["col"], g["vals"], n, DEV)
    vals = vals.to(dtype).requires_grad_(True)
    v = T(g["v"], dtype, True)
    (Q, (d, e)), (q, b) = lanczos.tridiag(op, k, reortho=reortho)(v, vals)
    pre = reortho + "_"
    ftol = 1e-8 if dtype == torch.float64 else 5e-3
    if reortho == "full" or dtype == torch.float64:
        assert close(d, g[pre + "d"], ftol) and close(e, g[pre + "e"], ftol, atol_rel=ftol)
    cot = [T(g[pre + s], dtype) for s in ("dQ", "dd", "de", "dq", "db")]
    dv, dvals = torch.autograd.grad((Q, d, e, q, b), (v, vals), cot)
    if dtype == torch.float64:
        assert close(dv, g[pre + "dv"], gtol, atol_rel=gtol)
        ref = g[pre + "dvals"][order.numpy()]
        assert close(dvals, ref, gtol, atol_rel=gtol)
    else:  # fp32 on cond(1138_bus) ~ 1e7: finite and the right size (the reference benchmark only times fp32)
        assert torch.isfinite(dv).all() and torch.isfinite(dvals).all()


# ------------------------------------------------------------------------------------------------
# SLQ integrand + Hutchinson (reference: tests/test_lanczos/test_integrand_spd_value_and_grad.py)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,vtol,gtol", [(torch.float64, 1e-10, 1e-7), (torch.float32, 1e-4, 2e-3)])
@pytest.mark.parametrize("reortho", ["full", "none"])
def test_integrand_spd_dense_golden(dtype, vtol, gtol, reortho):
    g = np.load(os.path.join(GOLD, "slq_dense.npz"))
    A, v0 = T(g["A"], dtype, True), T(g["v0"], dtype, True)
    integrand = lanczos.integrand_spd(torch.log, 6, DenseOp(), reortho=reortho)
    val = integrand(v0, A)
    dv0, dA = torch.autograd.grad(val, (v0, A))
    assert close(val, g[f"{reortho}_value"], vtol)
    ref = g[f"{reortho}_dA"]
    if reortho == "none":
        dA, ref = 0.5 * (dA + dA.T), 0.5 * (ref + ref.T)
    assert close(dA, ref, gtol, atol_rel=gtol)
    assert close(dv0, g[f"{reortho}_dv0"], gtol, atol_rel=gtol)
    if reortho == "full":  # k = n: the quadrature is exact, v^T log(A) v
        full = lanczos.integrand_spd(torch.log, 11, DenseOp())(v0, A)
        assert close(full, g["full_depth_value"], vtol * 10)


def test_integrand_symmetric_parametrisation_callable_matches_oracle():
    """test_integrand_spd_value_and_grad.py:10-38 with matvec (p + p^T) @ x as a Python callable."""
    n, k = 10, 6
    A = orc.symmetric_matrix_from_eigenvalues(np.arange(0.0, 1.0 + n) + 1.0, seed=5)
    P = np.triu(A) - 0.5 * np.diag(np.diag(A))
    v0 = orc.rademacher(2, 1, n + 1)[0]
    for use_adjoints in (True,):
        integrand = lanczos.integrand_spd(torch.log, k, lambda x, p: (p + p.T) @ x, use_adjoints_for_tridiag=use_adjoints)
        Pt = T(P, grad=True)
        x_like = torch.ones(n + 1, dtype=torch.float64, device=DEV)
        estimate = hutchinson.hutchinson(integrand, hutchinson.sampler_rademacher(x_like, num=1))
        val = estimate(2, Pt)
        (gP,) = torch.autograd.grad(val, Pt)
        rv, _, (rP,) = orc.integrand_spd_value_and_grad(orc.DenseSymOp(), k, v0, (P,))
        small = np.sqrt(np.finfo(np.float32).eps)
        assert close(val, rv, small) and close(gP, rP, small, atol_rel=1e-9)


def test_c1_config_dense_512():
    """BASELINE config 1: 512 x 512 SPD (diag + rank-4), 20 Lanczos steps, 1 probe."""
    g = np.load(os.path.join(GOLD, "slq_dense.npz"))
    A = orc.spd_diag_plus_lowrank(512, 4, seed=0)
    probe = orc.rademacher(1, 1, 512)[0]
    for dtype, vtol, gtol in ((torch.float64, 1e-10, 1e-7), (torch.float32, 1e-4, 2e-3)):
        At = T(A, dtype, True)
        val = lanczos.integrand_spd(torch.log, 20, DenseOp())(T(probe, dtype), At)
        (dA,) = torch.autograd.grad(val, At)
        assert close(val, g["c1_value"], vtol)
        assert close(torch.diagonal(dA), g["c1_dA_diag"], gtol, atol_rel=gtol)
        assert close(dA[0], g["c1_dA_row0"], gtol, atol_rel=gtol)


@pytest.mark.parametrize("dtype,vtol,gtol,precision", [(torch.float64, 1e-9, 1e-6, "fp32"), (torch.float32, 1e-4, 2e-3, "fp32"),
                                                       (torch.float32, 1e-4, 2e-3, "f16x3")])
@pytest.mark.parametrize("tag", ["ard", "iso"])
def test_slq_rbf_golden(dtype, vtol, gtol, precision, tag):
    g = np.load(os.path.join(GOLD, "slq_rbf_n96.npz"))
    op = gp_util.gram_operator(T(g["X"], dtype), noise_minval=float(g["noise_minval"]), precision=precision)
    params = [T(g[f"{tag}_raw_l"], dtype, True), T(g["raw_s"], dtype, True), T(g["raw_n"], dtype, True)]
    integrand = lanczos.integrand_spd(torch.log, 8, op)
    probes = T(g["probes"], dtype)
    vals = integrand(probes, *params)
    assert close(vals, g[f"{tag}_values"], vtol)
    value = hutchinson.hutchinson(integrand, lambda key: key)(probes, *params)
    grads = torch.autograd.grad(value, params)
    assert close(value, g[f"{tag}_value"], vtol)
    for gr, name in zip(grads, ("g_l", "g_s", "g_n")):
        assert close(gr.reshape(g[f"{tag}_{name}"].shape), g[f"{tag}_{name}"], gtol, atol_rel=gtol)


@pytest.mark.parametrize("kernel", ["rbf", "matern32"])
@pytest.mark.parametrize("precision", ["fp32", "f16x3-matvec", "f16x3"])
@pytest.mark.parametrize("n,d,k,p", [(1536, 8, 12, 64), (1000, 9, 10, 24)])
def test_slq_rbf_mfma_path_downsized_c4_c2(n, d, k, p, precision, kernel):
    """Down-sized BASELINE configs 4 (d=8, 64 probes) and 2 (d=9): MFMA Gram matvec (exact fp32 and the
    3 x f16 split) inside the full SLQ value-and-gradient, against the fp64 oracle on identical explicit probes."""
    rng = np.random.default_rng(4)
    X = rng.standard_normal((n, d))
    ls = 2.0
    raw = (np.array(np.log(np.expm1(ls))), np.array(np.log(np.expm1(1.0))), np.array(np.log(np.expm1(0.1))))
    probes = orc.rademacher(5, p, n)
    ref_val, ref_g, ref_vals = orc.hutchinson_value_and_grad(
        orc.RbfGramOp(X, kernel=kernel, eps=float(np.finfo(np.float32).eps)), k, probes, raw)
    op = gp_util.gram_operator(T(X, torch.float32), precision=precision, kernel=kernel)
    params = [T(r, torch.float32, True) for r in raw]
    integrand = lanczos.integrand_spd(torch.log, k, op)
    vals = integrand(T(probes, torch.float32), *params)
    grads = torch.autograd.grad(vals.mean(), params)
    assert close(vals, ref_vals, 1e-4)
    for gr, rr in zip(grads, ref_g):
        assert close(gr.reshape(np.shape(rr)), rr, 2e-3, atol_rel=2e-3)


def test_reuse_integrand_and_logdet_helpers():
    n, k = 64, 10
    rng = np.random.default_rng(6)
    A = orc.spd_diag_plus_lowrank(n, 3, seed=1)
    At = T(A, grad=True)
    probes = T(orc.rademacher(3, 8, n))
    exact = lanczos.integrand_spd(torch.log, k, DenseOp())
    reuse = lanczos.integrand_spd_custom_vjp_reuse(torch.log, k, DenseOp())
    v1, v2 = exact(probes, At), reuse(probes, At)
    assert close(v2, N(v1), 1e-12)
    with pytest.warns(UserWarning):
        (g2,) = torch.autograd.grad(v2.sum(), At)
    for b in range(2):
        _, _, (rg,) = orc.integrand_spd_reuse_value_and_grad(orc.DenseOp(), k, N(probes[b]), (A,))
        (gb,) = torch.autograd.grad(reuse(probes[b], At), At)
        assert close(gb, rg, 1e-8, atol_rel=1e-8)
    # krylov_logdet_slq (util/gp_util.py:552-576): info dict keys and value
    sample = hutchinson.sampler_rademacher(torch.empty(n, dtype=torch.float64, device=DEV), num=16)
    value, info = gp_util.krylov_logdet_slq(k, sample=sample, num_batches=1, checkpoint=False)(DenseOp().bind(At), 3)
    assert set(info) == {"std", "std_rel"} and abs(value.item() - np.linalg.slogdet(A)[1]) < 0.05 * abs(np.linalg.slogdet(A)[1]) + 1.0
    value2, info2 = gp_util.krylov_logdet_slq(k, sample=sample, num_batches=3, checkpoint=True)(DenseOp().bind(At), 3)
    assert set(info2) == {"std_abs", "std_rel"}
    (gA,) = torch.autograd.grad(value2, At)
    assert close(gA.diagonal().sum(), np.trace(np.linalg.inv(A)), 0.2)


def test_hutchinson_variants():
    """tests/test_hutchinson.py:9-37: custom-vjp forward == plain forward, backward differs but agrees statistically."""
    n, k = 24, 6
    A = orc.spd_diag_plus_lowrank(n, 2, seed=2)
    At = T(A, grad=True)
    integrand = lanczos.integrand_spd(torch.log, k, DenseOp())
    sample = hutchinson.sampler_rademacher(torch.empty(n, dtype=torch.float64, device=DEV), num=512)
    plain = hutchinson.hutchinson(integrand, sample)
    nograd = hutchinson.hutchinson_nograd(integrand, sample)
    custom = hutchinson.hutchinson_custom_vjp(integrand, sample)
    v0, v1, v2 = plain(1, At), nograd(1, At), custom(1, At)
    assert v0.item() == v1.item() == v2.item()
    (g0,) = torch.autograd.grad(v0, At)
    (g2,) = torch.autograd.grad(v2, At)
    assert not torch.equal(g0, g2)
    assert close(g2, N(g0), 0.25, atol_rel=0.25)
    batch = hutchinson.hutchinson_batch(plain, num=4)(1, At)
    keys = hutchinson.split(1, 4)
    assert close(batch, np.mean([plain(kk, At).item() for kk in keys]), 1e-12)


def test_no_cpu_fallback():
    with pytest.raises(_lib.MfxError, match="no CPU fallback"):
        arnoldi.hessenberg(DenseOp(), 2, reortho="full")(torch.ones(4), torch.eye(4))


def test_error_paths_on_device():
    n = 16
    A = T(np.eye(n) * 2.0)
    v = T(np.ones(n))

    def boom(x, p):
        raise ZeroDivisionError("matvec failed on purpose")

    with pytest.raises(ZeroDivisionError, match="on purpose"):  # exceptions in Python matvecs cross the C loop intact
        arnoldi.hessenberg(boom, 3, reortho="full")(v, A)
    # plain tridiagonalisation has no depth limit (the SLQ quadrature's: 2048, next test)
    big = T(orc.spd_diag_plus_lowrank(256, 2, seed=0))
    (Q, (d, e)), _ = lanczos.tridiag(DenseOp(), 200, reortho="full")(T(np.ones(256) + np.arange(256) * 1e-3), big)
    assert Q.shape == (200, 256) and torch.isfinite(d).all()
    with pytest.raises(_lib.MfxError, match="needs fp64 buffers"):  # the C-ABI states it; the Python layer casts (next test)
        lib = _lib.get()
        a32 = torch.ones((1, 130), dtype=torch.float32, device=DEV)
        out = torch.empty((1, 130 * 130 + 130), dtype=torch.float32, device=DEV)
        _lib.check(lib.mfx_tridiag_eigh(_lib.ptr(a32), _lib.ptr(a32), 129, 1, 130, _lib.dtype_code(torch.float32), _lib.ptr(out),
                                        _lib.ptr(out[:, 130:]), _lib.stream_ptr(DEV)))
    with pytest.raises(ValueError, match="square"):
        DenseOp()(v, T(np.ones((n, n + 1))))
    with pytest.raises(TypeError):
        arnoldi.hessenberg(DenseOp(), 3, reortho="full")(v.to(torch.float16), A.to(torch.float16))


@pytest.mark.parametrize("dtype,vtol,gtol", [(torch.float64, 1e-10, 1e-7), (torch.float32, 2e-5, 2e-3)])
@pytest.mark.parametrize("k", [121, 150, 255])
def test_slq_beyond_depth_120(k, dtype, vtol, gtol):
    """The reference's SuiteSparse sweeps run to depth 150 (experiments/benchmarks/.../benchmark.py:21,83; plot_quadrant.py:22) and
    integrand_spd has no depth limit (lanczos.py:14-61).  Up to k = 120 the on-device eigen-solver and the quadrature's VJP keep their
    k x k work matrices in LDS; beyond, the rotations are accumulated in the fp64 output and the divided differences evaluated in place
    (fp32 problems are cast for the k x k part).  Against the oracle, value and gradient w.r.t. a dense symmetric parameter; k = 255 of
    n = 256 is (nearly) the exact log-quadratic form v^T log(A) v."""
    n = 256
    A = orc.spd_diag_plus_lowrank(n, 4, seed=3)
    v = np.where(np.random.default_rng(k).random(n) < 0.5, -1.0, 1.0)
    At = torch.tensor(A, dtype=dtype, device=DEV, requires_grad=True)
    val = lanczos.integrand_spd(torch.log, k, DenseOp())(torch.tensor(v, dtype=dtype, device=DEV), At)
    (g,) = torch.autograd.grad(val, At)
    ref, _, (gref,) = orc.integrand_spd_value_and_grad(orc.DenseOp(), k, v, (A,))
    assert abs(val.item() - ref) <= vtol * abs(ref), (val.item(), ref)
    assert np.abs(N(g.double()) - gref).max() <= gtol * np.abs(gref).max()
    if k == 255:
        lam, U = np.linalg.eigh(A)
        exact = (U.T @ v) ** 2 @ np.log(lam)
        assert abs(val.item() - exact) <= 10 * vtol * abs(exact)


@pytest.mark.parametrize("which", ["arnoldi", "lanczos-none", "lanczos-full"])
def test_custom_vjp_equals_autodiff_through_the_loop(which):
    """The reference's own adjoint test (tests/test_arnoldi/test_hessenberg_adjoint.py, tests/test_lanczos/test_tridiag_adjoint.py):
    the custom VJP must agree with back-propagation through the forward loop (custom_vjp=False).  Symmetric parametrisation
    for the three-term recurrence (its adjoint applies A, not A^T: quirk Q4)."""
    n, k = 10, 4
    rng = np.random.default_rng(12)
    P = rng.standard_normal((n, n)) + n * np.eye(n)
    v = rng.standard_normal(n)
    sym = which != "arnoldi"

    class SymOp:  # matvec(v, P) = (P + P^T) v as a python callable: exercises the callback operator too
        def __call__(self, x, Pm):
            return (Pm + Pm.T) @ x

    def run(custom):
        vt, Pt = T(v, grad=True), T(P, grad=True)
        matvec = SymOp() if sym else DenseOp()
        if which == "arnoldi":
            outs = arnoldi.hessenberg(matvec, k, reortho="full", custom_vjp=custom)(vt, Pt)
        else:
            (Q, (d, e)), (q, b) = lanczos.tridiag(matvec, k, reortho=which.split("-")[1], custom_vjp=custom)(vt, Pt)
            outs = (Q, d, e, q, b)
        g = torch.Generator(device=DEV).manual_seed(0)
        cot = [torch.randn(o.shape, dtype=o.dtype, device=DEV, generator=g) for o in outs]
        return [N(o) for o in outs], [N(t) for t in torch.autograd.grad(outs, (vt, Pt), cot)]

    o1, g1 = run(True)
    o0, g0 = run(False)
    for a, b in zip(o1, o0):
        assert np.allclose(a, b, rtol=1e-10, atol=1e-12)
    for a, b in zip(g1, g0):
        assert np.allclose(a, b, rtol=1e-6, atol=1e-8 * np.abs(b).max()), np.abs(a - b).max()


def test_pytree_start_vector_and_matfuns():
    """lanczos.py:24,28-33: integrand_spd ravels a pytree v0 and unravels it for the matvec."""
    n1, n2, k = 5, 7, 6
    rng = np.random.default_rng(8)
    A = orc.symmetric_matrix_from_eigenvalues(np.linspace(1.0, 3.0, n1 + n2), seed=8)
    At = T(A, grad=True)

    def matvec(tree, p):
        flat = torch.cat([tree["a"].reshape(-1), tree["b"].reshape(-1)])
        out = p @ flat
        return {"a": out[:n1], "b": out[n1:].reshape(n2, 1)}

    v0 = {"a": T(rng.standard_normal(n1)), "b": T(rng.standard_normal((n2, 1)))}
    flat = np.concatenate([N(v0["a"]), N(v0["b"]).ravel()])
    for name, fn in (("log", torch.log), ("exp", torch.exp), ("inv", lambda x: 1.0 / x), ("sqrt", torch.sqrt)):
        val = lanczos.integrand_spd(fn, k, matvec)(v0, At)
        (g,) = torch.autograd.grad(val, At)
        rv, _, (rg,) = orc.integrand_spd_value_and_grad(orc.DenseOp(), k, flat, (A,), matfun=name)
        assert close(val, rv, 1e-9), name
        assert close(g, rg, 1e-7, atol_rel=1e-7), name


# ------------------------------------------------------------------------------------------------
# BASELINE config 4 at FULL size (n = 131072, d = 8, k = 40): size-independent properties, since the
# oracle cannot run at this size.  fp32, default precision mode.
# ------------------------------------------------------------------------------------------------
def _c4_operator(n=131072, d=8):
    g = torch.Generator(device=DEV).manual_seed(4)
    X = torch.randn((n, d), generator=g, device=DEV, dtype=torch.float32)
    inv = lambda v: float(np.log(np.expm1(v)))
    raw = [torch.tensor(inv(v), device=DEV, dtype=torch.float32) for v in (2.0, 1.0, 0.1)]
    return X, raw


def test_c4_full_size_matvec_is_linear_and_symmetric():
    X, raw = _c4_operator()
    n = X.shape[0]
    op = gp_util.gram_operator(X)
    g = torch.Generator(device=DEV).manual_seed(1)
    U = torch.randn((8, n), generator=g, device=DEV, dtype=torch.float32)
    V = torch.randn((8, n), generator=g, device=DEV, dtype=torch.float32)
    with torch.no_grad():
        KU, KV = op(U, *raw), op(V, *raw)
        # symmetry: u^T K v = v^T K u (each side a sum of 131072 products of O(1e2) numbers)
        a, b = (U.double() * KV.double()).sum(-1), (V.double() * KU.double()).sum(-1)
        assert torch.allclose(a, b, rtol=0, atol=1e-4 * float(KU.double().norm(dim=-1).max() * V.double().norm(dim=-1).max()))
        # linearity: K (2u - 3v) = 2 K u - 3 K v
        lin = op(2 * U - 3 * V, *raw)
        ref = 2 * KU - 3 * KV
        assert float((lin - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
        # the noise term: K e_i has noise + outputscale on the diagonal (self-distance 0)
        e = torch.zeros((1, n), device=DEV, dtype=torch.float32)
        e[0, 77] = 1.0
        col = op(e, *raw)[0]
        assert abs(float(col[77]) - (1.0 + 0.1)) < 1e-5
        # single right-hand side and 64 right-hand sides go through the same kernel family: row 0 agrees
        one = op(U[:1], *raw)
        assert float((one[0] - KU[0]).abs().max()) <= 1e-5 * float(KU[0].abs().max())


def test_c4_full_size_lanczos_identities_and_gradient_direction():
    X, raw = _c4_operator()
    n, k, p = X.shape[0], 40, 4
    op = gp_util.gram_operator(X)
    probes = hutchinson.sampler_rademacher(X[:, 0], num=p)(3)
    with torch.no_grad():
        (Q, (alpha, beta)), (q_rem, b_rem) = lanczos.tridiag(op, k, reortho="full")(probes, *raw)
        # orthonormal basis (lanczos.py:152-169 with full re-orthogonalisation): Q Q^T = I_k per probe
        G = Q.double() @ Q.double().transpose(-1, -2)
        assert float((G - torch.eye(k, device=DEV, dtype=torch.float64)).abs().max()) < 5e-5
        # three-term recurrence on the last column: A q_k = beta_{k-1} q_{k-1} + alpha_k q_k + b_rem q_rem
        AQ = op(Q[:, -1], *raw).double()
        rec = beta[:, -1, None].double() * Q[:, -2].double() + alpha[:, -1, None].double() * Q[:, -1].double() \
            + b_rem[:, None].double() * q_rem.double()
        assert float((AQ - rec).norm(dim=-1).max()) <= 1e-3 * float(AQ.norm(dim=-1).max())
    # gradient of the SLQ estimate along a direction in (raw_l, raw_s, raw_noise) vs a central difference of the value
    params = [q.clone().requires_grad_(True) for q in raw]
    integrand = lanczos.integrand_spd(torch.log, k, op)
    val = integrand(probes, *params).double().mean()
    val.backward()
    direction = [0.3, -0.5, 0.8]
    got = sum(float(q.grad) * dd for q, dd in zip(params, direction))
    h = 2e-2  # fp32 values of size 3e5 with ~1e-6 relative noise: the step must be large
    with torch.no_grad():
        up = integrand(probes, *[q + h * dd for q, dd in zip(raw, direction)]).double().mean()
        dn = integrand(probes, *[q - h * dd for q, dd in zip(raw, direction)]).double().mean()
    fd = float(up - dn) / (2 * h)
    assert abs(got - fd) <= 2e-2 * abs(fd), (got, fd)


def test_operator_and_vector_dtypes_must_agree():
    """A float64 vector on float32 operator data would make the kernels misread the operator (and read out of bounds)."""
    X = torch.randn((64, 3), device=DEV, dtype=torch.float32)
    raw = [torch.zeros((), device=DEV, dtype=torch.float32) for _ in range(3)]
    v64 = torch.randn(64, device=DEV, dtype=torch.float64)
    with pytest.raises(TypeError, match="dtype of the vectors"):
        RbfGramOp(X)(v64, *raw)
    with pytest.raises(TypeError, match="dtype of the vectors"):
        DenseOp()(v64, torch.eye(64, device=DEV, dtype=torch.float32))
    with pytest.raises(TypeError, match="dtype of the vectors"):
        lanczos.tridiag(DenseOp(), 4, reortho="full")(v64, torch.eye(64, device=DEV, dtype=torch.float32))


@pytest.mark.parametrize("d", [3, 20])  # (20: the pre-packed form of 16 < d <= 32 has the same guard and the same fallback)
@pytest.mark.parametrize("kernel", ["rbf", "matern32"])
def test_rbf_op_inputs_beyond_the_f16_range_fall_back(kernel, d):
    """Scaled inputs with |x/l|^2 > 6e4 overflow the f16 image of the distance operands: the matvec must notice on the device and
    take the fp32-distance kernel (finite and right up to the cancellation error fp32 has at these magnitudes anyway)."""
    rng = np.random.default_rng(11)
    n, p = 2304, 8
    X = 40.0 * np.sqrt(3.0 / d) + rng.standard_normal((n, d)) * np.sqrt(3.0 / d)  # offset cloud: |x/l|^2 ~ 2e5 with l = 0.157
    raw = (np.array(-1.77), np.array(0.4), np.array(-1.0))  # softplus(-1.77) = 0.157
    V = rng.standard_normal((p, n))
    o = orc.RbfGramOp(X.astype(np.float32).astype(np.float64), noise_minval=1e-4, kernel=kernel, eps=float(torch.finfo(torch.float32).eps))
    op = RbfGramOp(T(X, torch.float32), noise_minval=1e-4, kernel=kernel)
    y = op(T(V, torch.float32), *[T(r, torch.float32) for r in raw])
    want = o.apply(V, *raw)
    assert np.all(np.isfinite(N(y)))
    assert np.abs(N(y) - want).max() <= 0.1 * np.abs(want).max()


def test_c_abi_standalone_consumer(tmp_path):
    """include/mfx.h + libmfx.so used from plain C++ (no Python, no torch): tests/cabi/cabi_smoke.cpp is compiled with hipcc on the box,
    linked against the in-tree library and run; it checks the Arnoldi identities and an adjoint invariant on the host."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "cabi_smoke")
    build = subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(root, "include"),
                            os.path.join(root, "tests", "cabi", "cabi_smoke.cpp"), "-L", libdir, "-lmfx",
                            f"-Wl,-rpath,{libdir}", "-o", exe], capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "cabi smoke ok" in run.stdout


def test_slq_value_and_gradient_are_bit_reproducible():
    """No data-path atomics anywhere (per-slice partials, fixed-order reductions, column-split partials summed in order): two runs of
    the same SLQ value-and-gradient give identical bits, on the matrix-core path (fp32) and on the VALU path (fp64)."""
    for dtype, n in ((torch.float32, 4096), (torch.float64, 1500)):
        g = torch.Generator(device=DEV).manual_seed(3)
        X = torch.randn((n, 5), generator=g, device=DEV, dtype=dtype)
        op = gp_util.gram_operator(X, noise_minval=1e-4)
        integrand = lanczos.integrand_spd(torch.log, 12, op)
        probes = hutchinson.sampler_rademacher(X[:, 0], num=16)(5)
        outs = []
        for _ in range(2):
            params = [torch.tensor(v, device=DEV, dtype=dtype, requires_grad=True) for v in (0.3, 0.1, -1.0)]
            vals = integrand(probes, *params)
            grads = torch.autograd.grad(vals.sum(), params)
            outs.append((vals.detach().clone(), [t.clone() for t in grads]))
        assert torch.equal(outs[0][0], outs[1][0])
        for a, b in zip(outs[0][1], outs[1][1]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("m", [24, 23])  # n = 576 (16-byte vector loads) and n = 529 (odd: scalar loads)
@pytest.mark.parametrize("reortho", ["full", "none"])
@pytest.mark.parametrize("dtype,vtol,gtol", [(torch.float64, 1e-10, 1e-7), (torch.float32, 2e-4, 5e-3)])
def test_slq_csr_several_probes(m, reortho, dtype, vtol, gtol):
    """SLQ integrand over a batch of probes on the native CSR operator (fused step head k_csr_step, grid.y = probes):
    value per probe and the gradient w.r.t. all stored values of the summed estimate, against the oracle probe by probe."""
    r, c, vals, n = orc.laplacian_2d_plus_identity(m)
    k, p = 7, 5
    probes = orc.rademacher(3, p, n)
    o = orc.CooOp(r, c, n)
    ref_vals, ref_grad = [], 0.0
    for v in probes:
        val, _dv, (dp,) = orc.integrand_spd_value_and_grad(o, k, v, (vals,), reortho=reortho)
        ref_vals.append(val)
        ref_grad = ref_grad + dp
    op, vt, order = CsrOp.from_coo(r, c, vals, n, DEV)
    vt = vt.to(dtype).requires_grad_(True)
    integrand = lanczos.integrand_spd(torch.log, k, op, reortho=reortho)
    out = integrand(T(probes, dtype), vt)
    assert close(out, np.asarray(ref_vals), vtol)
    (g,) = torch.autograd.grad(out.sum(), vt)
    ref = ref_grad[order.numpy()]
    assert close(g, ref, gtol, atol_rel=gtol)


def _random_csr(n, rng, per_row=4):
    """non-symmetric sparse matrix with a dominant diagonal: (row, col, vals) in COO order, each row 1..per_row+1 entries"""
    rows, cols, vals = [], [], []
    for i in range(n):
        m = int(rng.integers(0, per_row + 1))
        js = set(int(j) for j in rng.integers(0, n, size=m)) - {i}
        rows += [i] * (len(js) + 1)
        cols += [i] + sorted(js)
        vals += [3.0 + rng.random()] + list(0.3 * rng.standard_normal(len(js)))
    return np.asarray(rows), np.asarray(cols), np.asarray(vals)


@pytest.mark.parametrize("n,k", [(1, 1), (2, 2), (3, 2), (63, 7), (64, 9), (65, 33), (511, 5), (513, 17), (1023, 40), (2047, 3),
                                 (2049, 12), (4097, 21), (10001, 8), (70000, 6)])
@pytest.mark.parametrize("reortho", ["full", "none"])
def test_hessenberg_shape_sweep_csr_against_the_oracle(n, k, reortho):
    """Slice geometry of the vector kernels (2048- and 512-element slices, ragged tails, scalar / 16-byte loads, row groups, the
    fused CSR step head, the transposed structure in the adjoint) over many shapes: arnoldi.hessenberg forward and its
    adjoint with cotangents on every output, fp64, against the oracle."""
    if reortho == "none":
        k = min(k, 8)  # without re-orthogonalisation deep recurrences are chaotic (gradients of 1e17 that differ in sign between
        # two fp64 implementations): only the stable depths are a parity test
    rng = np.random.default_rng(1000 * n + k)
    r, c, vals = _random_csr(n, rng)
    v = rng.standard_normal(n)
    o = orc.CooOp(r, c, n)
    Qo, Ho, ro, co = orc.arnoldi_forward(o, k, v, vals, reortho=reortho)
    cot = dict(dQ=rng.standard_normal(Qo.shape), dH=rng.standard_normal(Ho.shape), dr=rng.standard_normal(n), dc=rng.standard_normal())
    dv_ref, (dvals_ref,) = orc.arnoldi_adjoint(o, (vals,), Q=Qo, H=Ho, r=ro, c=co, reortho=reortho, **cot)
    op, vt, order = CsrOp.from_coo(r, c, vals, n, DEV)
    vt = vt.double().requires_grad_(True)
    x0 = T(v, grad=True)
    for _ in range(2):  # the second call replays a hipGraph when the allocator hands out the same buffers
        Q, H, rr, cc = arnoldi.hessenberg(op, k, reortho=reortho)(x0, vt)
        assert close(Q, Qo, 1e-9, atol_rel=1e-9) and close(H, Ho, 1e-9, atol_rel=1e-9) and close(cc, co, 1e-10)
        assert np.allclose(N(rr), ro, rtol=1e-8, atol=1e-9 * max(np.abs(ro).max(), 1e-30) + 1e-13)
        dv, dvals = torch.autograd.grad((Q, H, rr, cc), (x0, vt), [T(cot["dQ"]), T(cot["dH"]), T(cot["dr"]), T(cot["dc"])])
        assert close(dv, dv_ref, 1e-7, atol_rel=1e-8)
        assert close(dvals, dvals_ref[order.numpy()], 1e-7, atol_rel=1e-8)
        del Q, H, rr, cc, dv, dvals


@pytest.mark.parametrize("n,k,p", [(65, 9, 3), (513, 17, 2), (1030, 12, 5), (2049, 20, 3), (4100, 33, 2), (9000, 6, 7)])
def test_hessenberg_shape_sweep_fp32_batched(n, k, p):
    """the fp32 instantiations of the same kernels (4-element vector loads, 1024-element slices, fp64 dot accumulation) with a batch
    of vectors: forward against the fp64 oracle, and the adjoint against the fp64 HIP path (itself oracle-checked above)."""
    rng = np.random.default_rng(77 * n + k)
    r, c, vals = _random_csr(n, rng)
    V = rng.standard_normal((p, n))
    o = orc.CooOp(r, c, n)
    op, vt, order = CsrOp.from_coo(r, c, vals, n, DEV)
    outs = {}
    for dtype in (torch.float64, torch.float32):
        v_ = vt.to(dtype).requires_grad_(True)
        x_ = T(V, dtype, grad=True)
        Q, H, rr, cc = arnoldi.hessenberg(op, k, reortho="full")(x_, v_)
        g = torch.Generator(device=DEV).manual_seed(5)
        cot = [torch.randn(t.shape, dtype=torch.float64, device=DEV, generator=g).to(dtype) for t in (Q, H, rr, cc)]
        dv, dvals = torch.autograd.grad((Q, H, rr, cc), (x_, v_), cot)
        outs[dtype] = [t.detach().double() for t in (Q, H, rr, cc, dv, dvals)]
    for b in range(p):
        Qo, Ho, ro, co = orc.arnoldi_forward(o, k, V[b], vals, reortho="full")
        assert close(outs[torch.float64][0][b], Qo, 1e-9, atol_rel=1e-9) and close(outs[torch.float64][1][b], Ho, 1e-9, atol_rel=1e-9)
    for a32, a64 in zip(outs[torch.float32], outs[torch.float64]):
        scale = a64.abs().max().item()
        assert torch.allclose(a32, a64, rtol=2e-3, atol=2e-4 * scale), (a32 - a64).abs().max().item() / scale
