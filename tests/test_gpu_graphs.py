"""hipGraph replay of launch-bound driver calls (include/mfx.h: mfx_graph_stats): the captured graph must see the data that
is in the buffers WHEN IT IS REPLAYED, and give what the eager launches give -- checked against the CPU oracle on inputs
that change in place between calls (benchmark protocol: the same jitted function called again and again,
experiments/benchmarks/wall_times_vjp_through_lanczos_arnoldi/suite_sparse/benchmark.py:91-121)."""

import numpy as np
import pytest
import torch

from oracle import slq_oracle as orc

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from matfree_extensions import _lib, lanczos
    from matfree_extensions.operators import CsrOp, DenseOp

DEV = torch.device("cuda:0")


def _tridiag_rounds(op, oracle_op, k, v_buf, p_buf, inputs, reortho, to_param=lambda a: a, got_map=lambda g: g, ref_map=None):
    """inputs: list of (v, params) NumPy pairs written IN PLACE into the same device buffers before each call."""
    alg = lanczos.tridiag(op, k, reortho=reortho)
    rng = np.random.default_rng(11)
    n = v_buf.shape[0]
    (Q0, _), _ = orc.tridiag(oracle_op, k, inputs[0][0], inputs[0][1], reortho=reortho)
    cot = ((rng.standard_normal(Q0.shape), (rng.standard_normal(k), rng.standard_normal(k - 1))), (rng.standard_normal(n), rng.standard_normal()))
    tc = [torch.tensor(np.asarray(t), dtype=torch.float64, device=DEV) for t in (cot[0][0], cot[0][1][0], cot[0][1][1], cot[1][0], cot[1][1])]
    vjp = orc.tridiag_full_vjp if reortho == "full" else orc.tridiag_none_vjp
    for v, prm in inputs:
        with torch.no_grad():
            v_buf.copy_(torch.tensor(v, dtype=torch.float64))
            p_buf.copy_(torch.tensor(to_param(prm), dtype=torch.float64))
        (Q, (d, e)), (q, b) = alg(v_buf, p_buf)
        dv, dp = torch.autograd.grad((Q, d, e, q, b), (v_buf, p_buf), tc)
        (Qr, (dr, er)), (qr, br) = orc.tridiag(oracle_op, k, v, prm, reortho=reortho)
        assert Q.shape == Qr.shape
        dv_ref, (dp_ref,) = vjp(oracle_op, k, v, (prm,), cot)
        assert np.allclose(d.detach().cpu().numpy(), dr, rtol=1e-9) and np.allclose(e.detach().cpu().numpy(), er, rtol=1e-9)
        assert np.allclose(dv.cpu().numpy(), dv_ref, rtol=1e-6, atol=1e-8 * np.abs(dv_ref).max())
        got, ref = got_map(dp).cpu().numpy(), (ref_map or to_param)(dp_ref)
        assert np.allclose(got, ref, rtol=1e-6, atol=1e-8 * np.abs(ref).max())
        del Q, d, e, q, b, dv, dp  # outputs go back to the allocator: the next call gets the same addresses


@pytest.mark.parametrize("reortho", ["full", "none"])
def test_graph_replay_sees_new_data_dense(reortho):
    n, k = 96, 6
    rng = np.random.default_rng(5)
    inputs = []
    for i in range(5):
        B = rng.standard_normal((n, n))
        inputs.append((rng.standard_normal(n), B @ B.T / n + (1.0 + i) * np.eye(n)))
    v_buf = torch.zeros(n, dtype=torch.float64, device=DEV, requires_grad=True)
    A_buf = torch.zeros(n, n, dtype=torch.float64, device=DEV, requires_grad=True)
    cap0, rep0 = _lib.graph_stats()
    sym = (lambda g: 0.5 * (g + g.T)) if reortho == "none" else (lambda g: g)  # lanczos.py:131-133: the 3-term VJP is for symmetric A
    _tridiag_rounds(DenseOp(), orc.DenseOp(), k, v_buf, A_buf, inputs, reortho, got_map=sym, ref_map=lambda a: np.asarray(sym(torch.tensor(a))))
    cap1, rep1 = _lib.graph_stats()
    assert cap1 - cap0 >= 2, "forward and adjoint should each have been captured on their second call"
    assert rep1 - rep0 >= 6, "later calls should replay the graphs"


def test_graph_replay_sees_new_data_csr():
    r, c, vals, n = orc.laplacian_2d_plus_identity(24)
    k = 8
    rng = np.random.default_rng(6)
    op, vt, order = CsrOp.from_coo(r, c, vals, n, DEV)
    inputs = [(rng.standard_normal(n), vals * (1.0 + 0.1 * i)) for i in range(4)]
    v_buf = torch.zeros(n, dtype=torch.float64, device=DEV, requires_grad=True)
    p_buf = torch.zeros_like(vt, dtype=torch.float64).requires_grad_(True)
    cap0, rep0 = _lib.graph_stats()
    # device values are in CSR order: to_param maps oracle (COO order) arrays to that order
    _tridiag_rounds(op, orc.CooOp(r, c, n), k, v_buf, p_buf, inputs, "full", to_param=lambda a: np.asarray(a)[order.numpy()])
    cap1, rep1 = _lib.graph_stats()
    assert cap1 - cap0 >= 2 and rep1 - rep0 >= 4


@pytest.mark.parametrize("reortho", ["full", "none"])
def test_graph_replay_hessenberg_matches_the_eager_first_call(reortho):
    """arnoldi.hessenberg forward + adjoint called five times on the same buffers: the first call runs eagerly, the second is
    captured, the rest are replays (the reortho="none" adjoint has a 2-D device copy per step inside the capture)."""
    from matfree_extensions import arnoldi

    n, k = 80, 7
    rng = np.random.default_rng(9)
    A = torch.tensor(rng.standard_normal((n, n)) / np.sqrt(n) + 2.0 * np.eye(n), dtype=torch.float64, device=DEV, requires_grad=True)
    v = torch.tensor(rng.standard_normal(n), dtype=torch.float64, device=DEV, requires_grad=True)
    alg = arnoldi.hessenberg(DenseOp(), k, reortho=reortho)
    cot = None
    first = None
    cap0, rep0 = _lib.graph_stats()
    for it in range(5):
        Q, H, r, c = alg(v, A)
        if cot is None:
            g = torch.Generator(device=DEV).manual_seed(1)
            cot = [torch.randn(t.shape, dtype=torch.float64, device=DEV, generator=g) for t in (Q, H, r, c)]
        dv, dA = torch.autograd.grad((Q, H, r, c), (v, A), cot)
        got = [t.detach().clone() for t in (Q, H, r, c, dv, dA)]
        if first is None:
            first = got
            Qo, Ho, ro, co = orc.arnoldi_forward(orc.DenseOp(), k, v.detach().cpu().numpy(), A.detach().cpu().numpy(), reortho=reortho)
            assert np.allclose(H.detach().cpu().numpy(), Ho, rtol=1e-9, atol=1e-12)
        else:
            for a, b in zip(got, first):
                assert torch.equal(a, b)
        del Q, H, r, c, dv, dA, got
    cap1, rep1 = _lib.graph_stats()
    assert cap1 - cap0 >= 1 and rep1 - rep0 >= 2  # (which calls repeat an address pattern is the allocator's business)
