"""The committed golden fixtures (tests/golden/*.npz, produced by the oracle) must themselves satisfy
the reference's known-answer identities -- independent of the code that wrote them."""

import os

import numpy as np
import pytest

from oracle import slq_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("reortho", ["full", "none"])
def test_tridiag_forward_fixture_identities(reortho):
    g = np.load(os.path.join(GOLD, "tridiag_forward_n12.npz"))
    A = g["A"]
    assert np.allclose(np.linalg.eigvalsh(A), np.arange(1.0, 2.0, 1 / 12))
    for k in (1, 5, 11):
        Q, T = g[f"{reortho}_{k}_Q"], orc.dense_tridiag(g[f"{reortho}_{k}_d"], g[f"{reortho}_{k}_e"])
        eK = np.eye(k)[-1]
        assert np.allclose(A @ Q.T, Q.T @ T + np.outer(eK, g[f"{reortho}_{k}_q"] * g[f"{reortho}_{k}_b"]).T, atol=1e-5)
    Q, T = g[f"{reortho}_12_Q"], orc.dense_tridiag(g[f"{reortho}_12_d"], g[f"{reortho}_12_e"])
    tol = 1e-5 if reortho == "full" else 1e-1
    assert np.allclose(Q.T @ T @ Q, A, atol=tol)


def test_arnoldi_adjoint_fixture_matches_finite_differences():
    g = np.load(os.path.join(GOLD, "arnoldi_adjoint.npz"))
    tag, reortho = "rand10", "full"
    A, v = g[f"{tag}_A"], g[f"{tag}_v"]
    pre = f"{tag}_{reortho}_"
    k = g[pre + "H"].shape[0]

    def functional(A_, v_):
        Q, H, r, c = orc.arnoldi_forward(orc.DenseOp(), k, v_, A_, reortho=reortho)
        return (Q * g[pre + "dQ"]).sum() + (H * g[pre + "dH"]).sum() + r @ g[pre + "dr"] + c * g[pre + "dc"]

    rng = np.random.default_rng(0)
    E, w = rng.standard_normal(A.shape), rng.standard_normal(v.shape)
    h = 1e-6
    fd_A = (functional(A + h * E, v) - functional(A - h * E, v)) / (2 * h)
    fd_v = (functional(A, v + h * w) - functional(A, v - h * w)) / (2 * h)
    assert np.isclose(fd_A, (g[pre + "dA"] * E).sum(), rtol=1e-6)
    assert np.isclose(fd_v, g[pre + "dv"] @ w, rtol=1e-6)


def test_slq_fixtures_known_answers():
    g = np.load(os.path.join(GOLD, "slq_dense.npz"))
    lam, U = np.linalg.eigh(g["A"])
    assert np.isclose(g["full_depth_value"], g["v0"] @ (U * np.log(lam)) @ U.T @ g["v0"], rtol=1e-9)
    r = np.load(os.path.join(GOLD, "slq_rbf_n96.npz"))
    assert np.isclose(r["ard_value"], r["ard_values"].mean())
    op = orc.RbfGramOp(r["X"], noise_minval=float(r["noise_minval"]))
    raw = (r["iso_raw_l"], r["raw_s"], r["raw_n"])
    K = np.stack([op.apply(e, *raw) for e in np.eye(96)])
    assert np.allclose(K, K.T) and np.linalg.eigvalsh(K).min() > 0
    # 4 probes are few; the SLQ mean is within Monte-Carlo distance of the exact log-determinant
    assert abs(r["iso_value"] - np.linalg.slogdet(K)[1]) < 0.25 * abs(np.linalg.slogdet(K)[1])


def test_csr_fixture_is_symmetric_expansion_of_1138_bus():
    g = np.load(os.path.join(GOLD, "csr_1138_bus.npz"))
    n = g["v"].shape[0]
    assert n == 1138 and g["row"].shape[0] == 4054
    D = np.zeros((n, n))
    np.add.at(D, (g["row"], g["col"]), g["vals"])
    assert np.allclose(D, D.T)
