"""Generate the committed golden fixtures from the CPU oracle (fp64).

    python tests/golden/make_golden.py

The reference stores no golden vectors for this path and cannot be executed here
(JAX / matfree absent), so the fixtures are produced by ``oracle/slq_oracle.py`` -- which is pinned
against the reference's own property tests in tests/test_oracle_pinning.py -- and every fixture is
re-validated against those identities by tests/test_golden_fixtures.py.  Fixtures are DATA only
(inputs + expected outputs).  The one external input, SuiteSparse ``1138_bus`` (a data file of the
reference: data/matrices/1138_bus/1138_bus.mtx, read as util/exp_util.py:35-42 does), is embedded as
COO triplets because /root/reference does not exist on the GPU box.
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import slq_oracle as orc  # noqa: E402


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name), **arrays)
    print("wrote", name, {k: np.shape(v) for k, v in arrays.items()})


def tridiag_forward():
    A = orc.symmetric_matrix_from_eigenvalues(np.arange(1.0, 2.0, 1.0 / 12), seed=1)
    v = np.flip(np.arange(1.0, 13.0)).copy()
    out = {"A": A, "v": v}
    for reortho in ("full", "none"):
        for k in (1, 5, 11, 12):
            (Q, (d, e)), (q, b) = orc.tridiag(orc.DenseOp(), k, v, A, reortho=reortho)
            out.update({f"{reortho}_{k}_Q": Q, f"{reortho}_{k}_d": d, f"{reortho}_{k}_e": e,
                        f"{reortho}_{k}_q": q, f"{reortho}_{k}_b": b})
    save("tridiag_forward_n12.npz", **out)


def arnoldi_adjoint():
    rng = np.random.default_rng(3)
    out = {}
    for tag, n, k, which in (("rand3", 3, 2, "random"), ("rand10", 10, 4, "random"), ("hilbert15", 15, 10, "hilbert")):
        v = rng.standard_normal(n)
        if which == "random":
            A = rng.standard_normal((n, n))
        else:
            Hm = np.tril(orc.hilbert(n))
            P = Hm - 0.5 * np.diag(np.diag(Hm))
            A = P + P.T
        out.update({f"{tag}_A": A, f"{tag}_v": v})
        for reortho in ("full", "none"):
            if which == "hilbert" and reortho == "none":
                continue
            Q, H, r, c = orc.arnoldi_forward(orc.DenseOp(), k, v, A, reortho=reortho)
            dQ, dH, dr, dc = (rng.standard_normal(np.shape(x)) for x in (Q, H, r, c))
            dv, (dA,) = orc.arnoldi_adjoint(orc.DenseOp(), (A,), Q=Q, H=H, r=r, c=c, dQ=dQ, dH=dH, dr=dr, dc=dc,
                                            reortho=reortho)
            pre = f"{tag}_{reortho}_"
            out.update({pre + "Q": Q, pre + "H": H, pre + "r": r, pre + "c": c, pre + "dQ": dQ, pre + "dH": dH,
                        pre + "dr": dr, pre + "dc": dc, pre + "dv": dv, pre + "dA": dA})
    save("arnoldi_adjoint.npz", **out)


def tridiag_adjoint():
    n, k = 10, 4
    rng = np.random.default_rng(2)
    A = orc.symmetric_matrix_from_eigenvalues(rng.uniform(size=n) + 1.0, seed=2)
    v = rng.standard_normal(n)
    out = {"A": A, "v": v}
    for reortho in ("full", "none"):
        (Q, (d, e)), (q, b) = orc.tridiag(orc.DenseOp(), k, v, A, reortho=reortho)
        cot = ((rng.standard_normal(Q.shape), (rng.standard_normal(d.shape), rng.standard_normal(e.shape))),
               (rng.standard_normal(q.shape), rng.standard_normal()))
        vjp = orc.tridiag_full_vjp if reortho == "full" else orc.tridiag_none_vjp
        dv, (dA,) = vjp(orc.DenseOp(), k, v, (A,), cot)
        pre = reortho + "_"
        out.update({pre + "dQ": cot[0][0], pre + "dd": cot[0][1][0], pre + "de": cot[0][1][1], pre + "dq": cot[1][0],
                    pre + "db": cot[1][1], pre + "dv": dv, pre + "dA": dA})
    save("tridiag_adjoint_n10.npz", **out)


def slq_dense():
    n, k = 11, 6
    A = orc.symmetric_matrix_from_eigenvalues(np.arange(1.0, 1.0 + n), seed=5)
    v0 = orc.rademacher(2, 1, n)[0]
    out = {"A": A, "v0": v0}
    for reortho in ("full", "none"):
        val, dv0, (dA,) = orc.integrand_spd_value_and_grad(orc.DenseOp(), k, v0, (A,), reortho=reortho)
        out.update({f"{reortho}_value": val, f"{reortho}_dv0": dv0, f"{reortho}_dA": dA})
    val, _, _ = orc.integrand_spd_value_and_grad(orc.DenseOp(), n, v0, (A,))
    out["full_depth_value"] = val
    # config C1 of BASELINE.json: 512 x 512 SPD (diag + rank 4), 20 steps, 1 probe
    A1 = orc.spd_diag_plus_lowrank(512, 4, seed=0)
    p1 = orc.rademacher(1, 1, 512)[0]
    val1, _, (dA1,) = orc.integrand_spd_value_and_grad(orc.DenseOp(), 20, p1, (A1,))
    out.update({"c1_value": val1, "c1_dA_diag": np.diag(dA1).copy(), "c1_dA_row0": dA1[0].copy()})
    save("slq_dense.npz", **out)


def slq_rbf():
    rng = np.random.default_rng(4)
    n, d, k, p = 96, 3, 8, 4
    X = rng.standard_normal((n, d))
    probes = orc.rademacher(5, p, n)
    out = {"X": X, "probes": probes}
    for tag, raw_l in (("ard", np.array([0.5, -0.3, 1.2])), ("iso", np.array(0.4))):
        raw = (raw_l, np.array(0.2), np.array(-1.0))
        op = orc.RbfGramOp(X, noise_minval=1e-4)
        val, grads, vals = orc.hutchinson_value_and_grad(op, k, probes, raw)
        out.update({f"{tag}_raw_l": raw_l, f"{tag}_value": val, f"{tag}_values": vals, f"{tag}_g_l": grads[0],
                    f"{tag}_g_s": grads[1], f"{tag}_g_n": grads[2]})
    out.update({"raw_s": np.array(0.2), "raw_n": np.array(-1.0), "noise_minval": np.array(1e-4)})
    save("slq_rbf_n96.npz", **out)


def csr_1138_bus():
    csr_suite_sparse("1138_bus", 12)


def csr_bloweybq():
    """SuiteSparse bloweybq (n = 10 001, 69 991 stored values after symmetric expansion): the second local matrix SURVEY.md
    section 8(d) names for BASELINE config 3 parity."""
    csr_suite_sparse("bloweybq", 6)


def csr_suite_sparse(name, k):
    path = f"/root/reference/data/matrices/{name}/{name}.mtx"
    import scipy.io

    M = scipy.io.mmread(path)  # symmetric expansion, as util/exp_util.py:36
    row, col, vals, n = M.row.astype(np.int64), M.col.astype(np.int64), M.data.astype(np.float64), M.shape[0]
    rng = np.random.default_rng(1)
    v = rng.standard_normal(n)
    op = orc.CooOp(row, col, n)
    out = {"row": row, "col": col, "vals": vals, "v": v, "k": np.array(k)}
    for reortho in ("full", "none"):
        (Q, (d, e)), (q, b) = orc.tridiag(op, k, v, vals, reortho=reortho)
        cot = ((rng.standard_normal(Q.shape), (rng.standard_normal(d.shape), rng.standard_normal(e.shape))),
               (rng.standard_normal(q.shape), rng.standard_normal()))
        vjp = orc.tridiag_full_vjp if reortho == "full" else orc.tridiag_none_vjp
        dv, (dvals,) = vjp(op, k, v, (vals,), cot)
        pre = reortho + "_"
        out.update({pre + "d": d, pre + "e": e, pre + "b": b, pre + "dQ": cot[0][0], pre + "dd": cot[0][1][0],
                    pre + "de": cot[0][1][1], pre + "dq": cot[1][0], pre + "db": cot[1][1], pre + "dv": dv,
                    pre + "dvals": dvals})
    if name != "1138_bus":  # keep the larger fixture small: int32 indices
        out["row"], out["col"] = row.astype(np.int32), col.astype(np.int32)
    save(f"csr_{name}.npz", **out)


def pde_wave():
    """The ONE numeric fixture the reference itself holds for this tier: data/pde_wave/16x16_data_{inputs,parameter,targets}.npy,
    written by experiments/applications/partial_differential_equation/make_data.py:52-103 -- initial states (y0, dy0), the
    coefficient field and the states at t = 1 of the anisotropic wave system (Neumann boundary, scale = parameter^2, mesh
    linspace(0, 1, 16), Dopri8 with 128 steps, fp32).  Copied as data; the tests compare expm(A) y0 against `targets`."""
    d = "/root/reference/data/pde_wave/16x16_data_"
    save("pde_wave_16x16.npz", inputs=np.load(d + "inputs.npy"), parameter=np.load(d + "parameter.npy"),
         targets=np.load(d + "targets.npy"))


def uci_protein():
    """BASELINE config 2's actual input, down-sized: the first 2048 rows of data/uci/protein/data.csv.gz (45 730 x 10, no header;
    the first 9 columns are the inputs), z-scored with the statistics of ALL rows as uci_util.py:229-230 does, and
    the oracle's SLQ log-det value-and-gradient at C2's settings (k = 30, 8 probes, raw parameters 0, noise floor 1e-4)."""
    import pandas as pd

    raw = pd.read_csv("/root/reference/data/uci/protein/data.csv.gz", header=None).values.astype(np.float64)
    assert raw.shape == (45730, 10)
    Xall = raw[:, :9]
    Xall = (Xall - Xall.mean(0)) / Xall.std(0)
    X = Xall[:2048]
    k, p, seed = 30, 8, 2
    probes = orc.rademacher(seed, p, X.shape[0])
    out = {"X": X, "k": np.array(k), "seed": np.array(seed), "num_probes": np.array(p), "noise_minval": np.array(1e-4),
           "col_mean": raw[:, :9].mean(0), "col_std": raw[:, :9].std(0)}
    for tag, raw_l in (("ard", np.zeros(9)), ("iso", np.array(0.0))):
        params = (raw_l, np.array(0.0), np.array(0.0))
        val, grads, vals = orc.hutchinson_value_and_grad(orc.RbfGramOp(X, noise_minval=1e-4), k, probes, params)
        out.update({f"{tag}_value": np.array(val), f"{tag}_values": vals, f"{tag}_g_l": grads[0], f"{tag}_g_s": np.array(grads[1]),
                    f"{tag}_g_n": np.array(grads[2])})
    save("uci_protein_2048.npz", **out)


def uci_protein_full():
    """BASELINE config 2's input at FULL size: all 45 730 rows of data/uci/protein/data.csv.gz, the 9 input columns z-scored as
    uci_util.py:229-230 does, stored in fp32 (the dtype config 2 runs in): data only, 1.5 MB."""
    import pandas as pd

    raw = pd.read_csv("/root/reference/data/uci/protein/data.csv.gz", header=None).values.astype(np.float64)
    assert raw.shape == (45730, 10)
    X = raw[:, :9]
    X = (X - X.mean(0)) / X.std(0)
    save("uci_protein_X.npz", X=X.astype(np.float32), col_mean=raw[:, :9].mean(0), col_std=raw[:, :9].std(0))


def gp_logml():
    """The "next" tier: short PCG runs (<= 8 steps: deterministic across implementations),
    pivoted partial Cholesky, Woodbury preconditioner and the composed log-marginal likelihood with its gradient."""
    rng = np.random.default_rng(21)
    n, d, rank, steps, k, nprobes, seed = 256, 3, 12, 8, 8, 4, 7
    X = rng.uniform(-1, 1, (n, d))
    y = np.sin(X.sum(-1)) + 0.1 * rng.standard_normal(n)
    raw = (np.array([0.2, -0.1, 0.4]), np.float64(0.3), np.float64(-2.0))
    minval, cval = 1e-3, 0.2
    out = {"X": X, "y": y, "raw_l": raw[0], "raw_s": raw[1], "raw_n": raw[2], "minval": np.array(minval), "cval": np.array(cval),
           "rank": np.array(rank), "steps": np.array(steps), "k": np.array(k), "nprobes": np.array(nprobes), "seed": np.array(seed)}
    for kind in ("rbf", "matern32"):
        op = orc.RbfGramOp(X, noise_minval=minval, kernel=kind)
        ls, s, noise = op.constrained(*raw)
        K = orc.kernel_matrix(kind, X, X, ls, s, diag_offset=0)
        L, info = orc.cholesky_partial_pivot(lambda i, j: K[i, j], n, rank)
        assert info["success"]
        P = lambda v: orc.precondition_solve(L, v, noise)
        A = lambda v: op.apply(v, *raw)
        b = y - cval
        solver = lambda Af, rhs: orc.pcg_fixed_step(Af, rhs, P, num_matvecs=steps)
        x, sinfo = solver(A, b)
        x_plain, _ = orc.pcg_fixed_step(A, b, None, num_matvecs=steps)
        xa, ainfo = orc.pcg_adaptive(A, b, P, atol=1e-3, rtol=0.0, maxiter=100, miniter=2)
        probes = orc.rademacher(seed, nprobes, n)
        ld, ld_grads, _ = orc.hutchinson_value_and_grad(op, k, probes, raw)
        value, _ = orc.logpdf_krylov(y, np.full(n, cval), logdet_value=ld, solve=lambda rhs: solver(A, rhs))
        lam, dp = orc.linear_solve_vjp(op, raw, solver, x, -0.5 * b)
        db = -0.5 * x + lam
        grads = [-0.5 * g + q for g, q in zip(ld_grads, dp)]
        pre = kind + "_"
        out.update({pre + "L": L, pre + "pivots": info["pivots"], pre + "x_pcg": x, pre + "r_pcg": sinfo["residual_abs"],
                    pre + "x_cg": x_plain, pre + "x_adaptive": xa, pre + "steps_adaptive": np.array(ainfo["num_steps"]),
                    pre + "precond_b": P(b), pre + "logml": np.array(value), pre + "g_l": grads[0], pre + "g_s": np.array(grads[1]),
                    pre + "g_n": np.array(grads[2]), pre + "g_y": db, pre + "g_c": np.array(-db.sum())})
    save("gp_logml_n256.npz", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:  # regenerate single fixtures: python make_golden.py pde_wave csr_bloweybq uci_protein ...
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    tridiag_forward()
    arnoldi_adjoint()
    tridiag_adjoint()
    slq_dense()
    slq_rbf()
    csr_1138_bus()
    csr_bloweybq()
    pde_wave()
    uci_protein()
    uci_protein_full()
    gp_logml()
