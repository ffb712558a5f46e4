"""The experiment harnesses of SURVEY.md section 8(f)-4 run end to end on the GPU (they are scripts: each runs as ONE child process).

* exp_util.suite_sparse_load on a MatrixMarket file written from the embedded 1138_bus fixture (the reader of
  util/exp_util.py:35-42 with its symmetric expansion), then the SuiteSparse VJP timing sweep (benchmark.py:82-160) with
  its three output columns: forward, custom adjoint, backprop through the loop;
* the GP training CLI (optim_logml_adjoints_fixed.py flags) for two epochs on a small synthetic set.
"""

import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BENCH = os.path.join(ROOT, "experiments", "benchmarks", "wall_times_vjp_through_lanczos_arnoldi", "suite_sparse", "benchmark.py")
TRAIN = os.path.join(ROOT, "experiments", "applications", "gaussian_process", "train", "optim_logml_adjoints_fixed.py")


def _write_1138_bus(folder):
    import scipy.io
    import scipy.sparse

    g = np.load(os.path.join(GOLD, "csr_1138_bus.npz"))
    n = g["v"].shape[0]
    full = scipy.sparse.coo_matrix((g["vals"], (g["row"], g["col"])), shape=(n, n))
    lower = scipy.sparse.tril(full)  # MatrixMarket "symmetric" stores one triangle; mmread expands it (exp_util.py:36)
    os.makedirs(os.path.join(folder, "1138_bus"), exist_ok=True)
    scipy.io.mmwrite(os.path.join(folder, "1138_bus", "1138_bus.mtx"), lower, symmetry="symmetric")
    return g


def test_suite_sparse_load_reads_the_symmetric_expansion(tmp_path):
    from matfree_extensions.util import exp_util

    g = _write_1138_bus(str(tmp_path / "data" / "matrices"))
    op, vals = exp_util.suite_sparse_load("1138_bus", path=str(tmp_path / "data" / "matrices") + "/", device=torch.device("cuda:0"),
                                          dtype=torch.float64)
    assert op.n == 1138 and op.nnz == 4054  # expanded entries of 1138_bus
    v = torch.tensor(g["v"], dtype=torch.float64, device="cuda:0")
    dense = np.zeros((1138, 1138))
    np.add.at(dense, (g["row"], g["col"]), g["vals"])
    assert np.allclose(op(v, vals).cpu().numpy(), dense @ g["v"], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("alg,reortho", [("arnoldi", "full"), ("lanczos", "none")])
def test_suite_sparse_benchmark_cli_writes_all_three_columns(tmp_path, alg, reortho):
    _write_1138_bus(str(tmp_path / "data" / "matrices"))
    r = subprocess.run([sys.executable, BENCH, "--lanczos_or_arnoldi", alg, "--reortho", reortho, "--which_matrix", "1138_bus",
                        "--num_runs", "1", "--max_krylov_depth", "10", "--backprop_until", "10", "--dtype", "float64", "--precompile"],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "(synthetic stand-in matrix)" not in r.stdout  # the file was found and read
    outdir = os.path.dirname(BENCH).replace("experiments/", "results/")
    label = f"{alg}_1138_bus_reortho_{reortho}_precompile_True"
    depths = np.load(os.path.join(outdir, label + "_krylov_depths.npy"))
    assert list(depths) == list(range(1, 11))
    for col in ("fwdpass", "custom", "autodiff"):
        t = np.load(os.path.join(outdir, f"{label}_times_{col}.npy"))
        assert t.shape == depths.shape and np.all(t > 0), (col, t)


def test_gp_training_cli_two_epochs(tmp_path):
    r = subprocess.run([sys.executable, TRAIN, "--num_data", "3000", "--rank_precon", "20", "--num_matvecs", "8", "--num_samples", "4",
                        "--num_epochs", "2"], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("epoch ")]
    assert len(lines) == 2
    losses = [float(ln.split("loss")[1].split()[0]) for ln in lines]
    rmses = [float(ln.split("rmse")[1].split()[0]) for ln in lines]
    assert all(np.isfinite(losses)) and all(0 < v < 2.0 for v in rmses)  # standardised targets: an untrained constant gives 1.0
