"""MFX_GRAPHS=0 (eager launches instead of hipGraph replay of launch-bound driver calls) selects a real alternative
code path; it is read once per process, so ONE child process re-runs the parity tests of the Krylov drivers with it.  (The other
switch that survives, MFX_RBF_FAT=0, is covered by tests/test_gpu_matvec_kernels.py.)"""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_parity_subset_without_graph_replay():
    env = dict(os.environ, MFX_GRAPHS="0")
    keep = "(csr or dense_op or hessenberg or tridiag or arnoldi_adjoint or integrand_spd_dense) and not matern"
    out = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_parity.py"), "-m", "gpu", "-x", "-q", "-k", keep,
         "-p", "no:cacheprovider"],
        cwd=os.path.dirname(HERE), env=env, capture_output=True, text=True, timeout=900,
    )
    tail = out.stdout[-2000:] + out.stderr[-1000:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout.splitlines()[-1], tail
