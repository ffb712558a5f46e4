"""The A/B switches of DESIGN.md §3.8 select real alternative code paths (the un-fused CSR step, 2048-element slices, eager
launches instead of hipGraph replay, the earlier pipelined Gram matvec, the 256 x 128 gradient-GEMM tile with the LDS epilogue).  They are read once
per process, so ONE child process re-runs the parity tests of the operators and Krylov drivers with all of them flipped."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_parity_subset_with_every_switch_flipped():
    env = dict(os.environ)
    env.update(MFX_CSR_FUSED="0", MFX_FINE_SLICES="0", MFX_GRAPHS="0", MFX_RBF_PACK="0", MFX_GRAD_TILE="128", MFX_GRAD_REGEPI="0")
    keep = ("(csr or dense_op or hessenberg or tridiag or arnoldi_adjoint or integrand_spd_dense or rbf_op_dispatch or "
            "(rbf_op_apply and f16x3)) and not matern")
    out = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_parity.py"), "-m", "gpu", "-x", "-q", "-k", keep,
         "-p", "no:cacheprovider"],
        cwd=os.path.dirname(HERE), env=env, capture_output=True, text=True, timeout=900,
    )
    tail = out.stdout[-2000:] + out.stderr[-1000:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout.splitlines()[-1], tail
