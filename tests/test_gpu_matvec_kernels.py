"""The matrix-core Gram matvec kernels against the fp64 NumPy oracle: the fat-wave kernel (`k_rbf_fat_apply`,
csrc/mfx_rbf_fat.hip: the default for RBF with d <= 16 -- BASELINE config 4's matvec and config 2's -- in its two forms, chunks of 64 vectors and,
for at most 32 vectors, one 32-probe block) and the same-program kernel `k_rbf_mfma_apply_h3` (every other shape, and
MFX_RBF_FAT=0).

What is specific to these kernels and therefore tested here: two probe chunks with a ragged second one, ragged n (last tile and
last row block), row blocks that start inside a workgroup, column splits (small n) and the unsplit sweep with chain folds (more
than 128 tiles per sweep), every block position of a tile (the fat kernel's distance MFMAs are asm the compiler cannot check), the
Matern diagonal fix, and BIT-IDENTITY of the two kernels (the fat kernel keeps the order of every sum).  The switch is read once
per process, so the non-default kernel runs in child processes.  Reference kernel: util/gp_util.py:69-184, Gram matvec :525-549.
"""

import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import slq_oracle as orc

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from matfree_extensions import _lib
    from matfree_extensions.operators import RbfGramOp

DEV = torch.device("cuda:0")
HERE = os.path.dirname(os.path.abspath(__file__))


def _setup(n, d, p, kernel, ard, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    raw = (rng.standard_normal(d) * 0.2 + 0.8 if ard else np.array(0.9), np.array(0.4), np.array(-1.0))
    V = rng.standard_normal((p, n)) * np.exp(rng.standard_normal((p, 1)) * 3.0)  # rows of very different magnitude
    o = orc.RbfGramOp(X, noise_minval=1e-4, kernel=kernel, eps=float(torch.finfo(torch.float32).eps))
    op = RbfGramOp(torch.tensor(X, dtype=torch.float32, device=DEV), noise_minval=1e-4, kernel=kernel)
    params = [torch.tensor(np.asarray(r), dtype=torch.float32, device=DEV) for r in raw]
    return o, op, raw, params, V


def _rel_err_rows(y, ref):
    y = y.detach().cpu().numpy().astype(np.float64)
    return (np.abs(y - ref).max(axis=1) / np.abs(ref).max(axis=1)).max()


@pytest.mark.parametrize("kernel", ["rbf", "matern32", "matern12"])
@pytest.mark.parametrize("n,d,p,ard", [(1000, 8, 64, False), (2341, 3, 33, True), (5000, 11, 100, True), (4096, 8, 64, True),
                                       (777, 12, 40, False), (1000, 8, 32, False), (2341, 3, 7, True), (4096, 5, 20, True),
                                       (3000, 8, 1, False), (5000, 4, 3, True)])
def test_pc_matvec_against_the_oracle(kernel, n, d, p, ard):
    o, op, raw, params, V = _setup(n, d, p, kernel, ard, seed=n + p)
    y = op(torch.tensor(V, dtype=torch.float32, device=DEV), *params)
    assert _rel_err_rows(y, o.apply(V, *raw)) < 3e-5  # per vector, relative to its largest entry


def _apply_block(op, cparams, V, row0, nrows):
    p, n = V.shape
    desc = op.descriptor(cparams, V.dtype, n)
    desc.row0, desc.nrows = row0, nrows
    ws = _lib.workspace(desc, n, 1, p, V.device)
    y = torch.empty((p, nrows), dtype=V.dtype, device=V.device)
    _lib.check(_lib.get().mfx_op_apply(C.byref(desc), _lib.ptr(V), n, _lib.ptr(y), nrows, p, 0, _lib.ptr(ws), ws.numel(),
                                       _lib.stream_ptr(V.device)))
    return y


@pytest.mark.parametrize("p", [64, 20])
@pytest.mark.parametrize("kernel", ["rbf", "matern32"])
def test_pc_matvec_row_blocks(kernel, p):
    """Row blocks (the row-sharded layout, util/gp_util.py:496-509): starts on multiples of 64 inside a 256-row workgroup,
    ragged ends, a block shorter than one wave's 64 rows."""
    n, d = 3000, 8
    o, op, raw, params, V = _setup(n, d, p, kernel, True, seed=5)
    ref = o.apply(V, *raw)
    Vt = torch.tensor(V, dtype=torch.float32, device=DEV)
    cparams = op.constrain(*params)
    for row0, nrows in [(0, 1500), (1472, 1528), (64, 192), (2944, 56), (1920, 1)]:
        blk = _apply_block(op, cparams, Vt, row0, nrows)
        r = ref[:, row0 : row0 + nrows]
        err = (np.abs(blk.cpu().numpy().astype(np.float64) - r).max(axis=1) / np.abs(ref).max(axis=1)).max()
        assert err < 3e-5, (row0, nrows, err)


CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, {tests!r})
sys.path.insert(0, {root!r})
sys.path.insert(0, {pkg!r})
import test_gpu_matvec_kernels as t
n = 20000                   # 313 tiles in ONE sweep: two chain folds
for kernel, p, d in (("rbf", 64, 8), ("matern32", 64, 8), ("rbf", 24, 8), ("rbf", 64, 9), ("rbf", 24, 12), ("rbf", 64, 16), ("rbf", 9, 13)):
    o, op, raw, params, V = t._setup(n, d, p, kernel, True, seed=11)
    # smooth positive vectors: the accumulators grow monotonically, which is what the chain folds are for
    V[:8] = np.abs(V[:8])
    y = op(torch.tensor(V, dtype=torch.float32, device=t.DEV), *params)
    err = t._rel_err_rows(y, o.apply(V, *raw))
    print(kernel, "err", err)
    assert err < 3e-5, err
    np.save({out!r} + "_" + kernel + str(p) + "_d" + str(d) + ".npy", y.cpu().numpy())
print("child ok")
"""


def _run_child(tmp_path, tag, **env_over):
    env = dict(os.environ)
    env.update(env_over)
    out = str(tmp_path / tag)
    root = os.path.dirname(HERE)
    code = CHILD.format(tests=HERE, root=root, pkg=os.path.join(root, "experiments-lanczos-adjoints_amd"), out=out)
    r = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(HERE), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    return out


def test_unsplit_sweep_with_chain_folds_and_bit_identity_of_the_two_kernels(tmp_path):
    fat = _run_child(tmp_path, "fat", MFX_RBF_SPLIT="1", MFX_RBF_FAT="1")
    h3 = _run_child(tmp_path, "h3", MFX_RBF_SPLIT="1", MFX_RBF_FAT="0")
    for kernel in ("rbf64_d8", "matern3264_d8", "rbf24_d8", "rbf64_d9", "rbf24_d12", "rbf64_d16", "rbf9_d13"):
        # the fat-wave kernel keeps the order of every sum of the same-program kernel (blocks, k-steps, products, chain folds):
        # bit-identical, so the accuracy tables of profiles/r02a_accuracy carry over (Matern: the child runs h3 both times)
        assert np.array_equal(np.load(fat + "_" + kernel + ".npy"), np.load(h3 + "_" + kernel + ".npy")), kernel


def test_every_parity_case_on_the_same_program_kernel():
    """The oracle, row-block and block-position cases of this file, re-run with MFX_RBF_FAT=0 (k_rbf_mfma_apply_h3 for every shape)."""
    env = dict(os.environ, MFX_RBF_FAT="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "oracle or row_blocks or position",
                        "-p", "no:cacheprovider"], cwd=os.path.dirname(HERE), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]


@pytest.mark.parametrize("d", [8, 9, 14])
@pytest.mark.parametrize("p", [64, 32])
def test_every_block_position_of_a_tile(p, d):
    """Unit vectors: K[i][j] itself, for the columns of one tile at a time -- every (column block, row block) position of the fat
    kernel's block pipeline (the blocks whose distances are computed across the mid-tile barrier included), first, middle and last
    tile; with 32 vectors (the one-probe-block form) one column block of the tile at a time."""
    n = 1536  # d = 9: three distance MFMAs per block (config 2's shape), d = 14: four -- other slot orders
    o, op, raw, params, _ = _setup(n, d, p, "rbf", False, seed=3)
    for tile, half in [(t, h) for t in (0, 1, 11, n // 64 - 1) for h in range(64 // p)]:
        E = np.zeros((p, n))
        E[np.arange(p), tile * 64 + half * p + np.arange(p)] = 1.0
        y = op(torch.tensor(E, dtype=torch.float32, device=DEV), *params).cpu().numpy().astype(np.float64)
        ref = o.apply(E, *raw)
        assert np.abs(y - ref).max() < 3e-6 * np.abs(ref).max(), (tile, half, np.abs(y - ref).max())
