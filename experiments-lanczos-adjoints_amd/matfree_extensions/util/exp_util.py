"""Experiment utilities (subset of the reference's util/exp_util.py that the hot-path harnesses use).

``suite_sparse_load`` (:35-42) reads a MatrixMarket file into the native CSR operator; ``suite_sparse_synthetic`` stands
in when the file is absent (this build has no network: the reference's ``suite_sparse_download`` is out of scope).
"""

from __future__ import annotations

import os

import numpy as np
import torch

from ..operators import CsrOp

# (rows, stored entries of the symmetric expansion) of the matrices the reference benchmarks name
_KNOWN = {"1138_bus": (1138, 4054), "bcsstk18": (11948, 149090), "t2dal_e": (4257, 4257), "gyro": (17361, 1021159)}


def suite_sparse_load(which, /, path="./data/matrices/", suffix=".mtx", *, device, dtype=torch.float32):
    """-> (CsrOp, values): ``op(v, values)`` is the reference's ``BCOO((values, indices)) @ v`` (benchmark.py:64-68)."""
    import scipy.io

    m = scipy.io.mmread(f"{path}{which}/{which}{suffix}").tocoo()
    op, vals, _ = CsrOp.from_coo(m.row, m.col, m.data, m.shape[0], device)
    return op, vals.to(dtype)


def suite_sparse_synthetic(which, /, *, device, dtype=torch.float32, seed=0):
    """A symmetric, diagonally dominant sparse matrix with the row count and fill of the named SuiteSparse matrix."""
    n, nnz = _KNOWN[which] if isinstance(which, str) else which
    rng = np.random.default_rng(seed)
    m = max((nnz - n) // 2, 0)
    i = rng.integers(0, n, 2 * m + 16)
    j = np.clip(i + rng.integers(1, max(2, n // 50), i.size) * rng.choice([-1, 1], i.size), 0, n - 1)
    keep = i != j
    pairs = np.unique(np.stack([np.minimum(i, j)[keep], np.maximum(i, j)[keep]], 1), axis=0)[:m]
    w = -rng.uniform(0.1, 1.0, pairs.shape[0])
    diag = np.zeros(n)
    np.add.at(diag, pairs[:, 0], -w)
    np.add.at(diag, pairs[:, 1], -w)
    row = np.concatenate([pairs[:, 0], pairs[:, 1], np.arange(n)])
    col = np.concatenate([pairs[:, 1], pairs[:, 0], np.arange(n)])
    val = np.concatenate([w, w, diag + 1.0])
    op, vals, _ = CsrOp.from_coo(row, col, val, n, device)
    return op, vals.to(dtype)


def matching_directory(file, where, /, replace="experiments/"):
    """util/exp_util.py:102-110."""
    if where not in ["data/", "figures/", "results/"]:
        raise ValueError
    if replace not in ["experiments/"]:
        raise ValueError
    return (os.path.dirname(file) + "/").replace(replace, where)


def hilbert(ndim, /, *, device=None, dtype=torch.float64):
    """util/exp_util.py:113-115."""
    a = torch.arange(ndim, device=device, dtype=dtype)
    return 1 / (1 + a[:, None] + a[None, :])


def tree_random_like(seed, tree, *, generate_func=torch.randn):
    """util/exp_util.py:118-121 for (nested) tuples/lists/dicts of tensors; ``seed`` replaces the PRNG key."""
    gen = {}

    def make(t):
        g = gen.setdefault(t.device, torch.Generator(device=t.device).manual_seed(int(seed)))
        return generate_func(t.shape, dtype=t.dtype, device=t.device, generator=g)

    def walk(x):
        if torch.is_tensor(x):
            return make(x)
        if isinstance(x, dict):
            return {k: walk(v) for k, v in x.items()}
        return type(x)(walk(v) for v in x)

    return walk(tree)
