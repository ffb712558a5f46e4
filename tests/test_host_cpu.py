"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol of include/mfx.h,
the host wrappers raise the reference's errors before any launch, and nothing silently falls back."""

import ctypes
import os
import re

import pytest
import torch

from matfree_extensions import _lib, arnoldi, hutchinson, lanczos
from matfree_extensions.distributed import rows_per_rank, shard_probes
from matfree_extensions.operators import CsrOp, DenseOp, RbfGramOp, as_operator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.get()
    header = open(os.path.join(ROOT, "include", "mfx.h")).read()
    declared = set(re.findall(r"\b(mfx_[a-z_0-9]+)\s*\(", header)) - {"mfx_callback_fn"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mfx_version() == 201
    assert lib.mfx_last_error() is not None


def test_struct_layout_matches_header_field_order():
    header = open(os.path.join(ROOT, "include", "mfx.h")).read()
    body = header[header.index("typedef struct mfx_operator {") : header.index("} mfx_operator;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(\w+);", body)
    assert fields == [f[0] for f in _lib.Operator._fields_]
    gbody = header[header.index("typedef struct mfx_op_grads {") : header.index("} mfx_op_grads;")]
    assert re.findall(r"\b(\w+);", gbody) == [f[0] for f in _lib.OpGrads._fields_]
    # sizes/offsets as the C compiler lays the header out
    import subprocess
    import tempfile

    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "mfx.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n",' \
           'sizeof(mfx_operator), sizeof(mfx_op_grads), offsetof(mfx_operator, x), offsetof(mfx_operator, lengthscale),' \
           'offsetof(mfx_operator, callback));return 0;}'
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "s.c"), os.path.join(td, "s")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        sizes = [int(t) for t in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    O = _lib.Operator
    assert sizes == [ctypes.sizeof(O), ctypes.sizeof(_lib.OpGrads), O.x.offset, O.lengthscale.offset, O.callback.offset]


def test_argument_errors_are_raised_before_any_launch():
    # arnoldi.py:16-19,58-60 ; lanczos.py:148-149 (tests/test_arnoldi/test_hessenberg_forward.py:69-84)
    for bad in (True, "full_with_sparsity", "None"):
        with pytest.raises(TypeError, match="Unexpected input"):
            arnoldi.hessenberg(lambda s: s, 1, reortho=bad)
    for k in (0, 5):
        with pytest.raises(ValueError, match="depth"):
            arnoldi.hessenberg(DenseOp(), k, reortho="full")(torch.ones(4), torch.eye(4))
        with pytest.raises(ValueError, match="depth"):
            lanczos.tridiag(DenseOp(), k, reortho="none")(torch.ones(4), torch.eye(4))
    with pytest.raises(ValueError, match="unsupported"):
        lanczos.tridiag(DenseOp(), 1, reortho="half")


def test_product_path_has_no_cpu_fallback():
    with pytest.raises(_lib.MfxError, match="no CPU fallback"):
        arnoldi.hessenberg(DenseOp(), 2, reortho="full")(torch.ones(4), torch.eye(4))
    with pytest.raises(_lib.MfxError, match="no CPU fallback"):
        lanczos.integrand_spd(torch.log, 2, DenseOp())(torch.ones(4), torch.eye(4))
    with pytest.raises(_lib.MfxError, match="no CPU fallback"):
        hutchinson.sampler_rademacher(torch.ones(4), num=2)(0)
    from matfree_extensions import cg, low_rank

    with pytest.raises(_lib.MfxError, match="no CPU fallback"):
        cg.cg_fixed_step(2)(DenseOp().bind(torch.eye(4)), torch.ones(4))
    with pytest.raises(_lib.MfxError, match="no CPU fallback"):
        low_rank.cholesky_partial_pivot(rank=2)(torch.eye(4), 4)
    src = ""
    pkg = os.path.join(ROOT, "experiments-lanczos-adjoints_amd", "matfree_extensions")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src += open(os.path.join(dirpath, f)).read()
    assert "oracle" not in src.replace("the oracle", "").lower() or "import oracle" not in src
    assert "from oracle" not in src and "import oracle" not in src


def test_operator_plumbing_on_cpu():
    op, bound = as_operator(DenseOp().bind(torch.eye(3)))
    assert isinstance(op, DenseOp) and len(bound) == 1
    op2, bound2 = as_operator(lambda v, p: p @ v)
    assert bound2 is None and op2(torch.ones(3), torch.eye(3)).shape == (3,)
    with pytest.raises(TypeError):
        as_operator(3)
    crow = torch.tensor([0, 2, 3, 5])
    col = torch.tensor([0, 2, 1, 0, 2])
    csr = CsrOp(crow, col, 3)
    assert csr.row.tolist() == [0, 0, 1, 2, 2]
    assert csr.t_crow.tolist() == [0, 2, 3, 5] and csr.t_col.tolist() == [0, 2, 1, 0, 2]
    assert csr.t_perm.tolist() == [0, 3, 2, 1, 4]
    rbf = RbfGramOp(torch.zeros(5, 2), noise_minval=1e-4)
    ls, s, nz = rbf.constrain(torch.zeros(2), torch.zeros(()), torch.zeros(()))
    assert torch.allclose(ls, torch.full((2,), 0.6931472)) and abs(nz.item() - (1e-4 + 0.6931472)) < 1e-6
    with pytest.raises(ValueError):
        rbf.constrain(torch.zeros(3), torch.zeros(()), torch.zeros(()))


def test_key_splitting_and_probe_sharding():
    ks = hutchinson.split(7, 3)
    assert len(set(ks)) == 3 and ks == hutchinson.split(7, 3)
    probes = torch.arange(24.0).reshape(6, 4)
    parts = hutchinson.split(probes, 3)
    assert len(parts) == 3 and torch.equal(torch.cat(parts), probes)
    assert hutchinson.sampler_rademacher(torch.ones(4), num=6)(probes) is probes
    covered = []
    for r in range(8):
        first, count = shard_probes(64, r, 8)
        covered += list(range(first, first + count))
    assert covered == list(range(64))
    assert [shard_probes(10, r, 4) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]


def _kernel_metadata():
    """{demangled-ish kernel name: (vgpr_count, vgpr_spill_count, private_segment_fixed_size)} of every gfx950 kernel in the BUILT
    libmfx.so (the AMDGPU metadata notes of its code objects)."""
    import shutil
    import subprocess
    import tempfile

    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf) and os.path.exists(_lib.LIB_PATH)):
        pytest.skip("needs the ROCm llvm tools and a built libmfx.so")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        so = shutil.copy(_lib.LIB_PATH, os.path.join(tmp, "libmfx.so"))
        subprocess.run([objdump, "--offloading", so], cwd=tmp, check=True, capture_output=True)  # extracts the bundles next to the copy
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([readelf, "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            for block in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                name = re.search(r"\.name:\s+(\S+)", block)
                if not name:
                    continue
                num = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", block).group(1))  # noqa: E731
                out[name.group(1)] = (num("vgpr_count"), num("vgpr_spill_count"), num("private_segment_fixed_size"))
    return out


def test_row_layout_is_equal_64_aligned_shards_or_a_stated_refusal():
    """rows_per_rank: the same shard length on every rank, a multiple of 64, rows on EVERY rank -- an n that equal shards cannot cover that
    way (858 rows on 8 ranks: shards of 128 leave the eighth rank empty) is refused with the rank count that has a layout in the message."""
    assert rows_per_rank(131072, 8) == 16384 and rows_per_rank(45730, 8) == 5760 and rows_per_rank(65, 2) == 64
    for n in range(65, 3000, 7):
        for world in (2, 3, 5, 8):
            try:
                nloc = rows_per_rank(n, world)
            except ValueError as exc:
                fewer = int(re.search(r"(\d+) ranks have one", str(exc)).group(1))
                assert 1 <= fewer < world and (fewer == 1 or rows_per_rank(n, fewer) > 0)
                continue
            assert nloc % 64 == 0 and (world - 1) * nloc < n <= world * nloc
    with pytest.raises(ValueError, match="leaves the last 1 rank"):
        rows_per_rank(858, 8)


def test_hot_kernels_scratch_budget_from_the_code_object():
    """DESIGN.md states what scratch the matrix-core kernels use; this reads it from the code object so the statement cannot rot
    (round 3's "zero scratch" had become untrue when a kernel was re-templated).  What is allowed, and where it executes:
    * k_rbf_fat_apply<4 | 8, *, 2>: 68 B -- one 8-byte spill pair around the CHAIN FOLD (once per 128 tiles) and one around the sweep;
      nothing per tile.  <12, *, 2> (three distance MFMAs per block: 16 more resident operand registers): 132 B -- ONE 16-register block
      of the chain masters lives in scratch and is touched by the chain fold only; <16, *, 2> (four distance MFMAs): 208 B, the same way.
      The <*, *, 1> forms: none.
    * k_rbf_mfma_grad_h: the register-epilogue forms (RBF, one lengthscale, d <= 8: config 4) none; the 256 x 256 tile with the LDS
      epilogue (Matern / ARD, d <= 8): <= 64 B -- row quantities of the epilogue, reloaded once per TILE (80 stages), nothing in the
      stage loop; the 256 x 128 forms: none, except the one for padded dimension 32 (16 < d <= 32): <= 64 B, the same way.
    * k_rbf_mfma_grad<64> (16 < d <= 64 in fp32, end of round 5): no scratch; <= 8 registers moved to AGPRs.
    * every other kernel of the library: none."""
    meta = _kernel_metadata()
    assert len(meta) > 300, len(meta)
    hot = 0
    for name, (vgpr, spill, scratch) in meta.items():
        if "k_rbf_fat_apply" in name:
            hot += 1
            two_blocks = re.search(r"Lb[01]ELi2E", name) is not None
            budget = 208 if "k_rbf_fat_applyILi16E" in name else (132 if "k_rbf_fat_applyILi12E" in name else 68)
            assert scratch <= (budget if two_blocks else 0), (name, scratch)
            assert vgpr > 256  # one wave per SIMD
        elif "k_rbf_mfma_grad_h" in name:
            hot += 1
            big_tile_lds_epilogue = re.search(r"k_rbf_mfma_grad_hILi\d+ELi4ELb0E", name) is not None
            widest = "k_rbf_mfma_grad_hILi32E" in name  # 16 < d <= 32 (round 5): 32 + 32 + 34 epilogue registers per thread, 44 B parked per TILE
            assert scratch <= (64 if big_tile_lds_epilogue or widest else 0), (name, scratch)
            assert vgpr <= 256  # two waves per SIMD
        elif "k_rbf_mfma_gradILi64E" in name:
            # the exact-fp32 sweep at padded dimension 64 (one workgroup per CU: 130 KB of LDS): 66 fp64 sums per thread next to the 64 x 64
            # accumulator block of the wave -- a handful of registers parked in AGPRs by the allocator, no memory
            assert scratch == 0 and spill <= 8, (name, spill, scratch)
        else:
            assert scratch == 0 and spill == 0, (name, spill, scratch)
    assert hot >= 12, hot


def test_host_side_of_the_c_abi_under_address_and_ub_sanitizers():
    """"`-fsanitize=address` host build of the C-ABI shim".  `make asan` compiles the HOST code of every
    csrc/*.hip under AddressSanitizer + UndefinedBehaviorSanitizer (device code unchanged; GPU sanitizers are not available on
    this pool) and tests/cabi/cabi_host_checks.cpp with the same flags.  Run here, without a GPU: (1) that program -- every workspace
    query on every operator kind and padding, the argument checks of the drivers, timing / graph counters; (2) this file's
    symbol-export, struct-layout, argument-error and no-fallback tests in a child interpreter that loads asan/libmfx_asan.so.
    Any invalid access or undefined behaviour aborts the child (`-fno-sanitize-recover`)."""
    import shutil
    import subprocess
    import sys

    csrc = os.path.join(ROOT, "experiments-lanczos-adjoints_amd", "csrc")
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not (shutil.which("make") and os.path.exists("/opt/rocm/bin/hipcc") and os.path.exists(clang)):
        pytest.skip("needs make, hipcc and the ROCm clang")
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("the sanitizer runtime is not installed")
    build = subprocess.run(["make", "-C", csrc, "asan", "-j4"], capture_output=True, text=True, timeout=900)
    assert build.returncode == 0, build.stdout[-2000:] + build.stderr[-2000:]
    lib = os.path.join(csrc, "asan", "libmfx_asan.so")
    chk = subprocess.run([os.path.join(csrc, "asan", "cabi_host_checks")], capture_output=True, text=True, timeout=300)
    assert chk.returncode == 0 and "cabi_host_checks ok" in chk.stdout, chk.stdout[-2000:] + chk.stderr[-3000:]
    # CPython itself is not leak-clean: leak detection off for the interpreter child (on for the C++ program above)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               MFX_LIBRARY_PATH=lib)
    child = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider", "-k",
                            "exports_every or struct_layout or argument_errors or no_cpu_fallback"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert child.returncode == 0 and "4 passed" in child.stdout, child.stdout[-2000:] + child.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in child.stderr and "runtime error:" not in child.stderr, child.stderr[-3000:]


def test_traffic_file_is_not_older_than_the_kernels_it_describes():
    """profiles/traffic.json feeds `roofline.traffic` of the bench line (read from the file, not measured in the run): it must name
    the commit and the command it was collected on, at the config-4 batch, and that commit must contain the last change to the
    kernels it describes -- round 4 shipped a file collected at --k 3 on an intermediate build that still spilled (658 MB of writes)."""
    import json
    import shutil
    import subprocess

    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert tj.get("krylov_depth") == 40 and "--k 40" in tj.get("command", ""), "collect with tools/prof_traffic.sh (K = 40)"
    assert tj["k_rbf_mfma_grad_h_WRITE_SIZE_KB"] < 16 * 1024, "the gradient GEMM writes its partial sums only: a spilling build?"
    assert tj["k_rbf_mfma_grad_h_launches_FETCH_SIZE"] >= 1 and tj["k_rbf_fat_apply_launches_FETCH_SIZE"] >= 80
    if not (shutil.which("git") and os.path.isdir(os.path.join(ROOT, ".git"))):
        pytest.skip("no git history here (the GPU box): the commit check runs in the development container")
    git = lambda *a: subprocess.run(["git", "-C", ROOT, *a], capture_output=True, text=True)  # noqa: E731
    csrc = "experiments-lanczos-adjoints_amd/csrc/"
    last = git("log", "-1", "--format=%H", "--", csrc + "mfx_rbf_fat.hip", csrc + "mfx_rbf_mfma.hip", csrc + "mfx_rbf_common.h").stdout.strip()
    assert last, "no history for the kernels"
    assert git("cat-file", "-e", tj["commit"] + "^{commit}").returncode == 0, f"traffic.json names an unknown commit {tj['commit']!r}"
    assert git("merge-base", "--is-ancestor", last, tj["commit"]).returncode == 0, (
        f"profiles/traffic.json was collected at {tj['commit']}, before the last change to the Gram kernels ({last[:7]}): re-run tools/prof_traffic.sh")
