// Host-side paths of the C-ABI that need no GPU, for the sanitizer build (`make -C experiments-lanczos-adjoints_amd/csrc asan`,
// SURVEY.md section 5): version / error string, every workspace query on every operator kind (the carve-up arithmetic of the
// RBF workspace included), and the argument checks of the drivers, which must fail with MFX_ERR_* and a message BEFORE any
// launch -- the reference's call-time errors (arnoldi.py:58-60 depth out of range; lanczos.py:148-149).  Built with
// -fsanitize=address,undefined and linked against asan/libmfx_asan.so; exits non-zero on a wrong status, and the sanitizers abort on
// any invalid access / undefined behaviour on the way.  tests/test_host_cpu.py runs it in the GPU-less container.
#include <cstdio>
#include <cstring>
#include <vector>

#include "mfx.h"

static int failures = 0;
#define EXPECT(cond, what)                                                                  \
  do {                                                                                      \
    if (!(cond)) {                                                                          \
      std::printf("FAIL %s:%d %s  [last error: %s]\n", __FILE__, __LINE__, what, mfx_last_error()); \
      ++failures;                                                                           \
    }                                                                                       \
  } while (0)

static mfx_operator make_op(int kind, int dtype, int64_t n) {
  mfx_operator op;
  std::memset(&op, 0, sizeof(op));
  op.kind = kind;
  op.dtype = dtype;
  op.n = n;
  return op;
}

int main() {
  EXPECT(mfx_version() == MFX_VERSION, "mfx_version");
  EXPECT(mfx_last_error() != nullptr, "mfx_last_error is never NULL");

  // ---- workspace queries: pure host arithmetic, every operator kind, both dtypes, small and C4-sized shapes --------------------
  static float dummy[64];
  for (int dtype : {MFX_F32, MFX_F64}) {
    for (int kind : {MFX_OP_DENSE, MFX_OP_CSR, MFX_OP_RBF, MFX_OP_CALLBACK}) {
      for (int64_t n : {int64_t(12), int64_t(1000), int64_t(131072)}) {
        mfx_operator op = make_op(kind, dtype, n);
        op.lda = n;
        op.nnz = 5 * n;
        op.max_row_nnz = 5;
        op.x = dummy;
        op.d = 8;
        op.rbf_mode = MFX_RBF_F16X3;
        for (int64_t k : {int64_t(1), int64_t(12)})
          for (int64_t p : {int64_t(1), int64_t(8), int64_t(64), int64_t(100)}) {
            const int64_t w = mfx_workspace_bytes(&op, n, k, p);
            EXPECT(w > 0, "mfx_workspace_bytes > 0");
            EXPECT(mfx_pcg_workspace_bytes(&op, n, p, 0) > 0, "mfx_pcg_workspace_bytes > 0");
            EXPECT(mfx_pcg_workspace_bytes(&op, n, p, 4) >= mfx_pcg_workspace_bytes(&op, n, p, 0), "preconditioner rank adds workspace");
            if (kind == MFX_OP_DENSE || kind == MFX_OP_CALLBACK) {
              mfx_operator op2 = op;
              op2.n = 2 * n;
              op2.lda = 2 * n;
              EXPECT(mfx_complex_workspace_bytes(&op2, n, k, p) > 0, "mfx_complex_workspace_bytes > 0");
            }
            if (kind == MFX_OP_RBF) EXPECT(mfx_gram_cross_workspace_bytes(&op, 37) > 0, "mfx_gram_cross_workspace_bytes > 0");
          }
        // every d the matrix-core kernels take, ARD or not, and one beyond (VALU kernel): the RBF carve-up has a branch per padding
        if (kind == MFX_OP_RBF)
          for (int d = 1; d <= 20; ++d) {
            op.d = d;
            op.ard = d & 1;
            for (int mode : {MFX_RBF_FP32, MFX_RBF_F16X3_MATVEC, MFX_RBF_F16X3}) {
              op.rbf_mode = mode;
              EXPECT(mfx_workspace_bytes(&op, n, 4, 33) > 0, "RBF workspace for every d");
            }
          }
      }
    }
  }
  // row-sharded queries with a plain (callback-less) communicator descriptor
  {
    mfx_comm cm;
    std::memset(&cm, 0, sizeof(cm));
    cm.rank = 1;
    cm.world = 8;
    cm.nloc = 16384;
    mfx_operator op = make_op(MFX_OP_RBF, MFX_F32, 131072);
    op.x = dummy;
    op.d = 8;
    op.rbf_mode = MFX_RBF_F16X3;
    op.row0 = cm.rank * cm.nloc;
    op.nrows = cm.nloc;
    EXPECT(mfx_sharded_workspace_bytes(&op, &cm, 131072, 40, 64) > 0, "mfx_sharded_workspace_bytes > 0");
    EXPECT(mfx_pcg_sharded_workspace_bytes(&op, &cm, 131072, 64, 0) > 0, "mfx_pcg_sharded_workspace_bytes > 0");
  }

  // ---- argument errors: a negative status and a message, never a launch ----------------------------------------------------------
  {
    mfx_operator op = make_op(MFX_OP_DENSE, MFX_F64, 4);
    op.dense_a = dummy;
    op.lda = 4;
    char ws[256];
    // depth outside [1, n]  (arnoldi.py:58-60)
    for (int64_t k : {int64_t(0), int64_t(5), int64_t(-3)}) {
      EXPECT(mfx_arnoldi_forward(&op, dummy, 4, k, 1, 1, dummy, dummy, dummy, dummy, ws, sizeof(ws), nullptr) == MFX_ERR_INVALID, "arnoldi depth");
      EXPECT(std::strlen(mfx_last_error()) > 0, "message for a bad depth");
      EXPECT(mfx_lanczos_forward(&op, dummy, 4, k, 1, dummy, dummy, dummy, dummy, ws, sizeof(ws), nullptr) == MFX_ERR_INVALID, "lanczos depth");
    }
    // null operator / null vectors / p < 1
    EXPECT(mfx_arnoldi_forward(nullptr, dummy, 4, 2, 1, 1, dummy, dummy, dummy, dummy, ws, sizeof(ws), nullptr) < 0, "null operator");
    EXPECT(mfx_arnoldi_forward(&op, nullptr, 4, 2, 1, 1, dummy, dummy, dummy, dummy, ws, sizeof(ws), nullptr) < 0, "null start vector");
    EXPECT(mfx_arnoldi_forward(&op, dummy, 4, 2, 0, 1, dummy, dummy, dummy, dummy, ws, sizeof(ws), nullptr) < 0, "p = 0");
    EXPECT(mfx_op_apply(nullptr, dummy, 4, dummy, 4, 1, 0, ws, sizeof(ws), nullptr) < 0, "apply: null operator");
    EXPECT(mfx_op_apply(&op, nullptr, 4, dummy, 4, 1, 0, ws, sizeof(ws), nullptr) < 0, "apply: null input");
    // unknown operator kind / dtype
    mfx_operator bad = op;
    bad.kind = 17;
    EXPECT(mfx_op_apply(&bad, dummy, 4, dummy, 4, 1, 0, ws, sizeof(ws), nullptr) < 0, "unknown operator kind");
    bad = op;
    bad.dtype = 9;
    EXPECT(mfx_arnoldi_forward(&bad, dummy, 4, 2, 1, 1, dummy, dummy, dummy, dummy, ws, sizeof(ws), nullptr) < 0, "unknown dtype");
    // workspace too small: the query says how much, a smaller buffer must be refused
    const int64_t need = mfx_workspace_bytes(&op, 4, 2, 1);
    EXPECT(need > 0, "workspace query");
    EXPECT(mfx_arnoldi_forward(&op, dummy, 4, 2, 1, 1, dummy, dummy, dummy, dummy, ws, 0, nullptr) == MFX_ERR_WORKSPACE, "workspace too small");
    // eigen-solver bounds (k <= 120), sampler arguments
    EXPECT(mfx_tridiag_eigh(dummy, dummy, 3, 1, 121, MFX_F64, dummy, dummy, nullptr) < 0, "eigh k > 120");
    EXPECT(mfx_tridiag_eigh(dummy, dummy, 3, 1, 0, MFX_F64, dummy, dummy, nullptr) < 0, "eigh k = 0");
    EXPECT(mfx_rademacher(1, 0, 2, 4, 7, dummy, nullptr) < 0, "rademacher dtype");
    EXPECT(mfx_rademacher(1, 0, 2, 4, MFX_F32, nullptr, nullptr) < 0, "rademacher null output");
    // the parameter sweep is not available for callback operators
    mfx_operator cb = make_op(MFX_OP_CALLBACK, MFX_F32, 4);
    mfx_op_grads g;
    std::memset(&g, 0, sizeof(g));
    EXPECT(mfx_op_vjp_params(&cb, dummy, 4, dummy, 4, 1, &g, ws, sizeof(ws), nullptr) < 0, "vjp_params on a callback operator");
    // gather mode of a communicator that is not libmfx's own
    mfx_comm cm;
    std::memset(&cm, 0, sizeof(cm));
    EXPECT(mfx_comm_rccl_gather_mode(&cm, 1) < 0, "gather mode on a foreign communicator");
    EXPECT(mfx_comm_rccl_gather_mode(nullptr, 1) < 0, "gather mode on NULL");
    EXPECT(mfx_comm_destroy_rccl(nullptr) <= 0, "destroy NULL");
  }
  // timing / graph counters: host state only
  {
    double ms = -1;
    int64_t launches = -1, cap = -1, rep = -1;
    EXPECT(mfx_timing_enable(0) == MFX_OK, "timing off");
    EXPECT(mfx_timing_reset() == MFX_OK, "timing reset");
    EXPECT(mfx_timing_read(0, &ms, &launches) == MFX_OK && ms == 0.0 && launches == 0, "timing read");
    EXPECT(mfx_timing_read(99, &ms, &launches) == MFX_OK && launches == 0, "a timing class nobody recorded reads as empty");
    EXPECT(mfx_graph_stats(&cap, &rep) == MFX_OK && cap >= 0 && rep >= 0, "graph stats");
  }
  std::printf(failures ? "cabi_host_checks: %d FAILED\n" : "cabi_host_checks ok\n", failures);
  return failures ? 1 : 0;
}
