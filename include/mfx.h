/*
 * mfx.h -- C-ABI of the MI355X-native Lanczos/Arnoldi-with-adjoint engine (libmfx.so).
 *
 * This is the drop-in boundary for the hot path of pnkraemer/experiments-lanczos-adjoints.  The
 * reference is pure-Python JAX and has no FFI; the entry points below are what a binding of its
 * functional API would call, and each one cites the reference function it replaces
 * (paths relative to /root/reference/src/matfree_extensions).
 *
 * Conventions
 *   - plain pointers + sizes only; all array pointers are DEVICE pointers unless stated otherwise;
 *   - every buffer is owned by the caller (outputs and workspace are pre-allocated; the library keeps
 *     no device allocations between calls);
 *   - every entry point is asynchronous on the given hipStream_t (passed as void*) and re-entrant
 *     across streams; no host synchronisation inside (except callback operators, see below);
 *   - return value 0 = ok, <0 = error (MFX_ERR_*); mfx_last_error() gives a thread-local message;
 *   - batched layout: a batch of p vectors of length n is row-major (p, n); the Krylov basis is
 *     (p, k, n) -- the layout lanczos.tridiag returns per probe (lanczos.py:165, `Q.T`), batched over
 *     probes in place of jax.vmap (hutchinson.py:14,53).
 */
#ifndef MFX_H
#define MFX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFX_VERSION 201 /* 201: + mfx_comm_rccl_count */

enum { MFX_F32 = 0, MFX_F64 = 1 };
enum { MFX_OP_DENSE = 0, MFX_OP_CSR = 1, MFX_OP_RBF = 2, MFX_OP_CALLBACK = 3 };
enum { MFX_REORTHO_NONE = 0, MFX_REORTHO_FULL = 1 };
/* Arithmetic of the fp32 RBF Gram kernels (wide batches; fp64 operators ignore it):
 *   MFX_RBF_FP32          exact fp32 MFMA everywhere (v_mfma_f32_32x32x2_f32, a round-to-nearest fmaf chain)
 *   MFX_RBF_F16X3_MATVEC  Gram matvec emulated on the f16 matrix pipe: operands split hi + lo (2 x 11 bits),
 *                         hi*hi + hi*lo + lo*hi accumulated in fp32; the squared distances are the same 3-product
 *                         split on the f16 pipe, block signs alternating (falls back to fp32-MFMA distances when an
 *                         input exceeds the f16 range); parameter-gradient GEMM exact fp32.
 *   MFX_RBF_F16X3         additionally the parameter-gradient GEMM S = L^T R split the same way; the f16 MFMA's
 *                         truncation bias is decorrelated by pseudo-random column signs on both operands
 *                         (undone in the epilogue).  Fastest mode.
 * Accuracy of the three modes against the fp64 path at the C4 size: DESIGN.md section 3.2 / profiles/r02a_*. */
enum { MFX_RBF_FP32 = 0, MFX_RBF_F16X3_MATVEC = 1, MFX_RBF_F16X3 = 2 };
/* Kernel family of the Gram operator, with s = |x_i/l - x_j/l|^2 clamped at 0 (util/gp_util.py:69-184):
 *   MFX_KERNEL_RBF       sigma exp(-s/2)                                   kernel_scaled_rbf        :151-184
 *   MFX_KERNEL_MATERN12  sigma exp(-r),          r = sqrt(s + eps)         kernel_scaled_matern_12  :110-148
 *   MFX_KERNEL_MATERN32  sigma (1 + r) exp(-r),  r = sqrt(3 s + eps)       kernel_scaled_matern_32  :69-107
 * eps = machine epsilon of the dtype (util/gp_util.py:99,140).  The distance of a point to itself is taken as
 * exactly 0 (the reference's expanded form leaves round-off there, which sqrt amplifies to 3e-4 in fp32). */
enum { MFX_KERNEL_RBF = 0, MFX_KERNEL_MATERN12 = 1, MFX_KERNEL_MATERN32 = 2 };
enum {
  MFX_OK = 0,
  MFX_ERR_INVALID = -1,     /* bad argument (shape, null pointer, depth out of range) */
  MFX_ERR_UNSUPPORTED = -2, /* dtype / operator kind / size not supported by this build */
  MFX_ERR_HIP = -3,         /* a HIP runtime call failed */
  MFX_ERR_WORKSPACE = -4,   /* workspace too small */
  MFX_ERR_CALLBACK = -5     /* a callback operator returned non-zero */
};

/* Callback operator (generic Python `matvec(v, *params)` of the reference, e.g. lanczos.py:28-33).
 * x, y are device pointers to p vectors of length n with leading dimensions ldx, ldy (elements).
 * The callee must enqueue its work on `stream` (or synchronise it itself).
 *   mode 0: y = A x                                            (arnoldi.py:84, lanczos.py:254)
 *   mode 1: y = A^T x, and accumulate d/dtheta [x^T A(theta) aux]   (arnoldi.py:207-209; aux = q_idx)
 *   mode 2: y = A x,   and accumulate d/dtheta [aux^T A(theta) x]   (lanczos.py:328-329; aux = x_j)
 * Parameter-gradient accumulation lives on the callee's side (it owns theta). */
typedef int (*mfx_callback_fn)(void* ctx, int mode, const void* x, int64_t ldx, const void* aux,
                               int64_t ldaux, void* y, int64_t ldy, int64_t p, int64_t n,
                               void* stream);

/* Operator descriptor: a POD of borrowed pointers, valid for the duration of one call.
 * Replaces the reference's `matvec(v, *params)` closures:
 *   DENSE    p @ x                          tests/test_lanczos/test_tridiag_forward.py:18
 *   CSR      BCOO((vals, idx)) @ x          experiments/benchmarks/.../suite_sparse/benchmark.py:64-68
 *   RBF      (K(X,X) + noise I) x           util/gp_util.py:69-184 (RBF / Matern kernels),225-226,525-549
 *   CALLBACK any Python callable                                                              */
typedef struct mfx_operator {
  int32_t kind;  /* MFX_OP_* */
  int32_t dtype; /* MFX_F32 / MFX_F64: element type of vectors, matrix values and X */
  int64_t n;     /* operator is n x n */

  /* DENSE: row-major (n, n), leading dimension lda */
  const void* dense_a;
  int64_t lda;

  /* CSR (int32 indices).  `row` = COO row index per stored value (for the SDDMM gradient).  The
   * transpose structure (CSR of A^T: t_crow, t_col, and t_perm[e] = position of that value in `val`)
   * may be NULL for operators that are only applied forward. */
  const int32_t* crow;
  const int32_t* col;
  const int32_t* row;
  const void* val;
  int64_t nnz;
  int64_t max_row_nnz; /* longest row of A (of A and A^T when the transpose structure is present); 0 = unknown */
  const int32_t* t_crow;
  const int32_t* t_col;
  const int32_t* t_perm;

  /* RBF Gram: X row-major (n, d); constrained hyper-parameters live on the device so that no host
   * synchronisation is needed: lengthscale (d values if ard else 1), outputscale (1), noise (1).
   * d <= 1024 (MFX_ERR_UNSUPPORTED beyond; the reference's kernels take any d, util/gp_util.py:151-184).  Which kernels run: fp32, d <= 32:
   * the f16-split matrix-core kernels (rbf_mode); fp32, up to d = 128 (parameter sweep: 64): the exact-fp32
   * matrix-core kernels in every rbf_mode; everything else (fp64, wider inputs, 1-3 vectors at n < 2048): VALU kernels. */
  const void* x;
  int32_t d;
  int32_t ard;
  int32_t rbf_mode;  /* MFX_RBF_* arithmetic of the fp32 Gram kernels (ignored for fp64) */
  int32_t kernel_fn; /* MFX_KERNEL_*: which stationary kernel the Gram operator evaluates */
  const void* lengthscale;
  const void* outputscale;
  const void* noise;

  /* CALLBACK */
  mfx_callback_fn callback;
  void* ctx;

  /* Row block (multi-GPU row sharding; the reference's own row partition of the Gram matvec is
   * util/gp_util.py:496-509).  nrows == 0: the whole operator.  nrows > 0: mfx_op_apply computes only rows
   * [row0, row0 + nrows) of A x (or of A^T x): x keeps length n, y has nrows entries per vector; and
   * mfx_op_vjp_params sums only those rows: L_b has nrows entries (rows row0.. of the full L_b), R_b length n.
   * The matrix-core Gram kernels need row0 % 64 == 0.  The single-device drivers require nrows == 0. */
  int64_t row0;
  int64_t nrows;
} mfx_operator;

/* Parameter-gradient outputs (device pointers, ACCUMULATED into, caller zero-fills).  NULL = skip.
 *   dense_a (n, n) row-major, leading dimension = n;  val (nnz);
 *   lengthscale (d if ard else 1), outputscale (1), noise (1).
 * d/dtheta of  sum_b L_b^T A(theta) R_b   (arnoldi.py:207-209, lanczos.py:328-329). */
typedef struct mfx_op_grads {
  void* dense_a;
  void* val;
  void* lengthscale;
  void* outputscale;
  void* noise;
} mfx_op_grads;

const char* mfx_last_error(void);
int mfx_version(void);

/* Workspace (bytes) needed by the drivers below for an (n, k, p) problem on this operator. */
int64_t mfx_workspace_bytes(const mfx_operator* op, int64_t n, int64_t k, int64_t p);

/* y_b = A x_b (transpose = 0) or A^T x_b (transpose = 1) for b < p.
 * Replaces one call of the user matvec, vmapped over probes (hutchinson.py:14). */
int mfx_op_apply(const mfx_operator* op, const void* x, int64_t ldx, void* y, int64_t ldy,
                 int64_t p, int transpose, void* ws, int64_t ws_bytes, void* stream);

/* grads += d/dtheta sum_{b<batch} L_b^T A(theta) R_b, with L_b = L + b*ldl, R_b = R + b*ldr.
 * Replaces the parameter half of jax.vjp(matvec) (arnoldi.py:207-209, lanczos.py:328-329), deferred
 * to ONE sweep over all (probe, step) pairs.  Not available for CALLBACK operators. */
int mfx_op_vjp_params(const mfx_operator* op, const void* L, int64_t ldl, const void* R,
                      int64_t ldr, int64_t batch, const mfx_op_grads* grads, void* ws,
                      int64_t ws_bytes, void* stream);

/* arnoldi._forward (arnoldi.py:57-101), batched over p start vectors, live columns only.
 *   v0 (p, n) -> Q (p, k, n) [= reference Q^T], H (p, k, k) row-major, r (p, n) un-normalised,
 *   c (p) = 1/|v0|.  second_pass != 0 runs the re-orthogonalisation pass (reference quirk Q1:
 *   the reference forward always does unless reortho_vjp="none", arnoldi.py:26,91). */
int mfx_arnoldi_forward(const mfx_operator* op, const void* v0, int64_t n, int64_t k, int64_t p,
                        int second_pass, void* Q, void* H, void* r, void* c, void* ws,
                        int64_t ws_bytes, void* stream);

/* arnoldi._forward for COMPLEX vectors (arnoldi.py:57-101 with its .conj() at :66,87,92,95; the case
 * tests/test_arnoldi/test_hessenberg_forward.py:10-37 runs with dtype=complex).  Forward only -- the reference's adjoint is not
 * exercised on complex input either.  All buffers are interleaved (re, im) pairs of the operator's REAL dtype:
 *   v0 (p, n) complex -> Q (p, k, n) complex [= reference Q^T, un-conjugated], H (p, k, k) complex row-major,
 *   r (p, n) complex, c (p) complex (= 1/|v0| + 0 i).
 * `op` is the complex-linear map in its real form: op->n = 2 n, acting on the interleaved vector (a dense complex A becomes the
 * (2 n, 2 n) matrix of 2 x 2 blocks [[Re, -Im], [Im, Re]]; a CALLBACK sees 2 n reals per vector).  The recurrence runs on the
 * real kernels of mfx_arnoldi_forward with two basis slots (q, i q) per complex Krylov vector.
 * Workspace: mfx_complex_workspace_bytes(op, n, k, p). */
int64_t mfx_complex_workspace_bytes(const mfx_operator* op, int64_t n, int64_t k, int64_t p);
int mfx_arnoldi_forward_complex(const mfx_operator* op, const void* v0, int64_t n, int64_t k, int64_t p,
                                int second_pass, void* Q, void* H, void* r, void* c, void* ws,
                                int64_t ws_bytes, void* stream);

/* arnoldi._adjoint (arnoldi.py:104-220).  Cotangents: dQ (p, k, n) or NULL (= 0), dH (p, k, k),
 * dr (p, n) or NULL, dc (p) or NULL.  Outputs dv (p, n); parameter gradients accumulated into
 * `grads` (native operators: one deferred sweep; CALLBACK: inside the callback, per step).
 * reortho = MFX_REORTHO_FULL re-projects lambda every step (arnoldi.py:200-204).
 * Lambda (p, k, n) is caller-provided scratch that holds the adjoint states on return. */
int mfx_arnoldi_adjoint(const mfx_operator* op, int64_t n, int64_t k, int64_t p, const void* Q,
                        const void* H, const void* r, const void* c, const void* dQ,
                        const void* dH, const void* dr, const void* dc, int reortho, void* dv,
                        void* Lambda, const mfx_op_grads* grads, void* ws, int64_t ws_bytes,
                        void* stream);

/* lanczos._forward (lanczos.py:215-285): three-term recurrence, no re-orthogonalisation.
 *   v0 (p, n) -> xs (p, k+1, n), alpha (p, k), beta (p, k), vnorm (p) = |v0|. */
int mfx_lanczos_forward(const mfx_operator* op, const void* v0, int64_t n, int64_t k, int64_t p,
                        void* xs, void* alpha, void* beta, void* vnorm, void* ws,
                        int64_t ws_bytes, void* stream);

/* lanczos._adjoint (lanczos.py:288-335).  dxs (p, k+1, n) or NULL, dalpha (p, k), dbeta (p, k).
 * Outputs dv (p, n); Lambda (p, k, n) scratch (adjoint states).  Applies A (not A^T) as the
 * reference does (quirk Q4, lanczos.py:328). */
int mfx_lanczos_adjoint(const mfx_operator* op, int64_t n, int64_t k, int64_t p, const void* xs,
                        const void* alpha, const void* beta, const void* vnorm, const void* dxs,
                        const void* dalpha, const void* dbeta, void* dv, void* Lambda,
                        const mfx_op_grads* grads, void* ws, int64_t ws_bytes, void* stream);

/* jnp.linalg.eigh of the k x k tridiagonal (lanczos.py:48-53), batched: alpha (p, k),
 * beta (p, k-1 values, leading dimension ldbeta) -> evals (p, k) (unordered), evecs (p, k, k) with
 * evecs[b][i][a] = component i of eigenvector a.  fp64 arithmetic inside.  k <= 120 in either dtype (work matrix in LDS);
 * 120 < k <= 2048 with fp64 buffers only (the rotations are accumulated in evecs itself; MFX_ERR_UNSUPPORTED in fp32 -- the Python
 * layer casts the k x k problem to fp64 there).  The call is asynchronous, so
 * a QL iteration that fails to converge (200 sweeps per eigenvalue) cannot be reported through the return code:
 * that probe's evals are set to NaN instead (never silently wrong numbers). */
int mfx_tridiag_eigh(const void* alpha, const void* beta, int64_t ldbeta, int64_t p, int64_t k,
                     int dtype, void* evals, void* evecs, void* stream);

/* VJP of  value_b = sum_a evecs[b][0][a]^2 f(evals[b][a])  w.r.t. (alpha, beta) -- what autodiff
 * through eigh + vmap(matfun) + dot yields in lanczos.py:53-59, in divided-difference form.
 * fvals = f(evals), dfvals = f'(evals) (p, k); gout (p) upstream cotangent.  k <= 2048 (beyond 120 the divided differences are
 * evaluated where they are used instead of stored in LDS). */
int mfx_slq_quadform_bwd(const void* evals, const void* evecs, const void* fvals,
                         const void* dfvals, const void* gout, int64_t p, int64_t k, int dtype,
                         void* dalpha, void* dbeta, int64_t lddbeta, void* stream);

/* +-1 probes (matfree.hutchinson.sampler_rademacher call sites: util/gp_util.py:557,
 * optim_logml_adjoints_adaptive.py:109).  Counter-based: element (b, i) depends only on
 * (seed, first_probe + b, i), so probe shards on different GPUs form one global probe matrix. */
int mfx_rademacher(uint64_t seed, int64_t first_probe, int64_t p, int64_t n, int dtype, void* out,
                   void* stream);

/* ---- row-sharded Krylov drivers: one process per GPU, rows of every Krylov vector split over `world` ranks ---------
 *
 * Rank r owns rows [r nloc, min(n, (r+1) nloc)) of every vector (start vector, basis, remainder, adjoint states) and of
 * the operator; nloc is the same on every rank, a multiple of 64.  Per Krylov step the library needs
 *   - ONE all-gather of the (p, nloc) iterate (the operator's input is the full vector), and
 *   - sum-all-reduces of the (p, k+1) Gram-Schmidt coefficients / norms (the reference's `Q.T @ v`, arnoldi.py:87-92,
 *     becomes a partial sum per rank),
 * both enqueued on the caller's stream through the two function pointers below.  The host side supplies them:
 * matfree_extensions/distributed.py binds them to torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in tests).
 * Buffers handed to the callbacks lie inside the workspace `ws` or inside `Qfull` of the call in progress.
 * H, c and the coefficient buffers come out identical on every rank; parameter gradients are PARTIAL sums over this
 * rank's rows -- the caller adds them over the ranks (one small all-reduce, folded into the estimator's). */
typedef int (*mfx_allreduce_fn)(void* ctx, void* buf, int64_t count, int dtype, void* stream); /* in place, sum */
typedef int (*mfx_allgather_fn)(void* ctx, const void* in, void* out, int64_t count, int dtype,
                                void* stream); /* out[r * count + i] = in_of_rank_r[i] */
/* Optional neighbour exchange (sparse operators: a rank's rows read only a few entries owned by other ranks -- the halo of
 * a stencil -- so all-gathering the whole iterate moves far more than needed).  `full` holds p vectors of length n with leading
 * dimension ldfull; the library has already copied this rank's own rows (`local`, p rows with leading dimension ldlocal) into
 * their place [rank nloc, ...).  The callback must fill every entry of `full` that this rank's rows of A (transpose = 0) or of
 * A^T (transpose = 1) read and that another rank owns; it may leave the rest untouched (it is never read).  NULL: all-gather. */
typedef int (*mfx_exchange_fn)(void* ctx, const void* local, int64_t ldlocal, void* full, int64_t ldfull, int64_t p,
                               int dtype, int transpose, void* stream);
/* Optional: all p vectors of the iterate at once, straight from the row shards (`local`, leading dimension ldlocal) into the
 * operator input `full` (leading dimension ldfull): full[b][r count + i] = local_of_rank_r[b][i], count = nloc.  Used when every
 * rank owns exactly nloc rows (n == world nloc); saves the pack / unpack copies around `allgather`.  NULL: `allgather`. */
typedef int (*mfx_allgather_rows_fn)(void* ctx, const void* local, int64_t ldlocal, void* full, int64_t ldfull, int64_t p,
                                     int64_t count, int dtype, void* stream);
typedef struct mfx_comm {
  int32_t rank, world;
  int64_t nloc; /* rows per rank (last rank: n - rank * nloc >= 1) */
  mfx_allreduce_fn allreduce_sum;
  mfx_allgather_fn allgather;
  void* ctx;
  mfx_exchange_fn exchange;             /* optional, see above */
  mfx_allgather_rows_fn allgather_rows; /* optional, see above */
} mfx_comm;

/* Native communicator: the function pointers of an mfx_comm filled with C functions of libmfx that issue ncclAllReduce /
 * ncclAllGather (RCCL over xGMI) themselves, on the stream of the driver call -- no host-language callback on the path
 * (round 2 went libmfx -> ctypes -> torch.distributed, ~280 callbacks per SLQ step).  RCCL is loaded at run time
 * (dlopen "librccl.so.1"); MFX_ERR_UNSUPPORTED when it is not there.
 *   mfx_rccl_available   : 1 when librccl and the entry points used here could be bound in this process, else 0 (no GPU call;
 *                          lets every rank of a group agree on native vs. callback collectives BEFORE the collective creation);
 *   mfx_rccl_unique_id   : one rank makes the id (ncclGetUniqueId; id: >= 128 bytes of host memory) and hands it to the others
 *                          by any means (matfree_extensions/distributed.py broadcasts it with torch.distributed);
 *   mfx_comm_create_rccl : COLLECTIVE over the `world` ranks (ncclCommInitRank on the calling thread's current device);
 *   mfx_comm_destroy_rccl: releases the communicator (a no-op for an mfx_comm the caller filled in itself). */
int mfx_rccl_available(void);
int mfx_rccl_unique_id(void* id, int64_t bytes);
int mfx_comm_create_rccl(const void* id, int64_t bytes, int32_t rank, int32_t world, int64_t nloc, mfx_comm* out);
int mfx_comm_destroy_rccl(mfx_comm* comm);
/* How a native communicator moves the (p, nloc) iterate of a Krylov step into the (p, n) operator input -- two legs to measure on
 * an 8-GPU node (same results).  mfx_comm_create_rccl leaves the communicator in MFX_GATHER_GROUPED; the Python layer selects
 * MFX_GATHER_PACKED unless told otherwise (matfree_extensions/distributed.py, $MFX_GATHER):
 *   MFX_GATHER_GROUPED: ONE grouped launch of p ncclAllGather of nloc elements each, straight from the row shards into the
 *                       operator input (no pack / unpack copies; at config 4: 64 operations of 64 KB per rank);
 *   MFX_GATHER_PACKED : k_pack_shard, ONE ncclAllGather of p nloc elements (config 4: 4 MB per rank), k_unshard. */
#define MFX_GATHER_GROUPED 0
#define MFX_GATHER_PACKED 1
int mfx_comm_rccl_gather_mode(mfx_comm* comm, int mode);
/* What RCCL itself says about a native communicator: ncclCommCount -> *ranks, ncclCommUserRank -> *rank (the row group the drivers'
 * collectives really span -- reported by bench.py next to torch.distributed's world size, which is a different library's view).
 * MFX_ERR_INVALID for an mfx_comm the caller filled in itself.  No counterpart in the reference (no collective there). */
int mfx_comm_rccl_count(const mfx_comm* comm, int32_t* ranks, int32_t* rank);

int64_t mfx_sharded_workspace_bytes(const mfx_operator* op, const mfx_comm* comm, int64_t n, int64_t k, int64_t p);

/* mfx_arnoldi_forward on row shards.  v0, r: (p, nrows) packed; Q: (p, k, nrows); Qfull: (p, k, n) receives the
 * all-gathered basis (every rank holds all of it afterwards: it is the operator's input in the forward pass anyway, and
 * the R operand of the parameter-gradient sweep in the adjoint); H (p, k, k), c (p) replicated. */
int mfx_arnoldi_forward_sharded(const mfx_operator* op, const mfx_comm* comm, const void* v0, int64_t n, int64_t k,
                                int64_t p, int second_pass, void* Q, void* Qfull, void* H, void* r, void* c, void* ws,
                                int64_t ws_bytes, void* stream);

/* mfx_arnoldi_adjoint on row shards.  Q, dQ, Lambda: (p, k, nrows); r, dr, dv: (p, nrows); Qfull (p, k, n) from the
 * forward pass; H, dH, c, dc replicated.  grads: partial sums over this rank's rows (see above). */
int mfx_arnoldi_adjoint_sharded(const mfx_operator* op, const mfx_comm* comm, int64_t n, int64_t k, int64_t p,
                                const void* Q, const void* Qfull, const void* H, const void* r, const void* c,
                                const void* dQ, const void* dH, const void* dr, const void* dc, int reortho, void* dv,
                                void* Lambda, const mfx_op_grads* grads, void* ws, int64_t ws_bytes, void* stream);

/* The three-term recurrence and its adjoint (mfx_lanczos_forward / mfx_lanczos_adjoint below; lanczos.py:231-335) on row
 * shards.  Vectors hold this rank's rows: v0, dv (p, nrows); xs, dxs (p, k + 1, nrows); Lambda (p, k, nrows).  alpha, beta,
 * vnorm and their cotangents are replicated.  Lambdafull (p, k, n) receives the gathered adjoint states -- the operator input
 * of every adjoint step and the right factor of the parameter-gradient sweep, whose result (`grads`) is this rank's PARTIAL sum
 * over its rows: the caller adds the ranks up.  Collectives per step: one all-gather of the iterate, the dots as small
 * all-reduces.  Workspace: mfx_sharded_workspace_bytes. */
int mfx_lanczos_forward_sharded(const mfx_operator* op, const mfx_comm* comm, const void* v0, int64_t n, int64_t k,
                                int64_t p, void* xs, void* alpha, void* beta, void* vnorm, void* ws, int64_t ws_bytes,
                                void* stream);
int mfx_lanczos_adjoint_sharded(const mfx_operator* op, const mfx_comm* comm, int64_t n, int64_t k, int64_t p,
                                const void* xs, const void* alpha, const void* beta, const void* vnorm, const void* dxs,
                                const void* dalpha, const void* dbeta, void* dv, void* Lambda, void* Lambdafull,
                                const mfx_op_grads* grads, void* ws, int64_t ws_bytes, void* stream);

/* ---- linear solves ("next" tier: the Mahalanobis half of the GP log-marginal likelihood) -------------------------
 *
 * (Preconditioned) conjugate gradients on a (p, n) batch of right-hand sides  (cg.py:19-60 pcg_fixed_step,
 * :74-137 pcg_adaptive; _safe_divide :222-241).  adaptive = 0: exactly `maxiter` iterations (num_matvecs);
 * adaptive = 1: per right-hand side, iterate while  rms(r / (atol + |x| rtol)) > 1  or steps < miniter, and
 * steps < maxiter; right-hand sides that stop are frozen, so a batch equals p independent solves.
 * Optional preconditioner: the Woodbury solve of  s I + L L^T  (low_rank.py:31-43) given Lt = L^T (rank, n)
 * row-major, minv = (s I + L^T L)^{-1} (rank, rank) and the device scalar s; precond_lt = NULL: none.
 * Outputs: x, r (p, n) packed (final iterate and residual, cg.py:39), num_steps int64 (p) or NULL.
 * Differentiation follows jax.lax.custom_linear_solve (cg.py:23-25): the caller solves again with the cotangent
 * and feeds (-lambda, x) to mfx_op_vjp_params. */
int64_t mfx_pcg_workspace_bytes(const mfx_operator* op, int64_t n, int64_t p, int64_t rank);
int mfx_pcg_solve(const mfx_operator* op, const void* b, int64_t ldb, int64_t n, int64_t p,
                  const void* precond_lt, int64_t rank, const void* precond_minv,
                  const void* precond_shift, int64_t maxiter, int64_t miniter, double atol, double rtol,
                  int adaptive, void* x, void* r, void* num_steps, void* ws, int64_t ws_bytes,
                  void* stream);

/* mfx_pcg_solve on row shards (cg.py:19-137, rows partitioned as util/gp_util.py:496-509): b, x, r hold this rank's rows
 * (p, nrows; ldb >= nrows); precond_lt holds this rank's COLUMNS of L^T, (rank, nrows) row-major; minv, shift replicated (built
 * from the whole L).  Per iteration: one all-gather of the search direction, and p.Ap, r.z, the adaptive error sum and L^T r as
 * small all-reduces -- every rank sees the same scalars, takes the same steps and stops at the same iteration.  num_steps is
 * replicated.  The re-orthogonalising variant is not sharded. */
int64_t mfx_pcg_sharded_workspace_bytes(const mfx_operator* op, const mfx_comm* comm, int64_t n, int64_t p, int64_t rank);
int mfx_pcg_solve_sharded(const mfx_operator* op, const mfx_comm* comm, const void* b, int64_t ldb, int64_t n, int64_t p,
                          const void* precond_lt, int64_t rank, const void* precond_minv, const void* precond_shift,
                          int64_t maxiter, int64_t miniter, double atol, double rtol, int adaptive, void* x, void* r,
                          void* num_steps, void* ws, int64_t ws_bytes, void* stream);

/* Fixed-step PCG that re-orthogonalises the residual against the stored, normalised earlier residuals each step
 * (cg.pcg_fixed_step_reortho, cg.py:140-219): q (p, num_matvecs, n) receives the rows r_i / sqrt(r_i . z_i), the
 * transpose of the reference's info["Q"].  Workspace: mfx_pcg_workspace_bytes with rank = max(rank, num_matvecs). */
int mfx_pcg_solve_reortho(const mfx_operator* op, const void* b, int64_t ldb, int64_t n, int64_t p,
                          const void* precond_lt, int64_t rank, const void* precond_minv,
                          const void* precond_shift, int64_t num_matvecs, void* x, void* r, void* q,
                          void* ws, int64_t ws_bytes, void* stream);

/* z = (v - L (s I + L^T L)^{-1} L^T v) / s on a (p, n) batch: the `solve(v, s)` of low_rank.py:31-43.
 * Workspace: mfx_pcg_workspace_bytes of any operator of this n and dtype. */
int mfx_precond_apply(int dtype, int64_t n, int64_t rank, const void* lt, const void* minv,
                      const void* shift, const void* v, int64_t ldv, void* z, int64_t ldz, int64_t p,
                      void* ws, int64_t ws_bytes, void* stream);

/* Partial Cholesky factor of a dense or kernel-Gram operator, element access instead of the reference's
 * lazy_kernel(i, j) callable (low_rank.py:63-120 without pivoting, :123-228 with pivoting).  Output Lt (rank, n)
 * = the reference's factor transposed, already in the original row order (low_rank.py:227-228); pivots int64
 * (rank); success int32 (all pivots positive, :205).  with_noise adds the operator's noise to the diagonal
 * (likelihood_pdf, util/gp_util.py:225-226) or leaves it out (likelihood_pdf_p, :253-254). */
int mfx_partial_cholesky(const mfx_operator* op, int64_t rank, int pivot, int with_noise, void* lt,
                         void* pivots, void* success, void* ws, int64_t ws_bytes, void* stream);

/* y (p, m) = K(X_new, X) v for a kernel-Gram operator: the cross-covariance matvec of the posterior mean
 * (likelihood_condition[_p], util/gp_util.py:299-305,338-344: `matvec(kernel)(xs, inputs, weights)`), no noise
 * term.  xnew (m, d) row-major in the operator's dtype; v (p, n). */
int64_t mfx_gram_cross_workspace_bytes(const mfx_operator* op, int64_t m);
int mfx_gram_cross_apply(const mfx_operator* op, const void* xnew, int64_t m, const void* v, int64_t ldv,
                         void* y, int64_t ldy, int64_t p, void* ws, int64_t ws_bytes, void* stream);

/* Per-kernel-class device timing with hipEvents recorded on the caller's stream (no host syncs
 * while enabled; events are read back in mfx_timing_read, which synchronises the events).
 * classes: 0 = operator apply, 1 = operator parameter-gradient sweep, 2 = Krylov vector kernels. */
int mfx_timing_enable(int enable);
int mfx_timing_reset(void);
int mfx_timing_read(int cls, double* total_ms, int64_t* launches);

/* hipGraph replay of launch-bound driver calls.  The Krylov drivers enqueue a fixed sequence of launches that
 * depends only on their arguments; for small problems (fewer than 256 vector-kernel workgroups, native operator,
 * timing off) the second call with identical arguments is captured into a hipGraph and later ones are one
 * hipGraphLaunch on the caller's stream.  The reference's counterpart is jax.jit's compiled executable
 * (experiments/benchmarks/wall_times_vjp_through_lanczos_arnoldi/suite_sparse/benchmark.py:91-121 times the
 * jitted function).  Environment MFX_GRAPHS=0 disables it.  Counters since load: calls captured, calls replayed. */
int mfx_graph_stats(int64_t* captured, int64_t* replayed);

#ifdef __cplusplus
}
#endif
#endif /* MFX_H */
