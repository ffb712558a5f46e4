// libmfx: the Arnoldi forward loop and its adjoint scan as ONE cooperative kernel each (dense and CSR operators).
// EXPERIMENTAL, OFF BY DEFAULT (MFX_FUSED=1 enables it): correct (the parity suite passes with it), but slower than the
// separate kernels on MI355X -- kept with its measurements because the reason is instructive.
//
// Idea: for a single start vector of moderate length (BASELINE config 3: n = 102400, k = 50, fp64; config 1: n = 512) every
// vector kernel of mfx_krylov.hip runs for a few microseconds and the k-loop is a chain of ~6 dependent launches per step
// (57 us per step, 17 % of the HBM roofline).  Here the whole k-loop is one launch; the steps are separated by grid-wide
// barriers (cooperative groups), the running vector of a workgroup's slice stays in REGISTERS from one phase to the next, and
// the operator is applied inside the kernel (rows of the slice, gathered from the published iterate).
//
// Measured (profiles/r02e_fused_arnoldi_and_grid_barrier.log): a grid-wide barrier costs 7.3 us for 50 workgroups and 32 us
// for 256 (cg::grid.sync(); a hand-written one-atomic-per-workgroup barrier 4.2 / 19 us: ~75 ns per arriving workgroup,
// serialised on one L2 atomic) -- MORE than the launch gap it replaces (~5 us), and with 3 barriers per step the fused C3
// forward takes 5.3 ms against 2.8 ms for the separate kernels (C1: 3.5 vs 0.85 ms, its single workgroup applies the whole
// dense operator alone).  A grid barrier pays off on this part only below ~10 workgroups; the launch-bound configs need
// fewer dependent launches per step instead.
//
//   forward step i (arnoldi.py:80-99), 3 barriers (2 without the second Gram-Schmidt pass):
//     S   len = |w| from the norm partials; q_i = w / len -> Q[i]; w <- A w / len (rows of this slice); partial dots Q^T w
//     -- barrier --
//     U1  h = sum of partials -> H[:, i]; w -= Q h; partial dots of the second pass
//     -- barrier --
//     U2  w -= Q h2; norm partial; publish w (the next step's operator input / the remainder r)
//     -- barrier --
//   adjoint step idx (arnoldi.py:200-220), 3 barriers (2 without re-projection):
//     A   partial dots Q[:m]^T lam                                           -- barrier --
//     B   lam -= Q (P lam - dH[:, idx]) -> Lambda[idx]; publish lam          -- barrier --
//     C   z <- A^T lam (rows of this slice); partial dots Q[:idx+1]^T z       -- barrier --
//     D   Gamma row, xi, the next lam (registers): no barrier needed before A of the next step
//
// Same slicing (2048 elements per workgroup, 8 per thread), same partial-sum layout and the same arithmetic order as the
// separate kernels -- the two paths agree to round-off.  Only launched when the grid (slices x vectors) is co-resident.
#include <hip/hip_cooperative_groups.h>
#include <stdlib.h>

#include "mfx_vec.h"

namespace cg = cooperative_groups;

namespace mfx {

template <typename T>
struct FusedOp {
  int kind;  // MFX_OP_DENSE / MFX_OP_CSR
  int64_t n;
  const T* dense_a;
  int64_t lda;
  const int32_t *crow, *col, *perm;  // forward: (crow, col, null); transpose: (t_crow, t_col, t_perm)
  const T* val;
  int transpose;
};

// ybuf[r] (LDS, r < 2048) = scale * (A x)[slice0 + r]  (or A^T x); x = one published vector in global memory
template <typename T>
__device__ __forceinline__ void fused_apply(const FusedOp<T>& op, const T* __restrict__ x, T scale, T* __restrict__ ybuf,
                                            int64_t slice0) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t n = op.n;
  const int rows = (int)((n - slice0) < kSlice ? (n - slice0) : kSlice);
  if (op.kind == MFX_OP_CSR) {
    const int sub = tid & 7;
    for (int r = tid >> 3; r < rows; r += kBlock / 8) {  // 8 lanes per row, as k_csr_apply
      const int64_t row = slice0 + r;
      T acc = T(0);
      for (int32_t e = op.crow[row] + sub; e < op.crow[row + 1]; e += 8) acc += (op.perm ? op.val[op.perm[e]] : op.val[e]) * x[op.col[e]];
      acc += __shfl_down(acc, 4, 8);
      acc += __shfl_down(acc, 2, 8);
      acc += __shfl_down(acc, 1, 8);
      if (sub == 0) ybuf[r] = scale * acc;
    }
  } else if (!op.transpose) {  // dense, one wave per row (as k_dense_apply)
    for (int r = wid; r < rows; r += kBlock / 64) {
      const T* arow = op.dense_a + (slice0 + r) * op.lda;
      T acc = T(0);
      for (int64_t j = lane; j < n; j += 64) acc += arow[j] * x[j];
      acc = wave_sum(acc);
      if (lane == 0) ybuf[r] = scale * acc;
    }
  } else {  // dense transpose: thread = column of A, coalesced over the threads
    for (int r = tid; r < rows; r += kBlock) {
      const T* acol = op.dense_a + slice0 + r;
      T acc = T(0);
      for (int64_t i = 0; i < n; ++i) acc += acol[i * op.lda] * x[i];
      ybuf[r] = scale * acc;
    }
  }
}

// the elements of the slice this thread owns, out of the LDS image written by fused_apply (same ownership as load_own)
template <typename T, int VEC>
__device__ __forceinline__ void own_from_lds(T (&dst)[kEpt], const T* __restrict__ ybuf, int64_t slice0, int64_t n, int tid) {
  constexpr int U = kEpt / VEC;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int loc = (u * kBlock + tid) * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) dst[u * VEC + e] = (slice0 + loc + e < n) ? ybuf[loc + e] : T(0);
  }
}

// partial[b][j][blk] = rows_j[slice] . xr, j < m   (body of k_dots)
template <typename T, int VEC>
__device__ __forceinline__ void dots_phase(const T* __restrict__ rb, int64_t row_stride, int m, const T (&xr)[kEpt], int64_t slice0,
                                           int64_t n, T* __restrict__ sm, T* __restrict__ partial_bj /* + (b kmax) nblk */, int nblk,
                                           int blk) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int JT = 4;
  int j = 0;
  for (; j + JT <= m; j += JT) {
    T rr[JT][kEpt];
#pragma unroll
    for (int q = 0; q < JT; ++q) load_own<T, VEC>(rr[q], rb + (int64_t)(j + q) * row_stride, slice0, n, tid);
#pragma unroll
    for (int q = 0; q < JT; ++q) {
      T acc = T(0);
#pragma unroll
      for (int e = 0; e < kEpt; ++e) acc += rr[q][e] * xr[e];
      acc = wave_sum(acc);
      if (lane == 0) sm[wid * m + j + q] = acc;
    }
  }
  for (; j < m; ++j) {
    T rr[kEpt];
    load_own<T, VEC>(rr, rb + (int64_t)j * row_stride, slice0, n, tid);
    T acc = T(0);
#pragma unroll
    for (int e = 0; e < kEpt; ++e) acc += rr[e] * xr[e];
    acc = wave_sum(acc);
    if (lane == 0) sm[wid * m + j] = acc;
  }
  __syncthreads();
  for (int jj = tid; jj < m; jj += kBlock) {
    T sum = T(0);
    for (int w = 0; w < kBlock / 64; ++w) sum += sm[w * m + jj];
    partial_bj[(int64_t)jj * nblk + blk] = sum;
  }
  __syncthreads();  // sm is reused by the next phase
}

// xr -= sum_j coef[j] rows_j[slice]   (body of k_update)
template <typename T, int VEC>
__device__ __forceinline__ void axpy_phase(const T* __restrict__ rb, int64_t row_stride, int m, const T* __restrict__ coef,
                                           T (&xr)[kEpt], int64_t slice0, int64_t n) {
  const int tid = threadIdx.x;
  constexpr int JT = 4;
  int j = 0;
  for (; j + JT <= m; j += JT) {
    T rr[JT][kEpt];
#pragma unroll
    for (int q = 0; q < JT; ++q) load_own<T, VEC>(rr[q], rb + (int64_t)(j + q) * row_stride, slice0, n, tid);
#pragma unroll
    for (int q = 0; q < JT; ++q) {
      const T c = coef[j + q];
#pragma unroll
      for (int e = 0; e < kEpt; ++e) xr[e] -= c * rr[q][e];
    }
  }
  for (; j < m; ++j) {
    T rr[kEpt];
    load_own<T, VEC>(rr, rb + (int64_t)j * row_stride, slice0, n, tid);
    const T c = coef[j];
#pragma unroll
    for (int e = 0; e < kEpt; ++e) xr[e] -= c * rr[e];
  }
}

// coef[j] = s1 * sum_blk partial[b][j][:] + s2 * extra[j * extra_stride]  (prologue of k_update); every thread of the workgroup calls
template <typename T>
__device__ __forceinline__ void coef_phase(const T* __restrict__ partial_bj, int nblk, int m, T s1, const T* __restrict__ extra,
                                           int64_t extra_stride, T s2, T* __restrict__ coef) {
  for (int idx = threadIdx.x; idx < m * kRedG; idx += kBlock) {
    const int j = idx / kRedG, g = idx % kRedG;
    T c = T(0);
    if (partial_bj) c = s1 * reduce_partials_group<T, kRedG>(partial_bj + (int64_t)j * nblk, nblk, g);
    if (g == 0) {
      if (extra) c += s2 * extra[(int64_t)j * extra_stride];
      coef[j] = c;
    }
  }
  __syncthreads();
}

template <typename T>
__device__ __forceinline__ void norm_partial_phase(const T (&xr)[kEpt], T* __restrict__ smn, T* __restrict__ pn_b, int blk) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  T acc = T(0);
#pragma unroll
  for (int e = 0; e < kEpt; ++e) acc += xr[e] * xr[e];
  acc = wave_sum(acc);
  if (lane == 0) smn[wid] = acc;
  __syncthreads();
  if (tid == 0) {
    T sum = T(0);
    for (int w = 0; w < kBlock / 64; ++w) sum += smn[w];
    pn_b[blk] = sum;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
template <typename T>
struct FusedFwd {
  FusedOp<T> op;
  int64_t n;
  int k, second_pass, kmax, nblk;
  const T* v0;
  T *Q, *H, *r, *cinv, *xbuf, *P1, *P2, *PN;
};

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_arnoldi_fwd_fused(FusedFwd<T> a) {
  cg::grid_group grid = cg::this_grid();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* ybuf = reinterpret_cast<T*>(smem_raw);  // [kSlice]
  T* coef = ybuf + kSlice;                    // [kmax]
  T* sm = coef + a.kmax;                      // [4 kmax + 8]
  __shared__ T bc[2];
  const int tid = threadIdx.x, blk = blockIdx.x, b = blockIdx.y;
  const int64_t n = a.n, slice0 = (int64_t)blk * kSlice;
  const int k = a.k, kmax = a.kmax, nblk = a.nblk;
  T* Qb = a.Q + (int64_t)b * k * n;
  T* Hb = a.H + (int64_t)b * k * k;
  T* xb = a.xbuf + (int64_t)b * n;
  T* P1b = a.P1 + (int64_t)b * kmax * nblk;
  T* P2b = a.P2 + (int64_t)b * kmax * nblk;
  T* PNb = a.PN + (int64_t)b * nblk;

  T w[kEpt];
  load_own<T, VEC>(w, a.v0 + (int64_t)b * n, slice0, n, tid);
  norm_partial_phase<T>(w, sm, PNb, blk);
  store_own<T, VEC>(w, xb, slice0, n, tid);
  for (int ij = tid + blk * kBlock; ij < k * k; ij += kBlock * nblk) Hb[ij] = T(0);  // the driver's memset of H
  grid.sync();
  for (int i = 0; i < k; ++i) {
    // ---- S: normalise, store q_i, apply the operator to the published w (scaled), first-pass dots
    if (tid < 64) {
      const T len = sqrt(reduce_partials_group<T, 64>(PNb, nblk, tid));
      if (tid == 0) {
        bc[0] = len;
        bc[1] = T(1) / len;
        if (blk == 0) {
          if (i == 0) a.cinv[b] = T(1) / len;
          else Hb[(int64_t)i * k + (i - 1)] = len;
        }
      }
    }
    __syncthreads();
    const T inv = bc[1];
    {
      T q[kEpt];
#pragma unroll
      for (int e = 0; e < kEpt; ++e) q[e] = w[e] * inv;
      store_own<T, VEC>(q, Qb + (int64_t)i * n, slice0, n, tid);
    }
    fused_apply<T>(a.op, xb, inv, ybuf, slice0);
    __syncthreads();
    own_from_lds<T, VEC>(w, ybuf, slice0, n, tid);
    const int m = i + 1;
    dots_phase<T, VEC>(Qb, n, m, w, slice0, n, sm, P1b, nblk, blk);
    grid.sync();
    // ---- U1: h, H[:, i], first Gram-Schmidt pass (+ dots of the second)
    coef_phase<T>(P1b, nblk, m, T(1), nullptr, 0, T(0), coef);
    if (blk == 0)
      for (int j = tid; j < m; j += kBlock) Hb[(int64_t)j * k + i] = coef[j];
    axpy_phase<T, VEC>(Qb, n, m, coef, w, slice0, n);
    if (a.second_pass) {
      __syncthreads();
      dots_phase<T, VEC>(Qb, n, m, w, slice0, n, sm, P2b, nblk, blk);
      grid.sync();
      // ---- U2: second pass (its coefficients are not added to h, arnoldi.py:92)
      coef_phase<T>(P2b, nblk, m, T(1), nullptr, 0, T(0), coef);
      axpy_phase<T, VEC>(Qb, n, m, coef, w, slice0, n);
    }
    __syncthreads();
    norm_partial_phase<T>(w, sm, PNb, blk);
    store_own<T, VEC>(w, xb, slice0, n, tid);
    if (i + 1 == k) store_own<T, VEC>(w, a.r + (int64_t)b * n, slice0, n, tid);  // the un-normalised remainder (Q2)
    grid.sync();
  }
}

// ------------------------------------------------------------------------------------------------
template <typename T>
struct FusedAdj {
  FusedOp<T> opt;  // the TRANSPOSED operator
  int64_t n;
  int k, reortho, kmax, nblk;
  const T *Q, *H, *r, *cinv, *dQ, *dH, *pig, *eta;
  T *lam0;  // (p, n) initial lambda (from the set-up kernels); reused as the publish buffer
  T *Lam, *Gam, *dv, *P1, *P2;
};

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_arnoldi_adj_fused(FusedAdj<T> a) {
  cg::grid_group grid = cg::this_grid();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* ybuf = reinterpret_cast<T*>(smem_raw);  // [kSlice]
  T* coef = ybuf + kSlice;                    // [kmax]   g_j of the combine
  T* hp = coef + a.kmax;                      // [kmax]   H[idx][j], j > idx
  T* sm = hp + a.kmax;                        // [4 kmax + 8]
  const int tid = threadIdx.x, blk = blockIdx.x, b = blockIdx.y;
  const int64_t n = a.n, slice0 = (int64_t)blk * kSlice;
  const int k = a.k, kmax = a.kmax, nblk = a.nblk;
  const int64_t kk = (int64_t)k * k;
  const T* Qb = a.Q + (int64_t)b * k * n;
  const T* Hb = a.H + (int64_t)b * kk;
  const T* dHb = a.dH + (int64_t)b * kk;
  T* Lb = a.Lam + (int64_t)b * k * n;
  T* Gb = a.Gam + (int64_t)b * kk;
  T* xb = a.lam0 + (int64_t)b * n;
  T* P1b = a.P1 + (int64_t)b * kmax * nblk;
  T* P2b = a.P2 + (int64_t)b * kmax * nblk;

  T lam[kEpt];
  load_own<T, VEC>(lam, xb, slice0, n, tid);
  for (int idx = k - 1; idx >= 0; --idx) {
    if (a.reortho) {
      // ---- A / B: re-projection (arnoldi.py:200-204), rows of P not yet masked
      const int m = (idx + 2 < k) ? idx + 2 : k;
      dots_phase<T, VEC>(Qb, n, m, lam, slice0, n, sm, P1b, nblk, blk);
      grid.sync();
      coef_phase<T>(P1b, nblk, m, T(1), dHb + idx, k, T(-1), coef);
      axpy_phase<T, VEC>(Qb, n, m, coef, lam, slice0, n);
    }
    store_own<T, VEC>(lam, Lb + (int64_t)idx * n, slice0, n, tid);  // Lambda[:, idx]
    store_own<T, VEC>(lam, xb, slice0, n, tid);                      // published for the transposed operator
    grid.sync();
    // ---- C: z = A^T lambda (rows of this slice), dots with Q[:idx+1]
    fused_apply<T>(a.opt, xb, T(1), ybuf, slice0);
    __syncthreads();
    T z[kEpt];
    own_from_lds<T, VEC>(z, ybuf, slice0, n, tid);
    dots_phase<T, VEC>(Qb, n, idx + 1, z, slice0, n, sm, P2b, nblk, blk);
    grid.sync();
    // ---- D: combine (k_adj_combine)
    for (int jx = tid; jx < k * kRedG; jx += kBlock) {
      const int j = jx / kRedG, gl = jx % kRedG;
      T gj;
      if (j <= idx) {
        const T zq = reduce_partials_group<T, kRedG>(P2b + (int64_t)j * nblk, nblk, gl);
        if (gl != 0) continue;
        const T low = (j < idx) ? T(1) : T(0.5);
        const T gam = low * (a.pig[(int64_t)b * kk + (int64_t)idx * k + j] - zq);
        if (blk == 0) Gb[(int64_t)idx * k + j] = gam;
        gj = (j < idx) ? gam : T(2) * gam;
        hp[j] = T(0);
      } else {
        if (gl != 0) continue;
        gj = Gb[(int64_t)j * k + idx];  // written in step j by workgroup 0, two barriers ago at least
        hp[j] = Hb[(int64_t)idx * k + j];
      }
      coef[j] = gj;
    }
    __syncthreads();
    const T alpha = Hb[(int64_t)idx * k + idx];
    const T bminus = (idx == 0) ? T(1) : Hb[(int64_t)idx * k + idx - 1];
    const T eta_i = a.eta[(int64_t)b * k + idx];
    T acc[kEpt], t[kEpt];
#pragma unroll
    for (int e = 0; e < kEpt; ++e) acc[e] = z[e] - alpha * lam[e];
    load_own<T, VEC>(t, a.r + (int64_t)b * n, slice0, n, tid);
#pragma unroll
    for (int e = 0; e < kEpt; ++e) acc[e] += eta_i * t[e];
    if (a.dQ) {
      load_own<T, VEC>(t, a.dQ + ((int64_t)b * k + idx) * n, slice0, n, tid);
#pragma unroll
      for (int e = 0; e < kEpt; ++e) acc[e] += t[e];
    }
    // acc += sum_j g_j q_j  ==  acc -= sum_j (-g_j) q_j ; acc -= sum_{j > idx} H[idx][j] Lam_j
    for (int j = tid; j < k; j += kBlock) coef[j] = -coef[j];
    __syncthreads();
    axpy_phase<T, VEC>(Qb, n, k, coef, acc, slice0, n);
    if (idx + 1 < k) axpy_phase<T, VEC>(Lb + (int64_t)(idx + 1) * n, n, k - idx - 1, hp + idx + 1, acc, slice0, n);
    const T inv = T(1) / bminus;
#pragma unroll
    for (int e = 0; e < kEpt; ++e) lam[e] = acc[e] * inv;
    __syncthreads();  // coef / hp / sm are rewritten by the next step
  }
  const T c = a.cinv[b];
#pragma unroll
  for (int e = 0; e < kEpt; ++e) lam[e] *= c;
  store_own<T, VEC>(lam, a.dv + (int64_t)b * n, slice0, n, tid);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool fused_enabled() {
  static const int v = [] {
    const char* e = getenv("MFX_FUSED");
    return e ? atoi(e) : 0;  // off by default: slower than the separate kernels on MI355X (see the header)
  }();
  return v != 0;
}

static size_t fused_smem(int64_t k, size_t es, bool adjoint) { return (size_t)(kSlice + (adjoint ? 2 : 1) * (k + 1) + 4 * (k + 1) + 8) * es; }

template <typename K>
static bool coresident(K kernel, dim3 grid, size_t smem) {
  int dev = 0, per_cu = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  int coop = 0;
  if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess || !coop) return false;
  if (smem > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
    return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, smem) != hipSuccess) return false;
  return (int64_t)per_cu * cus >= (int64_t)grid.x * grid.y;
}

// 1 = launched, 0 = the runtime says the grid cannot be co-resident after all (take the separate kernels), < 0 = error
static int coop_launch(const void* kernel, dim3 grid, void** args, size_t smem, hipStream_t stream) {
  const hipError_t e = hipLaunchCooperativeKernel(kernel, grid, dim3(kBlock), args, (unsigned)smem, stream);
  if (e == hipErrorCooperativeLaunchTooLarge) {
    (void)hipGetLastError();
    return 0;
  }
  MFX_CHECK_HIP(e);
  return 1;
}

template <typename T>
static bool fill_op(const mfx_operator* op, int transpose, FusedOp<T>* f) {
  if (op->nrows != 0) return false;
  f->kind = op->kind;
  f->n = op->n;
  f->dense_a = (const T*)op->dense_a;
  f->lda = op->lda;
  f->val = (const T*)op->val;
  f->transpose = transpose;
  f->perm = nullptr;
  if (op->kind == MFX_OP_DENSE) return op->dense_a != nullptr;
  if (op->kind != MFX_OP_CSR || !op->crow || !op->col || !op->val) return false;
  if (!transpose) {
    f->crow = op->crow;
    f->col = op->col;
  } else {
    if (!op->t_crow || !op->t_col || !op->t_perm) return false;
    f->crow = op->t_crow;
    f->col = op->t_col;
    f->perm = op->t_perm;
  }
  return true;
}

// returns 1 when the fused kernel was launched, 0 when the caller should take the separate kernels, < 0 on error
template <typename T>
int arnoldi_forward_fused(const mfx_operator* op, const T* v0, int64_t n, int64_t k, int64_t p, int second_pass, T* Q, T* H,
                          T* r, T* cinv, T* xbuf, T* P1, T* P2, T* PN, int vec, hipStream_t stream) {
  if (!fused_enabled() || (op->kind != MFX_OP_DENSE && op->kind != MFX_OP_CSR) || k + 1 > 512) return 0;
  FusedFwd<T> a{};
  if (!fill_op<T>(op, 0, &a.op)) return 0;
  a.n = n; a.k = (int)k; a.second_pass = second_pass; a.kmax = (int)(k + 1);
  a.nblk = (int)((n + kSlice - 1) / kSlice);
  a.v0 = v0; a.Q = Q; a.H = H; a.r = r; a.cinv = cinv; a.xbuf = xbuf; a.P1 = P1; a.P2 = P2; a.PN = PN;
  const dim3 grid((unsigned)a.nblk, (unsigned)p);
  const size_t smem = fused_smem(k, sizeof(T), false);
  void* args[] = {&a};
  constexpr int V = VecWidth<T>::value;
  if (vec > 1) {
    if (!coresident(k_arnoldi_fwd_fused<T, V>, grid, smem)) return 0;
    return coop_launch(reinterpret_cast<const void*>(k_arnoldi_fwd_fused<T, V>), grid, args, smem, stream);
  }
  if (!coresident(k_arnoldi_fwd_fused<T, 1>, grid, smem)) return 0;
  return coop_launch(reinterpret_cast<const void*>(k_arnoldi_fwd_fused<T, 1>), grid, args, smem, stream);
}

template <typename T>
int arnoldi_adjoint_fused(const mfx_operator* op, int64_t n, int64_t k, int64_t p, const T* Q, const T* H, const T* r,
                          const T* cinv, const T* dQ, const T* dH, const T* pig, const T* eta, int reortho, T* lam0, T* Lam,
                          T* Gam, T* dv, T* P1, T* P2, int vec, hipStream_t stream) {
  if (!fused_enabled() || (op->kind != MFX_OP_DENSE && op->kind != MFX_OP_CSR) || k + 1 > 512) return 0;
  FusedAdj<T> a{};
  if (!fill_op<T>(op, 1, &a.opt)) return 0;
  a.n = n; a.k = (int)k; a.reortho = reortho == MFX_REORTHO_FULL; a.kmax = (int)(k + 1);
  a.nblk = (int)((n + kSlice - 1) / kSlice);
  a.Q = Q; a.H = H; a.r = r; a.cinv = cinv; a.dQ = dQ; a.dH = dH; a.pig = pig; a.eta = eta;
  a.lam0 = lam0; a.Lam = Lam; a.Gam = Gam; a.dv = dv; a.P1 = P1; a.P2 = P2;
  const dim3 grid((unsigned)a.nblk, (unsigned)p);
  const size_t smem = fused_smem(k, sizeof(T), true);
  void* args[] = {&a};
  constexpr int V = VecWidth<T>::value;
  if (vec > 1) {
    if (!coresident(k_arnoldi_adj_fused<T, V>, grid, smem)) return 0;
    return coop_launch(reinterpret_cast<const void*>(k_arnoldi_adj_fused<T, V>), grid, args, smem, stream);
  }
  if (!coresident(k_arnoldi_adj_fused<T, 1>, grid, smem)) return 0;
  return coop_launch(reinterpret_cast<const void*>(k_arnoldi_adj_fused<T, 1>), grid, args, smem, stream);
}

template int arnoldi_forward_fused<float>(const mfx_operator*, const float*, int64_t, int64_t, int64_t, int, float*, float*, float*,
                                          float*, float*, float*, float*, float*, int, hipStream_t);
template int arnoldi_forward_fused<double>(const mfx_operator*, const double*, int64_t, int64_t, int64_t, int, double*, double*,
                                           double*, double*, double*, double*, double*, double*, int, hipStream_t);
template int arnoldi_adjoint_fused<float>(const mfx_operator*, int64_t, int64_t, int64_t, const float*, const float*, const float*,
                                          const float*, const float*, const float*, const float*, const float*, int, float*, float*,
                                          float*, float*, float*, float*, int, hipStream_t);
template int arnoldi_adjoint_fused<double>(const mfx_operator*, int64_t, int64_t, int64_t, const double*, const double*,
                                           const double*, const double*, const double*, const double*, const double*, const double*,
                                           int, double*, double*, double*, double*, double*, double*, int, hipStream_t);

}  // namespace mfx
