// libmfx: operators A(theta) -- apply, transpose-apply and the deferred parameter-gradient sweep
//   d/dtheta sum_b L_b^T A(theta) R_b        (arnoldi.py:207-209, lanczos.py:328-329)
// for DENSE, CSR and matrix-free RBF-Gram operators, plus the CALLBACK trampoline.
//
// The RBF kernels in this file are the generic VALU path (any dtype, any number of probes); the
// fp32 MFMA path for wide probe batches lives in mfx_rbf_mfma.hip and is selected in rbf_apply().
#include "mfx_internal.h"
#include "mfx_kernel_fn.h"

namespace mfx {

// ================================================================================================
// DENSE   (tests/test_lanczos/test_tridiag_forward.py:18 `p @ s`)
// ================================================================================================
// y[b][i] = sum_j A[i][j] x[b][j] : one wave per row, PB probes per pass
template <typename T, int PB>
__global__ __launch_bounds__(256) void k_dense_apply(const T* __restrict__ A, int64_t lda, int64_t n,
                                                     const T* __restrict__ x, int64_t ldx, T* __restrict__ y,
                                                     int64_t ldy, int64_t p, int64_t row0, int64_t nrow) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + wid;  // local row: y[i] = (A x)[row0 + i]
  const int64_t b0 = (int64_t)blockIdx.y * PB;
  if (i >= nrow) return;
  T acc[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) acc[q] = T(0);
  const T* row = A + (row0 + i) * lda;
  for (int64_t j = lane; j < n; j += 64) {
    const T a = row[j];
#pragma unroll
    for (int q = 0; q < PB; ++q)
      if (b0 + q < p) acc[q] += a * x[(b0 + q) * ldx + j];
  }
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const T s = wave_sum(acc[q]);
    if (lane == 0 && b0 + q < p) y[(b0 + q) * ldy + i] = s;
  }
}

// y[b][j] = sum_i A[i][j] x[b][i] : lane = column, the 4 waves split the rows
template <typename T, int PB>
__global__ __launch_bounds__(256) void k_dense_apply_t(const T* __restrict__ A, int64_t lda, int64_t n,
                                                       const T* __restrict__ x, int64_t ldx, T* __restrict__ y,
                                                       int64_t ldy, int64_t p, int64_t row0, int64_t nrow) {
  __shared__ T sm[4][PB][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 64 + lane;  // local row of A^T: y[j] = (A^T x)[row0 + j]
  const int64_t b0 = (int64_t)blockIdx.y * PB;
  T acc[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) acc[q] = T(0);
  if (j < nrow) {
    for (int64_t i = wid; i < n; i += 4) {
      const T a = A[i * lda + row0 + j];
#pragma unroll
      for (int q = 0; q < PB; ++q)
        if (b0 + q < p) acc[q] += a * x[(b0 + q) * ldx + i];
    }
  }
#pragma unroll
  for (int q = 0; q < PB; ++q) sm[wid][q][lane] = acc[q];
  __syncthreads();
  if (wid == 0 && j < nrow) {
#pragma unroll
    for (int q = 0; q < PB; ++q)
      if (b0 + q < p) y[(b0 + q) * ldy + j] = sm[0][q][lane] + sm[1][q][lane] + sm[2][q][lane] + sm[3][q][lane];
  }
}

// dA[row0 + i][j] += sum_bt L[bt][i] R[bt][j]   (i < nrow: the rows this call owns)
template <typename T>
__global__ __launch_bounds__(256) void k_dense_grad(const T* __restrict__ L, int64_t ldl, const T* __restrict__ R,
                                                    int64_t ldr, int64_t batch, int64_t n, T* __restrict__ dA,
                                                    int64_t row0, int64_t nrow) {
  const int tj = threadIdx.x & 15, ti = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.y * 16 + ti, j = (int64_t)blockIdx.x * 16 + tj;
  if (i >= nrow || j >= n) return;
  double acc = 0.0;
  for (int64_t bt = 0; bt < batch; ++bt) acc += (double)L[bt * ldl + i] * (double)R[bt * ldr + j];
  dA[(row0 + i) * n + j] += (T)acc;
}

// ================================================================================================
// CSR   (experiments/benchmarks/.../suite_sparse/benchmark.py:64-68, exp_util.py:35-42)
// ================================================================================================
// 8 lanes per row; `perm` (optional) maps a stored position to its slot in val (transpose structure)
template <typename T>
__global__ __launch_bounds__(256) void k_csr_apply(const int32_t* __restrict__ crow, const int32_t* __restrict__ col,
                                                   const int32_t* __restrict__ perm, const T* __restrict__ val,
                                                   int64_t nrow, const T* __restrict__ x, int64_t ldx,
                                                   T* __restrict__ y, int64_t ldy, int64_t row0) {
  const int sub = threadIdx.x & 7;
  const int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);  // local: y[row] = (A x)[row0 + row]
  const int64_t b = blockIdx.y;
  T acc = T(0);
  if (row < nrow) {
    const T* xb = x + b * ldx;
    for (int32_t e = crow[row0 + row] + sub; e < crow[row0 + row + 1]; e += 8) {
      const T v = perm ? val[perm[e]] : val[e];
      acc += v * xb[col[e]];
    }
  }
  acc += __shfl_down(acc, 4, 8);
  acc += __shfl_down(acc, 2, 8);
  acc += __shfl_down(acc, 1, 8);
  if (row < nrow && sub == 0) y[b * ldy + row] = acc;
}

// dval[e] += sum_bt L[bt][row_e - row0] R[bt][col_e]   (SDDMM on the sparsity pattern; only entries of the rows
// [row0, row0 + nrow) this call owns -- the others belong to other row shards)
template <typename T>
__global__ __launch_bounds__(256) void k_csr_grad(const int32_t* __restrict__ row, const int32_t* __restrict__ col,
                                                  int64_t nnz, const T* __restrict__ L, int64_t ldl,
                                                  const T* __restrict__ R, int64_t ldr, int64_t batch,
                                                  T* __restrict__ dval, int64_t row0, int64_t nrow) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nnz) return;
  const int64_t r = (int64_t)row[e] - row0, c = col[e];
  if (r < 0 || r >= nrow) return;
  double acc = 0.0;
  for (int64_t bt = 0; bt < batch; ++bt) acc += (double)L[bt * ldl + r] * (double)R[bt * ldr + c];
  dval[e] += (T)acc;
}

// ================================================================================================
// RBF Gram  K_ij = s exp(-max(0, |x_i/l|^2 + |x_j/l|^2 - 2 (x_i/l).(x_j/l)) / 2) + noise delta_ij
//   (util/gp_util.py:160-176 kernel, :225-226 noise inside the lazy kernel, :525-549 gram matvec)
// ================================================================================================

// xs[i][c] = X[i][c] / l_c (zero padded to DPAD), sq[i] = |xs_i|^2
template <typename T>
__global__ void k_rbf_prep(const T* __restrict__ X, int64_t n, int d, int dpad, const T* __restrict__ ls, int ard,
                           T* __restrict__ xs, T* __restrict__ sq) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T s = T(0);
  for (int c = 0; c < dpad; ++c) {
    T v = T(0);
    if (c < d) v = X[i * d + c] / ls[ard ? c : 0];
    xs[i * dpad + c] = v;
    s += v * v;
  }
  sq[i] = s;
}

constexpr int kRbfTJ = 128;

template <typename T, int DPAD, int PB>
__global__ __launch_bounds__(256) void k_rbf_apply(const T* __restrict__ xs, const T* __restrict__ sq, int64_t n,
                                                   const T* __restrict__ outputscale, const T* __restrict__ noise,
                                                   const T* __restrict__ x, int64_t ldx, T* __restrict__ y,
                                                   int64_t ldy, int64_t p, int kind,
                                                   const T* __restrict__ xrow, const T* __restrict__ sqrow, int64_t m,
                                                   int64_t row0) {
  // rows i < m come from (xrow, sqrow).  row0 >= 0: they are the points row0 .. row0 + m of X itself -- the square Gram
  // operator (or a row block of it): self distances exactly 0, noise on the diagonal; row0 < 0: another point set, the
  // cross-covariance K(X_new, X) of the posterior mean (util/gp_util.py:299-301)
  const bool self = row0 >= 0;
  __shared__ __attribute__((aligned(16))) T xj[kRbfTJ][DPAD];
  __shared__ T sqj[kRbfTJ];
  __shared__ T vj[PB][kRbfTJ];
  const int tid = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * 256 + tid;
  const int64_t b0 = (int64_t)blockIdx.y * PB;
  const int64_t ic = i < m ? i : m - 1;
  T xi[DPAD];
#pragma unroll
  for (int c = 0; c < DPAD; ++c) xi[c] = xrow[ic * DPAD + c];
  const T sqi = sqrow[ic];
  T acc[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) acc[q] = T(0);
  for (int64_t j0 = 0; j0 < n; j0 += kRbfTJ) {
    for (int t = tid; t < kRbfTJ * DPAD; t += 256) {
      const int64_t g = j0 * DPAD + t;
      (&xj[0][0])[t] = g < n * DPAD ? xs[g] : T(0);
    }
    if (tid < kRbfTJ) sqj[tid] = (j0 + tid < n) ? sq[j0 + tid] : T(0);
    for (int t = tid; t < PB * kRbfTJ; t += 256) {
      const int q = t / kRbfTJ, jj = t % kRbfTJ;
      vj[q][jj] = (b0 + q < p && j0 + jj < n) ? x[(b0 + q) * ldx + j0 + jj] : T(0);
    }
    __syncthreads();
#pragma unroll 4
    for (int jj = 0; jj < kRbfTJ; ++jj) {
      T dot = T(0);
#pragma unroll
      for (int c = 0; c < DPAD; ++c) dot += xi[c] * xj[jj][c];
      T dist = sqi + sqj[jj] - T(2) * dot;
      dist = dist > T(0) ? dist : T(0);
      T kv, wl;
      if (kind == MFX_KERNEL_RBF) {
        kv = exp_neg_half(dist);
      } else {
        if (self && j0 + jj == row0 + i) dist = T(0);  // a point's distance to itself is exactly 0 (sqrt amplifies round-off)
        kernel_eval<T>(kind, dist, kv, wl);
      }
#pragma unroll
      for (int q = 0; q < PB; ++q) acc[q] += kv * vj[q][jj];
    }
    __syncthreads();
  }
  if (i < m) {
    const T s = outputscale[0], nz = noise[0];
#pragma unroll
    for (int q = 0; q < PB; ++q)
      if (b0 + q < p) y[(b0 + q) * ldy + i] = self ? s * acc[q] + nz * x[(b0 + q) * ldx + row0 + i] : s * acc[q];
  }
}

// Parameter-gradient sweep (generic VALU path): workgroup = 256 rows i, walks all j in tiles of 16,
// S_ij = sum_bt L[bt][i] R[bt][j] accumulated in registers, then W = S o K and the per-parameter
// reductions.  Per-workgroup partials (double) -> k_rbf_grad_final (deterministic).
constexpr int kGradTJ = 16;
constexpr int kGradBC = 32;

template <typename T, int DPAD>
__global__ __launch_bounds__(256) void k_rbf_grad(const T* __restrict__ xs, const T* __restrict__ sq, int64_t n,
                                                  int ard, int kind, const T* __restrict__ L, int64_t ldl,
                                                  const T* __restrict__ R, int64_t ldr, int64_t batch,
                                                  double* __restrict__ partial /* (nblocks, DPAD + 2) */,
                                                  int64_t row0, int64_t nrow) {
  __shared__ __attribute__((aligned(16))) T xj[kGradTJ][DPAD];
  __shared__ T sqj[kGradTJ];
  __shared__ T rj[kGradBC][kGradTJ];
  __shared__ double red[4][DPAD + 2];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t il = (int64_t)blockIdx.x * 256 + tid;  // local row: L is indexed by il, the point is row0 + il
  const bool live = il < nrow;
  const int64_t i = row0 + il;
  const int64_t ic = live ? i : row0 + nrow - 1;
  T xi[DPAD];
#pragma unroll
  for (int c = 0; c < DPAD; ++c) xi[c] = xs[ic * DPAD + c];
  const T sqi = sq[ic];
  double g[DPAD + 2];  // [0..DPAD): lengthscale dims (ARD) or [0] scalar; [DPAD]: outputscale; [DPAD+1]: noise
#pragma unroll
  for (int c = 0; c < DPAD + 2; ++c) g[c] = 0.0;
  for (int64_t j0 = 0; j0 < n; j0 += kGradTJ) {
    T S[kGradTJ];
#pragma unroll
    for (int jj = 0; jj < kGradTJ; ++jj) S[jj] = T(0);
    for (int64_t bt0 = 0; bt0 < batch; bt0 += kGradBC) {
      __syncthreads();
      for (int t = tid; t < kGradBC * kGradTJ; t += 256) {
        const int q = t / kGradTJ, jj = t % kGradTJ;
        rj[q][jj] = (bt0 + q < batch && j0 + jj < n) ? R[(bt0 + q) * ldr + j0 + jj] : T(0);
      }
      __syncthreads();
      const int qmax = (int)((batch - bt0) < kGradBC ? (batch - bt0) : kGradBC);
      for (int q = 0; q < qmax; ++q) {
        const T l = live ? L[(bt0 + q) * ldl + il] : T(0);
#pragma unroll
        for (int jj = 0; jj < kGradTJ; ++jj) S[jj] += l * rj[q][jj];
      }
    }
    __syncthreads();
    for (int t = tid; t < kGradTJ * DPAD; t += 256) {
      const int64_t gi = j0 * DPAD + t;
      (&xj[0][0])[t] = gi < n * DPAD ? xs[gi] : T(0);
    }
    if (tid < kGradTJ) sqj[tid] = (j0 + tid < n) ? sq[j0 + tid] : T(0);
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < kGradTJ; ++jj) {
      if (j0 + jj >= n) continue;
      T dot = T(0);
#pragma unroll
      for (int c = 0; c < DPAD; ++c) dot += xi[c] * xj[jj][c];
      T dist = sqi + sqj[jj] - T(2) * dot;
      dist = dist > T(0) ? dist : T(0);
      if (kind != MFX_KERNEL_RBF && j0 + jj == i) dist = T(0);
      T kv, wl;
      kernel_eval<T>(kind, dist, kv, wl);
      g[DPAD] += (double)(S[jj] * kv);
      const T w = S[jj] * wl;
      if (ard) {
#pragma unroll
        for (int c = 0; c < DPAD; ++c) {
          const T df = xi[c] - xj[jj][c];
          g[c] += (double)(w * df * df);
        }
      } else {
        g[0] += (double)(w * dist);
      }
      if (j0 + jj == i) g[DPAD + 1] += (double)S[jj];
    }
  }
#pragma unroll
  for (int c = 0; c < DPAD + 2; ++c) {
    const double v = wave_sum(live ? g[c] : 0.0);
    if (lane == 0) red[wid][c] = v;
  }
  __syncthreads();
  if (tid < DPAD + 2) partial[(int64_t)blockIdx.x * (DPAD + 2) + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// ------------------------------------------------------------------------------------------------
// Wide inputs, d > 32 (round 5).  The reference's kernel functions take any input dimension (util/gp_util.py:151-184) and its UCI
// loaders go to d = 90 (song) and 385 (slice) (util/uci_util.py:85-99,303-310): refusing them is not an option for a drop-in.  The
// kernels above keep a point in registers (xi[DPAD]); here the scaled points are padded to a multiple of 32 and the distance is a
// small GEMM: a workgroup = 256 rows i x a tile of columns j, the d axis in chunks staged TRANSPOSED through LDS (xiT[c][row],
// xjT[c][col]), each thread accumulating <x_i, x_j> for its row and every column of the tile.  VALU arithmetic in the operator's
// dtype, every mode: correctness and the API first -- these shapes are not on the matrix cores.
// ------------------------------------------------------------------------------------------------
template <typename T> struct Wide { static constexpr int CH = sizeof(T) == 4 ? 32 : 16; };  // d-chunk: 32 KB of LDS for xiT
constexpr int kWideTJ = 32;   // columns per tile, matvec
constexpr int kWideGJ = 16;   // columns per tile, parameter sweep
constexpr int kWideGC = 32;   // ARD dimensions a workgroup of the sweep accumulates (blockIdx.y selects them)

// dot[jj] += <x_i, x_j> over all chunks of the padded d axis; TJ columns j0 .. j0 + TJ
template <typename T, int TJ>
__device__ __forceinline__ void wide_dots(const T* __restrict__ xs, int64_t n, int dpad, const T* __restrict__ xrow, int64_t ic,
                                          int64_t j0, T (*xiT)[256], T (*xjT)[TJ], T (&dot)[TJ]) {
  constexpr int CH = Wide<T>::CH;
  const int tid = threadIdx.x;
#pragma unroll
  for (int jj = 0; jj < TJ; ++jj) dot[jj] = T(0);
  for (int c0 = 0; c0 < dpad; c0 += CH) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CH; ++c) xiT[c][tid] = xrow[ic * dpad + c0 + c];
    for (int t = tid; t < TJ * CH; t += 256) {
      const int jj = t / CH, c = t % CH;
      xjT[c][jj] = (j0 + jj < n) ? xs[(j0 + jj) * dpad + c0 + c] : T(0);
    }
    __syncthreads();
#pragma unroll 2
    for (int c = 0; c < CH; ++c) {
      const T xic = xiT[c][tid];
#pragma unroll
      for (int jj = 0; jj < TJ; ++jj) dot[jj] += xic * xjT[c][jj];
    }
  }
}

template <typename T, int PB>
__global__ __launch_bounds__(256) void k_rbf_apply_wide(const T* __restrict__ xs, const T* __restrict__ sq, int64_t n, int dpad,
                                                        const T* __restrict__ outputscale, const T* __restrict__ noise,
                                                        const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy,
                                                        int64_t p, int kind, const T* __restrict__ xrow,
                                                        const T* __restrict__ sqrow, int64_t m, int64_t row0) {
  // rows, row0: as k_rbf_apply
  constexpr int CH = Wide<T>::CH;
  const bool self = row0 >= 0;
  __shared__ __attribute__((aligned(16))) T xiT[CH][256];
  __shared__ __attribute__((aligned(16))) T xjT[CH][kWideTJ];
  __shared__ T sqj[kWideTJ];
  __shared__ T vj[PB][kWideTJ];
  const int tid = threadIdx.x;
  const int64_t i = (int64_t)blockIdx.x * 256 + tid;
  const int64_t b0 = (int64_t)blockIdx.y * PB;
  const int64_t ic = i < m ? i : m - 1;
  const T sqi = sqrow[ic];
  T acc[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) acc[q] = T(0);
  for (int64_t j0 = 0; j0 < n; j0 += kWideTJ) {
    __syncthreads();  // the previous tile's vj / sqj are read no more
    if (tid < kWideTJ) sqj[tid] = (j0 + tid < n) ? sq[j0 + tid] : T(0);
    for (int t = tid; t < PB * kWideTJ; t += 256) {
      const int q = t / kWideTJ, jj = t % kWideTJ;
      vj[q][jj] = (b0 + q < p && j0 + jj < n) ? x[(b0 + q) * ldx + j0 + jj] : T(0);
    }
    T dot[kWideTJ];
    wide_dots<T, kWideTJ>(xs, n, dpad, xrow, ic, j0, xiT, xjT, dot);  // (its barriers publish vj / sqj too)
#pragma unroll
    for (int jj = 0; jj < kWideTJ; ++jj) {
      T dist = sqi + sqj[jj] - T(2) * dot[jj];
      dist = dist > T(0) ? dist : T(0);
      T kv, wl;
      if (kind == MFX_KERNEL_RBF) {
        kv = exp_neg_half(dist);
      } else {
        if (self && j0 + jj == row0 + i) dist = T(0);
        kernel_eval<T>(kind, dist, kv, wl);
      }
#pragma unroll
      for (int q = 0; q < PB; ++q) acc[q] += kv * vj[q][jj];  // (columns j >= n carry v = 0)
    }
  }
  if (i < m) {
    const T s = outputscale[0], nz = noise[0];
#pragma unroll
    for (int q = 0; q < PB; ++q)
      if (b0 + q < p) y[(b0 + q) * ldy + i] = self ? s * acc[q] + nz * x[(b0 + q) * ldx + row0 + i] : s * acc[q];
  }
}

// Parameter sweep for wide inputs: as k_rbf_grad; with ARD the d lengthscale derivatives do not fit a thread's registers, so the
// grid's second axis selects kWideGC of them (S_ij and the kernel weights are re-evaluated per selection: d / 32 times the work of
// a scalar lengthscale -- the price of keeping the direct, cancellation-free form sum_ij w_ij (x_ic - x_jc)^2).
template <typename T>
__global__ __launch_bounds__(256) void k_rbf_grad_wide(const T* __restrict__ xs, const T* __restrict__ sq, int64_t n, int dpad,
                                                       int ard, int kind, const T* __restrict__ L, int64_t ldl,
                                                       const T* __restrict__ R, int64_t ldr, int64_t batch,
                                                       double* __restrict__ partial /* (nblocks, dpad + 2) */, int64_t row0,
                                                       int64_t nrow) {
  constexpr int CH = Wide<T>::CH;
  __shared__ __attribute__((aligned(16))) T xiT[CH][256];
  __shared__ __attribute__((aligned(16))) T xjT[CH][kWideGJ];
  __shared__ __attribute__((aligned(16))) T xjg[kWideGJ][kWideGC];
  __shared__ T sqj[kWideGJ];
  __shared__ T rj[kGradBC][kWideGJ];
  __shared__ double red[4][kWideGC + 2];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int gc = (int)blockIdx.y;  // ARD: dimensions gc * 32 .. gc * 32 + 31; scalar lengthscale: one selection
  const int64_t il = (int64_t)blockIdx.x * 256 + tid;
  const bool live = il < nrow;
  const int64_t i = row0 + il;
  const int64_t ic = live ? i : row0 + nrow - 1;
  const T sqi = sq[ic];
  T xig[kWideGC];
#pragma unroll
  for (int c = 0; c < kWideGC; ++c) xig[c] = ard ? xs[ic * dpad + gc * kWideGC + c] : T(0);
  double g[kWideGC + 2];  // [0 .. 32): this selection's lengthscale dims (scalar: [0]); [32]: outputscale; [33]: noise (selection 0 only)
#pragma unroll
  for (int c = 0; c < kWideGC + 2; ++c) g[c] = 0.0;
  for (int64_t j0 = 0; j0 < n; j0 += kWideGJ) {
    T S[kWideGJ];
#pragma unroll
    for (int jj = 0; jj < kWideGJ; ++jj) S[jj] = T(0);
    for (int64_t bt0 = 0; bt0 < batch; bt0 += kGradBC) {
      __syncthreads();
      for (int t = tid; t < kGradBC * kWideGJ; t += 256) {
        const int q = t / kWideGJ, jj = t % kWideGJ;
        rj[q][jj] = (bt0 + q < batch && j0 + jj < n) ? R[(bt0 + q) * ldr + j0 + jj] : T(0);
      }
      __syncthreads();
      const int qmax = (int)((batch - bt0) < kGradBC ? (batch - bt0) : kGradBC);
      for (int q = 0; q < qmax; ++q) {
        const T l = live ? L[(bt0 + q) * ldl + il] : T(0);
#pragma unroll
        for (int jj = 0; jj < kWideGJ; ++jj) S[jj] += l * rj[q][jj];
      }
    }
    __syncthreads();
    if (tid < kWideGJ) sqj[tid] = (j0 + tid < n) ? sq[j0 + tid] : T(0);
    if (ard)
      for (int t = tid; t < kWideGJ * kWideGC; t += 256) {
        const int jj = t / kWideGC, c = t % kWideGC;
        xjg[jj][c] = (j0 + jj < n) ? xs[(j0 + jj) * dpad + gc * kWideGC + c] : T(0);
      }
    T dot[kWideGJ];
    wide_dots<T, kWideGJ>(xs, n, dpad, xs, ic, j0, xiT, xjT, dot);
#pragma unroll
    for (int jj = 0; jj < kWideGJ; ++jj) {
      if (j0 + jj >= n) continue;
      T dist = sqi + sqj[jj] - T(2) * dot[jj];
      dist = dist > T(0) ? dist : T(0);
      if (kind != MFX_KERNEL_RBF && j0 + jj == i) dist = T(0);
      T kv, wl;
      kernel_eval<T>(kind, dist, kv, wl);
      const T w = S[jj] * wl;
      if (gc == 0) {
        g[kWideGC] += (double)(S[jj] * kv);
        if (j0 + jj == i) g[kWideGC + 1] += (double)S[jj];
      }
      if (ard) {
#pragma unroll
        for (int c = 0; c < kWideGC; ++c) {
          const T df = xig[c] - xjg[jj][c];
          g[c] += (double)(w * df * df);
        }
      } else {
        g[0] += (double)(w * dist);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < kWideGC + 2; ++c) {
    const double v = wave_sum(live ? g[c] : 0.0);
    if (lane == 0) red[wid][c] = v;
  }
  __syncthreads();
  double* out = partial + (int64_t)blockIdx.x * (dpad + 2);
  if (tid < kWideGC && (ard || tid == 0)) out[gc * kWideGC + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
  if (!ard)  // (the columns k_rbf_grad_final sums and drops)
    for (int c = 1 + tid; c < dpad; c += 256) out[c] = 0.0;
  if (gc == 0 && tid >= kWideGC && tid < kWideGC + 2)
    out[dpad + tid - kWideGC] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// grads += factors * sum_blocks partial  (chain to the constrained parameters l, s, noise)
template <typename T>
__global__ __launch_bounds__(256) void k_rbf_grad_final(const double* __restrict__ partial, int64_t nblocks, int dpad, int d,
                                                        int ard, const T* __restrict__ ls, const T* __restrict__ outputscale,
                                                        T* __restrict__ g_ls, T* __restrict__ g_s, T* __restrict__ g_noise,
                                                        const float* __restrict__ scales) {
  __shared__ double sm[4];
  const int c = blockIdx.x;  // one workgroup per output column, fixed summation order (deterministic)
  double acc = 0.0;
  for (int64_t q = threadIdx.x; q < nblocks; q += 256) acc += partial[q * (dpad + 2) + c];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x != 0) return;
  acc = sm[0] + sm[1] + sm[2] + sm[3];
  if (scales) acc *= (double)scales[1] * (double)scales[3];  // undo the power-of-two operand scales of the split GEMM
  const double s = (double)outputscale[0];
  if (c < dpad) {
    if (ard ? (c < d) : (c == 0)) {
      if (g_ls) g_ls[c] += (T)(s * acc / (double)ls[ard ? c : 0]);
    }
  } else if (c == dpad) {
    if (g_s) g_s[0] += (T)acc;
  } else {
    if (g_noise) g_noise[0] += (T)acc;
  }
}

constexpr int kRbfMaxD = 1024;  // wide inputs (d > 32): padded to a multiple of 32, k_rbf_apply_wide / k_rbf_grad_wide
static int rbf_dpad(int d) { return d <= 4 ? 4 : d <= 8 ? 8 : d <= 12 ? 12 : d <= 16 ? 16 : d <= kRbfMaxD ? (d + 31) / 32 * 32 : -1; }

struct RbfWs {
  void *xs, *sq;
  double* partial;
  float* vscale;  // (rows, 2) power-of-two scales of the f16-split path
  void* hws;      // f16 hi/lo packs of the split gradient sweep (sized for `batch_hint` rows)
  int64_t hws_bytes;
  void* pk;       // pre-packed f16 tile images of the pipelined matvec (sized for `p_apply` vectors)
  int64_t pk_bytes;
};

int64_t rbf_grad_h_ws_bytes(int64_t n, int64_t batch);
int64_t rbf_pack_ws_bytes(const mfx_operator* op, int64_t p);
int rbf_mode(const mfx_operator* op);

static int64_t rbf_carve(const mfx_operator* op, void* ws, int64_t ws_bytes, RbfWs* out, int64_t batch_hint = 0,
                         int64_t p_apply = 0) {
  const size_t es = dtype_size(op->dtype);
  const int dpad = rbf_dpad(op->d);
  Carver cv(ws, ws_bytes);
  RbfWs r;
  r.xs = cv.take(op->n * (dpad > 0 ? dpad : 1) * es);
  r.sq = cv.take(op->n * es);
  // per-workgroup gradient partials: VALU sweep n/256 rows, MFMA sweep 8 * n/128 rows, <= 34 doubles each
  r.partial = static_cast<double*>(cv.take(((op->n + 127) / 128) * 64 * (dpad > 32 && dpad <= 64 ? dpad + 2 : 34) * sizeof(double)));  // 8 XCDs x 8 sub-ranges (DPAD 64: the matrix-core sweep writes 66 per workgroup; wider: the VALU sweep, dpad + 2 per 256 rows)
  r.vscale = static_cast<float*>(cv.take(65536 * 3 * sizeof(float)));  // [s, 1/s] per row + |max| bit patterns
  r.hws_bytes = (op->dtype == MFX_F32 && rbf_mode(op) == MFX_RBF_F16X3 && batch_hint > 0) ? rbf_grad_h_ws_bytes(op->n, batch_hint) : 0;
  r.hws = r.hws_bytes ? cv.take(r.hws_bytes) : nullptr;
  r.pk_bytes = (op->dtype == MFX_F32 && p_apply > 0) ? rbf_pack_ws_bytes(op, p_apply) : 0;
  r.pk = r.pk_bytes ? cv.take(r.pk_bytes) : nullptr;
  if (out) *out = r;
  return cv.off;
}

// MFMA path (mfx_rbf_mfma.hip)
bool rbf_mfma_supported(const mfx_operator* op, int64_t p);
int rbf_mfma_apply(const mfx_operator* op, const float* xs, const float* sq, int dpad, const float* x, int64_t ldx,
                   float* y, int64_t ldy, int64_t p, hipStream_t stream);
int rbf_mfma_apply_h3(const mfx_operator* op, const float* xs, const float* sq, int dpad, const float* x, int64_t ldx,
                      float* y, int64_t ldy, int64_t p, float* vscale, void* pk, hipStream_t stream);
int rbf_mfma_grad_h(const mfx_operator* op, const float* xs, const float* sq, int dpad, const float* L, int64_t ldl,
                    const float* R, int64_t ldr, int64_t batch, int64_t inner, double* partial, int64_t* nblocks_out, void* hws,
                    const float** scales_out, hipStream_t stream);
bool rbf_mfma_grad_supported(const mfx_operator* op, int64_t batch);
bool rbf_mfma_exact_wide_supported(const mfx_operator* op, int64_t p);         // 16 < d <= 128: the exact-fp32 kernels, in every mode
bool rbf_mfma_h3_wide_supported(const mfx_operator* op, int64_t p);            // 16 < d <= 32, split modes: fp32 distances + f16x3 contraction
bool rbf_mfma_grad_exact_wide_supported(const mfx_operator* op, int64_t batch);  // 16 < d <= 64
int rbf_mfma_grad(const mfx_operator* op, const float* xs, const float* sq, int dpad, const float* L, int64_t ldl,
                  const float* R, int64_t ldr, int64_t batch, double* partial, int64_t* nblocks_out,
                  hipStream_t stream);

// what the last k_rbf_prep inside the current PrepScope of this thread prepared (null: nothing yet)
struct PrepKey {
  const void *x = nullptr, *ls = nullptr, *xs = nullptr;
  int64_t n = 0;
  int d = 0, ard = 0, dtype = 0;
  hipStream_t stream = nullptr;
  bool operator==(const PrepKey& o) const {
    return x == o.x && ls == o.ls && xs == o.xs && n == o.n && d == o.d && ard == o.ard && dtype == o.dtype && stream == o.stream;
  }
};
static thread_local int t_prep_depth = 0;
static thread_local PrepKey t_prep_done;

template <typename T>
static int rbf_prep(const mfx_operator* op, const RbfWs& w, int dpad, hipStream_t stream) {
  PrepKey key;
  key.x = op->x; key.ls = op->lengthscale; key.xs = w.xs; key.n = op->n; key.d = op->d; key.ard = op->ard; key.dtype = op->dtype;
  key.stream = stream;
  if (t_prep_depth > 0 && t_prep_done.xs != nullptr && t_prep_done == key) return MFX_OK;  // same driver call, same operator, same workspace
  k_rbf_prep<T><<<(unsigned)((op->n + 255) / 256), 256, 0, stream>>>(
      (const T*)op->x, op->n, op->d, dpad, (const T*)op->lengthscale, op->ard, (T*)w.xs, (T*)w.sq);
  MFX_CHECK_LAUNCH();
  if (t_prep_depth > 0) t_prep_done = key;
  return MFX_OK;
}

template <typename T, int DPAD>
static int rbf_apply_d(const mfx_operator* op, const RbfWs& w, const T* x, int64_t ldx, T* y, int64_t ldy, int64_t p,
                       hipStream_t stream, const T* xrow = nullptr, const T* sqrow = nullptr, int64_t m = 0) {
  int64_t row0 = -1;  // cross-covariance rows (another point set)
  if (!xrow) {        // rows row0 .. row0 + m of the square Gram operator
    row0 = op_row0(op);
    m = op_nrows(op);
    xrow = (const T*)w.xs + row0 * DPAD;
    sqrow = (const T*)w.sq + row0;
  }
  const unsigned gx = (unsigned)((m + 255) / 256);
#define MFX_RBF_LAUNCH(PB)                                                                           \
  k_rbf_apply<T, DPAD, PB><<<dim3(gx, (unsigned)((p + PB - 1) / PB)), 256, 0, stream>>>(               \
      (const T*)w.xs, (const T*)w.sq, op->n, (const T*)op->outputscale, (const T*)op->noise, x, ldx, y, ldy, p, \
      op->kernel_fn, xrow, sqrow, m, row0)
  if (p == 1) {
    MFX_RBF_LAUNCH(1);
  } else if (p == 2) {
    MFX_RBF_LAUNCH(2);
  } else if (p <= 4) {
    MFX_RBF_LAUNCH(4);
  } else {
    MFX_RBF_LAUNCH(8);
  }
#undef MFX_RBF_LAUNCH
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

template <typename T>
static int rbf_apply_wide(const mfx_operator* op, const RbfWs& w, int dpad, const T* x, int64_t ldx, T* y, int64_t ldy, int64_t p,
                          hipStream_t stream, const T* xrow = nullptr, const T* sqrow = nullptr, int64_t m = 0) {
  int64_t row0 = -1;
  if (!xrow) {
    row0 = op_row0(op);
    m = op_nrows(op);
    xrow = (const T*)w.xs + row0 * dpad;
    sqrow = (const T*)w.sq + row0;
  }
  const unsigned gx = (unsigned)((m + 255) / 256);
#define MFX_RBF_LAUNCH(PB)                                                                                 \
  k_rbf_apply_wide<T, PB><<<dim3(gx, (unsigned)((p + PB - 1) / PB)), 256, 0, stream>>>(                     \
      (const T*)w.xs, (const T*)w.sq, op->n, dpad, (const T*)op->outputscale, (const T*)op->noise, x, ldx, y, ldy, p, \
      op->kernel_fn, xrow, sqrow, m, row0)
  if (p == 1) {
    MFX_RBF_LAUNCH(1);
  } else if (p <= 4 || sizeof(T) == 8) {  // (8 fp64 vectors next to the 32 fp64 dot products of a tile would spill)
    MFX_RBF_LAUNCH(4);
  } else {
    if constexpr (sizeof(T) == 4) MFX_RBF_LAUNCH(8);
  }
#undef MFX_RBF_LAUNCH
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

template <typename T>
static int rbf_apply(const mfx_operator* op, const T* x, int64_t ldx, T* y, int64_t ldy, int64_t p, void* ws,
                     int64_t ws_bytes, hipStream_t stream) {
  const int dpad = rbf_dpad(op->d);
  MFX_REQUIRE(dpad > 0, MFX_ERR_UNSUPPORTED, "RBF operator supports d <= 1024 (got %d)", op->d);
  RbfWs w;
  MFX_REQUIRE(rbf_carve(op, ws, ws_bytes, &w, 0, p) <= ws_bytes && ws, MFX_ERR_WORKSPACE, "RBF workspace too small");
  MFX_TRY(rbf_prep<T>(op, w, dpad, stream));
  if constexpr (sizeof(T) == 4) {
    // 1-3 vectors (the CG solves of the log-marginal likelihood): the VALU kernel below costs 2-2.7x a matrix-core
    // sweep over one 32-probe block (measured, n = 131072: 12.9 vs 5.9 ms), whose probe guards handle any p >= 1
    if (p < 4 && op->n >= 2048 && rbf_mfma_supported(op, 4) && rbf_mode(op) >= MFX_RBF_F16X3_MATVEC)
      return rbf_mfma_apply_h3(op, (const float*)w.xs, (const float*)w.sq, dpad, x, ldx, y, ldy, p, w.vscale, w.pk, stream);
    if (rbf_mfma_supported(op, p)) {
      if (rbf_mode(op) >= MFX_RBF_F16X3_MATVEC)
        return rbf_mfma_apply_h3(op, (const float*)w.xs, (const float*)w.sq, dpad, x, ldx, y, ldy, p, w.vscale, w.pk, stream);
      return rbf_mfma_apply(op, (const float*)w.xs, (const float*)w.sq, dpad, x, ldx, y, ldy, p, stream);
    }
    if (rbf_mfma_h3_wide_supported(op, p) && rbf_mode(op) >= MFX_RBF_F16X3_MATVEC)
      return rbf_mfma_apply_h3(op, (const float*)w.xs, (const float*)w.sq, dpad, x, ldx, y, ldy, p, w.vscale, w.pk, stream);
    if (rbf_mfma_exact_wide_supported(op, p))
      return rbf_mfma_apply(op, (const float*)w.xs, (const float*)w.sq, dpad, x, ldx, y, ldy, p, stream);
  }
  switch (dpad) {
    case 4: return rbf_apply_d<T, 4>(op, w, x, ldx, y, ldy, p, stream);
    case 8: return rbf_apply_d<T, 8>(op, w, x, ldx, y, ldy, p, stream);
    case 12: return rbf_apply_d<T, 12>(op, w, x, ldx, y, ldy, p, stream);
    case 16: return rbf_apply_d<T, 16>(op, w, x, ldx, y, ldy, p, stream);
    case 32: return rbf_apply_d<T, 32>(op, w, x, ldx, y, ldy, p, stream);
    default: return rbf_apply_wide<T>(op, w, dpad, x, ldx, y, ldy, p, stream);
  }
}

// y (p, m) = K(X_new, X) v: the cross-covariance matvec of the posterior mean (util/gp_util.py:299-305), no noise term
template <typename T>
static int rbf_cross_apply(const mfx_operator* op, const T* xnew, int64_t m, const T* v, int64_t ldv, T* y, int64_t ldy,
                           int64_t p, void* ws, int64_t ws_bytes, hipStream_t stream) {
  const int dpad = rbf_dpad(op->d);
  MFX_REQUIRE(dpad > 0, MFX_ERR_UNSUPPORTED, "RBF operator supports d <= 1024 (got %d)", op->d);
  RbfWs w;
  const int64_t base = rbf_carve(op, ws, ws_bytes, &w);
  Carver cv(ws ? (char*)ws + base : nullptr, ws_bytes - base);
  T* xr = (T*)cv.take(m * dpad * sizeof(T));
  T* sqr = (T*)cv.take(m * sizeof(T));
  MFX_REQUIRE(ws && base + cv.off <= ws_bytes, MFX_ERR_WORKSPACE, "cross-Gram workspace too small");
  MFX_TRY(rbf_prep<T>(op, w, dpad, stream));
  k_rbf_prep<T><<<(unsigned)((m + 255) / 256), 256, 0, stream>>>(xnew, m, op->d, dpad, (const T*)op->lengthscale, op->ard,
                                                                xr, sqr);
  MFX_CHECK_LAUNCH();
  switch (dpad) {
    case 4: return rbf_apply_d<T, 4>(op, w, v, ldv, y, ldy, p, stream, xr, sqr, m);
    case 8: return rbf_apply_d<T, 8>(op, w, v, ldv, y, ldy, p, stream, xr, sqr, m);
    case 12: return rbf_apply_d<T, 12>(op, w, v, ldv, y, ldy, p, stream, xr, sqr, m);
    case 16: return rbf_apply_d<T, 16>(op, w, v, ldv, y, ldy, p, stream, xr, sqr, m);
    case 32: return rbf_apply_d<T, 32>(op, w, v, ldv, y, ldy, p, stream, xr, sqr, m);
    default: return rbf_apply_wide<T>(op, w, dpad, v, ldv, y, ldy, p, stream, xr, sqr, m);
  }
}

int64_t rbf_cross_ws_bytes(const mfx_operator* op, int64_t m) {
  const int dpad = rbf_dpad(op->d);
  return rbf_carve(op, nullptr, 0, nullptr) + align_up(m * (dpad > 0 ? dpad : 1) * dtype_size(op->dtype), 256) +
         align_up(m * dtype_size(op->dtype), 256);
}

int op_cross_apply(const mfx_operator* op, const void* xnew, int64_t m, const void* v, int64_t ldv, void* y, int64_t ldy,
                   int64_t p, void* ws, int64_t ws_bytes, hipStream_t stream) {
  MFX_REQUIRE(op->kind == MFX_OP_RBF, MFX_ERR_UNSUPPORTED, "cross-covariance matvec needs a kernel-Gram operator");
  ScopedTimer t(0, stream);
  if (op->dtype == MFX_F32)
    return rbf_cross_apply<float>(op, (const float*)xnew, m, (const float*)v, ldv, (float*)y, ldy, p, ws, ws_bytes, stream);
  return rbf_cross_apply<double>(op, (const double*)xnew, m, (const double*)v, ldv, (double*)y, ldy, p, ws, ws_bytes, stream);
}

template <typename T>
static int rbf_grad(const mfx_operator* op, const T* L, int64_t ldl, const T* R, int64_t ldr, int64_t batch, int64_t inner,
                    const mfx_op_grads* grads, void* ws, int64_t ws_bytes, hipStream_t stream) {
  const int dpad = rbf_dpad(op->d);
  MFX_REQUIRE(dpad > 0, MFX_ERR_UNSUPPORTED, "RBF operator supports d <= 1024 (got %d)", op->d);
  RbfWs w;
  MFX_REQUIRE(rbf_carve(op, ws, ws_bytes, &w, batch) <= ws_bytes && ws, MFX_ERR_WORKSPACE, "RBF workspace too small");
  MFX_TRY(rbf_prep<T>(op, w, dpad, stream));
  const int64_t row0 = op_row0(op), nrow = op_nrows(op);
  int64_t nblocks = (nrow + 255) / 256;
  bool done = false;
  const float* scales = nullptr;
  if constexpr (sizeof(T) == 4) {
    // the split GEMM stages its packed operands through 32-bit byte offsets: 2 B x padded batch x padded n < 4 GiB
    const bool fits32 = ((batch + 31) / 32 * 32) * ((op->n + 255) / 256 * 256) * 2 < ((int64_t)1 << 32);
    // (16 < d <= 32: the split sweep's 256 x 128 form with 64-column epilogue passes -- 249 registers, 140 KB of LDS)
    const bool split_ok = rbf_mfma_grad_supported(op, batch) || (rbf_mfma_grad_exact_wide_supported(op, batch) && op->d <= 32);
    if (split_ok && rbf_mode(op) == MFX_RBF_F16X3 && w.hws && fits32) {
      MFX_TRY(rbf_mfma_grad_h(op, (const float*)w.xs, (const float*)w.sq, dpad, L, ldl, R, ldr, batch, inner, w.partial, &nblocks,
                              w.hws, &scales, stream));
      done = true;
    } else if (rbf_mfma_grad_supported(op, batch) || rbf_mfma_grad_exact_wide_supported(op, batch)) {
      MFX_TRY(rbf_mfma_grad(op, (const float*)w.xs, (const float*)w.sq, dpad, L, ldl, R, ldr, batch, w.partial,
                            &nblocks, stream));
      done = true;
    }
  }
  if (!done) {
#define MFX_RBF_GRAD(D)                                                                                    \
  k_rbf_grad<T, D><<<(unsigned)nblocks, 256, 0, stream>>>((const T*)w.xs, (const T*)w.sq, op->n, op->ard, op->kernel_fn, L, ldl, R, \
                                                           ldr, batch, w.partial, row0, nrow)
    switch (dpad) {
      case 4: MFX_RBF_GRAD(4); break;
      case 8: MFX_RBF_GRAD(8); break;
      case 12: MFX_RBF_GRAD(12); break;
      case 16: MFX_RBF_GRAD(16); break;
      case 32: MFX_RBF_GRAD(32); break;
      default:
        k_rbf_grad_wide<T><<<dim3((unsigned)nblocks, op->ard ? (unsigned)(dpad / kWideGC) : 1u), 256, 0, stream>>>(
            (const T*)w.xs, (const T*)w.sq, op->n, dpad, op->ard, op->kernel_fn, L, ldl, R, ldr, batch, w.partial, row0, nrow);
        break;
    }
#undef MFX_RBF_GRAD
    MFX_CHECK_LAUNCH();
  }
  k_rbf_grad_final<T><<<dpad + 2, 256, 0, stream>>>(w.partial, nblocks, dpad, op->d, op->ard, (const T*)op->lengthscale,
                                            (const T*)op->outputscale, (T*)grads->lengthscale,
                                            (T*)grads->outputscale, (T*)grads->noise, scales);
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

// ================================================================================================
// dispatch
// ================================================================================================
PrepScope::PrepScope() {
  if (t_prep_depth++ == 0) t_prep_done = PrepKey{};
}
PrepScope::~PrepScope() {
  if (--t_prep_depth == 0) t_prep_done = PrepKey{};
}

int64_t op_workspace_bytes(const mfx_operator* op, int64_t batch_hint, int64_t p_apply) {
  if (op->kind == MFX_OP_RBF) return rbf_carve(op, nullptr, 0, nullptr, batch_hint, p_apply);
  return 256;
}

template <typename T>
static int op_apply_t(const mfx_operator* op, const T* x, int64_t ldx, T* y, int64_t ldy, int64_t p, int transpose,
                      void* ws, int64_t ws_bytes, hipStream_t stream) {
  const int64_t n = op->n;
  const int64_t row0 = op_row0(op), nrow = op_nrows(op);
  MFX_REQUIRE(row0 >= 0 && nrow >= 1 && row0 + nrow <= n, MFX_ERR_INVALID, "row block [%lld, +%lld) outside the operator (n = %lld)",
              (long long)row0, (long long)nrow, (long long)n);
  switch (op->kind) {
    case MFX_OP_DENSE: {
      MFX_REQUIRE(op->dense_a, MFX_ERR_INVALID, "dense operator without matrix");
      constexpr int PB = 4;
      if (!transpose) {
        k_dense_apply<T, PB><<<dim3((unsigned)((nrow + 3) / 4), (unsigned)((p + PB - 1) / PB)), 256, 0, stream>>>(
            (const T*)op->dense_a, op->lda, n, x, ldx, y, ldy, p, row0, nrow);
      } else {
        k_dense_apply_t<T, PB><<<dim3((unsigned)((nrow + 63) / 64), (unsigned)((p + PB - 1) / PB)), 256, 0, stream>>>(
            (const T*)op->dense_a, op->lda, n, x, ldx, y, ldy, p, row0, nrow);
      }
      MFX_CHECK_LAUNCH();
      return MFX_OK;
    }
    case MFX_OP_CSR: {
      MFX_REQUIRE(op->crow && op->col && op->val, MFX_ERR_INVALID, "CSR operator without structure");
      const dim3 grid((unsigned)((nrow + 31) / 32), (unsigned)p);
      if (!transpose) {
        k_csr_apply<T><<<grid, 256, 0, stream>>>(op->crow, op->col, nullptr, (const T*)op->val, nrow, x, ldx, y, ldy, row0);
      } else {
        MFX_REQUIRE(op->t_crow && op->t_col && op->t_perm, MFX_ERR_INVALID,
                    "CSR transpose structure required for the Arnoldi adjoint");
        k_csr_apply<T><<<grid, 256, 0, stream>>>(op->t_crow, op->t_col, op->t_perm, (const T*)op->val, nrow, x, ldx, y, ldy, row0);
      }
      MFX_CHECK_LAUNCH();
      return MFX_OK;
    }
    case MFX_OP_RBF:
      MFX_REQUIRE(op->x && op->lengthscale && op->outputscale && op->noise, MFX_ERR_INVALID,
                  "RBF operator with null pointers");
      return rbf_apply<T>(op, x, ldx, y, ldy, p, ws, ws_bytes, stream);  // symmetric: transpose ignored
    default:
      set_error("unknown operator kind %d", op->kind);
      return MFX_ERR_UNSUPPORTED;
  }
}

int op_apply(const mfx_operator* op, const void* x, int64_t ldx, void* y, int64_t ldy, int64_t p, int transpose,
             void* ws, int64_t ws_bytes, hipStream_t stream) {
  MFX_REQUIRE(p <= 65535, MFX_ERR_UNSUPPORTED, "p too large");
  if (op->dtype == MFX_F32) return op_apply_t<float>(op, (const float*)x, ldx, (float*)y, ldy, p, transpose, ws, ws_bytes, stream);
  if (op->dtype == MFX_F64) return op_apply_t<double>(op, (const double*)x, ldx, (double*)y, ldy, p, transpose, ws, ws_bytes, stream);
  set_error("unsupported dtype %d", op->dtype);
  return MFX_ERR_UNSUPPORTED;
}

int op_apply_cb(const mfx_operator* op, int mode, const void* x, int64_t ldx, const void* aux, int64_t ldaux, void* y,
                int64_t ldy, int64_t p, hipStream_t stream) {
  MFX_REQUIRE(op->callback, MFX_ERR_INVALID, "callback operator without function pointer");
  const int rc = op->callback(op->ctx, mode, x, ldx, aux, ldaux, y, ldy, p, op->n, stream);
  MFX_REQUIRE(rc == 0, MFX_ERR_CALLBACK, "operator callback failed with code %d", rc);
  return MFX_OK;
}

template <typename T>
static int op_vjp_params_t(const mfx_operator* op, const T* L, int64_t ldl, const T* R, int64_t ldr, int64_t batch,
                           int64_t inner, const mfx_op_grads* grads, void* ws, int64_t ws_bytes, hipStream_t stream) {
  const int64_t n = op->n;
  const int64_t row0 = op_row0(op), nrow = op_nrows(op);
  MFX_REQUIRE(row0 >= 0 && nrow >= 1 && row0 + nrow <= n, MFX_ERR_INVALID, "row block [%lld, +%lld) outside the operator (n = %lld)",
              (long long)row0, (long long)nrow, (long long)n);
  switch (op->kind) {
    case MFX_OP_DENSE:
      if (!grads->dense_a) return MFX_OK;
      k_dense_grad<T><<<dim3((unsigned)((n + 15) / 16), (unsigned)((nrow + 15) / 16)), 256, 0, stream>>>(
          L, ldl, R, ldr, batch, n, (T*)grads->dense_a, row0, nrow);
      MFX_CHECK_LAUNCH();
      return MFX_OK;
    case MFX_OP_CSR:
      if (!grads->val) return MFX_OK;
      MFX_REQUIRE(op->row && op->col, MFX_ERR_INVALID, "CSR gradient needs the COO row index");
      k_csr_grad<T><<<(unsigned)((op->nnz + 255) / 256), 256, 0, stream>>>(op->row, op->col, op->nnz, L, ldl, R, ldr,
                                                                         batch, (T*)grads->val, row0, nrow);
      MFX_CHECK_LAUNCH();
      return MFX_OK;
    case MFX_OP_RBF:
      if (!grads->lengthscale && !grads->outputscale && !grads->noise) return MFX_OK;
      return rbf_grad<T>(op, L, ldl, R, ldr, batch, inner, grads, ws, ws_bytes, stream);
    default:
      set_error("unknown operator kind %d", op->kind);
      return MFX_ERR_UNSUPPORTED;
  }
}

int op_vjp_params(const mfx_operator* op, const void* L, int64_t ldl, const void* R, int64_t ldr, int64_t batch,
                  const mfx_op_grads* grads, void* ws, int64_t ws_bytes, hipStream_t stream, int64_t inner) {
  if (op->dtype == MFX_F32)
    return op_vjp_params_t<float>(op, (const float*)L, ldl, (const float*)R, ldr, batch, inner, grads, ws, ws_bytes, stream);
  if (op->dtype == MFX_F64)
    return op_vjp_params_t<double>(op, (const double*)L, ldl, (const double*)R, ldr, batch, inner, grads, ws, ws_bytes, stream);
  set_error("unsupported dtype %d", op->dtype);
  return MFX_ERR_UNSUPPORTED;
}

}  // namespace mfx
