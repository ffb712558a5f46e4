// libmfx: the Krylov inner loops -- Arnoldi/Lanczos forward recurrences and their adjoints.
//
// Layout: a batch of p vectors is (p, n) row-major; the basis is (p, k, n).  Every HBM-bound kernel
// below uses grid = (slices, p): one workgroup (4 waves) owns a slice of ONE probe's vectors -- 2048 elements,
// 8 per thread, or 512 / 1024 (one 16-byte load per thread and row) when that leaves most CUs idle (Ctx::fine) --
// and keeps its elements in registers across the whole multi-row sweep (double-buffered, mfx_vec.h: sweep_rows).
// Dot products are accumulated in fp64, reduced lane -> wave (DPP, several rows at once: wave_sums) -> workgroup (LDS)
// and written as per-slice partials (p, kmax, nslices); the consumer kernel re-reduces the partials in its
// prologue (deterministic, no atomics, and no host round trip: the whole k-loop is enqueued on one stream,
// and replayed from a hipGraph when the call repeats: mfx_core.hip).
//
// Reference: arnoldi.py:57-101 (forward), :104-220 (adjoint); lanczos.py:215-285, :288-335.
#include "mfx_vec.h"

namespace mfx {


// ------------------------------------------------------------------------------------------------
// Arnoldi-adjoint small setup, one workgroup per probe (arnoldi.py:120,128):
//   eta_j = dH[j][k-1] - (Q^T dr)_j ;  Pi_gamma = -dc c e1 e1^T + H dH^T   (dQ^T Q subtracted later)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_adj_setup(const T* __restrict__ H, const T* __restrict__ dH, const T* __restrict__ c,
                            const T* __restrict__ dc, const T* __restrict__ part_qtdr, int kmax,
                            int nblk, int k, T* __restrict__ eta, T* __restrict__ pig) {
  // grid (p, k): workgroup (b, i) owns row i of Pi_gamma; the workgroups of row 0 also write eta
  const int b = blockIdx.x, i = blockIdx.y;
  const T* Hb = H + (int64_t)b * k * k;
  const T* dHb = dH + (int64_t)b * k * k;
  if (i == 0) {
    for (int idx = threadIdx.x; idx < k * kRedG; idx += blockDim.x) {
      const int j = idx / kRedG, g = idx % kRedG;
      T e = dHb[(int64_t)j * k + (k - 1)];
      if (part_qtdr) e -= reduce_partials_group<T, kRedG>(part_qtdr + ((int64_t)b * kmax + j) * nblk, nblk, g);
      if (g == 0) eta[(int64_t)b * k + j] = e;
    }
  }
  for (int j = threadIdx.x; j < k; j += blockDim.x) {
    double acc = 0.0;
    for (int l = 0; l < k; ++l) acc += (double)Hb[(int64_t)i * k + l] * (double)dHb[(int64_t)j * k + l];
    if (i == 0 && j == 0 && dc) acc -= (double)dc[b] * (double)c[b];
    pig[(int64_t)b * k * k + (int64_t)i * k + j] = (T)acc;
  }
}

// Pi_gamma[b][col][j] -= (dQ[:,col] . Q[:,j])  from dots partials   (arnoldi.py:128, "- dQ.T @ Q")
// grid (p, cols of this batch): column col0 + blockIdx.y from the partials at partial + blockIdx.y * part_zstride
template <typename T>
__global__ void k_pig_sub(T* __restrict__ pig, int k, int col0, const T* __restrict__ partial, int kmax,
                          int nblk, int64_t part_zstride) {
  const int b = blockIdx.x, col = col0 + (int)blockIdx.y;
  const T* part = partial + (int64_t)blockIdx.y * part_zstride;
  for (int idx = threadIdx.x; idx < k * kRedG; idx += blockDim.x) {
    const int j = idx / kRedG, g = idx % kRedG;
    const T v = reduce_partials_group<T, kRedG>(part + ((int64_t)b * kmax + j) * nblk, nblk, g);
    if (g == 0) pig[((int64_t)b * k + col) * k + j] -= v;
  }
}

// ------------------------------------------------------------------------------------------------
// Arnoldi-adjoint combine (arnoldi.py:212-220), step idx:
//   Gamma[idx][j] = lower[idx][j] (Pi_gamma[idx][j] - (z^T Q)_j)            j <= idx
//   g_j = (Gamma + Gamma^T)[idx][j]                                          all j < k
//   out = (dQ[idx] + eta_idx r + sum_j g_j q_j - alpha lam + z - sum_{j>idx} H[idx][j] Lam_j) / beta_minus
// ------------------------------------------------------------------------------------------------
template <typename T>
struct CombineArgs {
  const T *Q, *Lam, *H, *pig, *eta, *r, *dQ, *z, *partial;
  T* Gam;
  T* out;
  int64_t n;
  int k, idx, kmax, nblk;
  // fused epilogue: partial_out[b][j][slice] = q_j . out, j < m_out -- the re-projection dots of the NEXT step (arnoldi.py:204),
  // whose lambda this kernel has just produced; null = none
  T* partial_out;
  int m_out, nblk_out;
};

template <typename T, int VEC, int EPT>
__global__ __launch_bounds__(kBlock) void k_adj_combine(CombineArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* g = reinterpret_cast<T*>(smem_raw);  // [k]
  T* hp = g + a.k;                         // [k]
  T* sm = hp + a.k;                        // [4][m_out] (fused dots)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y, blk = blockIdx.x;
  const int k = a.k, idx = a.idx;
  const int64_t kk = (int64_t)k * k;
  const T* Hb = a.H + (int64_t)b * kk;
  T* Gb = a.Gam + (int64_t)b * kk;
  for (int jx = tid; jx < k * kRedG; jx += (int)blockDim.x) {
    const int j = jx / kRedG, gl = jx % kRedG;
    T gj;
    if (j <= idx) {
      const T zq = reduce_partials_group<T, kRedG>(a.partial + ((int64_t)b * a.kmax + j) * a.nblk, a.nblk, gl);
      if (gl != 0) continue;
      const T low = (j < idx) ? T(1) : T(0.5);
      const T gam = low * (a.pig[(int64_t)b * kk + (int64_t)idx * k + j] - zq);
      if (blk == 0) Gb[(int64_t)idx * k + j] = gam;
      gj = (j < idx) ? gam : T(2) * gam;
      hp[j] = T(0);
    } else {
      if (gl != 0) continue;
      gj = Gb[(int64_t)j * k + idx];  // written by the combine kernel of step j (earlier launch)
      hp[j] = Hb[(int64_t)idx * k + j];
    }
    g[j] = gj;
  }
  __syncthreads();
  const T alpha = Hb[(int64_t)idx * k + idx];
  const T bminus = (idx == 0) ? T(1) : Hb[(int64_t)idx * k + idx - 1];
  const T eta_i = a.eta[(int64_t)b * k + idx];
  const int64_t slice0 = (int64_t)blk * ((int64_t)blockDim.x * EPT);
  const int64_t n = a.n;
  const T* Qb = a.Q + (int64_t)b * k * n;
  const T* Lb = a.Lam + (int64_t)b * k * n;
  T acc[EPT], t[EPT];
  load_own<T, VEC>(acc, a.z + (int64_t)b * n, slice0, n, tid);
  load_own<T, VEC>(t, Lb + (int64_t)idx * n, slice0, n, tid);
#pragma unroll
  for (int e = 0; e < EPT; ++e) acc[e] -= alpha * t[e];
  load_own<T, VEC>(t, a.r + (int64_t)b * n, slice0, n, tid);
#pragma unroll
  for (int e = 0; e < EPT; ++e) acc[e] += eta_i * t[e];
  if (a.dQ) {
    load_own<T, VEC>(t, a.dQ + ((int64_t)b * k + idx) * n, slice0, n, tid);
#pragma unroll
    for (int e = 0; e < EPT; ++e) acc[e] += t[e];
  }
  constexpr int JT = RowsInFlight<EPT>::value;
  sweep_rows<T, VEC, EPT, JT>(Qb, n, 0, k, slice0, n, tid, [&](int j, const T (&row)[JT][EPT], int nvalid) {
#pragma unroll
    for (int q = 0; q < JT; ++q) {
      const T gj = q < nvalid ? g[j + q] : T(0);
#pragma unroll
      for (int e = 0; e < EPT; ++e) acc[e] += gj * row[q][e];
    }
  });
  sweep_rows<T, VEC, EPT, JT>(Lb, n, idx + 1, k, slice0, n, tid, [&](int j, const T (&row)[JT][EPT], int nvalid) {
#pragma unroll
    for (int q = 0; q < JT; ++q) {
      const T hj = q < nvalid ? hp[j + q] : T(0);
#pragma unroll
      for (int e = 0; e < EPT; ++e) acc[e] -= hj * row[q][e];
    }
  });
  const T inv = T(1) / bminus;
#pragma unroll
  for (int e = 0; e < EPT; ++e) acc[e] *= inv;
  store_own<T, VEC>(acc, a.out + (int64_t)b * n, slice0, n, tid);
  if (a.partial_out) {
    const int m = a.m_out;
    sweep_rows<T, VEC, EPT, JT>(Qb, n, 0, m, slice0, n, tid, [&](int j, const T (&row)[JT][EPT], int nvalid) {
      DotAcc<T> d[JT];
#pragma unroll
      for (int q = 0; q < JT; ++q) {
        d[q] = DotAcc<T>(0);
#pragma unroll
        for (int e = 0; e < EPT; ++e) d[q] += (DotAcc<T>)row[q][e] * (DotAcc<T>)acc[e];
      }
      const T wsum = (T)wave_sums<JT>(d, lane);
      const int q = row16_index<JT>(lane);
      if (wave_sums_writer<JT>(lane) && q < nvalid) sm[wid * m + j + q] = wsum;
    });
    __syncthreads();
    for (int j = tid; j < m; j += (int)blockDim.x) {
      T sum = T(0);
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += sm[w * m + j];
      a.partial_out[((int64_t)b * a.kmax + j) * a.nblk_out + blk] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Lanczos (no re-orthogonalisation) adjoint kernels, lanczos.py:317-335
//   dots:  d0 = lam_plus . x_j ; d1 = x_{j+1} . xi ; d2 = x_j . xi        (xi not yet divided by b_j)
//   lam :  mu = db_j - d0 + d1/b_j ; nu = da_j + d2/b_j ; lam = -xi/b_j + mu x_{j+1} + nu x_j
//   xi' : -dx_j - A lam + a_j lam + b_j lam_plus - b_j nu x_{j+1}
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_lz_adj_dots(const T* __restrict__ xj, const T* __restrict__ xj1,
                                                        int64_t ldxs, const T* __restrict__ xi,
                                                        const T* __restrict__ lam_plus, int64_t ldlp,
                                                        int64_t n, T* __restrict__ partial, int nblk) {
  __shared__ T sm[4][3];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y, blk = blockIdx.x;
  const int64_t slice0 = (int64_t)blk * ((int64_t)blockDim.x * kEpt);
  T x0[kEpt], x1[kEpt], xv[kEpt], lp[kEpt];
  load_own<T, VEC>(x0, xj + (int64_t)b * ldxs, slice0, n, tid);
  load_own<T, VEC>(x1, xj1 + (int64_t)b * ldxs, slice0, n, tid);
  load_own<T, VEC>(xv, xi + (int64_t)b * n, slice0, n, tid);
  if (lam_plus) {
    load_own<T, VEC>(lp, lam_plus + (int64_t)b * ldlp, slice0, n, tid);
  } else {
#pragma unroll
    for (int e = 0; e < kEpt; ++e) lp[e] = T(0);
  }
  T d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
  for (int e = 0; e < kEpt; ++e) {
    d0 += lp[e] * x0[e];
    d1 += x1[e] * xv[e];
    d2 += x0[e] * xv[e];
  }
  d0 = wave_sum(d0);
  d1 = wave_sum(d1);
  d2 = wave_sum(d2);
  if (lane == 0) {
    sm[wid][0] = d0;
    sm[wid][1] = d1;
    sm[wid][2] = d2;
  }
  __syncthreads();
  if (tid < 3) {
    T sum = T(0);
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += sm[w][tid];
    partial[((int64_t)b * 3 + tid) * nblk + blk] = sum;
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_lz_adj_lambda(const T* __restrict__ xj, const T* __restrict__ xj1,
                                                          int64_t ldxs, const T* __restrict__ xi,
                                                          int64_t n, const T* __restrict__ partial, int nblk,
                                                          const T* __restrict__ beta, const T* __restrict__ dalpha,
                                                          const T* __restrict__ dbeta, int k, int j,
                                                          T* __restrict__ lam_out, int64_t ldlam,
                                                          T* __restrict__ munu /* (p,2) */) {
  __shared__ T sc[3];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, blk = blockIdx.x;
  if (tid < 64) {
    const T d0 = reduce_partials_group<T, 64>(partial + ((int64_t)b * 3 + 0) * nblk, nblk, tid);
    const T d1 = reduce_partials_group<T, 64>(partial + ((int64_t)b * 3 + 1) * nblk, nblk, tid);
    const T d2 = reduce_partials_group<T, 64>(partial + ((int64_t)b * 3 + 2) * nblk, nblk, tid);
    const T bj = beta[(int64_t)b * k + j];
    const T mu = dbeta[(int64_t)b * k + j] - d0 + d1 / bj;
    const T nu = dalpha[(int64_t)b * k + j] + d2 / bj;
    sc[0] = mu;
    sc[1] = nu;
    sc[2] = T(1) / bj;
    if (blk == 0) {
      munu[(int64_t)b * 2 + 0] = mu;
      munu[(int64_t)b * 2 + 1] = nu;
    }
  }
  __syncthreads();
  const T mu = sc[0], nu = sc[1], ib = sc[2];
  const int64_t slice0 = (int64_t)blk * ((int64_t)blockDim.x * kEpt);
  T x0[kEpt], x1[kEpt], xv[kEpt];
  load_own<T, VEC>(x0, xj + (int64_t)b * ldxs, slice0, n, tid);
  load_own<T, VEC>(x1, xj1 + (int64_t)b * ldxs, slice0, n, tid);
  load_own<T, VEC>(xv, xi + (int64_t)b * n, slice0, n, tid);
#pragma unroll
  for (int e = 0; e < kEpt; ++e) xv[e] = -xv[e] * ib + mu * x1[e] + nu * x0[e];
  store_own<T, VEC>(xv, lam_out + (int64_t)b * ldlam, slice0, n, tid);
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_lz_adj_xi(const T* __restrict__ dxj, int64_t lddxs,
                                                      const T* __restrict__ Alam, const T* __restrict__ lam,
                                                      int64_t ldlam, const T* __restrict__ lam_plus,
                                                      int64_t ldlp, const T* __restrict__ xj1, int64_t ldxs,
                                                      int64_t n, const T* __restrict__ alpha,
                                                      const T* __restrict__ beta, int k, int j,
                                                      const T* __restrict__ munu, T* __restrict__ xi_out) {
  const int tid = threadIdx.x;
  const int b = blockIdx.y, blk = blockIdx.x;
  const T aj = alpha[(int64_t)b * k + j], bj = beta[(int64_t)b * k + j];
  const T nu = munu[(int64_t)b * 2 + 1];
  const int64_t slice0 = (int64_t)blk * ((int64_t)blockDim.x * kEpt);
  T acc[kEpt], t[kEpt];
  load_own<T, VEC>(acc, Alam + (int64_t)b * n, slice0, n, tid);
#pragma unroll
  for (int e = 0; e < kEpt; ++e) acc[e] = -acc[e];
  if (dxj) {
    load_own<T, VEC>(t, dxj + (int64_t)b * lddxs, slice0, n, tid);
#pragma unroll
    for (int e = 0; e < kEpt; ++e) acc[e] -= t[e];
  }
  load_own<T, VEC>(t, lam + (int64_t)b * ldlam, slice0, n, tid);
#pragma unroll
  for (int e = 0; e < kEpt; ++e) acc[e] += aj * t[e];
  if (lam_plus) {
    load_own<T, VEC>(t, lam_plus + (int64_t)b * ldlp, slice0, n, tid);
#pragma unroll
    for (int e = 0; e < kEpt; ++e) acc[e] += bj * t[e];
  }
  load_own<T, VEC>(t, xj1 + (int64_t)b * ldxs, slice0, n, tid);
#pragma unroll
  for (int e = 0; e < kEpt; ++e) acc[e] -= bj * nu * t[e];
  store_own<T, VEC>(acc, xi_out + (int64_t)b * n, slice0, n, tid);
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_lz_adj_dvec(const T* __restrict__ x0, int64_t ldxs,
                                                        const T* __restrict__ xi, int64_t n,
                                                        const T* __restrict__ partial, int kmax, int nblk,
                                                        const T* __restrict__ vnorm, T* __restrict__ dv) {
  __shared__ T sc[2];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, blk = blockIdx.x;
  if (tid < 64) {
    const T d = reduce_partials_group<T, 64>(partial + ((int64_t)b * kmax) * nblk, nblk, tid);
    if (tid == 0) {
      sc[0] = d;
      sc[1] = T(1) / vnorm[b];
    }
  }
  __syncthreads();
  const T d = sc[0], inv = sc[1];
  const int64_t slice0 = (int64_t)blk * ((int64_t)blockDim.x * kEpt);
  T a[kEpt], x[kEpt];
  load_own<T, VEC>(a, xi + (int64_t)b * n, slice0, n, tid);
  load_own<T, VEC>(x, x0 + (int64_t)b * ldxs, slice0, n, tid);
#pragma unroll
  for (int e = 0; e < kEpt; ++e) a[e] = (d * x[e] - a[e]) * inv;
  store_own<T, VEC>(a, dv + (int64_t)b * n, slice0, n, tid);
}


// ------------------------------------------------------------------------------------------------
// Fused step head for the native CSR operator (few-launch regime: one vector, n ~ 1e5 -- every launch is 3-5 us of fixed
// cost there): ONE kernel does what k_scale + k_csr_apply + k_dots do one after the other,
//   PRE : f = 1 / |x| from the norm partials, qout = f x (the normalised basis vector, arnoldi.py:80 / lanczos.py:258),
//         len_out = |x|;  the operator is applied to f x
//   y   = A (f x) for the rows of this workgroup's slice -- each thread computes the rows it owns in the vector
//         kernels' layout, straight into registers (entries of a row in order, CH at a time so that the index/value
//         loads and then the gathers of all owned rows are in flight together)
//   DOTS: partial[b][j][slice] = rows_j . y over the slice, j < m  (h = Q^T w, arnoldi.py:87; a = x_i . A x_i, lanczos.py:279)
// `perm` (optional) maps a stored position to its slot in val (the transpose structure).  x and y must not overlap
// (other workgroups gather from all of x).
// ------------------------------------------------------------------------------------------------
template <typename T>
struct CsrStepArgs {
  const int32_t *crow, *col, *perm;
  const T* val;
  const T* x;
  int64_t ldx;
  T* y;
  int64_t ldy;
  int64_t n;
  const T* partial_norm;  // PRE: (p, nblk_in)
  int nblk_in;
  T* qout;
  int64_t ldq;
  T* len_out;
  int64_t len_ld;
  const T* rows;  // DOTS
  int64_t rows_ldb, row_stride;
  int m;
  T* partial;
  int kmax, nblk;
};

template <typename T, int VEC, int EPT, bool PRE, bool DOTS>
__global__ __launch_bounds__(kBlock) void k_csr_step(CsrStepArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);  // [4][m] (DOTS)
  __shared__ T f_sh;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y, blk = blockIdx.x;
  const int64_t slice0 = (int64_t)blk * ((int64_t)blockDim.x * EPT);
  const int64_t n = a.n;
  T f = T(1);
  if constexpr (PRE) {
    if (tid < 64) {
      const T len = sqrt(reduce_partials_group<T, 64>(a.partial_norm + (int64_t)b * a.nblk_in, a.nblk_in, tid));
      if (tid == 0) {
        f_sh = T(1) / len;
        if (blk == 0 && a.len_out) a.len_out[(int64_t)b * a.len_ld] = len;
      }
    }
    __syncthreads();
    f = f_sh;
  }
  const T* xb = a.x + (int64_t)b * a.ldx;
  constexpr int U = EPT / VEC;
  constexpr int CH = EPT >= 8 ? 2 : 16 / EPT;  // entries per owned row and round
  int32_t s[EPT], t[EPT];
  int32_t longest = 0;
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int64_t row = slice0 + (int64_t)(u * (int)blockDim.x + tid) * VEC + e;
      const int64_t rr = row < n ? row : 0;
      const int32_t s0 = a.crow[rr], t0 = a.crow[rr + 1];
      s[u * VEC + e] = s0;
      t[u * VEC + e] = row < n ? t0 : s0;  // rows past the end are empty
      longest = max(longest, t[u * VEC + e] - s0);
    }
  T yr[EPT];
#pragma unroll
  for (int r = 0; r < EPT; ++r) yr[r] = T(0);
  for (int32_t o = 0; o < longest; o += CH) {
    int32_t c[EPT][CH];
    T v[EPT][CH];
#pragma unroll
    for (int r = 0; r < EPT; ++r)
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int32_t ee = s[r] + o + i;
        const bool ok = ee < t[r];
        const int32_t e1 = ok ? ee : 0;  // entry 0 exists whenever the loop runs
        c[r][i] = a.col[e1];
        const T vv = a.perm ? a.val[a.perm[e1]] : a.val[e1];
        v[r][i] = ok ? vv : T(0);
      }
#pragma unroll
    for (int r = 0; r < EPT; ++r)
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        T xv = xb[c[r][i]];
        if constexpr (PRE) xv *= f;
        yr[r] += v[r][i] * xv;
      }
  }
  store_own<T, VEC>(yr, a.y + (int64_t)b * a.ldy, slice0, n, tid);
  if constexpr (PRE) {
    T q[EPT];
    load_own<T, VEC>(q, xb, slice0, n, tid);
#pragma unroll
    for (int r = 0; r < EPT; ++r) q[r] *= f;
    store_own<T, VEC>(q, a.qout + (int64_t)b * a.ldq, slice0, n, tid);
  }
  if constexpr (DOTS) {
    constexpr int JT = RowsInFlight<EPT>::value;
    const int m = a.m;
    sweep_rows<T, VEC, EPT, JT>(a.rows + (int64_t)b * a.rows_ldb, a.row_stride, 0, m, slice0, n, tid,
                                [&](int j, const T (&row)[JT][EPT], int nvalid) {
                                  DotAcc<T> acc[JT];
#pragma unroll
                                  for (int q = 0; q < JT; ++q) {
                                    acc[q] = DotAcc<T>(0);
#pragma unroll
                                    for (int e = 0; e < EPT; ++e) acc[q] += (DotAcc<T>)row[q][e] * (DotAcc<T>)yr[e];
                                  }
                                  const T wsum = (T)wave_sums<JT>(acc, lane);
                                  const int q = row16_index<JT>(lane);
                                  if (wave_sums_writer<JT>(lane) && q < nvalid) sm[wid * m + j + q] = wsum;
                                });
    __syncthreads();
    for (int j = tid; j < m; j += (int)blockDim.x) {
      T sum = T(0);
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += sm[w * m + j];
      a.partial[((int64_t)b * a.kmax + j) * a.nblk + blk] = sum;
    }
  }
}


// Columns of dQ projected onto the basis per launch (arnoldi.py:128, dQ^T Q): one when a single column already fills the
// chip, up to k when there are few slices (one vector, n ~ 1e5: k launches of 50 workgroups were 1 ms of config 3's adjoint).
static int64_t dq_batch_cols(int64_t n, int64_t k, int64_t p, const mfx_comm* comm) {
  if (comm) return 1;
  const int64_t wgs = num_slices(n) * p;
  const int64_t cb = 1024 / (wgs > 0 ? wgs : 1);
  return cb < 1 ? 1 : (cb > k ? k : cb);
}

// n = length of the vectors this process holds (the operator size, or the rows of its shard: comm != NULL)
static int64_t carve_ws(const mfx_operator* op, int64_t n, int64_t k, int64_t p, void* ws, int64_t ws_bytes,
                        KrylovWs* out, const mfx_comm* comm = nullptr) {
  const size_t es = dtype_size(op->dtype);
  const int64_t nblk = (n + 64 * kEpt - 1) / (64 * kEpt);  // the finest slicing the drivers may choose
  const int64_t kmax = k + 1;
  Carver cv(ws, ws_bytes);
  KrylovWs r{};
  r.w = cv.take(2 * p * n * es);                // two (p, n) scratch vectors
  r.p1 = cv.take(p * kmax * nblk * es);         // dots partials
  r.p2 = cv.take(p * kmax * nblk * es);         // second-pass partials
  r.pn = cv.take(p * 3 * nblk * es);            // norm / 3-dot partials
  r.small = cv.take((2 * p * k * k + 4 * p * k + 8 * p) * es);  // Gamma, Pi_gamma, eta, coefficients
  r.opws_bytes = op_workspace_bytes(op, p * (k + 1), p);  // the deferred gradient sweep batches all (probe, step) pairs
  r.opws = cv.take(r.opws_bytes);
  const int64_t cb = dq_batch_cols(n, k, p, comm);
  r.pb = cb > 1 ? cv.take(cb * p * kmax * nblk * es) : nullptr;
  if (comm) {
    r.stage = cv.take(p * (kmax > 3 ? kmax : 3) * nblk * es);  // producers' per-slice partials before the all-reduce (3: the three-term adjoint's dots)
    r.send = cv.take(p * comm->nloc * es);                   // this rank's (p, nloc) iterate, zero padded
    r.gathered = cv.take(comm->world * p * comm->nloc * es); // (world, p, nloc)
    r.xfull = cv.take(p * op->n * es);                       // the full iterate (p, n) (adjoint; the forward writes Qfull)
  }
  if (out) *out = r;
  return cv.off;
}

// ------------------------------------------------------------------------------------------------
// fused CSR step head (k_csr_step): when it applies, and its launcher
// ------------------------------------------------------------------------------------------------
template <typename T>
static bool csr_fusable(const mfx_operator* op, const Ctx<T>& c, int transpose) {
  if (op->kind != MFX_OP_CSR || c.comm || op->nrows != 0) return false;
  if (!op->crow || !op->col || !op->val || op->nnz < 1) return false;
  // One thread walks a whole row in the fused step, and a workgroup waits for its longest: skewed matrices (one dense row, a
  // power-law graph) belong to the 8-lanes-per-row kernel.  Decided on the LONGEST row when the caller states it
  // (mfx_operator.max_row_nnz; CsrOp computes it once), on the mean row length otherwise.
  if (op->max_row_nnz > 0 ? op->max_row_nnz > 64 : op->nnz > 24 * op->n) return false;
  if (transpose && !(op->t_crow && op->t_col && op->t_perm)) return false;
  return true;
}

template <typename T>
static int launch_csr_step(const Ctx<T>& c, const mfx_operator* op, int transpose, CsrStepArgs<T> a, bool pre, bool dots) {
  a.crow = transpose ? op->t_crow : op->crow;
  a.col = transpose ? op->t_col : op->col;
  a.perm = transpose ? op->t_perm : nullptr;
  a.val = static_cast<const T*>(op->val);
  a.n = c.n;
  a.kmax = c.kmax;
  a.nblk = c.nblk;
  a.nblk_in = c.nblk_in;
  ScopedTimer t(0, c.stream);
  const size_t sh = dots ? (size_t)4 * a.m * sizeof(T) : 0;
  MFX_REQUIRE(sh <= 64 * 1024, MFX_ERR_UNSUPPORTED, "Krylov depth %d too large for the fused CSR step (%zu B of LDS > 64 KiB)", a.m, sh);
  if (pre && dots) {
    MFX_VEC_EPT_SWITCH(c, (k_csr_step<T, VEC, EPT, true, true><<<c.grid(), c.wg, sh, c.stream>>>(a)));
  } else if (pre) {
    MFX_VEC_EPT_SWITCH(c, (k_csr_step<T, VEC, EPT, true, false><<<c.grid(), c.wg, sh, c.stream>>>(a)));
  } else if (dots) {
    MFX_VEC_EPT_SWITCH(c, (k_csr_step<T, VEC, EPT, false, true><<<c.grid(), c.wg, sh, c.stream>>>(a)));
  } else {
    MFX_VEC_EPT_SWITCH(c, (k_csr_step<T, VEC, EPT, false, false><<<c.grid(), c.wg, sh, c.stream>>>(a)));
  }
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}


// ------------------------------------------------------------------------------------------------
// drivers
// ------------------------------------------------------------------------------------------------
// i * (re, im) = (-im, re) on interleaved complex vectors held as 2 n reals: the second basis slot of a complex Arnoldi step
template <typename T>
__global__ __launch_bounds__(256) void k_times_i(const T* __restrict__ q, T* __restrict__ jq, int64_t ld, int64_t ncomplex) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ncomplex) return;
  const T* src = q + (int64_t)blockIdx.y * ld + 2 * i;
  T* dst = jq + (int64_t)blockIdx.y * ld + 2 * i;
  const T re = src[0], im = src[1];
  dst[0] = -im;
  dst[1] = re;
}

// comm != NULL: row-sharded (n = rows of this rank, op->n = operator size, Qfull (p, k, op->n) receives the gathered basis)
//
// S = 2: COMPLEX vectors of n / 2 entries, held interleaved as n reals, and an operator that is complex-linear on them.  A
// complex Krylov vector q takes TWO real basis slots, q and i q: the complex projection h = q^H w is (<q, w>, <i q, w>) in
// real inner products, and w -= h q is w -= Re(h) q + Im(h) (i q) -- so the complex recurrence (arnoldi.py:66,87,92,95 with
// .conj()) is the real one on 2 (i + 1) slots, run by the same dots / update kernels.  Q is then (p, 2 k, n) and H is
// (p, 2 k, k): rows 2 a and 2 a + 1 of column i hold Re and Im of H[a][i].
template <typename T>
static int arnoldi_forward_t(const mfx_operator* op, const T* v0, int64_t n, int64_t k, int64_t p,
                             int second_pass, T* Q, T* H, T* r, T* cinv, const KrylovWs& ws,
                             hipStream_t stream, const mfx_comm* comm = nullptr, T* Qfull = nullptr, int S = 1) {
  Ctx<T> c(n, S * k, p, pick_vec<T>(n, {v0, Q, r, ws.w}), stream);
  if (comm) c.shard(comm, static_cast<T*>(ws.stage));
  c.fine();
  T* P1 = static_cast<T*>(ws.p1);
  T* P2 = static_cast<T*>(ws.p2);
  T* PN = static_cast<T*>(ws.pn);
  const int64_t ldq = S * k * n;
  const int64_t ldh = S * k * k;
  auto second_slot = [&](int64_t i) -> int {  // slot S i + 1 = i * slot S i
    if (S == 1) return MFX_OK;
    k_times_i<T><<<dim3((unsigned)((n / 2 + 255) / 256), (unsigned)p), 256, 0, stream>>>(Q + S * i * n, Q + (S * i + 1) * n, ldq, n / 2);
    MFX_CHECK_LAUNCH();
    return MFX_OK;
  };
  MFX_TRY(zero_async(H, sizeof(T) * p * ldh, stream));
  {
    ScopedTimer t(2, stream);
    MFX_TRY(launch_sumsq<T>(c, v0, n, PN));
    MFX_TRY(launch_scale<T>(c, v0, n, Q, ldq, PN, nullptr, 0, nullptr, 0, cinv));  // q_0, c = 1/|v|
    MFX_TRY(second_slot(0));
  }
  T* w = r;  // the running vector lives in the remainder output (arnoldi.py:75 returns it as r)
  // CSR operator, few slices: normalisation of the previous step, operator application and h = Q^T w in one launch
  // (k_csr_step).  The un-normalised w of step i - 1 is the input of step i, so w alternates between r and a scratch
  // vector, arranged so that the last step ends in r.
  const bool fused = S == 1 && csr_fusable<T>(op, c, 0);
  T* const scratch_w = static_cast<T*>(ws.w);
  for (int64_t i = 0; i < k; ++i) {
    const int m = (int)(S * (i + 1));
    if (fused) {
      T* const w_prev = w;
      w = (((i ^ (k - 1)) & 1) != 0) ? scratch_w : r;
      CsrStepArgs<T> cs{};
      cs.y = w; cs.ldy = n;
      cs.rows = Q; cs.rows_ldb = ldq; cs.row_stride = n; cs.m = m; cs.partial = P1;
      if (i == 0) {
        cs.x = Q; cs.ldx = ldq;
        MFX_TRY(launch_csr_step<T>(c, op, 0, cs, false, true));
      } else {  // q_i = w / |w|, H[i][i-1] = |w| (arnoldi.py:80,99) from the norm partials of step i - 1
        cs.x = w_prev; cs.ldx = n;
        cs.partial_norm = PN; cs.qout = Q + i * n; cs.ldq = ldq; cs.len_out = H + i * k + (i - 1); cs.len_ld = k * k;
        MFX_TRY(launch_csr_step<T>(c, op, 0, cs, true, true));
      }
    } else if (comm) {
      MFX_TRY(apply_sharded<T>(op, comm, 0, Q + i * n, ldq, w, n, p, Qfull + i * op->n, k * op->n, ws, stream));
    } else {
      MFX_TRY(apply_any(op, 0, Q + S * i * n, ldq, nullptr, 0, w, n, p, ws.opws, ws.opws_bytes, stream));
    }
    ScopedTimer t(2, stream);
    if (!fused) MFX_TRY(launch_dots<T>(c, Q, ldq, n, m, w, n, P1));
    UpdateArgs<T> a{};
    a.rows = Q; a.rows_ldb = ldq; a.row_stride = n; a.m = m;
    a.partial_in = P1; a.s1 = T(1);
    a.hout = H + i; a.hout_ldb = ldh; a.hout_stride = k;  // H[:, i] (arnoldi.py:99)
    a.x = w; a.ldx = n; a.y = w; a.ldy = n;
    a.partial_out = P2; a.partial_norm = PN;
    if (second_pass) {
      MFX_TRY(launch_update<T>(c, a, true, false));
      UpdateArgs<T> a2{};
      a2.rows = Q; a2.rows_ldb = ldq; a2.row_stride = n; a2.m = m;
      a2.partial_in = P2; a2.s1 = T(1);  // second-pass coefficients are not added to h (arnoldi.py:92)
      a2.x = w; a2.ldx = n; a2.y = w; a2.ldy = n; a2.partial_norm = PN;
      MFX_TRY(launch_update<T>(c, a2, false, true));
    } else {
      MFX_TRY(launch_update<T>(c, a, false, true));
    }
    if (i + 1 < k && !fused) {  // Q2: H[k][k-1] does not exist; the last vector stays un-normalised in r
      MFX_TRY(launch_scale<T>(c, w, n, Q + S * (i + 1) * n, ldq, PN, nullptr, 0, H + S * (i + 1) * k + i, ldh, nullptr));
      MFX_TRY(second_slot(i + 1));
    }
  }
  return MFX_OK;
}

// (p, 2 k, k) real rows (Re, Im interleaved by ROW) -> (p, k, k) complex entries; c -> (c, 0)
template <typename T>
__global__ __launch_bounds__(256) void k_complex_small(const T* __restrict__ Hext, const T* __restrict__ cinv, T* __restrict__ H,
                                                       T* __restrict__ c, int k) {
  const int b = blockIdx.x;
  for (int e = threadIdx.x; e < k * k; e += 256) {
    const int a = e / k, i = e % k;
    H[((int64_t)b * k * k + e) * 2] = Hext[(int64_t)b * 2 * k * k + (2 * a) * k + i];
    H[((int64_t)b * k * k + e) * 2 + 1] = Hext[(int64_t)b * 2 * k * k + (2 * a + 1) * k + i];
  }
  if (threadIdx.x == 0) {
    c[2 * b] = cinv[b];
    c[2 * b + 1] = T(0);
  }
}

struct ComplexWs {
  void* Qext;  // (p, 2 k, 2 n)
  void* Hext;  // (p, 2 k, k)
  void* cinv;  // (p)
  KrylovWs kws;
};
static int64_t carve_complex_ws(const mfx_operator* op, int64_t n, int64_t k, int64_t p, void* ws, int64_t ws_bytes, ComplexWs* out) {
  const size_t es = dtype_size(op->dtype);
  Carver cv(ws, ws_bytes);
  ComplexWs r{};
  r.Qext = cv.take(p * 2 * k * 2 * n * es);
  r.Hext = cv.take(p * 2 * k * k * es);
  r.cinv = cv.take(p * es);
  const int64_t inner = carve_ws(op, 2 * n, 2 * k, p, ws ? static_cast<char*>(ws) + cv.off : nullptr, ws ? ws_bytes - cv.off : 0, &r.kws);
  if (out) *out = r;
  return cv.off + inner;
}

template <typename T>
static int arnoldi_forward_complex_t(const mfx_operator* op, const T* v0, int64_t n, int64_t k, int64_t p, int second_pass, T* Q,
                                     T* H, T* r, T* c, const ComplexWs& w, hipStream_t stream) {
  T* Qext = static_cast<T*>(w.Qext);
  T* Hext = static_cast<T*>(w.Hext);
  MFX_TRY(arnoldi_forward_t<T>(op, v0, 2 * n, k, p, second_pass, Qext, Hext, r, static_cast<T*>(w.cinv), w.kws, stream, nullptr,
                               nullptr, 2));
  // the complex basis = the even slots; (b, i) -> one row of 2 n reals
  for (int64_t b = 0; b < p; ++b)
    MFX_TRY(copy_rows_async(Q + b * k * 2 * n, sizeof(T) * 2 * n, Qext + b * 2 * k * 2 * n, sizeof(T) * 4 * n, sizeof(T) * 2 * n, (size_t)k, stream));
  k_complex_small<T><<<(unsigned)p, 256, 0, stream>>>(Hext, static_cast<const T*>(w.cinv), H, c, (int)k);
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

template <typename T>
static int arnoldi_adjoint_t(const mfx_operator* op, int64_t n, int64_t k, int64_t p, const T* Q, const T* H,
                             const T* r, const T* cinv, const T* dQ, const T* dH, const T* dr, const T* dc,
                             int reortho, T* dv, T* Lam, const mfx_op_grads* grads, const KrylovWs& ws,
                             hipStream_t stream, const mfx_comm* comm = nullptr, const T* Qfull = nullptr) {
  Ctx<T> c(n, k, p, pick_vec<T>(n, {Q, r, dQ, dr, dv, Lam, ws.w}), stream);
  if (comm) c.shard(comm, static_cast<T*>(ws.stage));
  c.fine();
  T* P1 = static_cast<T*>(ws.p1);
  T* P2 = static_cast<T*>(ws.p2);
  T* lam = static_cast<T*>(ws.w);  // current lambda (p, n)
  T* z = lam + p * n;              // A^T lambda     (p, n)
  T* Gam = static_cast<T*>(ws.small);
  T* pig = Gam + p * k * k;
  T* eta = pig + p * k * k;
  const int64_t ldq = k * n;
  const bool fused = csr_fusable<T>(op, c, 1);
  MFX_TRY(zero_async(Gam, sizeof(T) * p * k * k, stream));
  {
    ScopedTimer t(2, stream);
    if (dr) MFX_TRY(launch_dots<T>(c, Q, ldq, n, (int)k, dr, n, P1));
    k_adj_setup<T><<<dim3((unsigned)p, (unsigned)k), 64, 0, stream>>>(H, dH, cinv, dc, dr ? P1 : nullptr, c.kmax, c.nblk_in, (int)k, eta, pig);
    MFX_CHECK_LAUNCH();
    if (dQ) {
      const int64_t cb = ws.pb ? dq_batch_cols(n, k, p, comm) : 1;
      T* PB = ws.pb ? static_cast<T*>(ws.pb) : P1;
      const int64_t pstride = p * (int64_t)c.kmax * c.nblk;
      for (int64_t col = 0; col < k; col += cb) {
        const int nc = (int)(col + cb <= k ? cb : k - col);
        MFX_TRY(launch_dots<T>(c, Q, ldq, n, (int)k, dQ + col * n, ldq, PB, nc, n, pstride));
        k_pig_sub<T><<<dim3((unsigned)p, (unsigned)nc), 512, 0, stream>>>(pig, (int)k, (int)col, PB, c.kmax, c.nblk_in, pstride);  // latency-bound: 8 lanes per coefficient
        MFX_CHECK_LAUNCH();
      }
    }
    // lambda_k = dr + Q eta  (arnoldi.py:121)
    UpdateArgs<T> a{};
    a.rows = Q; a.rows_ldb = ldq; a.row_stride = n; a.m = (int)k;
    a.extra = eta; a.extra_ldb = k; a.extra_stride = 1; a.s2 = T(-1);
    a.x = dr; a.ldx = n; a.y = lam; a.ldy = n;
    // full re-orthogonalisation: whoever produces a lambda also takes its dots with the basis (Q^T lambda, arnoldi.py:204),
    // the first thing the next step needs -- here for step k - 1 (all k rows), then in the epilogue of k_adj_combine
    a.partial_out = P1;
    MFX_TRY(launch_update<T>(c, a, reortho == MFX_REORTHO_FULL, false));
  }
  for (int64_t idx = k - 1; idx >= 0; --idx) {
    T* lam_idx = Lam + idx * n;  // Lambda[:, idx] (arnoldi.py:216), leading dimension ldq
    {
      ScopedTimer t(2, stream);
      if (reortho == MFX_REORTHO_FULL) {
        const int m = (int)((idx + 2 < k) ? idx + 2 : k);  // rows of P not yet masked (arnoldi.py:201); their dots are in P1
        UpdateArgs<T> a{};
        a.rows = Q; a.rows_ldb = ldq; a.row_stride = n; a.m = m;
        a.partial_in = P1; a.s1 = T(1);
        a.extra = dH + idx; a.extra_ldb = k * k; a.extra_stride = k; a.s2 = T(-1);  // - P^T (mask o dH[:, idx])
        a.x = lam; a.ldx = n; a.y = lam_idx; a.ldy = ldq;
        MFX_TRY(launch_update<T>(c, a, false, false));
      } else {
        MFX_TRY(copy_rows_async(lam_idx, sizeof(T) * ldq, lam, sizeof(T) * n, sizeof(T) * n, p, stream));
      }
    }
    // z = A^T lambda (+ parameter gradient for callback operators), arnoldi.py:207-209
    if (fused) {  // z = A^T lambda and z^T Q (arnoldi.py:207-212) in one launch
      CsrStepArgs<T> cs{};
      cs.x = lam_idx; cs.ldx = ldq; cs.y = z; cs.ldy = n;
      cs.rows = Q; cs.rows_ldb = ldq; cs.row_stride = n; cs.m = (int)(idx + 1); cs.partial = P2;
      MFX_TRY(launch_csr_step<T>(c, op, 1, cs, false, true));
    } else if (comm) {
      MFX_TRY(apply_sharded<T>(op, comm, 1, lam_idx, ldq, z, n, p, static_cast<T*>(ws.xfull), op->n, ws, stream));
    } else {
      MFX_TRY(apply_any(op, 1, lam_idx, ldq, Q + idx * n, ldq, z, n, p, ws.opws, ws.opws_bytes, stream));
    }
    ScopedTimer t(2, stream);
    if (!fused) MFX_TRY(launch_dots<T>(c, Q, ldq, n, (int)(idx + 1), z, n, P2));
    CombineArgs<T> ca{Q, Lam, H, pig, eta, r, dQ, z, P2, Gam, lam, n, (int)k, (int)idx, c.kmax, c.nblk_in, nullptr, 0, c.nblk};
    if (reortho == MFX_REORTHO_FULL && idx > 0) {  // dots of the lambda of step idx - 1: rows min(idx + 1, k)
      ca.partial_out = c.producer(P1);
      ca.m_out = (int)((idx + 1 < k) ? idx + 1 : k);
    }
    const size_t sh = (size_t)(2 * k + 4 * ca.m_out) * sizeof(T);
    MFX_REQUIRE(sh <= 64 * 1024, MFX_ERR_UNSUPPORTED, "Krylov depth %lld too large for the adjoint combine kernel (%zu B of LDS > 64 KiB)",
                (long long)k, sh);
    MFX_VEC_EPT_SWITCH(c, (k_adj_combine<T, VEC, EPT><<<c.grid(), c.wg, sh, stream>>>(ca)));
    MFX_CHECK_LAUNCH();
    if (ca.partial_out) MFX_TRY(c.finish(P1, c.kmax, ca.m_out));
  }
  {
    ScopedTimer t(2, stream);
    MFX_TRY(launch_scale<T>(c, lam, n, dv, n, nullptr, cinv, 1, nullptr, 0, nullptr));  // dv = lambda c
  }
  if (op->kind != MFX_OP_CALLBACK && grads) {
    ScopedTimer t(1, stream);
    if (comm) {  // this rank's rows of S = Lambda^T Q against all columns: L = local adjoint states, R = the gathered basis
      mfx_operator rows = *op;
      rows.row0 = shard_row0(comm);
      rows.nrows = n;
      MFX_TRY(op_vjp_params(&rows, Lam, n, Qfull, op->n, p * k, grads, ws.opws, ws.opws_bytes, stream, k));
    } else {
      MFX_TRY(op_vjp_params(op, Lam, n, Q, n, p * k, grads, ws.opws, ws.opws_bytes, stream, k));
    }
  }
  return MFX_OK;
}

// comm != NULL: row-sharded (n = rows of this rank; every x_i is gathered into the scratch ws.xfull for the operator)
template <typename T>
static int lanczos_forward_t(const mfx_operator* op, const T* v0, int64_t n, int64_t k, int64_t p, T* xs,
                             T* alpha, T* beta, T* vnorm, const KrylovWs& ws, hipStream_t stream,
                             const mfx_comm* comm = nullptr) {
  Ctx<T> c(n, k, p, pick_vec<T>(n, {v0, xs, ws.w}), stream);
  if (comm) c.shard(comm, static_cast<T*>(ws.stage));
  c.fine();
  const int64_t prow = comm ? 1 : c.nblk;  // stride of a coefficient row in the partials (sharded: already summed)
  T* P1 = static_cast<T*>(ws.p1);
  T* P2 = static_cast<T*>(ws.p2);
  T* PN = static_cast<T*>(ws.pn);
  T* w = static_cast<T*>(ws.w);
  const int64_t ldx = (k + 1) * n;
  // r = w - a x_i - b_{i-1} x_{i-1} (lanczos.py:281-282) in ONE update over the rows (x_{i-1}, x_i): coefficient 0 is
  // 0 * partials + beta[i-1], coefficient 1 is the dot a = x_i . w (its partials in row 1 of P1, row 0 stays zero) + beta[i],
  // which is still zero when step i reads it -- so the three-term step needs no separate scalar kernel.
  MFX_TRY(zero_async(beta, sizeof(T) * p * k, stream));
  MFX_TRY(zero_async(P1, sizeof(T) * p * c.kmax * c.nblk, stream));
  {
    ScopedTimer t(2, stream);
    MFX_TRY(launch_sumsq<T>(c, v0, n, PN));
    MFX_TRY(launch_scale<T>(c, v0, n, xs, ldx, PN, nullptr, 0, vnorm, 1, nullptr));
  }
  // CSR operator, few slices: x_i = w / |w|, b_{i-1} = |w|, A x_i and a = x_i . A x_i in one launch (k_csr_step); w alternates
  // between the two scratch vectors because the un-normalised w of step i - 1 is the input of step i.
  const bool fused = csr_fusable<T>(op, c, 0);
  T* const w0 = w;
  for (int64_t i = 0; i < k; ++i) {
    const int m = i == 0 ? 1 : 2;
    T* P = i == 0 ? P2 : P1;
    if (fused) {
      T* const w_prev = w;
      w = (i & 1) ? w0 + p * n : w0;
      CsrStepArgs<T> cs{};
      cs.y = w; cs.ldy = n;
      cs.rows = xs + i * n; cs.rows_ldb = ldx; cs.row_stride = n; cs.m = 1; cs.partial = P + (m - 1) * prow;
      if (i == 0) {
        cs.x = xs; cs.ldx = ldx;
        MFX_TRY(launch_csr_step<T>(c, op, 0, cs, false, true));
      } else {
        cs.x = w_prev; cs.ldx = n;
        cs.partial_norm = PN; cs.qout = xs + i * n; cs.ldq = ldx; cs.len_out = beta + (i - 1); cs.len_ld = k;
        MFX_TRY(launch_csr_step<T>(c, op, 0, cs, true, true));
      }
    } else if (comm) {
      MFX_TRY(apply_sharded<T>(op, comm, 0, xs + i * n, ldx, w, n, p, static_cast<T*>(ws.xfull), op->n, ws, stream));
    } else {
      MFX_TRY(apply_any(op, 0, xs + i * n, ldx, nullptr, 0, w, n, p, ws.opws, ws.opws_bytes, stream));
    }
    ScopedTimer t(2, stream);
    if (!fused) MFX_TRY(launch_dots<T>(c, xs + i * n, ldx, n, 1, w, n, P + (m - 1) * prow));  // a = x_i . A x_i
    UpdateArgs<T> a{};
    a.rows = xs + (i == 0 ? 0 : (i - 1) * n); a.rows_ldb = ldx; a.row_stride = n; a.m = m;
    a.partial_in = P; a.s1 = T(1);
    if (i > 0) { a.extra = beta + (i - 1); a.extra_ldb = k; a.extra_stride = 1; a.s2 = T(1); }
    a.hout = alpha + i; a.hout_ldb = k; a.hout_stride = 0; a.hout_from = m - 1;
    a.x = w; a.ldx = n; a.y = w; a.ldy = n; a.partial_norm = PN;
    MFX_TRY(launch_update<T>(c, a, false, true));
    if (!fused || i + 1 == k) MFX_TRY(launch_scale<T>(c, w, n, xs + (i + 1) * n, ldx, PN, nullptr, 0, beta + i, k, nullptr));
  }
  return MFX_OK;
}

// comm != NULL: row-sharded; Lamfull (p, k, op->n) receives the gathered adjoint states (the operator input of every step and
// the right factor of the parameter-gradient sweep)
template <typename T>
static int lanczos_adjoint_t(const mfx_operator* op, int64_t n, int64_t k, int64_t p, const T* xs, const T* alpha,
                             const T* beta, const T* vnorm, const T* dxs, const T* dalpha, const T* dbeta, T* dv,
                             T* Lam, const mfx_op_grads* grads, const KrylovWs& ws, hipStream_t stream,
                             const mfx_comm* comm = nullptr, T* Lamfull = nullptr) {
  Ctx<T> c(n, k, p, pick_vec<T>(n, {xs, dxs, dv, Lam, ws.w}), stream);
  if (comm) c.shard(comm, static_cast<T*>(ws.stage));
  T* P1 = static_cast<T*>(ws.p1);
  T* PN = static_cast<T*>(ws.pn);
  T* xi = static_cast<T*>(ws.w);
  T* y = xi + p * n;
  T* munu = static_cast<T*>(ws.small);
  const int64_t ldx = (k + 1) * n, ldl = k * n;
  // xi = -dx_k (lanczos.py:304)
  if (dxs) {
    MFX_TRY(launch_scale<T>(c, dxs + k * n, ldx, xi, n, nullptr, nullptr, 2, nullptr, 0, nullptr));
  } else {
    MFX_TRY(zero_async(xi, sizeof(T) * p * n, stream));
  }
  for (int64_t j = k - 1; j >= 0; --j) {
    const T* xj = xs + j * n;
    const T* xj1 = xs + (j + 1) * n;
    const T* lam_plus = (j + 1 < k) ? Lam + (j + 1) * n : nullptr;
    T* lam_j = Lam + j * n;
    {
      ScopedTimer t(2, stream);
      MFX_VEC_SWITCH(c.vec, (k_lz_adj_dots<T, VEC><<<c.grid(), c.wg, 0, stream>>>(xj, xj1, ldx, xi, lam_plus, ldl, n, c.producer(PN), c.nblk)));
      MFX_CHECK_LAUNCH();
      MFX_TRY(c.finish(PN, 3, 3));
      MFX_VEC_SWITCH(c.vec, (k_lz_adj_lambda<T, VEC><<<c.grid(), c.wg, 0, stream>>>(
                                xj, xj1, ldx, xi, n, PN, c.nblk_in, beta, dalpha, dbeta, (int)k, (int)j, lam_j, ldl, munu)));
      MFX_CHECK_LAUNCH();
    }
    // A lambda (Q4: not A^T), parameter gradient of x_j^T A(theta) lambda (lanczos.py:328-329)
    if (comm) {
      MFX_TRY(apply_sharded<T>(op, comm, 0, lam_j, ldl, y, n, p, Lamfull + j * op->n, k * op->n, ws, stream));
    } else {
      MFX_TRY(apply_any(op, 2, lam_j, ldl, xj, ldx, y, n, p, ws.opws, ws.opws_bytes, stream));
    }
    ScopedTimer t(2, stream);
    MFX_VEC_SWITCH(c.vec, (k_lz_adj_xi<T, VEC><<<c.grid(), c.wg, 0, stream>>>(
                              dxs ? dxs + j * n : nullptr, ldx, y, lam_j, ldl, lam_plus, ldl, xj1, ldx, n, alpha, beta,
                              (int)k, (int)j, munu, xi)));
    MFX_CHECK_LAUNCH();
  }
  {
    ScopedTimer t(2, stream);
    MFX_TRY(launch_dots<T>(c, xs, ldx, n, 1, xi, n, P1));  // xi . x_0 (Q3: "lambda_1" is the final xi)
    MFX_VEC_SWITCH(c.vec, (k_lz_adj_dvec<T, VEC><<<c.grid(), c.wg, 0, stream>>>(xs, ldx, xi, n, P1, c.kmax, c.nblk_in, vnorm, dv)));
    MFX_CHECK_LAUNCH();
  }
  if (op->kind != MFX_OP_CALLBACK && grads) {
    // d/dtheta sum_j x_j^T A(theta) lambda_j : L = xs (ld (k+1) n per probe), R = Lambda
    ScopedTimer t(1, stream);
    if (comm) {  // this rank's rows of the x_j against ALL entries of the lambda_j
      mfx_operator rows = *op;
      rows.row0 = shard_row0(comm);
      rows.nrows = n;
      for (int64_t b = 0; b < p; ++b)
        MFX_TRY(op_vjp_params(&rows, xs + b * ldx, n, Lamfull + b * k * op->n, op->n, k, grads, ws.opws, ws.opws_bytes, stream));
    } else {
      for (int64_t b = 0; b < p; ++b)
        MFX_TRY(op_vjp_params(op, xs + b * ldx, n, Lam + b * ldl, n, k, grads, ws.opws, ws.opws_bytes, stream));
    }
  }
  return MFX_OK;
}

static int check_common(const mfx_operator* op, int64_t n, int64_t k, int64_t p) {
  MFX_REQUIRE(op != nullptr, MFX_ERR_INVALID, "operator is NULL");
  MFX_REQUIRE(op->dtype == MFX_F32 || op->dtype == MFX_F64, MFX_ERR_UNSUPPORTED, "unsupported dtype %d", op->dtype);
  MFX_REQUIRE(n >= 1 && p >= 1, MFX_ERR_INVALID, "n=%lld, p=%lld must be positive", (long long)n, (long long)p);
  MFX_REQUIRE(op->n == n, MFX_ERR_INVALID, "operator size %lld != n=%lld", (long long)op->n, (long long)n);
  MFX_REQUIRE(op->nrows == 0, MFX_ERR_INVALID, "the Krylov drivers take the whole operator (nrows = 0); row shards go through mfx_*_sharded");
  MFX_REQUIRE(k >= 1 && k <= n, MFX_ERR_INVALID, "Parameter depth %lld is outside the expected range", (long long)k);
  MFX_REQUIRE(p <= 65535, MFX_ERR_UNSUPPORTED, "p=%lld exceeds the grid limit 65535", (long long)p);
  MFX_REQUIRE((int64_t)(4 * (k + 1) + (k + 1) + 4) * 8 <= 64 * 1024, MFX_ERR_UNSUPPORTED, "k=%lld too large", (long long)k);
  return MFX_OK;
}

}  // namespace mfx

using namespace mfx;

extern "C" {

int64_t mfx_workspace_bytes(const mfx_operator* op, int64_t n, int64_t k, int64_t p) {
  if (!op) return -1;
  return carve_ws(op, n, k, p, nullptr, 0, nullptr);
}

int mfx_op_apply(const mfx_operator* op, const void* x, int64_t ldx, void* y, int64_t ldy, int64_t p,
                 int transpose, void* ws, int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(op && x && y, MFX_ERR_INVALID, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (op->kind == MFX_OP_CALLBACK) return op_apply_cb(op, transpose ? 1 : 0, x, ldx, nullptr, 0, y, ldy, p, s);
  ScopedTimer t(0, s);
  return op_apply(op, x, ldx, y, ldy, p, transpose, ws, ws_bytes, s);
}

int mfx_op_vjp_params(const mfx_operator* op, const void* L, int64_t ldl, const void* R, int64_t ldr,
                      int64_t batch, const mfx_op_grads* grads, void* ws, int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(op && L && R && grads, MFX_ERR_INVALID, "null argument");
  MFX_REQUIRE(op->kind != MFX_OP_CALLBACK, MFX_ERR_UNSUPPORTED, "callback operators own their parameter gradients");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ScopedTimer t(1, s);
  return op_vjp_params(op, L, ldl, R, ldr, batch, grads, ws, ws_bytes, s);
}

// launch-bound regime: few slices and a native operator (a callback enqueues work the capture cannot see)
static bool graph_eligible(const mfx_operator* op, int64_t n, int64_t p) {
  return op->kind != MFX_OP_CALLBACK && num_slices(n) * p < 256;
}
static GraphKey driver_key(int fn_id, const mfx_operator* op, int64_t n, int64_t k, int64_t p, const mfx_op_grads* grads,
                           const void* ws, int64_t ws_bytes, const void* stream) {
  GraphKey key;
  key.add(fn_id).add(*op).add(n).add(k).add(p).add(ws).add(ws_bytes).add(stream);
  const mfx_op_grads none{};
  key.add(grads ? *grads : none).add(grads != nullptr);
  return key;
}

#define MFX_DRIVER_PROLOGUE()                                                                   \
  MFX_TRY(check_common(op, n, k, p));                                                           \
  KrylovWs kws;                                                                                 \
  const int64_t need = carve_ws(op, n, k, p, ws, ws_bytes, &kws);                                \
  MFX_REQUIRE(ws && need <= ws_bytes, MFX_ERR_WORKSPACE, "workspace too small: need %lld bytes", \
              (long long)need);                                                                 \
  hipStream_t s = static_cast<hipStream_t>(stream);                                               \
  PrepScope prep_scope

int mfx_arnoldi_forward(const mfx_operator* op, const void* v0, int64_t n, int64_t k, int64_t p,
                        int second_pass, void* Q, void* H, void* r, void* c, void* ws, int64_t ws_bytes,
                        void* stream) {
  MFX_REQUIRE(v0 && Q && H && r && c, MFX_ERR_INVALID, "null argument");
  MFX_DRIVER_PROLOGUE();
  GraphKey key = driver_key(1, op, n, k, p, nullptr, ws, ws_bytes, stream);
  key.add(v0).add(second_pass).add(Q).add(H).add(r).add(c);
  return run_graphed(graph_eligible(op, n, p), key, s, [&](hipStream_t st) {
    if (op->dtype == MFX_F32)
      return arnoldi_forward_t<float>(op, (const float*)v0, n, k, p, second_pass, (float*)Q, (float*)H, (float*)r, (float*)c, kws, st);
    return arnoldi_forward_t<double>(op, (const double*)v0, n, k, p, second_pass, (double*)Q, (double*)H, (double*)r, (double*)c, kws, st);
  });
}

int64_t mfx_complex_workspace_bytes(const mfx_operator* op, int64_t n, int64_t k, int64_t p) {
  if (!op) return -1;
  return carve_complex_ws(op, n, k, p, nullptr, 0, nullptr);
}

int mfx_arnoldi_forward_complex(const mfx_operator* op, const void* v0, int64_t n, int64_t k, int64_t p, int second_pass, void* Q,
                                void* H, void* r, void* c, void* ws, int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(op && v0 && Q && H && r && c, MFX_ERR_INVALID, "null argument");
  MFX_REQUIRE(k >= 1 && k <= n, MFX_ERR_INVALID, "Parameter depth %lld is outside the expected range", (long long)k);
  MFX_TRY(check_common(op, 2 * n, 2 * k, p));  // the operator is the real form (size 2 n) of the complex-linear map
  MFX_REQUIRE(k <= 65535, MFX_ERR_UNSUPPORTED, "k=%lld exceeds the grid limit 65535", (long long)k);
  ComplexWs cw;
  const int64_t need = carve_complex_ws(op, n, k, p, ws, ws_bytes, &cw);
  MFX_REQUIRE(ws && need <= ws_bytes, MFX_ERR_WORKSPACE, "workspace too small: need %lld bytes", (long long)need);
  hipStream_t s = static_cast<hipStream_t>(stream);
  PrepScope prep_scope;
  if (op->dtype == MFX_F32)
    return arnoldi_forward_complex_t<float>(op, (const float*)v0, n, k, p, second_pass, (float*)Q, (float*)H, (float*)r, (float*)c, cw, s);
  return arnoldi_forward_complex_t<double>(op, (const double*)v0, n, k, p, second_pass, (double*)Q, (double*)H, (double*)r, (double*)c, cw, s);
}

int mfx_arnoldi_adjoint(const mfx_operator* op, int64_t n, int64_t k, int64_t p, const void* Q, const void* H,
                        const void* r, const void* c, const void* dQ, const void* dH, const void* dr,
                        const void* dc, int reortho, void* dv, void* Lambda, const mfx_op_grads* grads, void* ws,
                        int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(Q && H && r && c && dH && dv && Lambda, MFX_ERR_INVALID, "null argument");
  MFX_REQUIRE(reortho == MFX_REORTHO_NONE || reortho == MFX_REORTHO_FULL, MFX_ERR_INVALID, "bad reortho flag %d", reortho);
  MFX_DRIVER_PROLOGUE();
  GraphKey key = driver_key(2, op, n, k, p, grads, ws, ws_bytes, stream);
  key.add(Q).add(H).add(r).add(c).add(dQ).add(dH).add(dr).add(dc).add(reortho).add(dv).add(Lambda);
  return run_graphed(graph_eligible(op, n, p), key, s, [&](hipStream_t st) {
    if (op->dtype == MFX_F32)
      return arnoldi_adjoint_t<float>(op, n, k, p, (const float*)Q, (const float*)H, (const float*)r, (const float*)c,
                                      (const float*)dQ, (const float*)dH, (const float*)dr, (const float*)dc, reortho,
                                      (float*)dv, (float*)Lambda, grads, kws, st);
    return arnoldi_adjoint_t<double>(op, n, k, p, (const double*)Q, (const double*)H, (const double*)r, (const double*)c,
                                     (const double*)dQ, (const double*)dH, (const double*)dr, (const double*)dc, reortho,
                                     (double*)dv, (double*)Lambda, grads, kws, st);
  });
}

static int check_sharded(const mfx_operator* op, const mfx_comm* cm, int64_t n, int64_t k, int64_t p) {
  MFX_TRY(check_common(op, n, k, p));
  MFX_REQUIRE(cm && cm->allreduce_sum && cm->allgather, MFX_ERR_INVALID, "communicator without callbacks");
  MFX_REQUIRE(op->kind != MFX_OP_CALLBACK, MFX_ERR_UNSUPPORTED, "row sharding needs a native operator");
  MFX_REQUIRE(cm->world >= 1 && cm->rank >= 0 && cm->rank < cm->world, MFX_ERR_INVALID, "bad rank %d of %d", cm->rank, cm->world);
  MFX_REQUIRE(cm->nloc >= 64 && cm->nloc % 64 == 0, MFX_ERR_INVALID, "rows per rank (%lld) must be a positive multiple of 64", (long long)cm->nloc);
  MFX_REQUIRE((int64_t)cm->world * cm->nloc >= n && (int64_t)(cm->world - 1) * cm->nloc < n, MFX_ERR_INVALID,
              "%d ranks x %lld rows do not tile n = %lld with a non-empty last shard", cm->world, (long long)cm->nloc, (long long)n);
  return MFX_OK;
}

#define MFX_SHARDED_PROLOGUE()                                                                  \
  MFX_TRY(check_sharded(op, comm, n, k, p));                                                    \
  const int64_t nrows = shard_nrows(comm, n);                                                   \
  KrylovWs kws;                                                                                 \
  const int64_t need = carve_ws(op, nrows, k, p, ws, ws_bytes, &kws, comm);                      \
  MFX_REQUIRE(ws && need <= ws_bytes, MFX_ERR_WORKSPACE, "workspace too small: need %lld bytes", \
              (long long)need);                                                                 \
  hipStream_t s = static_cast<hipStream_t>(stream);                                               \
  PrepScope prep_scope

int64_t mfx_sharded_workspace_bytes(const mfx_operator* op, const mfx_comm* comm, int64_t n, int64_t k, int64_t p) {
  if (!op || !comm || comm->nloc < 1) return -1;
  return carve_ws(op, comm->nloc, k, p, nullptr, 0, nullptr, comm);
}

int mfx_arnoldi_forward_sharded(const mfx_operator* op, const mfx_comm* comm, const void* v0, int64_t n, int64_t k,
                                int64_t p, int second_pass, void* Q, void* Qfull, void* H, void* r, void* c, void* ws,
                                int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(v0 && Q && Qfull && H && r && c, MFX_ERR_INVALID, "null argument");
  MFX_SHARDED_PROLOGUE();
  if (op->dtype == MFX_F32)
    return arnoldi_forward_t<float>(op, (const float*)v0, nrows, k, p, second_pass, (float*)Q, (float*)H, (float*)r, (float*)c,
                                    kws, s, comm, (float*)Qfull);
  return arnoldi_forward_t<double>(op, (const double*)v0, nrows, k, p, second_pass, (double*)Q, (double*)H, (double*)r,
                                   (double*)c, kws, s, comm, (double*)Qfull);
}

int mfx_arnoldi_adjoint_sharded(const mfx_operator* op, const mfx_comm* comm, int64_t n, int64_t k, int64_t p,
                                const void* Q, const void* Qfull, const void* H, const void* r, const void* c,
                                const void* dQ, const void* dH, const void* dr, const void* dc, int reortho, void* dv,
                                void* Lambda, const mfx_op_grads* grads, void* ws, int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(Q && Qfull && H && r && c && dH && dv && Lambda, MFX_ERR_INVALID, "null argument");
  MFX_REQUIRE(reortho == MFX_REORTHO_NONE || reortho == MFX_REORTHO_FULL, MFX_ERR_INVALID, "bad reortho flag %d", reortho);
  MFX_SHARDED_PROLOGUE();
  if (op->dtype == MFX_F32)
    return arnoldi_adjoint_t<float>(op, nrows, k, p, (const float*)Q, (const float*)H, (const float*)r, (const float*)c,
                                    (const float*)dQ, (const float*)dH, (const float*)dr, (const float*)dc, reortho,
                                    (float*)dv, (float*)Lambda, grads, kws, s, comm, (const float*)Qfull);
  return arnoldi_adjoint_t<double>(op, nrows, k, p, (const double*)Q, (const double*)H, (const double*)r, (const double*)c,
                                   (const double*)dQ, (const double*)dH, (const double*)dr, (const double*)dc, reortho,
                                   (double*)dv, (double*)Lambda, grads, kws, s, comm, (const double*)Qfull);
}

int mfx_lanczos_forward_sharded(const mfx_operator* op, const mfx_comm* comm, const void* v0, int64_t n, int64_t k, int64_t p,
                                void* xs, void* alpha, void* beta, void* vnorm, void* ws, int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(v0 && xs && alpha && beta && vnorm, MFX_ERR_INVALID, "null argument");
  MFX_SHARDED_PROLOGUE();
  if (op->dtype == MFX_F32)
    return lanczos_forward_t<float>(op, (const float*)v0, nrows, k, p, (float*)xs, (float*)alpha, (float*)beta, (float*)vnorm, kws, s, comm);
  return lanczos_forward_t<double>(op, (const double*)v0, nrows, k, p, (double*)xs, (double*)alpha, (double*)beta, (double*)vnorm, kws, s, comm);
}

int mfx_lanczos_adjoint_sharded(const mfx_operator* op, const mfx_comm* comm, int64_t n, int64_t k, int64_t p, const void* xs,
                                const void* alpha, const void* beta, const void* vnorm, const void* dxs, const void* dalpha,
                                const void* dbeta, void* dv, void* Lambda, void* Lambdafull, const mfx_op_grads* grads, void* ws,
                                int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(xs && alpha && beta && vnorm && dalpha && dbeta && dv && Lambda && Lambdafull, MFX_ERR_INVALID, "null argument");
  MFX_SHARDED_PROLOGUE();
  if (op->dtype == MFX_F32)
    return lanczos_adjoint_t<float>(op, nrows, k, p, (const float*)xs, (const float*)alpha, (const float*)beta, (const float*)vnorm,
                                    (const float*)dxs, (const float*)dalpha, (const float*)dbeta, (float*)dv, (float*)Lambda, grads,
                                    kws, s, comm, (float*)Lambdafull);
  return lanczos_adjoint_t<double>(op, nrows, k, p, (const double*)xs, (const double*)alpha, (const double*)beta, (const double*)vnorm,
                                   (const double*)dxs, (const double*)dalpha, (const double*)dbeta, (double*)dv, (double*)Lambda, grads,
                                   kws, s, comm, (double*)Lambdafull);
}

int mfx_lanczos_forward(const mfx_operator* op, const void* v0, int64_t n, int64_t k, int64_t p, void* xs,
                        void* alpha, void* beta, void* vnorm, void* ws, int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(v0 && xs && alpha && beta && vnorm, MFX_ERR_INVALID, "null argument");
  MFX_DRIVER_PROLOGUE();
  GraphKey key = driver_key(3, op, n, k, p, nullptr, ws, ws_bytes, stream);
  key.add(v0).add(xs).add(alpha).add(beta).add(vnorm);
  return run_graphed(graph_eligible(op, n, p), key, s, [&](hipStream_t st) {
    if (op->dtype == MFX_F32)
      return lanczos_forward_t<float>(op, (const float*)v0, n, k, p, (float*)xs, (float*)alpha, (float*)beta, (float*)vnorm, kws, st);
    return lanczos_forward_t<double>(op, (const double*)v0, n, k, p, (double*)xs, (double*)alpha, (double*)beta, (double*)vnorm, kws, st);
  });
}

int mfx_lanczos_adjoint(const mfx_operator* op, int64_t n, int64_t k, int64_t p, const void* xs, const void* alpha,
                        const void* beta, const void* vnorm, const void* dxs, const void* dalpha, const void* dbeta,
                        void* dv, void* Lambda, const mfx_op_grads* grads, void* ws, int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(xs && alpha && beta && vnorm && dalpha && dbeta && dv && Lambda, MFX_ERR_INVALID, "null argument");
  MFX_DRIVER_PROLOGUE();
  GraphKey key = driver_key(4, op, n, k, p, grads, ws, ws_bytes, stream);
  key.add(xs).add(alpha).add(beta).add(vnorm).add(dxs).add(dalpha).add(dbeta).add(dv).add(Lambda);
  return run_graphed(graph_eligible(op, n, p), key, s, [&](hipStream_t st) {
    if (op->dtype == MFX_F32)
      return lanczos_adjoint_t<float>(op, n, k, p, (const float*)xs, (const float*)alpha, (const float*)beta,
                                      (const float*)vnorm, (const float*)dxs, (const float*)dalpha, (const float*)dbeta,
                                      (float*)dv, (float*)Lambda, grads, kws, st);
    return lanczos_adjoint_t<double>(op, n, k, p, (const double*)xs, (const double*)alpha, (const double*)beta,
                                     (const double*)vnorm, (const double*)dxs, (const double*)dalpha, (const double*)dbeta,
                                     (double*)dv, (double*)Lambda, grads, kws, st);
  });
}

}  // extern "C"
