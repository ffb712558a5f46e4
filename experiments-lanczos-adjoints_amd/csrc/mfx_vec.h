// libmfx: the shared HBM-bound vector kernels (dots / update / sumsq / scale) and their launch helpers.
// Included by the Krylov drivers (mfx_krylov.hip) and the CG / preconditioner drivers (mfx_cg.hip): every
// kernel uses grid = (slices, p) and per-slice partials, see the header comment of mfx_krylov.hip.
#pragma once
#include <stdlib.h>

#include <initializer_list>

#include "mfx_internal.h"

namespace mfx {

// ------------------------------------------------------------------------------------------------
// thread-owned elements
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int EPT>
__device__ __forceinline__ void load_own(T (&dst)[EPT], const T* __restrict__ base, int64_t slice0,
                                         int64_t n, int tid) {
  constexpr int U = EPT / VEC;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t off = slice0 + (int64_t)(u * (int)blockDim.x + tid) * VEC;
    if (off < n) {
      Pack<T, VEC> p = load_pack<T, VEC>(base + off);
#pragma unroll
      for (int e = 0; e < VEC; ++e) dst[u * VEC + e] = p.v[e];
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) dst[u * VEC + e] = T(0);
    }
  }
}

template <typename T, int VEC, int EPT>
__device__ __forceinline__ void store_own(const T (&src)[EPT], T* __restrict__ base, int64_t slice0,
                                          int64_t n, int tid) {
  constexpr int U = EPT / VEC;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t off = slice0 + (int64_t)(u * (int)blockDim.x + tid) * VEC;
    if (off < n) {
      Pack<T, VEC> p;
#pragma unroll
      for (int e = 0; e < VEC; ++e) p.v[e] = src[u * VEC + e];
      store_pack<T, VEC>(base + off, p);
    }
  }
}

template <typename T>
__device__ __forceinline__ T reduce_partials(const T* __restrict__ part, int nblk) {
  double acc = 0.0;  // few hundred terms at most; fp64 keeps the reduction order-insensitive
  for (int q = 0; q < nblk; ++q) acc += (double)part[q];
  return (T)acc;
}

// The same sum by an aligned group of G consecutive lanes (G a power of two <= 64, every lane of the group calls, `g` = lane
// within the group); result in every lane of the group.  The consumers' prologues run in EVERY slice workgroup, so with one
// lane per coefficient they cost nblk serial reads each -- 977 at n = 2e6 (BASELINE config 5), where they dominated the
// vector kernels.  Fixed lane-strided order + xor tree: deterministic.
// sum over an aligned group of G = 8 or 64 lanes, result in every lane of the group (DPP, see wave_sum_all)
template <int G>
__device__ __forceinline__ double group_sum_all(double v) {
  static_assert(G == 8 || G == 64, "group size");
  if constexpr (G == 64) {
    return wave_sum_all(v);
  } else {
    v += dpp_mov<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xF>(v);  // row_half_mirror
    return v;
  }
}
template <typename T, int G>
__device__ __forceinline__ T reduce_partials_group(const T* __restrict__ part, int nblk, int g) {
  double acc = 0.0;
  constexpr int U = 8;  // loads in flight per lane: the prologues are a chain of L2 round trips otherwise
  for (int q0 = g; q0 < nblk; q0 += U * G) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (q0 + u * G < nblk) ? part[q0 + u * G] : T(0);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += (double)v[u];  // same order as a plain lane-strided loop
  }
  return (T)group_sum_all<G>(acc);
}
constexpr int kRedG = 8;  // lanes per coefficient in the multi-coefficient prologues

// Accumulator type of the Gram-Schmidt dot products and norms inside a slice (products, per-thread sums, cross-lane tree): fp64
// also for fp32 vectors.  The kernels are HBM-bound (C4: +1.3 ms of 48 ms per step), and at the C4 size the worst gradient
// component against fp64 over four probe sets drops from 3.8e-5 ... 1.25e-4 to 1.2e-5 ... 6.0e-5 (profiles/r02a_accuracy/
// table_other_probe_sets.log); what remains is the rounding of the fp32-stored vectors themselves.
template <typename T>
using DotAcc = double;

// Rows j0 <= j < j1 of a (rows, n) panel, this thread's elements of the slice, in order, JT rows at a time:
// f(j, rows[JT][EPT], nvalid) --
// double-buffered: the loads of the next JT rows are issued before the current JT are consumed, so a sweep is
// ceil(rows / JT) overlapped round trips instead of that many serial ones (what bounds these kernels when there
// are few workgroups: one vector, n ~ 1e5).  Every load is issued on every path -- rows past the end re-read the
// last row, lanes past n re-read element 0 and are zeroed by a select -- because a load under a branch or an exec
// mask makes the compiler fall back to s_waitcnt vmcnt(0), which serialises the two buffers again.
template <typename T, int VEC, int EPT, int JT, typename F>
__device__ __forceinline__ void sweep_rows(const T* __restrict__ rb, int64_t row_stride, int j0, int j1,
                                           int64_t slice0, int64_t n, int tid, F&& f) {
  if (j0 >= j1) return;
  constexpr int U = EPT / VEC;
  int64_t off[U];
  bool ok[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t o = slice0 + (int64_t)(u * (int)blockDim.x + tid) * VEC;
    ok[u] = o < n;
    off[u] = ok[u] ? o : 0;
  }
  auto load = [&](T (&dst)[JT][EPT], int j) {
#pragma unroll
    for (int q = 0; q < JT; ++q) {
      const int jj = (j + q < j1) ? j + q : j1 - 1;
      const T* row = rb + (int64_t)jj * row_stride;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        Pack<T, VEC> pk = load_pack<T, VEC>(row + off[u]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) dst[q][u * VEC + e] = pk.v[e];
      }
    }
  };
  auto use = [&](T (&src)[JT][EPT], int j) {
#pragma unroll
    for (int q = 0; q < JT; ++q)
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e) src[q][u * VEC + e] = ok[u] ? src[q][u * VEC + e] : T(0);
    f(j, src, (j1 - j < JT) ? j1 - j : JT);  // rows j .. j + nvalid - 1 (the others repeat the last row)
  };
  T ra[JT][EPT];
  if (j1 - j0 <= JT) {
    load(ra, j0);
    use(ra, j0);
    return;
  }
  T rc[JT][EPT];
  int j = j0;
  load(ra, j);
  for (;;) {
    load(rc, j + JT);
    use(ra, j);
    j += JT;
    if (j >= j1) break;
    load(ra, j + JT);
    use(rc, j);
    j += JT;
    if (j >= j1) break;
  }
}
template <int EPT>
struct RowsInFlight {
  static constexpr int value = EPT <= 4 ? 16 : 4;  // per buffer; two buffers in flight
};

// ------------------------------------------------------------------------------------------------
// K-dots: partial[b][j][blk] = sum_{i in slice} rows[b][j][i] * x[b][i],  j < m
//   forward  h = Q^T w            (arnoldi.py:87)      adjoint  P lam, z^T Q   (arnoldi.py:204,212)
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int EPT, int CT>
__global__ __launch_bounds__(kBlock) void k_dots(const T* __restrict__ rows, int64_t rows_ldb,
                                                 int64_t row_stride, int m,
                                                 const T* __restrict__ x, int64_t ldx, int64_t n,
                                                 T* __restrict__ partial, int kmax, int nblk,
                                                 int jchunk, int ngroups, int ncols, int64_t x_zstride,
                                                 int64_t part_zstride) {
  // blockIdx.z = ctile * ngroups + group: `group` owns rows [group * jchunk, +jchunk) (more workgroups, shorter sweeps when
  // there are few slices); `ctile` selects CT of the ncols right-hand vectors x + col * x_zstride of the same probe (the
  // dQ^T Q projection of the adjoint: every row that is loaded meets CT vectors), partials at partial + col * part_zstride.
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);  // [4][CT][jchunk]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y, blk = blockIdx.x;
  const int col0 = ((int)blockIdx.z / ngroups) * CT, grp = (int)blockIdx.z % ngroups;
  const int j0 = grp * jchunk, j1 = (j0 + jchunk < m) ? j0 + jchunk : m;
  const int64_t slice0 = (int64_t)blk * ((int64_t)blockDim.x * EPT);
  T xr[CT][EPT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = col0 + ct < ncols ? col0 + ct : ncols - 1;  // a ragged last tile repeats its last column (not stored)
    load_own<T, VEC>(xr[ct], x + (int64_t)b * ldx + (int64_t)col * x_zstride, slice0, n, tid);
  }
  constexpr int JT = RowsInFlight<EPT>::value;
  sweep_rows<T, VEC, EPT, JT>(
      rows + (int64_t)b * rows_ldb, row_stride, j0, j1, slice0, n, tid, [&](int j, const T (&row)[JT][EPT], int nvalid) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          DotAcc<T> acc[JT];  // all rows of the buffer in one block of straight-line code: the JT reductions interleave
#pragma unroll
          for (int q = 0; q < JT; ++q) {
            acc[q] = DotAcc<T>(0);
#pragma unroll
            for (int e = 0; e < EPT; ++e) acc[q] += (DotAcc<T>)row[q][e] * (DotAcc<T>)xr[ct][e];
          }
          const T wsum = (T)wave_sums<JT>(acc, lane);
          const int q = row16_index<JT>(lane);
          if (wave_sums_writer<JT>(lane) && q < nvalid) sm[(wid * CT + ct) * jchunk + (j - j0) + q] = wsum;
        }
      });
  __syncthreads();
  const int nj = j1 - j0;
  for (int i = tid; i < CT * nj; i += (int)blockDim.x) {
    const int ct = i / nj, j = j0 + i % nj;
    if (col0 + ct >= ncols) break;
    T sum = T(0);
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += sm[(w * CT + ct) * jchunk + (j - j0)];
    partial[(int64_t)(col0 + ct) * part_zstride + ((int64_t)b * kmax + j) * nblk + blk] = sum;
  }
}

// ------------------------------------------------------------------------------------------------
// K-update: coef_j = s1 * sum_blk partial_in[b][j][:] + s2 * extra[b][j];   y = x - sum_j coef_j row_j
//   optional: hout[b][j] = coef_j (H column, arnoldi.py:99), second store y2, fused second dots
//   (re-orthogonalisation pass, arnoldi.py:91-92) and fused |y|^2 partial (arnoldi.py:95).
// ------------------------------------------------------------------------------------------------
template <typename T>
struct UpdateArgs {
  const T* rows;
  int64_t rows_ldb, row_stride;
  int m;
  const T* partial_in;  // (p, kmax, nblk) or null
  T s1;
  const T* extra;  // null or extra[b*extra_ldb + j*extra_stride]
  int64_t extra_ldb, extra_stride;
  T s2;
  T* hout;  // null or hout[b*hout_ldb + j*hout_stride], j >= hout_from
  int64_t hout_ldb, hout_stride;
  int hout_from;
  const T* x;  // null = zeros
  int64_t ldx;
  T* y;
  int64_t ldy;
  T* y2;  // optional second destination
  int64_t ldy2;
  int64_t n;
  T* partial_out;   // DOTS
  T* partial_norm;  // NORM: (p, nblk)
  int kmax, nblk;
  int nblk_in;  // slices of partial_in (1 when the partials were summed over the row shards, see Ctx::comm)
};

template <typename T, int VEC, bool DOTS, bool NORM, int EPT>
__global__ __launch_bounds__(kBlock) void k_update(UpdateArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* coef = reinterpret_cast<T*>(smem_raw);  // [m]
  T* sm = coef + a.m;                         // [4][m] (DOTS) ; [4] (NORM) after that
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y, blk = blockIdx.x;
  const int64_t slice0 = (int64_t)blk * ((int64_t)blockDim.x * EPT);
  const int m = a.m;
  for (int idx = tid; idx < m * kRedG; idx += (int)blockDim.x) {  // kRedG lanes per coefficient
    const int j = idx / kRedG, g = idx % kRedG;
    T c = T(0);
    if (a.partial_in)
      c = a.s1 * reduce_partials_group<T, kRedG>(a.partial_in + ((int64_t)b * a.kmax + j) * a.nblk_in, a.nblk_in, g);
    if (g == 0) {
      if (a.extra) c += a.s2 * a.extra[(int64_t)b * a.extra_ldb + (int64_t)j * a.extra_stride];
      coef[j] = c;
      if (a.hout && blk == 0 && j >= a.hout_from) a.hout[(int64_t)b * a.hout_ldb + (int64_t)j * a.hout_stride] = c;
    }
  }
  __syncthreads();
  T xr[EPT];
  if (a.x) {
    load_own<T, VEC>(xr, a.x + (int64_t)b * a.ldx, slice0, a.n, tid);
  } else {
#pragma unroll
    for (int e = 0; e < EPT; ++e) xr[e] = T(0);
  }
  const T* rb = a.rows + (int64_t)b * a.rows_ldb;
  constexpr int JT = RowsInFlight<EPT>::value;
  sweep_rows<T, VEC, EPT, JT>(rb, a.row_stride, 0, m, slice0, a.n, tid, [&](int j, const T (&row)[JT][EPT], int nvalid) {
#pragma unroll
    for (int q = 0; q < JT; ++q) {
      const T c = q < nvalid ? coef[j + q] : T(0);
#pragma unroll
      for (int e = 0; e < EPT; ++e) xr[e] -= c * row[q][e];
    }
  });
  store_own<T, VEC>(xr, a.y + (int64_t)b * a.ldy, slice0, a.n, tid);
  if (a.y2) store_own<T, VEC>(xr, a.y2 + (int64_t)b * a.ldy2, slice0, a.n, tid);
  if constexpr (DOTS) {
    sweep_rows<T, VEC, EPT, JT>(rb, a.row_stride, 0, m, slice0, a.n, tid, [&](int j, const T (&row)[JT][EPT], int nvalid) {
      DotAcc<T> acc[JT];
#pragma unroll
      for (int q = 0; q < JT; ++q) {
        acc[q] = DotAcc<T>(0);
#pragma unroll
        for (int e = 0; e < EPT; ++e) acc[q] += (DotAcc<T>)row[q][e] * (DotAcc<T>)xr[e];
      }
      const T wsum = (T)wave_sums<JT>(acc, lane);
      const int q = row16_index<JT>(lane);
      if (wave_sums_writer<JT>(lane) && q < nvalid) sm[wid * m + j + q] = wsum;
    });
    __syncthreads();
    for (int j = tid; j < m; j += (int)blockDim.x)
    {
      T sum = T(0);
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += sm[w * m + j];
      a.partial_out[((int64_t)b * a.kmax + j) * a.nblk + blk] = sum;
    }
  }
  if constexpr (NORM) {
    T* smn = sm + (DOTS ? 4 * m : 0);
    DotAcc<T> acc = DotAcc<T>(0);
#pragma unroll
    for (int e = 0; e < EPT; ++e) acc += (DotAcc<T>)xr[e] * (DotAcc<T>)xr[e];
    acc = wave_sum(acc);
    if (lane == 0) smn[wid] = (T)acc;
    __syncthreads();
    if (tid == 0) {
      T sum = T(0);
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += smn[w];
      a.partial_norm[(int64_t)b * a.nblk + blk] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K-sumsq: partial_norm[b][blk] = sum x^2          (arnoldi.py:67, lanczos.py:222)
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int EPT>
__global__ __launch_bounds__(kBlock) void k_sumsq(const T* __restrict__ x, int64_t ldx, int64_t n,
                                                  T* __restrict__ partial_norm, int nblk) {
  __shared__ T smn[4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.y, blk = blockIdx.x;
  T xr[EPT];
  load_own<T, VEC>(xr, x + (int64_t)b * ldx, (int64_t)blk * ((int64_t)blockDim.x * EPT), n, tid);
  DotAcc<T> acc = DotAcc<T>(0);
#pragma unroll
  for (int e = 0; e < EPT; ++e) acc += (DotAcc<T>)xr[e] * (DotAcc<T>)xr[e];
  acc = wave_sum(acc);
  if (lane == 0) smn[wid] = (T)acc;
  __syncthreads();
  if (tid == 0) {
    T sum = T(0);
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += smn[w];
    partial_norm[(int64_t)b * nblk + blk] = sum;
  }
}

// ------------------------------------------------------------------------------------------------
// K-scale: len = sqrt(sum_blk partial_norm[b][:]) (or given scale[b]);  y = x * f,
//   mode 0: f = 1/len   (normalise: arnoldi.py:80, lanczos.py:258)      mode 1: f = scale[b]      mode 2: f = -1
//   optional scalar outputs: len_out[b*ld] = len, inv_out[b] = 1/len.  y may be null (scalars only).
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int EPT>
__global__ __launch_bounds__(kBlock) void k_scale(const T* __restrict__ x, int64_t ldx,
                                                  T* __restrict__ y, int64_t ldy, int64_t n,
                                                  const T* __restrict__ partial_norm, int nblk,
                                                  const T* __restrict__ scale, int mode,
                                                  T* __restrict__ len_out, int64_t len_ld,
                                                  T* __restrict__ inv_out) {
  __shared__ T f_sh;
  const int tid = threadIdx.x;
  const int b = blockIdx.y, blk = blockIdx.x;
  if (tid < 64) {  // wave 0
    T f;
    if (mode == 0) {
      const T len = sqrt(reduce_partials_group<T, 64>(partial_norm + (int64_t)b * nblk, nblk, tid));
      f = T(1) / len;
      if (blk == 0 && tid == 0) {
        if (len_out) len_out[(int64_t)b * len_ld] = len;
        if (inv_out) inv_out[b] = f;
      }
    } else if (mode == 1) {
      f = scale[b];
    } else {
      f = T(-1);
    }
    if (tid == 0) f_sh = f;
  }
  __syncthreads();
  if (!y) return;
  const T f = f_sh;
  T xr[EPT];
  load_own<T, VEC>(xr, x + (int64_t)b * ldx, (int64_t)blk * ((int64_t)blockDim.x * EPT), n, tid);
#pragma unroll
  for (int e = 0; e < EPT; ++e) xr[e] *= f;
  store_own<T, VEC>(xr, y + (int64_t)b * ldy, (int64_t)blk * ((int64_t)blockDim.x * EPT), n, tid);
}

// ------------------------------------------------------------------------------------------------
// host-side launch helpers
// ------------------------------------------------------------------------------------------------
template <typename T>
static int pick_vec(int64_t n, std::initializer_list<const void*> ptrs) {
  constexpr int V = VecWidth<T>::value;
  if (n % V != 0) return 1;
  for (const void* p : ptrs)
    if (p && (reinterpret_cast<uintptr_t>(p) % 16) != 0) return 1;
  return V;
}

#define MFX_VEC_SWITCH(vec, ...)                \
  if ((vec) > 1) {                              \
    constexpr int VEC = VecWidth<T>::value;     \
    __VA_ARGS__;                                \
  } else {                                      \
    constexpr int VEC = 1;                      \
    __VA_ARGS__;                                \
  }

// VEC and the elements per thread EPT: kEpt (2048-element slices), or -- Ctx::fine -- one 16-byte load per thread and row
// (512 / 1024-element slices) when the coarse slicing leaves most CUs idle
#define MFX_VEC_EPT_SWITCH(c, ...)              \
  if ((c).vec > 1 && (c).ept != kEpt) {         \
    constexpr int VEC = VecWidth<T>::value;     \
    constexpr int EPT = VEC;                    \
    __VA_ARGS__;                                \
  } else if ((c).vec > 1) {                     \
    constexpr int VEC = VecWidth<T>::value;     \
    constexpr int EPT = kEpt;                   \
    __VA_ARGS__;                                \
  } else {                                      \
    constexpr int VEC = 1;                      \
    constexpr int EPT = kEpt;                   \
    __VA_ARGS__;                                \
  }

// workgroup size of the vector kernels: 256 threads (2048-element slices); one wave (512-element slices) only for
// tiny problems.  (Measured on config 3, n = 102400, p = 1: 50 workgroups x 4 waves beat 200 x 1 wave -- what
// counts is loads in flight per CU, not the number of CUs touched.)
static int pick_wg(int64_t n, int64_t p) { return ((n + kSlice - 1) / kSlice) * p >= 16 ? kBlock : 64; }

// ------------------------------------------------------------------------------------------------
// Row-sharded mode (one process per GPU, rows of every vector split over the ranks of an mfx_comm): a producer's
// per-slice partials are summed on the device into ONE number per coefficient, all-reduced over the ranks, and the
// consumers read them with a single "slice" (nblk_in = 1).  Same kernels, same partial layout (b, j, slice).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_compact_partials(const T* __restrict__ in, int kmax, int nblk, int m,
                                                          T* __restrict__ out) {
  const int b = blockIdx.x;
  for (int idx = threadIdx.x; idx < m * kRedG; idx += (int)blockDim.x) {
    const int j = idx / kRedG, g = idx % kRedG;
    const T v = reduce_partials_group<T, kRedG>(in + ((int64_t)b * kmax + j) * nblk, nblk, g);
    if (g == 0) out[(int64_t)b * kmax + j] = v;
  }
}

template <typename T>
struct Ctx {
  int64_t n, k, p;
  int wg, nblk, kmax, vec;
  int ept = kEpt;  // elements per thread of the kernels launched through MFX_VEC_EPT_SWITCH
  hipStream_t stream;
  const mfx_comm* comm = nullptr;  // row-sharded mode
  T* stage = nullptr;              // (p, kmax, nblk) producer scratch of the sharded mode
  int nblk_in;                     // slices the consumers sum over
  Ctx(int64_t n_, int64_t k_, int64_t p_, int vec_, hipStream_t s)
      : n(n_), k(k_), p(p_), wg(pick_wg(n_, p_)), nblk((int)((n_ + (int64_t)wg * kEpt - 1) / ((int64_t)wg * kEpt))),
        kmax((int)(k_ + 1)), vec(vec_), stream(s), nblk_in(nblk) {}
  // Finer slices (one 16-byte load per thread and row) when the 2048-element slicing gives fewer than 128 workgroups: the
  // vector kernels are bandwidth-bound PER CU (measured on config 3, n = 102400, one vector: 50 workgroups stream 1.4 TB/s).
  // Only drivers whose every kernel goes through MFX_VEC_EPT_SWITCH may call this (the partial layout changes with nblk).
  // First measured WITHOUT the other changes of DESIGN.md section 3.1 (profiles/r02g_*): no net gain -- every consumer re-reduced 4x more
  // partials in a serial prologue.  With the pipelined partial sums, the double-buffered sweeps and the DPP reductions it pays
  // (config 3: forward + adjoint 8.2 -> 3.6 ms altogether, profiles/r02k_*).
  void fine() {
    if (vec <= 1 || wg != kBlock || (int64_t)nblk * p >= 128) return;
    ept = VecWidth<T>::value;
    nblk = (int)((n + (int64_t)wg * ept - 1) / ((int64_t)wg * ept));
    if (!comm) nblk_in = nblk;
  }
  void shard(const mfx_comm* cm, T* stage_) {
    comm = cm;
    stage = stage_;
    nblk_in = 1;
  }
  dim3 grid() const { return dim3(nblk, (unsigned)p); }
  // where a producer writes its (b, j, slice) partials that the caller wants in `dst`
  T* producer(T* dst) const { return comm ? stage : dst; }
  // sharded: dst[b][j] (j < m, row stride kmax_) = sum over the ranks and slices of the partials just produced in `stage`
  int finish(T* dst, int kmax_, int m) const { return finish_from(stage, dst, kmax_, m); }
  // the same for a producer that wrote its per-slice partials into a staging buffer of its own (a kernel with two outputs)
  int finish_from(const T* stage_, T* dst, int kmax_, int m) const {
    if (!comm) return MFX_OK;
    k_compact_partials<T><<<(unsigned)p, 256, 0, stream>>>(stage_, kmax_, nblk, m, dst);
    MFX_CHECK_LAUNCH();
    // the span just written: rows b < p of stride kmax_, m entries each (dst may point INTO a coefficient row)
    const int rc = comm->allreduce_sum(comm->ctx, dst, (p - 1) * (int64_t)kmax_ + m, sizeof(T) == 4 ? MFX_F32 : MFX_F64, stream);
    MFX_REQUIRE(rc == 0, MFX_ERR_CALLBACK, "all-reduce callback failed with code %d", rc);
    return MFX_OK;
  }
};

// ncols > 1: the same rows against x + col * x_zstride, col < ncols, partials at partial + col * part_zstride (not in
// the row-sharded mode: one staging buffer)
constexpr int kDotsColTile = 4;
template <typename T>
static int launch_dots(const Ctx<T>& c, const T* rows, int64_t rows_ldb, int64_t row_stride, int m,
                       const T* x, int64_t ldx, T* partial, int ncols = 1, int64_t x_zstride = 0,
                       int64_t part_zstride = 0) {
  if (m <= 0) return MFX_OK;
  MFX_REQUIRE(ncols == 1 || !c.comm, MFX_ERR_UNSUPPORTED, "batched dots in the row-sharded mode");
  const bool tiled = ncols > 1 && c.vec > 1 && c.ept == VecWidth<T>::value;  // CT vectors of EPT elements in registers beside the row buffers
  const int ctiles = tiled ? (ncols + kDotsColTile - 1) / kDotsColTile : ncols;
  // few slices (one vector, n ~ 1e5): split the rows over workgroups as well
  const int64_t wgs = (int64_t)c.nblk * c.p * ctiles;
  const int64_t maxgroups = wgs < 256 ? 1024 / wgs : 1;
  int jchunk = (int)((m + maxgroups - 1) / maxgroups);
  const int jmin = c.ept <= 4 ? 16 : 8;  // one buffer of sweep_rows at least
  if (jchunk < jmin) jchunk = m < jmin ? m : jmin;
  if (jchunk > 256) jchunk = 256;  // the per-wave partials in LDS: 4 x CT x jchunk values (deep bases just get more row groups)
  const int ngroups = (m + jchunk - 1) / jchunk;
  const size_t sh = (size_t)4 * (tiled ? kDotsColTile : 1) * jchunk * sizeof(T);
  dim3 grid(c.nblk, (unsigned)c.p, (unsigned)(ctiles * ngroups));
  if (tiled) {
    constexpr int VEC = VecWidth<T>::value;  // Ctx::fine: ept == VEC, 16-byte loads
    k_dots<T, VEC, VEC, kDotsColTile><<<grid, c.wg, sh, c.stream>>>(rows, rows_ldb, row_stride, m, x, ldx, c.n, c.producer(partial),
                                                                   c.kmax, c.nblk, jchunk, ngroups, ncols, x_zstride, part_zstride);
  } else {
    MFX_VEC_EPT_SWITCH(c, (k_dots<T, VEC, EPT, 1><<<grid, c.wg, sh, c.stream>>>(
                              rows, rows_ldb, row_stride, m, x, ldx, c.n, c.producer(partial), c.kmax, c.nblk, jchunk,
                              ngroups, ncols, x_zstride, part_zstride)));
  }
  MFX_CHECK_LAUNCH();
  return c.finish(partial, c.kmax, m);
}

template <typename T>
static int launch_update(const Ctx<T>& c, UpdateArgs<T> a, bool dots, bool norm) {
  a.n = c.n;
  a.kmax = c.kmax;
  a.nblk = c.nblk;
  a.nblk_in = c.nblk_in;
  T* const want_dots = a.partial_out;
  T* const want_norm = a.partial_norm;
  MFX_REQUIRE(!(c.comm && dots && norm), MFX_ERR_UNSUPPORTED, "sharded update: fused dots + norm share one staging buffer");
  if (dots) a.partial_out = c.producer(want_dots);
  if (norm) a.partial_norm = c.producer(want_norm);
  const size_t sh = (size_t)(a.m + (dots ? 4 * a.m : 0) + 4) * sizeof(T);
  MFX_REQUIRE(sh <= 64 * 1024, MFX_ERR_UNSUPPORTED, "Krylov depth %d too large for the update kernels' coefficient buffers (%zu B of LDS > 64 KiB)",
              a.m, sh);
  if (dots && norm) {
    MFX_VEC_EPT_SWITCH(c, (k_update<T, VEC, true, true, EPT><<<c.grid(), c.wg, sh, c.stream>>>(a)));
  } else if (dots) {
    MFX_VEC_EPT_SWITCH(c, (k_update<T, VEC, true, false, EPT><<<c.grid(), c.wg, sh, c.stream>>>(a)));
  } else if (norm) {
    MFX_VEC_EPT_SWITCH(c, (k_update<T, VEC, false, true, EPT><<<c.grid(), c.wg, sh, c.stream>>>(a)));
  } else {
    MFX_VEC_EPT_SWITCH(c, (k_update<T, VEC, false, false, EPT><<<c.grid(), c.wg, sh, c.stream>>>(a)));
  }
  MFX_CHECK_LAUNCH();
  if (dots) MFX_TRY(c.finish(want_dots, c.kmax, a.m));
  if (norm) MFX_TRY(c.finish(want_norm, 1, 1));
  return MFX_OK;
}

template <typename T>
static int launch_sumsq(const Ctx<T>& c, const T* x, int64_t ldx, T* partial_norm) {
  MFX_VEC_EPT_SWITCH(c, (k_sumsq<T, VEC, EPT><<<c.grid(), c.wg, 0, c.stream>>>(x, ldx, c.n, c.producer(partial_norm), c.nblk)));
  MFX_CHECK_LAUNCH();
  return c.finish(partial_norm, 1, 1);
}

template <typename T>
static int launch_scale(const Ctx<T>& c, const T* x, int64_t ldx, T* y, int64_t ldy, const T* partial_norm,
                        const T* scale, int mode, T* len_out, int64_t len_ld, T* inv_out) {
  dim3 grid = y ? c.grid() : dim3(1, (unsigned)c.p);
  MFX_VEC_EPT_SWITCH(c, (k_scale<T, VEC, EPT><<<grid, c.wg, 0, c.stream>>>(x, ldx, y, ldy, c.n, partial_norm, c.nblk_in,
                                                                        scale, mode, len_out, len_ld, inv_out)));
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

static int apply_any(const mfx_operator* op, int mode, const void* x, int64_t ldx, const void* aux,
                     int64_t ldaux, void* y, int64_t ldy, int64_t p, void* ws, int64_t ws_bytes,
                     hipStream_t stream) {
  if (op->kind == MFX_OP_CALLBACK) return op_apply_cb(op, mode, x, ldx, aux, ldaux, y, ldy, p, stream);
  ScopedTimer t(0, stream);
  return op_apply(op, x, ldx, y, ldy, p, mode == 1 ? 1 : 0, ws, ws_bytes, stream);
}

// workspace of the Krylov drivers (carved by mfx_krylov.hip; the CG driver fills the fields apply_sharded reads)
struct KrylovWs {
  void *w, *p1, *p2, *pn, *small, *opws;
  void* pb;  // partials of a batch of dQ columns (adjoint, few-slice problems), null when the columns go one by one
  int64_t opws_bytes;
  // row-sharded drivers only
  void *stage, *send, *gathered, *xfull;
};

// ------------------------------------------------------------------------------------------------
// row-sharded operator application: all-gather the iterate, apply the rows this rank owns
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_pack_shard(const T* __restrict__ x, int64_t ldx, int64_t nrows, int64_t nloc,
                                                    T* __restrict__ send) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i < nloc) send[b * nloc + i] = i < nrows ? x[b * ldx + i] : T(0);
}

// gathered (world, p, nloc) -> full[b][g nloc + i], the first n columns
template <typename T>
__global__ __launch_bounds__(256) void k_unshard(const T* __restrict__ gathered, int64_t nloc, int64_t p, int64_t n,
                                                 T* __restrict__ full, int64_t ldfull) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (j >= n) return;
  const int64_t g = j / nloc, i = j - g * nloc;
  full[b * ldfull + j] = gathered[(g * p + b) * nloc + i];
}

static inline int64_t shard_row0(const mfx_comm* cm) { return (int64_t)cm->rank * cm->nloc; }
static inline int64_t shard_nrows(const mfx_comm* cm, int64_t n) {
  const int64_t left = n - shard_row0(cm);
  return left < cm->nloc ? left : cm->nloc;
}

// y (p rows of nrows, this rank's rows of A x or A^T x) from the sharded x; `full` (p, n) receives the gathered iterate
template <typename T>
static int apply_sharded(const mfx_operator* op, const mfx_comm* cm, int transpose, const T* x, int64_t ldx, T* y,
                         int64_t ldy, int64_t p, T* full, int64_t ldfull, const KrylovWs& ws, hipStream_t stream) {
  const int64_t n = op->n, nloc = cm->nloc, nrows = shard_nrows(cm, n);
  if (cm->exchange) {  // neighbour exchange: own rows into place, then only the entries this rank's rows read from other ranks
    ScopedTimer t(3, stream);
    MFX_CHECK_HIP(hipMemcpy2DAsync(full + shard_row0(cm), sizeof(T) * ldfull, x, sizeof(T) * ldx, sizeof(T) * nrows, p,
                                   hipMemcpyDeviceToDevice, stream));
    const int rc = cm->exchange(cm->ctx, x, ldx, full, ldfull, p, op->dtype, transpose, stream);
    MFX_REQUIRE(rc == 0, MFX_ERR_CALLBACK, "exchange callback failed with code %d", rc);
  } else if (cm->allgather_rows && nrows == nloc && (int64_t)cm->world * nloc == n) {
    // every rank owns exactly nloc rows: the shards go straight into the (p, n) operator input, no pack / unpack copies
    ScopedTimer t(3, stream);
    const int rc = cm->allgather_rows(cm->ctx, x, ldx, full, ldfull, p, nloc, op->dtype, stream);
    MFX_REQUIRE(rc == 0, MFX_ERR_CALLBACK, "all-gather (rows) failed with code %d", rc);
  } else {
    ScopedTimer t(3, stream);
    k_pack_shard<T><<<dim3((unsigned)((nloc + 255) / 256), (unsigned)p), 256, 0, stream>>>(x, ldx, nrows, nloc, (T*)ws.send);
    MFX_CHECK_LAUNCH();
    const int rc = cm->allgather(cm->ctx, ws.send, ws.gathered, p * nloc, op->dtype, stream);
    MFX_REQUIRE(rc == 0, MFX_ERR_CALLBACK, "all-gather callback failed with code %d", rc);
    k_unshard<T><<<dim3((unsigned)((n + 255) / 256), (unsigned)p), 256, 0, stream>>>((const T*)ws.gathered, nloc, p, n, full, ldfull);
    MFX_CHECK_LAUNCH();
  }
  mfx_operator rows = *op;
  rows.row0 = shard_row0(cm);
  rows.nrows = nrows;
  ScopedTimer t(0, stream);
  return op_apply(&rows, full, ldfull, y, ldy, p, transpose, ws.opws, ws.opws_bytes, stream);
}

}  // namespace mfx
