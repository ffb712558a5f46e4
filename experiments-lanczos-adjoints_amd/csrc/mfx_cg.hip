// libmfx: (preconditioned) conjugate gradients, the partial (pivoted) Cholesky factor and the Woodbury
// preconditioner built from it -- the linear-solve half of the GP log-marginal likelihood ("next" tier,
// SURVEY.md §8f-1).
//
// Reference: cg.py:19-60 (pcg_fixed_step), :74-137 (pcg_adaptive), :222-241 (_safe_divide);
//            low_rank.py:10-60 (preconditioner), :63-120 (cholesky_partial), :123-228 (cholesky_partial_pivot).
//
// Layout: right-hand sides are a (p, n) row-major batch exactly like the probes of the Krylov drivers; the
// low-rank factor is stored TRANSPOSED, Lt (rank, n) row-major, so that L^T r and r - L u are the same
// "dots against rows" / "subtract rows" sweeps the Gram-Schmidt kernels do (k_dots with a zero batch stride).
// Every vector kernel: grid (ceil(n / 2048), p), per-slice partials, scalars re-reduced in the consumer's
// prologue; the fixed-step solver enqueues its whole loop without a host round trip, the adaptive one reads
// back ONE flag per iteration (the reference's while_loop condition).
#include "mfx_kernel_fn.h"
#include "mfx_vec.h"

namespace mfx {

// cg.py:222-241: a / b where |b| > eps^2, else a (the iteration may run past convergence: 0/0 -> 0)
template <typename T>
__device__ __forceinline__ T safe_div(T a, T b) {
  const T eps = dtype_eps<T>() * dtype_eps<T>();
  const bool ok = b > eps || b < -eps;
  return ok ? a / b : a;
}

template <typename T>
__device__ __forceinline__ T block_sum(T acc, T* smn) {  // valid in thread 0
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  acc = wave_sum(acc);
  if (lane == 0) smn[wid] = acc;
  __syncthreads();
  T sum = T(0);
  if (tid == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += smn[w];
  return sum;
}

// ------------------------------------------------------------------------------------------------
// CG step, first half (cg.py:44-50):  a = (r.z) / (p.Ap);  x += a p;  r -= a Ap
//   fused partials: |r|^2 (the next r.z when there is no preconditioner) and the adaptive error
//   sum (r / (atol + |x| rtol))^2 (cg.py:101-103)
// ------------------------------------------------------------------------------------------------
template <typename T>
struct CgArgs {
  T *x, *r, *pv, *z;
  const T* Ap;
  int64_t n;
  const T* part_pap;  // (p, kmax, nblk) partials of p.Ap in row 0
  T* part_rz;         // (p, nblk)
  T* part_err;        // (p, nblk) or null
  const T* rz_cur;    // (p)
  T* rz_next;         // (p)
  const int* active;  // (p) or null = all active
  T atol, rtol;
  int kmax, nblk, first, has_z;
  int nblk_in;   // slices the consumers sum over (= nblk; 1 in the row-sharded mode, where the partials arrive summed)
  T* qrow;       // re-orthogonalising variant: row i of Q = r_old / sqrt(r_old . z_old)  (cg.py:200), or null
  int64_t ldq;   // batch stride of Q
};

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_cg_xr(CgArgs<T> a) {
  __shared__ T smn[4];
  __shared__ T alpha_sh;
  const int tid = threadIdx.x;
  const int b = blockIdx.y, blk = blockIdx.x;
  if (a.active && !a.active[b]) return;
  const int64_t slice0 = (int64_t)blk * ((int64_t)blockDim.x * kEpt);
  if (tid < 64) {
    const T pap = reduce_partials_group<T, 64>(a.part_pap + (int64_t)b * a.kmax * a.nblk_in, a.nblk_in, tid);
    if (tid == 0) {
      alpha_sh = safe_div(a.rz_cur[b], pap);
      const T rzv = a.rz_cur[b];
      smn[0] = sqrt(rzv > T(0) ? rzv : T(0));  // _safe_sqrt (cg.py:244-246)
    }
  }
  __syncthreads();
  const T alpha = alpha_sh;
  const T qden = smn[0];
  __syncthreads();
  T xr[kEpt], rr[kEpt], pr[kEpt], ar[kEpt];
  load_own<T, VEC>(xr, a.x + (int64_t)b * a.n, slice0, a.n, tid);
  load_own<T, VEC>(rr, a.r + (int64_t)b * a.n, slice0, a.n, tid);
  if (a.qrow) {
    T qr[kEpt];
#pragma unroll
    for (int e = 0; e < kEpt; ++e) qr[e] = safe_div(rr[e], qden);
    store_own<T, VEC>(qr, a.qrow + (int64_t)b * a.ldq, slice0, a.n, tid);
  }
  load_own<T, VEC>(pr, a.pv + (int64_t)b * a.n, slice0, a.n, tid);
  load_own<T, VEC>(ar, a.Ap + (int64_t)b * a.n, slice0, a.n, tid);
  T s_rr = T(0), s_err = T(0);
#pragma unroll
  for (int e = 0; e < kEpt; ++e) {
    xr[e] += alpha * pr[e];
    rr[e] -= alpha * ar[e];
    s_rr += rr[e] * rr[e];
    const T q = rr[e] / (a.atol + (xr[e] < T(0) ? -xr[e] : xr[e]) * a.rtol);
    s_err += q * q;  // zero-padded lanes contribute 0 / atol = 0
  }
  store_own<T, VEC>(xr, a.x + (int64_t)b * a.n, slice0, a.n, tid);
  store_own<T, VEC>(rr, a.r + (int64_t)b * a.n, slice0, a.n, tid);
  if (!a.has_z) {
    const T s = block_sum(s_rr, smn);
    if (tid == 0) a.part_rz[(int64_t)b * a.nblk + blk] = s;
    __syncthreads();
  }
  if (a.part_err) {
    const T s = block_sum(s_err, smn);
    if (tid == 0) a.part_err[(int64_t)b * a.nblk + blk] = s;
  }
}

// second half (cg.py:52-57):  beta = (r.z)_new / (r.z)_old;  p = z + beta p      (first: p = z)
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_cg_dir(CgArgs<T> a) {
  __shared__ T beta_sh;
  const int tid = threadIdx.x;
  const int b = blockIdx.y, blk = blockIdx.x;
  if (a.active && !a.active[b]) return;
  const int64_t slice0 = (int64_t)blk * ((int64_t)blockDim.x * kEpt);
  if (tid < 64) {
    const T rz_new = reduce_partials_group<T, 64>(a.part_rz + (int64_t)b * a.nblk_in, a.nblk_in, tid);
    if (tid == 0) {
      beta_sh = a.first ? T(0) : safe_div(rz_new, a.rz_cur[b]);
      if (blk == 0) a.rz_next[b] = rz_new;
    }
  }
  __syncthreads();
  const T beta = beta_sh;
  T zr[kEpt], pr[kEpt];
  load_own<T, VEC>(zr, (a.has_z ? a.z : a.r) + (int64_t)b * a.n, slice0, a.n, tid);
  if (a.first) {
#pragma unroll
    for (int e = 0; e < kEpt; ++e) pr[e] = zr[e];
  } else {
    load_own<T, VEC>(pr, a.pv + (int64_t)b * a.n, slice0, a.n, tid);
#pragma unroll
    for (int e = 0; e < kEpt; ++e) pr[e] = zr[e] + beta * pr[e];
  }
  store_own<T, VEC>(pr, a.pv + (int64_t)b * a.n, slice0, a.n, tid);
}

// initial adaptive error sum with x = 0 (cg.py:101 on the initial state)
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_cg_err0(const T* __restrict__ r, int64_t n, T atol,
                                                    T* __restrict__ part_err, int nblk) {
  __shared__ T smn[4];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, blk = blockIdx.x;
  T rr[kEpt];
  load_own<T, VEC>(rr, r + (int64_t)b * n, (int64_t)blk * ((int64_t)blockDim.x * kEpt), n, tid);
  T acc = T(0);
#pragma unroll
  for (int e = 0; e < kEpt; ++e) {
    const T q = rr[e] / atol;
    acc += q * q;
  }
  const T s = block_sum(acc, smn);
  if (tid == 0) part_err[(int64_t)b * nblk + blk] = s;
}

// while_loop condition (cg.py:97-108), one thread per right-hand side
template <typename T>
__global__ void k_cg_cond(const T* __restrict__ part_err, int nblk, int64_t n, int64_t p, int64_t miniter,
                          int64_t maxiter, int* __restrict__ active, int64_t* __restrict__ nsteps,
                          int* __restrict__ any_active) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p) return;
  double acc = 0.0;
  for (int q = 0; q < nblk; ++q) acc += (double)part_err[b * nblk + q];
  const bool large = sqrt(acc / (double)n) > 1.0;  // false for NaN, like jnp
  const bool proceed = (large || nsteps[b] < miniter) && nsteps[b] < maxiter;
  active[b] = proceed ? 1 : 0;
  if (proceed) {
    nsteps[b] += 1;
    atomicOr(any_active, 1);
  }
}

__global__ void k_set_i32(int* dst, int v) { *dst = v; }

template <typename T>
__global__ void k_fill_i64(int64_t* dst, int64_t p, int64_t v) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < p) dst[b] = v;
}

// ------------------------------------------------------------------------------------------------
// Woodbury preconditioner (low_rank.py:31-43):  z = (v - L (s I + L^T L)^{-1} L^T v) / s
//   t = Lt v (k_dots, zero batch stride) -> u = Minv t (k_pre_small) -> z (k_pre_finish, fused r.z partial)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_pre_small(const T* __restrict__ part_t, int kmax, int nblk, int rank,
                                                   const T* __restrict__ minv, T* __restrict__ u,
                                                   const int* __restrict__ active) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* t = reinterpret_cast<T*>(smem_raw);  // [rank]
  const int b = blockIdx.x, tid = threadIdx.x;
  if (active && !active[b]) return;
  for (int i = tid; i < rank; i += 256) t[i] = reduce_partials(part_t + ((int64_t)b * kmax + i) * nblk, nblk);
  __syncthreads();
  for (int j = tid; j < rank; j += 256) {
    double acc = 0.0;
    for (int i = 0; i < rank; ++i) acc += (double)minv[(int64_t)i * rank + j] * (double)t[i];  // Minv is symmetric
    u[(int64_t)b * rank + j] = (T)acc;
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_pre_finish(const T* __restrict__ lt, int rank, const T* __restrict__ u,
                                                       const T* __restrict__ shift, const T* __restrict__ v,
                                                       int64_t ldv, T* __restrict__ z, int64_t ldz, int64_t n,
                                                       T* __restrict__ part_rz, int nblk,
                                                       const int* __restrict__ active) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* coef = reinterpret_cast<T*>(smem_raw);  // [rank] then [4]
  T* smn = coef + rank;
  const int tid = threadIdx.x;
  const int b = blockIdx.y, blk = blockIdx.x;
  if (active && !active[b]) return;
  const int64_t slice0 = (int64_t)blk * ((int64_t)blockDim.x * kEpt);
  for (int j = tid; j < rank; j += (int)blockDim.x) coef[j] = u[(int64_t)b * rank + j];
  __syncthreads();
  T vr[kEpt], acc[kEpt];
  load_own<T, VEC>(vr, v + (int64_t)b * ldv, slice0, n, tid);
#pragma unroll
  for (int e = 0; e < kEpt; ++e) acc[e] = vr[e];
  constexpr int JT = 4;
  int j = 0;
  for (; j + JT <= rank; j += JT) {
    T rr[JT][kEpt];
#pragma unroll
    for (int q = 0; q < JT; ++q) load_own<T, VEC>(rr[q], lt + (int64_t)(j + q) * n, slice0, n, tid);
#pragma unroll
    for (int q = 0; q < JT; ++q) {
      const T c = coef[j + q];
#pragma unroll
      for (int e = 0; e < kEpt; ++e) acc[e] -= c * rr[q][e];
    }
  }
  for (; j < rank; ++j) {
    T rr[kEpt];
    load_own<T, VEC>(rr, lt + (int64_t)j * n, slice0, n, tid);
    const T c = coef[j];
#pragma unroll
    for (int e = 0; e < kEpt; ++e) acc[e] -= c * rr[e];
  }
  const T inv_s = T(1) / shift[0];
  T dot = T(0);
#pragma unroll
  for (int e = 0; e < kEpt; ++e) {
    acc[e] *= inv_s;
    dot += vr[e] * acc[e];
  }
  store_own<T, VEC>(acc, z + (int64_t)b * ldz, slice0, n, tid);
  if (part_rz) {
    const T s = block_sum(dot, smn);
    if (tid == 0) part_rz[(int64_t)b * nblk + blk] = s;
  }
}

struct Precond {
  const void *lt, *minv, *shift;
  int64_t rank;
};

// row-sharded (c.comm): lt holds this rank's columns of L^T (rank, nrows); L^T v is summed over the ranks (launch_dots), and
// the r.z partials go through `stage_rz` (per-slice) into part_rz (summed over slices and ranks)
template <typename T>
static int precond_apply_t(const Ctx<T>& c, const Precond& pc, const T* v, int64_t ldv, T* z, int64_t ldz, T* part_t,
                           T* u, T* part_rz, const int* active, T* stage_rz = nullptr) {
  const int rank = (int)pc.rank;
  MFX_TRY(launch_dots<T>(c, (const T*)pc.lt, 0, c.n, rank, v, ldv, part_t));
  k_pre_small<T><<<(unsigned)c.p, 256, rank * sizeof(T), c.stream>>>(part_t, c.kmax, c.nblk_in, rank, (const T*)pc.minv, u,
                                                                    active);
  MFX_CHECK_LAUNCH();
  const size_t sh = (size_t)(rank + 4) * sizeof(T);
  T* const prod_rz = (c.comm && part_rz) ? stage_rz : part_rz;
  if (c.comm && part_rz && active) MFX_TRY(zero_async(stage_rz, sizeof(T) * c.p * c.nblk, c.stream));  // frozen right-hand sides write nothing
  MFX_VEC_SWITCH(c.vec, (k_pre_finish<T, VEC><<<c.grid(), c.wg, sh, c.stream>>>(
                            (const T*)pc.lt, rank, u, (const T*)pc.shift, v, ldv, z, ldz, c.n, prod_rz, c.nblk, active)));
  MFX_CHECK_LAUNCH();
  if (c.comm && part_rz) MFX_TRY(c.finish_from(stage_rz, part_rz, 1, 1));
  return MFX_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
struct CgWs {
  void *Ap, *pv, *z, *part_a, *part_rz, *part_err, *u, *rz, *active, *nsteps, *flag, *opws;
  int64_t opws_bytes;
  // row-sharded solve only
  void *stage, *stage_rz, *stage_err, *send, *gathered, *xfull;
};

// n = length of the vectors this process holds (the operator size, or the rows of its shard: comm != NULL)
static int64_t cg_carve(const mfx_operator* op, int64_t n, int64_t p, int64_t rank, void* ws, int64_t ws_bytes,
                        CgWs* out, const mfx_comm* comm = nullptr) {
  const size_t es = op ? dtype_size(op->dtype) : 8;
  const int64_t nblk = (n + 64 * kEpt - 1) / (64 * kEpt);
  const int64_t kmax = (rank > 0 ? rank : 1) + 1;
  Carver cv(ws, ws_bytes);
  CgWs r;
  r.Ap = cv.take(p * n * es);
  r.pv = cv.take(p * n * es);
  r.z = cv.take(p * n * es);
  r.part_a = cv.take(p * kmax * nblk * es);
  r.part_rz = cv.take(p * nblk * es);
  r.part_err = cv.take(p * nblk * es);
  r.u = cv.take(p * kmax * es);
  r.rz = cv.take(2 * p * es);
  r.active = cv.take(p * sizeof(int));
  r.nsteps = cv.take(p * sizeof(int64_t));
  r.flag = cv.take(256);
  r.opws_bytes = op ? op_workspace_bytes(op, p, p) : 0;
  r.opws = cv.take(r.opws_bytes);
  r.stage = r.stage_rz = r.stage_err = r.send = r.gathered = r.xfull = nullptr;
  if (comm) {
    r.stage = cv.take(p * kmax * nblk * es);
    r.stage_rz = cv.take(p * nblk * es);
    r.stage_err = cv.take(p * nblk * es);
    r.send = cv.take(p * comm->nloc * es);
    r.gathered = cv.take(comm->world * p * comm->nloc * es);
    r.xfull = cv.take(p * op->n * es);
  }
  if (out) *out = r;
  return cv.off;
}

// ------------------------------------------------------------------------------------------------
// PCG driver
// ------------------------------------------------------------------------------------------------
template <typename T>
static int pcg_t(const mfx_operator* op, const T* b, int64_t ldb, int64_t n, int64_t p, const Precond* pc,
                 int64_t maxiter, int64_t miniter, double atol, double rtol, int adaptive, T* x, T* r,
                 int64_t* num_steps, T* Q, const CgWs& ws, hipStream_t stream, const mfx_comm* comm = nullptr) {
  // comm != NULL: row-sharded -- n = this rank's rows of every vector (and columns of L^T), op->n = the system size; the
  // operator input is gathered, every inner product (p.Ap, r.z, the error sum, L^T v) is summed over the ranks, so all ranks
  // take the same steps and stop together
  const int64_t rank = pc ? pc->rank : 0;
  const int64_t kdim = Q ? (rank > maxiter ? rank : maxiter) : rank;
  Ctx<T> c(n, kdim > 0 ? kdim : 1, p, pick_vec<T>(n, {x, r, ws.Ap, ws.pv, ws.z, pc ? pc->lt : nullptr, Q}), stream);
  if (comm) c.shard(comm, static_cast<T*>(ws.stage));
  KrylovWs kws{};
  kws.opws = ws.opws; kws.opws_bytes = ws.opws_bytes; kws.send = ws.send; kws.gathered = ws.gathered;
  T* const stage_rz = (T*)ws.stage_rz;
  T* const stage_err = (T*)ws.stage_err;
  T* Ap = (T*)ws.Ap;
  T* pv = (T*)ws.pv;
  T* z = (T*)ws.z;
  T* rz = (T*)ws.rz;
  int* active = adaptive ? (int*)ws.active : nullptr;
  int64_t* nsteps = (int64_t*)ws.nsteps;
  int* flag = (int*)ws.flag;

  // x = 0, r = b - A 0 = b (cg.py:29-33)
  MFX_CHECK_HIP(hipMemsetAsync(x, 0, sizeof(T) * p * n, stream));
  MFX_CHECK_HIP(hipMemcpy2DAsync(r, sizeof(T) * n, b, sizeof(T) * ldb, sizeof(T) * n, p, hipMemcpyDeviceToDevice, stream));
  CgArgs<T> a{};
  a.x = x; a.r = r; a.pv = pv; a.z = z; a.Ap = Ap; a.n = n;
  a.part_pap = (T*)ws.part_a; a.part_rz = (T*)ws.part_rz; a.part_err = adaptive ? (T*)ws.part_err : nullptr;
  a.active = nullptr; a.atol = (T)atol; a.rtol = (T)rtol; a.kmax = c.kmax; a.nblk = c.nblk; a.nblk_in = c.nblk_in; a.has_z = pc ? 1 : 0;
  if (comm) {  // producers write per-slice partials into their staging buffers; the summed values land where the consumers read
    a.part_rz = stage_rz;
    if (adaptive) a.part_err = stage_err;
  }
  T* const part_rz = (T*)ws.part_rz;
  T* const part_err = (T*)ws.part_err;
  {
    ScopedTimer t(2, stream);
    if (pc) {
      MFX_TRY(precond_apply_t<T>(c, *pc, r, n, z, n, (T*)ws.part_a, (T*)ws.u, part_rz, nullptr, stage_rz));
    } else {
      MFX_TRY(launch_sumsq<T>(c, r, n, part_rz));
    }
    CgArgs<T> d = a;  // k_cg_dir only READS the r.z partials: always the summed ones
    d.part_rz = part_rz;
    d.first = 1; d.rz_cur = rz; d.rz_next = rz;
    MFX_VEC_SWITCH(c.vec, (k_cg_dir<T, VEC><<<c.grid(), c.wg, 0, stream>>>(d)));
    MFX_CHECK_LAUNCH();
    a.first = 0;
    if (adaptive) {
      MFX_VEC_SWITCH(c.vec, (k_cg_err0<T, VEC><<<c.grid(), c.wg, 0, stream>>>(r, n, (T)atol, comm ? stage_err : part_err, c.nblk)));
      MFX_CHECK_LAUNCH();
      if (comm) MFX_TRY(c.finish_from(stage_err, part_err, 1, 1));
      MFX_CHECK_HIP(hipMemsetAsync(nsteps, 0, sizeof(int64_t) * p, stream));
    } else {
      k_fill_i64<T><<<(unsigned)((p + 255) / 256), 256, 0, stream>>>(nsteps, p, maxiter);
      MFX_CHECK_LAUNCH();
    }
  }
  a.active = active;
  for (int64_t it = 0; it < maxiter; ++it) {
    if (adaptive) {
      MFX_CHECK_HIP(hipMemsetAsync(flag, 0, sizeof(int), stream));
      k_cg_cond<T><<<(unsigned)((p + 255) / 256), 256, 0, stream>>>(part_err, c.nblk_in, op->n, p, miniter, maxiter,
                                                                   active, nsteps, flag);
      MFX_CHECK_LAUNCH();
      int host_flag = 0;
      MFX_CHECK_HIP(hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, stream));
      MFX_CHECK_HIP(hipStreamSynchronize(stream));
      if (!host_flag) break;
    }
    if (comm) {
      MFX_TRY(apply_sharded<T>(op, comm, 0, pv, n, Ap, n, p, static_cast<T*>(ws.xfull), op->n, kws, stream));
    } else {
      MFX_TRY(apply_any(op, 0, pv, n, nullptr, 0, Ap, n, p, ws.opws, ws.opws_bytes, stream));
    }
    ScopedTimer t(2, stream);
    MFX_TRY(launch_dots<T>(c, Ap, n, 0, 1, pv, n, (T*)ws.part_a));
    a.rz_cur = rz + (it & 1) * p;
    a.rz_next = rz + ((it + 1) & 1) * p;
    if (Q) {
      a.qrow = Q + it * n;
      a.ldq = maxiter * n;
    }
    if (comm && active) {  // frozen right-hand sides write no partials: their staged slices must read as zero
      if (!pc) MFX_TRY(zero_async(stage_rz, sizeof(T) * p * c.nblk, stream));
      MFX_TRY(zero_async(stage_err, sizeof(T) * p * c.nblk, stream));
    }
    MFX_VEC_SWITCH(c.vec, (k_cg_xr<T, VEC><<<c.grid(), c.wg, 0, stream>>>(a)));
    MFX_CHECK_LAUNCH();
    if (comm) {
      if (!pc) MFX_TRY(c.finish_from(stage_rz, part_rz, 1, 1));
      if (adaptive) MFX_TRY(c.finish_from(stage_err, part_err, 1, 1));
    }
    if (pc) MFX_TRY(precond_apply_t<T>(c, *pc, r, n, z, n, (T*)ws.part_a, (T*)ws.u, part_rz, active, stage_rz));
    if (Q) {
      // re-orthogonalise against the stored (normalised) residuals, then precondition again (cg.py:199-205):
      //   r -= Q (Q^T z);  z = P(r);  r.z
      const int m = (int)(it + 1);
      MFX_TRY(launch_dots<T>(c, Q, maxiter * n, n, m, pc ? z : r, n, (T*)ws.part_a));
      UpdateArgs<T> ua{};
      ua.rows = Q; ua.rows_ldb = maxiter * n; ua.row_stride = n; ua.m = m;
      ua.partial_in = (T*)ws.part_a; ua.s1 = T(1);
      ua.x = r; ua.ldx = n; ua.y = r; ua.ldy = n;
      ua.partial_norm = part_rz;
      MFX_TRY(launch_update<T>(c, ua, false, true));
      if (pc) MFX_TRY(precond_apply_t<T>(c, *pc, r, n, z, n, (T*)ws.part_a, (T*)ws.u, part_rz, active, stage_rz));
    }
    CgArgs<T> d = a;
    d.part_rz = part_rz;
    MFX_VEC_SWITCH(c.vec, (k_cg_dir<T, VEC><<<c.grid(), c.wg, 0, stream>>>(d)));
    MFX_CHECK_LAUNCH();
  }
  if (num_steps)
    MFX_CHECK_HIP(hipMemcpyAsync(num_steps, nsteps, sizeof(int64_t) * p, hipMemcpyDeviceToDevice, stream));
  return MFX_OK;
}

// ------------------------------------------------------------------------------------------------
// partial Cholesky (low_rank.py:63-228), in the ORIGINAL row order: pivot pi_i = argmax_j |d_j| of the residual
// diagonal d_j = K_jj - sum_c L_jc^2 (recomputed from L each step like the reference, :185-188), then
//   L[:, i] = (K[:, pi_i] - L L[pi_i, :]^T) / sqrt(d_pi)
// (the reference carries row permutations and undoes them at the end, :227-228: same numbers).
// ------------------------------------------------------------------------------------------------
template <typename T>
struct ElemArgs {
  int kind;  // MFX_OP_DENSE / MFX_OP_RBF
  const T* A;
  int64_t lda;
  const T* X;
  int d, ard, kernel_fn;
  const T *ls, *os, *noise;
  int with_noise;
};

template <typename T>
__device__ __forceinline__ T elem(const ElemArgs<T>& a, int64_t i, int64_t j) {
  if (a.kind == MFX_OP_DENSE) return a.A[i * a.lda + j];
  T dist = T(0);
  if (i != j) {
    for (int c = 0; c < a.d; ++c) {
      const T l = a.ls[a.ard ? c : 0];
      const T df = a.X[i * a.d + c] / l - a.X[j * a.d + c] / l;
      dist += df * df;
    }
  }
  T kv, wl;
  kernel_eval<T>(a.kernel_fn, dist, kv, wl);
  T v = a.os[0] * kv;
  if (a.with_noise && i == j) v += a.noise[0];
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void k_pchol_diag(ElemArgs<T> ea, const T* __restrict__ lt, int64_t n, int step,
                                                    T* __restrict__ pval, int64_t* __restrict__ pidx) {
  __shared__ T sv[256];
  __shared__ int64_t si[256];
  const int tid = threadIdx.x;
  const int64_t j = (int64_t)blockIdx.x * 256 + tid;
  T best = T(-1);
  int64_t bi = n;
  if (j < n) {
    T dj = elem(ea, j, j);
    for (int c = 0; c < step; ++c) {
      const T l = lt[(int64_t)c * n + j];
      dj -= l * l;
    }
    best = dj < T(0) ? -dj : dj;
    if (!(best == best)) best = T(-1);  // NaN never wins
    bi = j;
  }
  sv[tid] = best;
  si[tid] = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      const T ov = sv[tid + o];
      const int64_t oi = si[tid + o];
      if (ov > sv[tid] || (ov == sv[tid] && oi < si[tid])) {
        sv[tid] = ov;
        si[tid] = oi;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    pval[blockIdx.x] = sv[0];
    pidx[blockIdx.x] = si[0];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_pchol_col(ElemArgs<T> ea, T* __restrict__ lt, int64_t n, int step, int pivot,
                                                   const T* __restrict__ pval, const int64_t* __restrict__ pidx,
                                                   int nparts, int64_t* __restrict__ pivots,
                                                   int* __restrict__ success) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* lrow = reinterpret_cast<T*>(smem_raw);  // [step]
  __shared__ T sv[256];
  __shared__ int64_t si[256];
  __shared__ T lii_sh;
  const int tid = threadIdx.x;
  int64_t pi = step;
  if (pivot) {
    T best = T(-2);
    int64_t bi = n;
    for (int q = tid; q < nparts; q += 256) {
      const T ov = pval[q];
      const int64_t oi = pidx[q];
      if (ov > best || (ov == best && oi < bi)) {
        best = ov;
        bi = oi;
      }
    }
    sv[tid] = best;
    si[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        const T ov = sv[tid + o];
        const int64_t oi = si[tid + o];
        if (ov > sv[tid] || (ov == sv[tid] && oi < si[tid])) {
          sv[tid] = ov;
          si[tid] = oi;
        }
      }
      __syncthreads();
    }
    pi = si[0] < n ? si[0] : 0;
  }
  for (int c = tid; c < step; c += 256) lrow[c] = lt[(int64_t)c * n + pi];
  __syncthreads();
  if (tid == 0) {
    T dpi = elem(ea, pi, pi);
    for (int c = 0; c < step; ++c) dpi -= lrow[c] * lrow[c];
    lii_sh = sqrt(dpi);  // NaN for a non-positive residual, exactly like the reference (low_rank.py:203-204)
    if (blockIdx.x == 0) {
      pivots[step] = pi;
      if (!(dpi > T(0))) *success = 0;
    }
  }
  __syncthreads();
  const T lii = lii_sh;
  const int64_t j = (int64_t)blockIdx.x * 256 + tid;
  if (j < n) {
    T col = elem(ea, j, pi);
    for (int c = 0; c < step; ++c) col -= lt[(int64_t)c * n + j] * lrow[c];
    lt[(int64_t)step * n + j] = col / lii;
  }
}

template <typename T>
static int pchol_t(const mfx_operator* op, int64_t rank, int pivot, int with_noise, T* lt, int64_t* pivots,
                   int* success, void* ws, int64_t ws_bytes, hipStream_t stream) {
  const int64_t n = op->n;
  const int nparts = (int)((n + 255) / 256);
  Carver cv(ws, ws_bytes);
  T* pval = (T*)cv.take(nparts * sizeof(T));
  int64_t* pidx = (int64_t*)cv.take(nparts * sizeof(int64_t));
  MFX_REQUIRE(ws && cv.ok(), MFX_ERR_WORKSPACE, "pivoted Cholesky workspace too small");
  ElemArgs<T> ea{};
  ea.kind = op->kind;
  ea.A = (const T*)op->dense_a; ea.lda = op->lda;
  ea.X = (const T*)op->x; ea.d = op->d; ea.ard = op->ard; ea.kernel_fn = op->kernel_fn;
  ea.ls = (const T*)op->lengthscale; ea.os = (const T*)op->outputscale; ea.noise = (const T*)op->noise;
  ea.with_noise = with_noise;
  k_set_i32<<<1, 1, 0, stream>>>(success, 1);
  MFX_CHECK_LAUNCH();
  for (int step = 0; step < (int)rank; ++step) {
    if (pivot) {
      k_pchol_diag<T><<<nparts, 256, 0, stream>>>(ea, lt, n, step, pval, pidx);
      MFX_CHECK_LAUNCH();
    }
    k_pchol_col<T><<<nparts, 256, (size_t)(step + 1) * sizeof(T), stream>>>(ea, lt, n, step, pivot, pval, pidx, nparts,
                                                                          pivots, success);
    MFX_CHECK_LAUNCH();
  }
  return MFX_OK;
}

}  // namespace mfx

using namespace mfx;

static int check_precond(int64_t n, int64_t rank, const void* lt, const void* minv, const void* shift) {
  MFX_REQUIRE(rank >= 1 && rank <= n, MFX_ERR_INVALID, "preconditioner rank %lld outside [1, n = %lld]", (long long)rank,
              (long long)n);
  // L^T v runs through k_dots with m = rank rows: 4 * rank * 8 B of LDS for its per-wave partials
  MFX_REQUIRE(rank <= 1024, MFX_ERR_UNSUPPORTED, "preconditioner rank %lld > 1024", (long long)rank);
  MFX_REQUIRE(lt && minv && shift, MFX_ERR_INVALID, "preconditioner needs lt, minv and shift");
  return MFX_OK;
}

extern "C" {

int64_t mfx_pcg_workspace_bytes(const mfx_operator* op, int64_t n, int64_t p, int64_t rank) {
  if (!op || n <= 0 || p <= 0 || rank < 0) return -1;
  const int64_t a = cg_carve(op, n, p, rank, nullptr, 0, nullptr);
  const int64_t b = ((n + 255) / 256) * 16 + 1024;  // pivoted Cholesky partials
  return (a > b ? a : b) + 256;
}

int mfx_pcg_solve(const mfx_operator* op, const void* b, int64_t ldb, int64_t n, int64_t p, const void* precond_lt,
                  int64_t rank, const void* precond_minv, const void* precond_shift, int64_t maxiter, int64_t miniter,
                  double atol, double rtol, int adaptive, void* x, void* r, void* num_steps, void* ws,
                  int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(op && b && x && r, MFX_ERR_INVALID, "mfx_pcg_solve: null argument");
  MFX_REQUIRE(op->n == n && n >= 1 && p >= 1 && ldb >= n, MFX_ERR_INVALID, "mfx_pcg_solve: bad sizes n=%lld p=%lld ldb=%lld",
              (long long)n, (long long)p, (long long)ldb);
  MFX_REQUIRE(p <= 65535, MFX_ERR_UNSUPPORTED, "at most 65535 right-hand sides per call");
  MFX_REQUIRE(maxiter >= 0 && miniter >= 0, MFX_ERR_INVALID, "negative iteration count");
  MFX_REQUIRE(op->dtype == MFX_F32 || op->dtype == MFX_F64, MFX_ERR_INVALID, "bad dtype");
  Precond pc{precond_lt, precond_minv, precond_shift, rank};
  if (precond_lt) MFX_TRY(check_precond(n, rank, precond_lt, precond_minv, precond_shift));
  CgWs w;
  MFX_REQUIRE(ws && cg_carve(op, n, p, precond_lt ? rank : 0, ws, ws_bytes, &w) <= ws_bytes, MFX_ERR_WORKSPACE,
              "mfx_pcg_solve: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  PrepScope prep_scope;
  if (op->dtype == MFX_F32)
    return pcg_t<float>(op, (const float*)b, ldb, n, p, precond_lt ? &pc : nullptr, maxiter, miniter, atol, rtol, adaptive,
                        (float*)x, (float*)r, (int64_t*)num_steps, nullptr, w, s);
  return pcg_t<double>(op, (const double*)b, ldb, n, p, precond_lt ? &pc : nullptr, maxiter, miniter, atol, rtol, adaptive,
                       (double*)x, (double*)r, (int64_t*)num_steps, nullptr, w, s);
}

static int64_t pcg_shard_rows(const mfx_comm* cm, int64_t n) {
  const int64_t left = n - (int64_t)cm->rank * cm->nloc;
  return left < cm->nloc ? left : cm->nloc;
}

int64_t mfx_pcg_sharded_workspace_bytes(const mfx_operator* op, const mfx_comm* comm, int64_t n, int64_t p, int64_t rank) {
  if (!op || !comm || comm->nloc < 1 || n <= 0 || p <= 0 || rank < 0) return -1;
  return cg_carve(op, comm->nloc, p, rank, nullptr, 0, nullptr, comm) + 256;
}

int mfx_pcg_solve_sharded(const mfx_operator* op, const mfx_comm* comm, const void* b, int64_t ldb, int64_t n, int64_t p,
                          const void* precond_lt, int64_t rank, const void* precond_minv, const void* precond_shift,
                          int64_t maxiter, int64_t miniter, double atol, double rtol, int adaptive, void* x, void* r,
                          void* num_steps, void* ws, int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(op && comm && b && x && r, MFX_ERR_INVALID, "mfx_pcg_solve_sharded: null argument");
  MFX_REQUIRE(op->n == n && n >= 1 && p >= 1, MFX_ERR_INVALID, "mfx_pcg_solve_sharded: bad sizes n=%lld p=%lld", (long long)n, (long long)p);
  MFX_REQUIRE(op->kind != MFX_OP_CALLBACK, MFX_ERR_UNSUPPORTED, "row sharding needs a native operator");
  MFX_REQUIRE(op->nrows == 0, MFX_ERR_INVALID, "pass the whole operator: the driver takes this rank's rows itself");
  MFX_REQUIRE(comm->allreduce_sum && comm->allgather, MFX_ERR_INVALID, "communicator without callbacks");
  MFX_REQUIRE(comm->world >= 1 && comm->rank >= 0 && comm->rank < comm->world, MFX_ERR_INVALID, "bad rank %d of %d", comm->rank, comm->world);
  MFX_REQUIRE(comm->nloc >= 64 && comm->nloc % 64 == 0, MFX_ERR_INVALID, "rows per rank (%lld) must be a positive multiple of 64", (long long)comm->nloc);
  MFX_REQUIRE((int64_t)comm->world * comm->nloc >= n && (int64_t)(comm->world - 1) * comm->nloc < n, MFX_ERR_INVALID,
              "%d ranks x %lld rows do not tile n = %lld with a non-empty last shard", comm->world, (long long)comm->nloc, (long long)n);
  const int64_t nrows = pcg_shard_rows(comm, n);
  MFX_REQUIRE(ldb >= nrows, MFX_ERR_INVALID, "ldb = %lld < this rank's %lld rows", (long long)ldb, (long long)nrows);
  MFX_REQUIRE(p <= 65535, MFX_ERR_UNSUPPORTED, "at most 65535 right-hand sides per call");
  MFX_REQUIRE(maxiter >= 0 && miniter >= 0, MFX_ERR_INVALID, "negative iteration count");
  MFX_REQUIRE(op->dtype == MFX_F32 || op->dtype == MFX_F64, MFX_ERR_INVALID, "bad dtype");
  Precond pc{precond_lt, precond_minv, precond_shift, rank};
  if (precond_lt) MFX_TRY(check_precond(n, rank, precond_lt, precond_minv, precond_shift));
  CgWs w;
  MFX_REQUIRE(ws && cg_carve(op, nrows, p, precond_lt ? rank : 0, ws, ws_bytes, &w, comm) <= ws_bytes, MFX_ERR_WORKSPACE,
              "mfx_pcg_solve_sharded: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  PrepScope prep_scope;
  if (op->dtype == MFX_F32)
    return pcg_t<float>(op, (const float*)b, ldb, nrows, p, precond_lt ? &pc : nullptr, maxiter, miniter, atol, rtol, adaptive,
                        (float*)x, (float*)r, (int64_t*)num_steps, nullptr, w, s, comm);
  return pcg_t<double>(op, (const double*)b, ldb, nrows, p, precond_lt ? &pc : nullptr, maxiter, miniter, atol, rtol, adaptive,
                       (double*)x, (double*)r, (int64_t*)num_steps, nullptr, w, s, comm);
}

int mfx_pcg_solve_reortho(const mfx_operator* op, const void* b, int64_t ldb, int64_t n, int64_t p,
                          const void* precond_lt, int64_t rank, const void* precond_minv, const void* precond_shift,
                          int64_t num_matvecs, void* x, void* r, void* q, void* ws, int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(op && b && x && r && q, MFX_ERR_INVALID, "mfx_pcg_solve_reortho: null argument");
  MFX_REQUIRE(op->n == n && n >= 1 && p >= 1 && ldb >= n, MFX_ERR_INVALID, "mfx_pcg_solve_reortho: bad sizes");
  MFX_REQUIRE(p <= 65535, MFX_ERR_UNSUPPORTED, "at most 65535 right-hand sides per call");
  MFX_REQUIRE(num_matvecs >= 1 && num_matvecs <= 1024, MFX_ERR_INVALID, "num_matvecs %lld outside [1, 1024]",
              (long long)num_matvecs);
  MFX_REQUIRE(op->dtype == MFX_F32 || op->dtype == MFX_F64, MFX_ERR_INVALID, "bad dtype");
  Precond pc{precond_lt, precond_minv, precond_shift, rank};
  if (precond_lt) MFX_TRY(check_precond(n, rank, precond_lt, precond_minv, precond_shift));
  const int64_t kdim = (precond_lt && rank > num_matvecs) ? rank : num_matvecs;
  CgWs w;
  MFX_REQUIRE(ws && cg_carve(op, n, p, kdim, ws, ws_bytes, &w) <= ws_bytes, MFX_ERR_WORKSPACE,
              "mfx_pcg_solve_reortho: workspace too small (size it with rank = max(rank, num_matvecs))");
  hipStream_t s = static_cast<hipStream_t>(stream);
  PrepScope prep_scope;
  if (op->dtype == MFX_F32)
    return pcg_t<float>(op, (const float*)b, ldb, n, p, precond_lt ? &pc : nullptr, num_matvecs, 0, 1.0, 0.0, 0, (float*)x,
                        (float*)r, nullptr, (float*)q, w, s);
  return pcg_t<double>(op, (const double*)b, ldb, n, p, precond_lt ? &pc : nullptr, num_matvecs, 0, 1.0, 0.0, 0, (double*)x,
                       (double*)r, nullptr, (double*)q, w, s);
}

int mfx_precond_apply(int dtype, int64_t n, int64_t rank, const void* lt, const void* minv, const void* shift,
                      const void* v, int64_t ldv, void* z, int64_t ldz, int64_t p, void* ws, int64_t ws_bytes,
                      void* stream) {
  MFX_REQUIRE(v && z && n >= 1 && p >= 1 && ldv >= n && ldz >= n, MFX_ERR_INVALID, "mfx_precond_apply: bad arguments");
  MFX_REQUIRE(dtype == MFX_F32 || dtype == MFX_F64, MFX_ERR_INVALID, "bad dtype");
  MFX_REQUIRE(p <= 65535, MFX_ERR_UNSUPPORTED, "at most 65535 vectors per call");
  MFX_TRY(check_precond(n, rank, lt, minv, shift));
  mfx_operator fake{};
  fake.dtype = dtype;
  fake.kind = MFX_OP_DENSE;
  fake.n = n;
  CgWs w;
  MFX_REQUIRE(ws && cg_carve(&fake, n, p, rank, ws, ws_bytes, &w) <= ws_bytes, MFX_ERR_WORKSPACE,
              "mfx_precond_apply: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  Precond pc{lt, minv, shift, rank};
  if (dtype == MFX_F32) {
    Ctx<float> c(n, rank, p, pick_vec<float>(n, {v, z, lt}), s);
    return precond_apply_t<float>(c, pc, (const float*)v, ldv, (float*)z, ldz, (float*)w.part_a, (float*)w.u, nullptr,
                                  nullptr);
  }
  Ctx<double> c(n, rank, p, pick_vec<double>(n, {v, z, lt}), s);
  return precond_apply_t<double>(c, pc, (const double*)v, ldv, (double*)z, ldz, (double*)w.part_a, (double*)w.u, nullptr,
                                 nullptr);
}

int64_t mfx_gram_cross_workspace_bytes(const mfx_operator* op, int64_t m) {
  if (!op || op->kind != MFX_OP_RBF || m <= 0) return -1;
  return rbf_cross_ws_bytes(op, m) + 256;
}

int mfx_gram_cross_apply(const mfx_operator* op, const void* xnew, int64_t m, const void* v, int64_t ldv, void* y,
                         int64_t ldy, int64_t p, void* ws, int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(op && xnew && v && y, MFX_ERR_INVALID, "mfx_gram_cross_apply: null argument");
  MFX_REQUIRE(m >= 1 && p >= 1 && ldv >= op->n && ldy >= m, MFX_ERR_INVALID, "mfx_gram_cross_apply: bad sizes");
  MFX_REQUIRE(p <= 65535 * 8, MFX_ERR_UNSUPPORTED, "too many vectors per call");
  return op_cross_apply(op, xnew, m, v, ldv, y, ldy, p, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int mfx_partial_cholesky(const mfx_operator* op, int64_t rank, int pivot, int with_noise, void* lt, void* pivots,
                         void* success, void* ws, int64_t ws_bytes, void* stream) {
  MFX_REQUIRE(op && lt && pivots && success, MFX_ERR_INVALID, "mfx_partial_cholesky: null argument");
  MFX_REQUIRE(op->kind == MFX_OP_DENSE || op->kind == MFX_OP_RBF, MFX_ERR_UNSUPPORTED,
              "partial Cholesky needs element access: dense or kernel-Gram operators only");
  MFX_REQUIRE(rank <= op->n, MFX_ERR_INVALID, "Rank exceeds n: %lld >= %lld.", (long long)rank, (long long)op->n);
  MFX_REQUIRE(rank >= 1, MFX_ERR_INVALID, "Rank must be positive, but %lld < 1.", (long long)rank);
  MFX_REQUIRE(rank <= 1024, MFX_ERR_UNSUPPORTED, "rank %lld > 1024 (the preconditioner applies at most 1024 columns)", (long long)rank);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (op->dtype == MFX_F32)
    return pchol_t<float>(op, rank, pivot, with_noise, (float*)lt, (int64_t*)pivots, (int*)success, ws, ws_bytes, s);
  return pchol_t<double>(op, rank, pivot, with_noise, (double*)lt, (int64_t*)pivots, (int*)success, ws, ws_bytes, s);
}

}  // extern "C"
