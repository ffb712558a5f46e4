// libmfx: the small dense pieces of stochastic Lanczos quadrature and the probe sampler.
//   - batched symmetric-tridiagonal eigensolver (lanczos.py:48-53), one wave per probe
//   - VJP of  sum_a U[0][a]^2 f(lam_a)  w.r.t. the tridiagonal entries (lanczos.py:53-59)
//   - counter-based Rademacher probes (matfree sampler_rademacher call sites, util/gp_util.py:557)
#include <math.h>

#include "mfx_internal.h"

namespace mfx {

// Implicit-shift QL on (d, e) with accumulation of the rotations into Z (EISPACK tql2 scheme).
// The scalar recurrence is evaluated redundantly by all 64 lanes (wave-uniform, LDS broadcast reads);
// the O(k) row updates of Z per rotation are spread over the lanes (lane = row of Z).
//
// DEEP (k > 120, fp64 only): k (k | 1) doubles of Z no longer fit the 160 KB of LDS -- the rotations are accumulated in the caller's evecs
// itself (same layout, Z[row][col], leading dimension k; a lane reads back only what it wrote: program order is enough), d and e stay in LDS.
// The reference's SuiteSparse sweeps go to depth 150 (benchmark.py:21,83; plot_quadrant.py:22).
template <typename T, bool DEEP>
__global__ __launch_bounds__(64) void k_tridiag_eigh(const T* __restrict__ alpha, const T* __restrict__ beta,
                                                     int64_t ldbeta, int k, T* __restrict__ evals,
                                                     T* __restrict__ evecs) {
  static_assert(!DEEP || sizeof(T) == 8, "the deep variant accumulates in the fp64 output");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* d = reinterpret_cast<double*>(smem_raw);
  double* e = d + k;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  double* Z = DEEP ? reinterpret_cast<double*>(evecs) + b * k * k : e + k;
  const int ldz = DEEP ? k : (k | 1);  // LDS: odd leading dimension, conflict-free column access
  for (int i = lane; i < k; i += 64) {
    d[i] = (double)alpha[b * k + i];
    e[i] = (i < k - 1) ? (double)beta[b * ldbeta + i] : 0.0;
  }
  if (DEEP) {  // by the lane that will rotate the row
    for (int row = lane; row < k; row += 64)
      for (int col = 0; col < k; ++col) Z[row * ldz + col] = row == col ? 1.0 : 0.0;
  } else {
    for (int t = lane; t < k * k; t += 64) Z[(t / k) * ldz + (t % k)] = (t / k == t % k) ? 1.0 : 0.0;
  }
  __syncthreads();
  const double eps = 2.220446049250313e-16;
  bool converged = true;  // wave-uniform
  for (int l = 0; l < k; ++l) {
    int iter = 0;
    while (true) {
      int m = l;
      for (; m < k - 1; ++m) {
        const double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(e[m]) <= eps * dd) break;
      }
      if (m == l) break;
      if (++iter > 200) {  // no silent garbage: this probe's eigenvalues become NaN (documented in mfx.h)
        converged = false;
        break;
      }
      double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
      double r = hypot(g, 1.0);
      g = d[m] - d[l] + e[l] / (g + copysign(r, g));
      double s = 1.0, c = 1.0, p = 0.0;
      int i;
      for (i = m - 1; i >= l; --i) {
        double f = s * e[i];
        const double bb = c * e[i];
        r = hypot(f, g);
        e[i + 1] = r;
        if (r == 0.0) {
          d[i + 1] -= p;
          e[m] = 0.0;
          break;
        }
        s = f / r;
        c = g / r;
        g = d[i + 1] - p;
        r = (d[i] - g) * s + 2.0 * c * bb;
        p = s * r;
        d[i + 1] = g + p;
        g = c * r - bb;
        for (int row = lane; row < k; row += 64) {
          const double z1 = Z[row * ldz + i + 1], z0 = Z[row * ldz + i];
          Z[row * ldz + i + 1] = s * z0 + c * z1;
          Z[row * ldz + i] = c * z0 - s * z1;
        }
      }
      if (r == 0.0 && i >= l) continue;
      d[l] -= p;
      e[l] = g;
      e[m] = 0.0;
    }
  }
  __syncthreads();
  for (int i = lane; i < k; i += 64) evals[b * k + i] = converged ? (T)d[i] : (T)NAN;
  if (!DEEP)
    for (int t = lane; t < k * k; t += 64) evecs[b * k * k + t] = (T)Z[(t / k) * ldz + (t % k)];
}

// G = gout * U (F o u0 u0^T) U^T with F the divided differences of f at the eigenvalues;
// dalpha_i = G_ii, dbeta_i = 2 G_{i,i+1}.
// DEEP (k > 120): M = F o u0 u0^T is not stored (k^2 doubles of LDS) but evaluated where it is used -- k^3 divided differences per probe
// instead of k^2, irrelevant next to the Krylov passes at such depths; LDS holds lam, f, f' and u0 (4 k doubles).
template <typename T, bool DEEP>
__global__ __launch_bounds__(64) void k_quadform_bwd(const T* __restrict__ evals, const T* __restrict__ evecs,
                                                     const T* __restrict__ fvals, const T* __restrict__ dfvals,
                                                     const T* __restrict__ gout, int k, T* __restrict__ dalpha,
                                                     T* __restrict__ dbeta, int64_t lddbeta) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* M = reinterpret_cast<double*>(smem_raw);  // [k][k]
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const T* lam = evals + b * k;
  const T* U = evecs + b * k * k;
  const T* f = fvals + b * k;
  const T* df = dfvals + b * k;
  double lmax = 0.0;
  for (int a = 0; a < k; ++a) lmax = fmax(lmax, fabs((double)lam[a]));
  const double tol = (sizeof(T) == 4 ? 1e-6 : 1e-13) * lmax;
  auto entry = [&](int a, int c, double la, double lc, double fa, double fc, double dfa, double dfc, double ua, double uc) {
    const double dl = la - lc;
    const double F = (a == c || fabs(dl) <= tol) ? 0.5 * (dfa + dfc) : (fa - fc) / dl;
    return F * ua * uc;  // U[0][a] U[0][c]
  };
  double *sl = M, *sf = M + k, *sd = M + 2 * k, *su = M + 3 * k;  // DEEP only
  if (DEEP) {
    for (int a = lane; a < k; a += 64) {
      sl[a] = (double)lam[a];
      sf[a] = (double)f[a];
      sd[a] = (double)df[a];
      su[a] = (double)U[a];
    }
  } else {
    for (int t = lane; t < k * k; t += 64) {
      const int a = t / k, c = t % k;
      M[t] = entry(a, c, (double)lam[a], (double)lam[c], (double)f[a], (double)f[c], (double)df[a], (double)df[c], (double)U[a], (double)U[c]);
    }
  }
  __syncthreads();
  const double go = (double)gout[b];
  for (int i = lane; i < k; i += 64) {
    double gii = 0.0, gi1 = 0.0;
    for (int a = 0; a < k; ++a) {
      double t = 0.0;
      if (DEEP) {
        for (int c = 0; c < k; ++c) t += entry(a, c, sl[a], sl[c], sf[a], sf[c], sd[a], sd[c], su[a], su[c]) * (double)U[i * k + c];
      } else {
        for (int c = 0; c < k; ++c) t += M[a * k + c] * (double)U[i * k + c];
      }
      // t = (M U_i)_a ; G_ij = sum_a U_ja t_a
      gii += (double)U[i * k + a] * t;
      if (i + 1 < k) gi1 += (double)U[(i + 1) * k + a] * t;
    }
    dalpha[b * k + i] = (T)(go * gii);
    if (i + 1 < k) dbeta[b * lddbeta + i] = (T)(2.0 * go * gi1);
  }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <typename T>
__global__ void k_rademacher(uint64_t seed, int64_t first_probe, int64_t n, T* __restrict__ out) {
  const int64_t b = blockIdx.y;
  const uint64_t key = splitmix64(seed ^ ((uint64_t)(first_probe + b) * 0xD1342543DE82EF95ull));
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t bits = splitmix64(key + (uint64_t)i);
    out[b * n + i] = (bits >> 63) ? T(1) : T(-1);
  }
}

constexpr int kSmallLdsDepth = 120;   // up to here the k x k work matrices of the two kernels above live in LDS
constexpr int kSmallMaxDepth = 2048;  // d, e (or lam, f, f', u0) still do: 4 k doubles

template <typename F>
static int allow_big_lds(F fn, size_t bytes) {
  if (bytes > 64 * 1024) {
    MFX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)bytes));
  }
  return MFX_OK;
}

}  // namespace mfx

using namespace mfx;

extern "C" {

int mfx_tridiag_eigh(const void* alpha, const void* beta, int64_t ldbeta, int64_t p, int64_t k, int dtype,
                     void* evals, void* evecs, void* stream) {
  MFX_REQUIRE(alpha && evals && evecs && (beta || k == 1), MFX_ERR_INVALID, "null argument");
  MFX_REQUIRE(p >= 1 && k >= 1, MFX_ERR_INVALID, "p, k must be positive");
  const bool deep = k > kSmallLdsDepth;
  MFX_REQUIRE(!deep || dtype == MFX_F64, MFX_ERR_UNSUPPORTED,
              "tridiagonal eigensolver: k > %d needs fp64 buffers (the rotations are accumulated in evecs itself; got k = %lld in fp32)",
              kSmallLdsDepth, (long long)k);
  MFX_REQUIRE(k <= kSmallMaxDepth, MFX_ERR_UNSUPPORTED, "tridiagonal eigensolver supports k <= %d (got %lld)", kSmallMaxDepth, (long long)k);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t sh = (size_t)(2 * k + (deep ? 0 : k * (k | 1))) * sizeof(double);
  if (dtype == MFX_F32) {
    MFX_TRY(allow_big_lds(k_tridiag_eigh<float, false>, sh));
    k_tridiag_eigh<float, false><<<(unsigned)p, 64, sh, s>>>((const float*)alpha, (const float*)beta, ldbeta, (int)k,
                                                             (float*)evals, (float*)evecs);
  } else if (dtype == MFX_F64 && deep) {
    k_tridiag_eigh<double, true><<<(unsigned)p, 64, sh, s>>>((const double*)alpha, (const double*)beta, ldbeta, (int)k,
                                                             (double*)evals, (double*)evecs);
  } else if (dtype == MFX_F64) {
    MFX_TRY(allow_big_lds(k_tridiag_eigh<double, false>, sh));
    k_tridiag_eigh<double, false><<<(unsigned)p, 64, sh, s>>>((const double*)alpha, (const double*)beta, ldbeta, (int)k,
                                                              (double*)evals, (double*)evecs);
  } else {
    set_error("unsupported dtype %d", dtype);
    return MFX_ERR_UNSUPPORTED;
  }
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

int mfx_slq_quadform_bwd(const void* evals, const void* evecs, const void* fvals, const void* dfvals,
                         const void* gout, int64_t p, int64_t k, int dtype, void* dalpha, void* dbeta,
                         int64_t lddbeta, void* stream) {
  MFX_REQUIRE(evals && evecs && fvals && dfvals && gout && dalpha && (dbeta || k == 1), MFX_ERR_INVALID, "null argument");
  MFX_REQUIRE(k >= 1 && k <= kSmallMaxDepth, MFX_ERR_UNSUPPORTED, "quadform backward supports k <= %d (got %lld)", kSmallMaxDepth, (long long)k);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool deep = k > kSmallLdsDepth;
  const size_t sh = (size_t)(deep ? 4 * k : k * k) * sizeof(double);
#define MFX_QF_LAUNCH(T, DEEP)                                                                                                 \
  k_quadform_bwd<T, DEEP><<<(unsigned)p, 64, sh, s>>>((const T*)evals, (const T*)evecs, (const T*)fvals, (const T*)dfvals,    \
                                                      (const T*)gout, (int)k, (T*)dalpha, (T*)dbeta, lddbeta)
  if (dtype == MFX_F32) {
    if (deep) {
      MFX_QF_LAUNCH(float, true);
    } else {
      MFX_TRY(allow_big_lds(k_quadform_bwd<float, false>, sh));
      MFX_QF_LAUNCH(float, false);
    }
  } else if (dtype == MFX_F64) {
    if (deep) {
      MFX_QF_LAUNCH(double, true);
    } else {
      MFX_TRY(allow_big_lds(k_quadform_bwd<double, false>, sh));
      MFX_QF_LAUNCH(double, false);
    }
  } else {
    set_error("unsupported dtype %d", dtype);
    return MFX_ERR_UNSUPPORTED;
  }
#undef MFX_QF_LAUNCH
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

int mfx_rademacher(uint64_t seed, int64_t first_probe, int64_t p, int64_t n, int dtype, void* out, void* stream) {
  MFX_REQUIRE(out && p >= 1 && n >= 1 && p <= 65535, MFX_ERR_INVALID, "bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int64_t gx = (n + 255) / 256;
  if (gx > 2048) gx = 2048;
  if (dtype == MFX_F32) {
    k_rademacher<float><<<dim3((unsigned)gx, (unsigned)p), 256, 0, s>>>(seed, first_probe, n, (float*)out);
  } else if (dtype == MFX_F64) {
    k_rademacher<double><<<dim3((unsigned)gx, (unsigned)p), 256, 0, s>>>(seed, first_probe, n, (double*)out);
  } else {
    set_error("unsupported dtype %d", dtype);
    return MFX_ERR_UNSUPPORTED;
  }
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

}  // extern "C"
