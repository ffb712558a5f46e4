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
template <typename T>
__global__ __launch_bounds__(64) void k_tridiag_eigh(const T* __restrict__ alpha, const T* __restrict__ beta,
                                                     int64_t ldbeta, int k, T* __restrict__ evals,
                                                     T* __restrict__ evecs) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* d = reinterpret_cast<double*>(smem_raw);
  double* e = d + k;
  double* Z = e + k;
  const int ldz = k | 1;  // odd leading dimension: conflict-free column access
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  for (int i = lane; i < k; i += 64) {
    d[i] = (double)alpha[b * k + i];
    e[i] = (i < k - 1) ? (double)beta[b * ldbeta + i] : 0.0;
  }
  for (int t = lane; t < k * k; t += 64) Z[(t / k) * ldz + (t % k)] = (t / k == t % k) ? 1.0 : 0.0;
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
  for (int t = lane; t < k * k; t += 64) evecs[b * k * k + t] = (T)Z[(t / k) * ldz + (t % k)];
}

// G = gout * U (F o u0 u0^T) U^T with F the divided differences of f at the eigenvalues;
// dalpha_i = G_ii, dbeta_i = 2 G_{i,i+1}.
template <typename T>
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
  for (int t = lane; t < k * k; t += 64) {
    const int a = t / k, c = t % k;
    const double dl = (double)lam[a] - (double)lam[c];
    double F;
    if (a == c || fabs(dl) <= tol) {
      F = 0.5 * ((double)df[a] + (double)df[c]);
    } else {
      F = ((double)f[a] - (double)f[c]) / dl;
    }
    M[t] = F * (double)U[a] * (double)U[c];  // U[0][a] U[0][c]
  }
  __syncthreads();
  const double go = (double)gout[b];
  for (int i = lane; i < k; i += 64) {
    double gii = 0.0, gi1 = 0.0;
    for (int a = 0; a < k; ++a) {
      double t = 0.0;
      for (int c = 0; c < k; ++c) t += M[a * k + c] * (double)U[i * k + c];
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
  MFX_REQUIRE(k <= 120, MFX_ERR_UNSUPPORTED, "tridiagonal eigensolver supports k <= 120 (got %lld)", (long long)k);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t sh = (size_t)(2 * k + k * (k | 1)) * sizeof(double);
  if (dtype == MFX_F32) {
    MFX_TRY(allow_big_lds(k_tridiag_eigh<float>, sh));
    k_tridiag_eigh<float><<<(unsigned)p, 64, sh, s>>>((const float*)alpha, (const float*)beta, ldbeta, (int)k,
                                                      (float*)evals, (float*)evecs);
  } else if (dtype == MFX_F64) {
    MFX_TRY(allow_big_lds(k_tridiag_eigh<double>, sh));
    k_tridiag_eigh<double><<<(unsigned)p, 64, sh, s>>>((const double*)alpha, (const double*)beta, ldbeta, (int)k,
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
  MFX_REQUIRE(k >= 1 && k <= 120, MFX_ERR_UNSUPPORTED, "quadform backward supports k <= 120");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t sh = (size_t)k * k * sizeof(double);
  if (dtype == MFX_F32) {
    MFX_TRY(allow_big_lds(k_quadform_bwd<float>, sh));
    k_quadform_bwd<float><<<(unsigned)p, 64, sh, s>>>((const float*)evals, (const float*)evecs, (const float*)fvals,
                                                      (const float*)dfvals, (const float*)gout, (int)k,
                                                      (float*)dalpha, (float*)dbeta, lddbeta);
  } else if (dtype == MFX_F64) {
    MFX_TRY(allow_big_lds(k_quadform_bwd<double>, sh));
    k_quadform_bwd<double><<<(unsigned)p, 64, sh, s>>>((const double*)evals, (const double*)evecs,
                                                       (const double*)fvals, (const double*)dfvals,
                                                       (const double*)gout, (int)k, (double*)dalpha, (double*)dbeta,
                                                       lddbeta);
  } else {
    set_error("unsupported dtype %d", dtype);
    return MFX_ERR_UNSUPPORTED;
  }
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
