// Stand-alone consumer of the C-ABI (include/mfx.h): no Python, no torch -- what a maintainer binding libmfx.so from another
// host language would write.  Dense symmetric operator, Arnoldi forward (arnoldi.py:57-101) + adjoint (:104-220) through the
// exported entry points; checks the decomposition identities on the host.  Built and run by tests/test_gpu_parity.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mfx.h"

#define HIP_OK(e)                                                             \
  do {                                                                        \
    hipError_t _e = (e);                                                      \
    if (_e != hipSuccess) {                                                   \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e));           \
      return 2;                                                               \
    }                                                                         \
  } while (0)
#define MFX_OK_(e)                                                            \
  do {                                                                        \
    int _rc = (e);                                                            \
    if (_rc != 0) {                                                           \
      std::fprintf(stderr, "%s -> %d: %s\n", #e, _rc, mfx_last_error());     \
      return 3;                                                               \
    }                                                                         \
  } while (0)

int main() {
  const int64_t n = 96, k = 7, p = 3;
  std::vector<double> A(n * n), V(p * n);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1u << 24) - 0.5; };
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j <= i; ++j) {
      const double v = rnd() / n + (i == j ? 1.0 + 0.05 * i : 0.0);
      A[i * n + j] = v;
      A[j * n + i] = v;
    }
  for (auto& v : V) v = rnd();

  double *dA, *dV, *dQ, *dH, *dr, *dc, *ddv, *dLam, *dgA, *ddQ, *ddH;
  HIP_OK(hipMalloc(&dA, sizeof(double) * n * n));
  HIP_OK(hipMalloc(&dV, sizeof(double) * p * n));
  HIP_OK(hipMalloc(&dQ, sizeof(double) * p * k * n));
  HIP_OK(hipMalloc(&dH, sizeof(double) * p * k * k));
  HIP_OK(hipMalloc(&dr, sizeof(double) * p * n));
  HIP_OK(hipMalloc(&dc, sizeof(double) * p));
  HIP_OK(hipMalloc(&ddv, sizeof(double) * p * n));
  HIP_OK(hipMalloc(&dLam, sizeof(double) * p * k * n));
  HIP_OK(hipMalloc(&dgA, sizeof(double) * n * n));
  HIP_OK(hipMalloc(&ddQ, sizeof(double) * p * k * n));
  HIP_OK(hipMalloc(&ddH, sizeof(double) * p * k * k));
  HIP_OK(hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dV, V.data(), sizeof(double) * p * n, hipMemcpyHostToDevice));
  HIP_OK(hipMemset(dgA, 0, sizeof(double) * n * n));
  HIP_OK(hipMemset(ddQ, 0, sizeof(double) * p * k * n));

  mfx_operator op = {};
  op.kind = MFX_OP_DENSE;
  op.dtype = MFX_F64;
  op.n = n;
  op.dense_a = dA;
  op.lda = n;
  const int64_t ws_bytes = mfx_workspace_bytes(&op, n, k, p);
  if (ws_bytes <= 0) return 4;
  void* ws;
  HIP_OK(hipMalloc(&ws, ws_bytes));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));

  MFX_OK_(mfx_arnoldi_forward(&op, dV, n, k, p, /*second_pass=*/1, dQ, dH, dr, dc, ws, ws_bytes, stream));
  HIP_OK(hipStreamSynchronize(stream));
  std::vector<double> Q(p * k * n), H(p * k * k), r(p * n), c(p);
  HIP_OK(hipMemcpy(Q.data(), dQ, sizeof(double) * Q.size(), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(H.data(), dH, sizeof(double) * H.size(), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(r.data(), dr, sizeof(double) * r.size(), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(c.data(), dc, sizeof(double) * c.size(), hipMemcpyDeviceToHost));

  double worst = 0.0;
  for (int64_t b = 0; b < p; ++b) {
    const double* Qb = &Q[b * k * n];
    const double* Hb = &H[b * k * k];
    // orthonormal basis, first vector = c v  (arnoldi.py:66-70)
    for (int64_t i = 0; i < k; ++i)
      for (int64_t j = 0; j < k; ++j) {
        double d = 0.0;
        for (int64_t l = 0; l < n; ++l) d += Qb[i * n + l] * Qb[j * n + l];
        worst = std::fmax(worst, std::fabs(d - (i == j ? 1.0 : 0.0)));
      }
    for (int64_t l = 0; l < n; ++l) worst = std::fmax(worst, std::fabs(Qb[l] - c[b] * V[b * n + l]));
    // A Q = Q H + r e_k^T with Q stored (k, n):  (A q_i)_l = sum_j H[j][i] q_j[l] + [i == k-1] r_l
    for (int64_t i = 0; i < k; ++i)
      for (int64_t l = 0; l < n; ++l) {
        double lhs = 0.0, rhs = (i == k - 1) ? r[b * n + l] : 0.0;
        for (int64_t m = 0; m < n; ++m) lhs += A[l * n + m] * Qb[i * n + m];
        for (int64_t j = 0; j < k; ++j) rhs += Hb[j * k + i] * Qb[j * n + l];
        worst = std::fmax(worst, std::fabs(lhs - rhs));
      }
  }
  std::printf("forward identities: max deviation %.3e\n", worst);
  if (!(worst < 1e-10)) return 5;

  // adjoint with cotangent dH = I (d trace(H)): finite and, for this symmetric operator, a symmetric-ish parameter gradient
  std::vector<double> dHh(p * k * k, 0.0);
  for (int64_t b = 0; b < p; ++b)
    for (int64_t i = 0; i < k; ++i) dHh[b * k * k + i * k + i] = 1.0;
  HIP_OK(hipMemcpy(ddH, dHh.data(), sizeof(double) * dHh.size(), hipMemcpyHostToDevice));
  mfx_op_grads grads = {};
  grads.dense_a = dgA;
  MFX_OK_(mfx_arnoldi_adjoint(&op, n, k, p, dQ, dH, dr, dc, /*dQ=*/nullptr, ddH, /*dr=*/nullptr, /*dc=*/nullptr, MFX_REORTHO_FULL,
                              ddv, dLam, &grads, ws, ws_bytes, stream));
  HIP_OK(hipStreamSynchronize(stream));
  std::vector<double> gA(n * n), dv(p * n);
  HIP_OK(hipMemcpy(gA.data(), dgA, sizeof(double) * gA.size(), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(dv.data(), ddv, sizeof(double) * dv.size(), hipMemcpyDeviceToHost));
  // d trace(H)/dA = sum_i q_i q_i^T (trace(Q^T A Q) with orthonormal Q held fixed is the leading term): compare the trace
  double tr = 0.0, want = 0.0, nonfinite = 0.0;
  for (int64_t i = 0; i < n; ++i) tr += gA[i * n + i];
  for (double v : gA) nonfinite += std::isfinite(v) ? 0.0 : 1.0;
  for (double v : dv) nonfinite += std::isfinite(v) ? 0.0 : 1.0;
  want = (double)(p * k);  // trace(sum_b Q_b^T Q_b) = p k
  std::printf("adjoint: trace(dA) = %.12f (p k = %.1f), non-finite entries %.0f\n", tr, want, nonfinite);
  if (nonfinite != 0.0 || std::fabs(tr - want) > 1e-8) return 6;
  std::printf("cabi smoke ok (libmfx version %d)\n", mfx_version());
  return 0;
}
