// Statistical accuracy of v_mfma_f32_32x32x16_f16 chains on random data (gfx950): signed relative error of
// D = sum_k A B (+ C chain of NCHAIN MFMAs) against an fp64 host reference on the same f16 operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// A: [chain][32 rows][16 k], B: [chain][16 k][32 cols] in f16; out D[32][32]
__global__ void k(const _Float16* A, const _Float16* B, float* D, int nchain) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  floatx16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  for (int t = 0; t < nchain; ++t) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = A[(t * 32 + r) * 16 + 8 * h + j];
      b[j] = B[(t * 16 + 8 * h + j) * 32 + r];
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    D[row * 32 + r] = c[i];
  }
}

int main() {
  for (int mode = 0; mode < 3; ++mode) {     // 0: random sign both, 1: A >= 0 (kernel-like), 2: both >= 0
    for (int nchain : {1, 16, 256, 4096}) {
      std::vector<_Float16> A((size_t)nchain * 32 * 16), B((size_t)nchain * 16 * 32);
      srand(1);
      auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
      for (auto& v : A) { float x = rnd() * 1000.f; if (mode >= 1) x = fabsf(x); v = (_Float16)x; }
      for (auto& v : B) { float x = rnd() * 1000.f; if (mode >= 2) x = fabsf(x); v = (_Float16)x; }
      _Float16 *dA, *dB; float* dD;
      hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, 32 * 32 * 4);
      hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
      k<<<1, 64>>>(dA, dB, dD, nchain);
      std::vector<float> D(1024);
      hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
      double sum_signed = 0, sum_abs = 0, sum_ref = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double ref = 0, mag = 0;
          for (int t = 0; t < nchain; ++t)
            for (int kk = 0; kk < 16; ++kk) {
              const double p = (double)(float)A[((size_t)t * 32 + i) * 16 + kk] * (double)(float)B[((size_t)t * 16 + kk) * 32 + j];
              ref += p; mag += fabs(p);
            }
          sum_signed += (D[i * 32 + j] - ref) * (ref >= 0 ? 1 : -1);  // > 0: magnitude over-estimated
          sum_abs += fabs(D[i * 32 + j] - ref);
          sum_ref += fabs(ref);
        }
      printf("mode %d chain %5d: mean |err| / mean |ref| = %.3e   signed (towards larger magnitude) = %+.3e\n", mode, nchain,
             sum_abs / sum_ref, sum_signed / sum_ref);
      hipFree(dA); hipFree(dB); hipFree(dD);
    }
  }
  return 0;
}
