// How does v_mfma_f32_32x32x16_f16 round when small addends are aligned to a large one?  Sign matrix (gfx950).
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f16_trunc.hip -o tools/mfma_f16_trunc.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
// product k = 0 is big (sign sb), products k = 1..nsmall are small = ss * 2^-e (others zero); accumulator input c0
__global__ void k(float* out, float big, float small_a, float small_b, int nsmall, float c0) {
  const int lane = threadIdx.x;
  half8 a, b;
  for (int j = 0; j < 8; ++j) {
    const int kk = 8 * (lane >> 5) + j;
    a[j] = (_Float16)(kk == 0 ? big : (kk <= nsmall ? small_a : 0.f));
    b[j] = (_Float16)(kk == 0 ? 1.f : (kk <= nsmall ? small_b : 0.f));
  }
  floatx16 c;
  for (int r = 0; r < 16; ++r) c[r] = c0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (lane == 0) out[0] = c[0];
}
// chain of MFMAs on random data: A: [chain][32][16], B: [chain][16][32]
__global__ void kchain(const _Float16* A, const _Float16* B, float* D, int nchain, float sgn) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  floatx16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  for (int t = 0; t < nchain; ++t) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (_Float16)(sgn * (float)A[(t * 32 + r) * 16 + 8 * h + j]);
      b[j] = B[(t * 16 + 8 * h + j) * 32 + r];
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}
#include <vector>
#include <stdlib.h>
static void chains() {
  printf("\nchains of MFMAs, all-positive operands: D+ from (A, B), D- from (-A, B), fp64 reference on the same f16 operands\n");
  for (int nchain : {16, 256, 4096, 24576}) {
    std::vector<_Float16> A((size_t)nchain * 32 * 16), B((size_t)nchain * 16 * 32);
    srand(1);
    for (auto& v : A) v = (_Float16)((float)rand() / RAND_MAX * 1000.f);
    for (auto& v : B) v = (_Float16)((float)rand() / RAND_MAX * 1000.f);
    _Float16 *dA, *dB; float *dP, *dM;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dP, 4096); hipMalloc(&dM, 4096);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    kchain<<<1, 64>>>(dA, dB, dP, nchain, 1.f);
    kchain<<<1, 64>>>(dA, dB, dM, nchain, -1.f);
    std::vector<float> P(1024), M(1024);
    hipMemcpy(P.data(), dP, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(M.data(), dM, 4096, hipMemcpyDeviceToHost);
    double ep = 0, em = 0, sym = 0, ref_sum = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int t = 0; t < nchain; ++t)
          for (int kk = 0; kk < 16; ++kk)
            ref += (double)(float)A[((size_t)t * 32 + i) * 16 + kk] * (double)(float)B[((size_t)t * 16 + kk) * 32 + j];
        ep += P[i * 32 + j] - ref; em += M[i * 32 + j] + ref; sym += (double)P[i * 32 + j] + (double)M[i * 32 + j]; ref_sum += ref;
      }
    printf("chain %6d: mean(D+ - ref)/mean ref = %+.3e   mean(D- + ref)/mean ref = %+.3e   mean(D+ + D-)/mean ref = %+.3e\n", nchain,
           ep / ref_sum, em / ref_sum, sym / ref_sum);
    hipFree(dA); hipFree(dB); hipFree(dP); hipFree(dM);
  }
}
int main() {
  chains();
  float* d; hipMalloc(&d, 4);
  const float ulp = 1.1920928955078125e-7f;
  printf("%-58s %14s %10s\n", "case", "got - exact [ulp(1)]", "got");
  for (int e : {13, 14, 15}) {            // small product = 2^-12 * 2^-e : 2^-25, 2^-26, 2^-27
    for (int ns : {1, 3, 7, 8, 15}) {
      for (int sb = 1; sb >= -1; sb -= 2)
        for (int ss = 1; ss >= -1; ss -= 2)
          for (int cmode = 0; cmode < 2; ++cmode) {   // big term as a product (C = 0) or as the accumulator input
            const float sa = ss * ldexpf(1.f, -12), sbv = ldexpf(1.f, -e);
            const float big = cmode == 0 ? (float)sb : 0.f, c0 = cmode == 0 ? 0.f : (float)sb;
            k<<<1, 64>>>(d, big, sa, sbv, ns, c0);
            float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
            const double exact = (double)sb + ns * (double)ss * ldexp(1.0, -12 - e);
            char name[128];
            snprintf(name, sizeof name, "big %+d (%s) + %2d x %+d*2^-%d", sb, cmode ? "C input" : "product", ns, ss, 12 + e);
            printf("%-58s %+14.4f %+.9f\n", name, (h - exact) / ulp, h);
          }
    }
  }
  return 0;
}
