#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef short shortx2 __attribute__((ext_vector_type(2)));
// A (32 x 64) row-major floats, B (64 x 32) [k][n]; assumed layout: lane l: row / col = l % 32, k = 32 (l / 32) + v, byte v of the 32-byte operand
__global__ void k(const float* A, const float* B, float* D, float* cv) {
  const int l = threadIdx.x, rc = l & 31, g = l >> 5;
  intx8 a, b;
  for (int d = 0; d < 8; ++d) {
    shortx2 pa = {0, 0}, pb = {0, 0};
    // entries 4d .. 4d+3
    pa = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(pa, A[rc * 64 + 32 * g + 4 * d + 0], A[rc * 64 + 32 * g + 4 * d + 1], 128.0f, false);
    pa = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(pa, A[rc * 64 + 32 * g + 4 * d + 2], A[rc * 64 + 32 * g + 4 * d + 3], 128.0f, true);
    int wb = 0;
    wb = __builtin_amdgcn_cvt_pk_fp8_f32(32.f * B[(32 * g + 4 * d + 0) * 32 + rc], 32.f * B[(32 * g + 4 * d + 1) * 32 + rc], wb, false);
    wb = __builtin_amdgcn_cvt_pk_fp8_f32(32.f * B[(32 * g + 4 * d + 2) * 32 + rc], 32.f * B[(32 * g + 4 * d + 3) * 32 + rc], wb, true);
    a[d] = __builtin_bit_cast(int, pa);
    b[d] = wb;
  }
  if (l == 0) { cv[0] = __builtin_bit_cast(float, a[0]); cv[1] = __builtin_bit_cast(float, b[0]); }
  floatx16 c = {0};
  // scale_a = 2^7 (E8M0 134), scale_b = 2^-5 (122), byte 0
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x86868686, 0, 0x7a7a7a7a);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + rc] = c[r];
}
int main() {
  float hA[32 * 64], hB[64 * 32], hD[1024], hc[2];
  srand(1);
  for (int i = 0; i < 2048; ++i) { hA[i] = (rand() % 16) * 1024.f; hB[i] = ((rand() % 15) - 7) * 0.5f; }   // exactly representable in e4m3 after scaling
  hA[0] = 3.f * 4096.f; hA[1] = 1024.f; hA[2] = 2048.f; hA[3] = 5 * 1024.f;
  float *A, *B, *D, *cv;
  hipMalloc(&A, sizeof(hA)); hipMalloc(&B, sizeof(hB)); hipMalloc(&D, sizeof(hD)); hipMalloc(&cv, 8);
  hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(B, hB, sizeof(hB), hipMemcpyHostToDevice);
  k<<<1, 64>>>(A, B, D, cv);
  hipMemcpy(hD, D, sizeof(hD), hipMemcpyDeviceToHost); hipMemcpy(hc, cv, 8, hipMemcpyDeviceToHost);
  unsigned ua, ub; memcpy(&ua, &hc[0], 4); memcpy(&ub, &hc[1], 4);
  printf("a dword0 %08x (A[0..3] = %g %g %g %g / 128), b dword0 %08x\n", ua, hA[0], hA[1], hA[2], hA[3], ub);
  double maxerr = 0, maxref = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double ref = 0; for (int kk = 0; kk < 64; ++kk) ref += (double)hA[i * 64 + kk] * hB[kk * 32 + j];
    maxerr = fmax(maxerr, fabs(ref - hD[i * 32 + j])); maxref = fmax(maxref, fabs(ref));
  }
  printf("max |D - A B| = %g (max |A B| = %g); D[0][0] = %g\n", maxerr, maxref, hD[0]);
  return 0;
}
