// Can an MFMA-only wave and a VALU-only wave that share a SIMD run concurrently on gfx950?
// 512-thread workgroups: waves 0-3 and waves 4-7 land on the same 4 SIMDs.  mode 0: all waves MFMA;
// 1: all waves VALU; 2: waves 0-3 MFMA, waves 4-7 VALU (same per-wave work as modes 0/1).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float* out, int iters, float seed, int mode) {
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_mfma = mode == 0 || (mode == 2 && wid < 4);
  const bool do_valu = mode == 1 || (mode == 2 && wid >= 4);
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + j); b[j] = (_Float16)(seed - j); }
  floatx16 c[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
  float v[8];
  for (int q = 0; q < 8; ++q) v[q] = seed * (q + 1);
  if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) c[q % 4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[q % 4], 0, 0, 0);
    }
  }
  if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 64; ++q) v[q % 8] = fmaf(v[q % 8], 1.0001f, 0.5f);
    }
  }
  float s = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += c[q][r];
  for (int q = 0; q < 8; ++q) s += v[q];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float* d; hipMalloc(&d, 64);
  const int iters = 20000;
  const char* names[] = {"8 waves: all MFMA (8 f16 mfma/iter)", "8 waves: all VALU (64 fma/iter)", "4 waves MFMA + 4 waves VALU"};
  for (int mode = 0; mode < 3; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, 512>>>(d, 100, 1.f, mode);
    hipEventRecord(e0);
    k<<<256, 512>>>(d, iters, 1.f, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s wall %.3f ms -> %.1f cycles/iter @2.4GHz\n", names[mode], ms, ms * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
