// Which gfx950 execution resources overlap inside one wave / across two waves of a SIMD?
// Each variant runs ITER iterations of: NF f16 MFMAs, NS fp32 MFMAs, NV VALU fmas, NE v_exp (independent chains),
// in an interleaved program order.  Reports cycles per iteration per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NF, int NS, int NV, int NE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + j); b[j] = (_Float16)(seed - j); }
  floatx16 cf[4], cs[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) { cf[q][r] = 0.f; cs[q][r] = 0.f; }
  float v[8];
  for (int q = 0; q < 8; ++q) v[q] = seed * (q + 1);
  float e[4] = {seed, seed + 1, seed + 2, seed + 3};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int q = 0; q < NF; ++q) if (q % 4 == u || NF <= 4 && q == u) cf[q % 4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, cf[q % 4], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NS; ++q) if (q % 4 == u) cs[q % 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], v[1], cs[q % 4], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) if (q % 4 == u) v[2 + q % 6] = fmaf(v[2 + q % 6], 1.0001f, 0.5f);
#pragma unroll
      for (int q = 0; q < NE; ++q) if (q % 4 == u) e[q % 4] = __builtin_amdgcn_exp2f(e[q % 4]) * 0.5f;
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += cf[q][r] + cs[q][r];
  for (int q = 0; q < 8; ++q) s += v[q];
  for (int q = 0; q < 4; ++q) s += e[q];
  if (s == 12345.678f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0) / iters;
}

template <int NF, int NS, int NV, int NE>
void run(const char* name, float* d, int waves_per_simd) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NF, NS, NV, NE><<<256 * waves_per_simd, 256>>>(d, 100, 1.f);
  hipEventRecord(e0);
  k<NF, NS, NV, NE><<<256 * waves_per_simd, 256>>>(d, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("%-34s waves/SIMD %d : %8.1f wave-clock cycles/iter (s_memtime), wall %.3f ms -> %.1f cyc/iter @2.4GHz\n", name, waves_per_simd, h[1], ms,
         ms * 1e-3 * 2.4e9 / iters);
}

int main() {
  float* d; hipMalloc(&d, 64);
  for (int w = 1; w <= 2; ++w) {
    run<8, 0, 0, 0>("8 f16 MFMA", d, w);
    run<0, 4, 0, 0>("4 fp32 MFMA", d, w);
    run<0, 0, 64, 0>("64 v_fma", d, w);
    run<0, 0, 0, 32>("32 v_exp(+mul)", d, w);
    run<8, 0, 64, 0>("8 f16 MFMA + 64 v_fma", d, w);
    run<0, 4, 64, 0>("4 fp32 MFMA + 64 v_fma", d, w);
    run<8, 4, 0, 0>("8 f16 MFMA + 4 fp32 MFMA", d, w);
    run<8, 4, 64, 0>("8 f16 + 4 fp32 MFMA + 64 v_fma", d, w);
    run<8, 0, 32, 16>("8 f16 MFMA + 32 fma + 16 exp", d, w);
  }
  return 0;
}
