// Cycles per MFMA instruction in a bare loop of independent accumulators, one wave per SIMD (4 waves per CU, 256 CUs):
// v_mfma_f32_32x32x16_f16 against the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 and with e2m3 (FP6) operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256, 1) void k(const int* in, float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x;
  intx8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = in[lane * 8 + i] & (int)0xBFBFBFBF;
    b[i] = in[4096 + lane * 8 + i] & (int)0xBFBFBFBF;
  }
  half8 ha = __builtin_bit_cast(half8, __builtin_shufflevector(a, a, 0, 1, 2, 3));
  half8 hb = __builtin_bit_cast(half8, __builtin_shufflevector(b, b, 0, 1, 2, 3));
  floatx16 acc[4];
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 3) {  // mixed stream: 3 f16 + 1 FP8, independent accumulators (expected 3 x 32 + 64 = 160 cycles per group of 4)
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[3], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      continue;
    }
    if (KIND == 4) {  // the FP8 MFMA accumulating into a register the f16 MFMA before it wrote (as a block of the matvec would)
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[2], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[2], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      continue;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (KIND == 0) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[q], 0, 0, 0);
      if (KIND == 1) acc[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[q], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      if (KIND == 2) {  // e2m3 (FP6): cbsz = blgp = 2, 6 dwords per operand
        acc[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[q], 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);  // reads 6 of the 8 dwords
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  float s = 0;
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  if (s == 1234.5f) out[0] = s;
}

template <int KIND>
void run(const char* name, const int* in) {
  float* d;
  long long* cyc;
  hipMalloc(&d, 64);
  hipMalloc(&cyc, 256 * 8);
  const int iters = 400000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<KIND><<<256, 256>>>(in, d, cyc, 1000);
  hipEventRecord(e0);
  k<KIND><<<256, 256>>>(in, d, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < 256; ++i) mean += (double)h[i];
  mean /= 256;
  printf("%-44s %6.1f cycles per MFMA, %.2f GHz, %6.1f ns per MFMA\n", name, mean / (iters * 4.0), mean / (ms * 1e-3) * 1e-9, ms * 1e6 / (iters * 4.0));
}

int main() {
  int h[8192];
  srand(5);
  for (int i = 0; i < 8192; ++i) h[i] = rand() * 65537 + rand();
  int* in;
  hipMalloc(&in, sizeof(h));
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("v_mfma_f32_32x32x16_f16 (k = 16)", in);
    run<1>("v_mfma_scale_f32_32x32x64_f8f6f4 e4m3 (k = 64)", in);
    run<2>("v_mfma_scale_f32_32x32x64_f8f6f4 e2m3 (k = 64)", in);
    run<3>("3 x f16 + 1 x e4m3, independent (x4)", in);
    run<4>("3 x f16 + 1 x e4m3 into the last f16's acc (x4)", in);
  }
  return 0;
}
