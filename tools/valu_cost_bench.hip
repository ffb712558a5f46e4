// Issue cost (cycles per instruction, one wave per SIMD, independent instructions) of the VALU instructions the Gram matvec's split chain
// is made of: v_exp_f32, v_cvt_pk_f16_f32, v_fma_mixlo_f16, v_cvt_scalef32_pk_fp8_f32, v_cvt_pk_fp8_f32, v_fma_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short shortx2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.001f * (threadIdx.x + i);
  unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) x[i] = __builtin_amdgcn_exp2f(x[i]);
      if (KIND == 1) { half2v h = {(_Float16)x[i], (_Float16)x[(i + 1) & 7]}; acc[i] ^= __builtin_bit_cast(unsigned, h); }
      if (KIND == 2) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]), "v"(x[i]));
      if (KIND == 3) { shortx2 p = __builtin_bit_cast(shortx2, acc[i]); p = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(p, x[i], x[(i + 1) & 7], 128.0f, false); acc[i] = __builtin_bit_cast(unsigned, p); }
      if (KIND == 4) acc[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[i], x[(i + 1) & 7], (int)acc[i], false);
      if (KIND == 5) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
    }
    if (KIND == 1 || KIND == 3 || KIND == 4) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + (float)acc[i];
  if (s == 1.2345f) out[0] = s;
}
template <int KIND>
void run(const char* name) {
  float* d; long long* c;
  hipMalloc(&d, 64); hipMalloc(&c, 256 * 8);
  const int iters = 200000;
  k<KIND><<<256, 256>>>(d, c, 1000);
  k<KIND><<<256, 256>>>(d, c, iters);
  long long h[256];
  hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; ++i) m += (double)h[i]; m /= 256;
  printf("%-28s %5.2f cycles per instruction\n", name, m / (iters * 8.0));
}
int main() {
  run<0>("v_exp_f32"); run<1>("v_cvt_pk_f16_f32 (+xor)"); run<2>("v_fma_mixlo_f16"); run<3>("v_cvt_scalef32_pk_fp8_f32"); run<4>("v_cvt_pk_fp8_f32"); run<5>("v_fma_f32");
  return 0;
}
