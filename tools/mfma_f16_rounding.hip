// Probe how v_mfma_f32_32x32x16_f16 rounds its internal sum (gfx950).  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float small_a, float small_b, float c0) {
  const int lane = threadIdx.x;
  // A[row][k]: row = lane & 31, k = 8*(lane>>5)+j ; B[k][col] col = lane & 31
  half8 a, b;
  for (int j = 0; j < 8; ++j) {
    const int kk = 8 * (lane >> 5) + j;
    a[j] = (_Float16)(kk == 0 ? 1.f : small_a);
    b[j] = (_Float16)(kk == 0 ? 1.f : small_b);
  }
  floatx16 c;
  for (int r = 0; r < 16; ++r) c[r] = c0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (lane == 0) out[0] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  struct { float a, b, c; const char* what; } cases[] = {
    {0.0001220703125f /*2^-13*/, 0.000244140625f /*2^-12*/, 0.f, "1 + 15*2^-25, C=0  (exact 1+3.75ulp)"},
    {0.0001220703125f, 0.000244140625f, 1.f, "1 + 15*2^-25 + C=1 (exact 2+1.875ulp(2))"},
    {0.00048828125f /*2^-11*/, 0.000244140625f /*2^-12*/, 0.f, "1 + 15*2^-23, C=0 (exact 1+15ulp)"},
    {0.0001220703125f, 0.0001220703125f /*2^-13*/, 0.f, "1 + 15*2^-26, C=0 (exact 1+1.875ulp)"},
  };
  for (auto& cs : cases) {
    k<<<1, 64>>>(d, cs.a, cs.b, cs.c);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    double exact = 1.0 + 15.0 * (double)cs.a * (double)cs.b + cs.c;
    printf("%-45s got %.10f (%a) exact %.10f diff_ulp(2^-23) %.3f\n", cs.what, h, h, exact, (h - exact) / 1.1920928955078125e-7);
  }
  return 0;
}
