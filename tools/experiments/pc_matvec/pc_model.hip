// The producer / consumer Gram matvec kernel ALONE, on synthetic operands, with per-wave cycle stamps: how many shader cycles
// does a producer / a consumer wave spend per stage (32 columns)?  Build with the kernel's A/B macros, e.g.
//   tools/experiments/pc_matvec/build.sh base ""      tools/experiments/pc_matvec/build.sh d1 "-DMFX_PC_DIAG=1"   ...
// Results are NOT checked here (tests/test_gpu_pc_matvec.py does that through libmfx); this is a timing harness.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "mfx_rbf_pc.hip"

namespace mfx {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
}
}  // namespace mfx

int main(int argc, char** argv) {
  using namespace mfx;
  const int64_t n = argc > 1 ? atoll(argv[1]) : 131072, p = 64;
  constexpr int DPAD = 8;
  using S = PcSmem<DPAD>;
  const int64_t ntile = (n + 63) / 64;
  const size_t pkv_bytes = (size_t)ntile * S::kVBytes, pka_bytes = (size_t)ntile * S::kABytes;
  std::vector<_Float16> hv(pkv_bytes / 2), ha(pka_bytes / 2);
  srand(1);
  for (auto& v : hv) v = (_Float16)((rand() % 2001 - 1000) * 8.0f);         // like scaled probe values (|max| ~ 2^13)
  for (auto& v : ha) v = (_Float16)((rand() % 2001 - 1000) * 1e-3f);        // distances come out in [-30, 30]: exp2 in range
  std::vector<float> hx(n * DPAD), hsq(n), hvs(2 * p + 8, 1.f), hin(p * n, 1.f);
  for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hsq) v = (rand() % 1000) * 1e-3f;
  float *xs, *sq, *vs, *x, *y, *part, *os;
  void *pkv, *pka;
  hipMalloc(&xs, hx.size() * 4); hipMalloc(&sq, hsq.size() * 4); hipMalloc(&vs, hvs.size() * 4);
  hipMalloc(&x, hin.size() * 4); hipMalloc(&y, hin.size() * 4); hipMalloc(&part, 1 << 22); hipMalloc(&os, 16);
  hipMalloc(&pkv, pkv_bytes); hipMalloc(&pka, pka_bytes);
  hipMemcpy(xs, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(sq, hsq.data(), hsq.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(vs, hvs.data(), hvs.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(x, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(pkv, hv.data(), pkv_bytes, hipMemcpyHostToDevice);
  hipMemcpy(pka, ha.data(), pka_bytes, hipMemcpyHostToDevice);
  const float one[2] = {1.f, 0.1f};
  hipMemcpy(os, one, 8, hipMemcpyHostToDevice);
  const dim3 grid((unsigned)((n + 255) / 256), 1, 1);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(e0);
    int rc = rbf_pc_launch(DPAD, MFX_KERNEL_RBF, true, grid, 0, xs, sq, n, os, os + 1, vs, x, n, y, n, p, pkv, pka, part, nullptr, n, 0, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    if (rc != 0 || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it > 0 && ms < best) best = ms;
  }
  std::vector<long long> st(grid.x * 8);
  hipMemcpy(st.data(), part, st.size() * 8, hipMemcpyDeviceToHost);
  double cp = 0, cc = 0;
  const bool swap = MFX_PC_SWAP != 0;
  for (unsigned b = 0; b < grid.x; ++b)
    for (int w = 0; w < 8; ++w) ((w < 4) != swap ? cp : cc) += (double)st[b * 8 + w];
  const double stages = 2.0 * ntile, waves = 4.0 * grid.x;
  printf("n=%lld: %.3f ms per launch; cycles per stage: producer loop %.1f, consumer loop %.1f; implied clock %.2f GHz (2 workgroups per CU in sequence)\n",
         (long long)n, best, cp / waves / stages, cc / waves / stages, (cc / waves) * (grid.x / 256.0) / (best * 1e-3) * 1e-9);
  return 0;
}
