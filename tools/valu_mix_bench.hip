// One wave per SIMD (256-thread workgroups, 512 registers): how many cycles does a block of 12 (+2) f16 MFMAs cost with the
// matvec's VALU mix (16 v_exp_f32, 16 v_cvt_pk_f16_f32, 16 v_fma_mix_f32 per block) issued between them, by placement?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// MODE 0: MFMA only.  1: + exps (independent).  2: + exps + 32 fma (independent).  3: the real dependent chain, four-step pipeline
// over the pairs (as in the kernel).  4: the real chain, pair by pair (exp exp cvt fma fma cvt) behind each MFMA.
// 5: as 3 with the exps of a pair split over two gaps (at most one 8-cycle instruction per gap)
template <int MODE, int NMFMA>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int blocks, float seed) {
  const int lane = threadIdx.x & 63;
  half8 a[4], b[4];
  for (int q = 0; q < 4; ++q)
    for (int j = 0; j < 8; ++j) { a[q][j] = (_Float16)(seed + j + q); b[q][j] = (_Float16)(seed - j - q); }
  floatx16 c[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
  float w[16];
  for (int q = 0; q < 16; ++q) w[q] = seed * 1e-3f * (q + 1) + lane * 1e-4f;
  half2v hh[8], ll[8];
  for (int q = 0; q < 8; ++q) { hh[q] = half2v{(_Float16)0, (_Float16)0}; ll[q] = hh[q]; }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < blocks; ++it) {
#pragma unroll
    for (int m = 0; m < NMFMA; ++m) {
      __builtin_amdgcn_sched_barrier(0);
      c[(m / 3) % 4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m % 4], b[(m / 2) % 4], c[(m / 3) % 4], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (e * 12 / 16 == m) w[e] = __builtin_amdgcn_exp2f(w[e]);
      }
      if (MODE == 2) {
#pragma unroll
        for (int e = 0; e < 32; ++e)
          if (e * 12 / 32 == m) w[e % 16] = fmaf(w[e % 16], 0.999f, 1e-3f);
      }
      if (MODE == 3 || MODE == 4 || MODE == 5) {
        constexpr int gap3[8] = {0, 1, 2, 3, 5, 6, 7, 8};
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            const int r0 = 2 * p;
            bool now;
            if (MODE == 3) now = gap3[p] + st == m;
            else if (MODE == 4) now = (p * 12 / 8 == m);
            else now = false;
            if (MODE == 5) {  // steps: 0a exp(w0) at g, 0b exp(w1) at g+1, 1 at g+2, 2 at g+3, 3 at g+4; g = p (p < 8), wraps to the next block
              const int g = p;
              if (st == 0) {
                if (g == m) w[r0] = __builtin_amdgcn_exp2f(w[r0]);
                if (g + 1 == m) w[r0 + 1] = __builtin_amdgcn_exp2f(w[r0 + 1]);
                continue;
              }
              now = (g + 1 + st) % 12 == m;
            }
            if (!now) continue;
            if (st == 0) { w[r0] = __builtin_amdgcn_exp2f(w[r0]); w[r0 + 1] = __builtin_amdgcn_exp2f(w[r0 + 1]); }
            else if (st == 1) hh[p] = half2v{(_Float16)w[r0], (_Float16)w[r0 + 1]};
            else if (st == 2) {
              const unsigned hb = __builtin_bit_cast(unsigned, hh[p]);
              float l0, l1;
              asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hb), "v"(w[r0]));
              asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hb), "v"(w[r0 + 1]));
              w[r0] = l0 * 1e3f; w[r0 + 1] = l1 * 1e3f;   // (keeps the values in range; two extra multiplies per pair)
            } else ll[p] = half2v{(_Float16)w[r0], (_Float16)w[r0 + 1]};
          }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && threadIdx.x < 64) cyc[blockIdx.x] = t1 - t0;
  float s = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += c[q][r];
  for (int q = 0; q < 16; ++q) s += w[q];
  for (int q = 0; q < 8; ++q) s += (float)hh[q][0] + (float)ll[q][1];
  if (s == 12345.678f) out[0] = s;
}

// MODE 6 / 7: table-driven CYCLIC schedule over the NMFMA gaps of a block (ops of the pairs that finish late run in the first gaps of
// the NEXT block: lag 1), two blocks per loop iteration so that the two blocks' data never need a register copy
struct Sched { int e[16], h[8], f[16], l[8]; };   // gap + 14 * lag
constexpr Sched kSchedA = {{0, 0, 1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13}, {1, 2, 4, 6, 8, 10, 12, 14},
                           {2, 2, 3, 3, 5, 5, 7, 7, 9, 9, 11, 11, 13, 13, 15, 15}, {3, 4, 6, 8, 10, 12, 14, 16}};
// B: the doubled exps in the middle of the block, the tails of pairs 6 and 7 in the next block
constexpr Sched kSchedB = {{0, 1, 2, 3, 4, 5, 5, 6, 7, 8, 9, 10, 10, 11, 12, 13}, {2, 4, 6, 7, 9, 11, 12, 14},
                           {3, 3, 5, 5, 7, 7, 8, 8, 10, 10, 12, 12, 13, 13, 15, 15}, {4, 6, 8, 9, 11, 13, 14, 16}};
template <int MODE, int NMFMA>
__global__ __launch_bounds__(256, 1) void k2(float* out, long long* cyc, int blocks, float seed) {
  constexpr Sched S = MODE == 6 ? kSchedA : kSchedB;
  const int lane = threadIdx.x & 63;
  half8 a[4], b[4];
  for (int q = 0; q < 4; ++q)
    for (int j = 0; j < 8; ++j) { a[q][j] = (_Float16)(seed + j + q); b[q][j] = (_Float16)(seed - j - q); }
  floatx16 c[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
  float w[2][16];
  for (int q = 0; q < 32; ++q) w[q / 16][q % 16] = seed * 1e-3f * (q + 1) + lane * 1e-4f;
  half2v hh[2][8], ll[2][8];
  for (int q = 0; q < 16; ++q) { hh[q / 8][q % 8] = half2v{(_Float16)0, (_Float16)0}; ll[q / 8][q % 8] = hh[0][0]; }
  auto op = [&](int par, int t, int i) {
    float (&ww)[16] = w[par];
    if (t == 0) ww[i] = __builtin_amdgcn_exp2f(ww[i]);
    else if (t == 1) hh[par][i] = half2v{(_Float16)ww[2 * i], (_Float16)ww[2 * i + 1]};
    else if (t == 2) {
      const unsigned hb = __builtin_bit_cast(unsigned, hh[par][i / 2]);
      float l;
      if (i & 1) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hb), "v"(ww[i]));
      else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hb), "v"(ww[i]));
      ww[i] = l;
    } else ll[par][i] = half2v{(_Float16)ww[2 * i], (_Float16)ww[2 * i + 1]};
  };
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < blocks / 2; ++it) {
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int m = 0; m < NMFMA; ++m) {
        __builtin_amdgcn_sched_barrier(0);
        c[(m / 3) % 4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m % 4], b[(m / 2) % 4], c[(m / 3) % 4], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int lag = 1; lag >= 0; --lag) {   // the older block's leftovers first
#pragma unroll
          for (int i = 0; i < 16; ++i) if (S.e[i] == m + NMFMA * lag) op(lag ? 1 - par : par, 0, i);
#pragma unroll
          for (int i = 0; i < 8; ++i) if (S.h[i] == m + NMFMA * lag) op(lag ? 1 - par : par, 1, i);
#pragma unroll
          for (int i = 0; i < 16; ++i) if (S.f[i] == m + NMFMA * lag) op(lag ? 1 - par : par, 2, i);
#pragma unroll
          for (int i = 0; i < 8; ++i) if (S.l[i] == m + NMFMA * lag) op(lag ? 1 - par : par, 3, i);
        }
        if (m == NMFMA - 1) {   // the finished block's data are "consumed" and fresh distances arrive
#pragma unroll
          for (int i = 0; i < 16; ++i) if (S.e[i] < NMFMA) {}
        }
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && threadIdx.x < 64) cyc[blockIdx.x] = t1 - t0;
  float s = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += c[q][r];
  for (int q = 0; q < 32; ++q) s += w[q / 16][q % 16];
  for (int q = 0; q < 16; ++q) s += (float)hh[q / 8][q % 8][0] + (float)ll[q / 8][q % 8][1];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE, int NMFMA>
void run(const char* name) {
  float* d; long long* cyc;
  hipMalloc(&d, 64); hipMalloc(&cyc, 256 * 8);
  const int blocks = 40000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  if (MODE >= 6) k2<MODE, NMFMA><<<256, 256>>>(d, cyc, 200, 1.f); else k<MODE, NMFMA><<<256, 256>>>(d, cyc, 200, 1.f);
  hipEventRecord(e0);
  if (MODE >= 6) k2<MODE, NMFMA><<<256, 256>>>(d, cyc, blocks, 1.f); else k<MODE, NMFMA><<<256, 256>>>(d, cyc, blocks, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < 256; ++i) mean += (double)h[i]; mean /= 256;
  printf("%-66s %7.1f cycles/block  (%.2f GHz)\n", name, mean / blocks, mean / (ms * 1e-3) * 1e-9);
  hipFree(d); hipFree(cyc);
}

int main() {
  run<0, 12>("12 MFMA");
  run<1, 12>("12 MFMA + 16 v_exp (independent, spread)");
  run<2, 12>("12 MFMA + 16 v_exp + 32 v_fma (independent, spread)");
  run<3, 12>("12 MFMA + real chain, four-step pipeline over the pairs");
  run<4, 12>("12 MFMA + real chain, pair by pair");
  run<5, 12>("12 MFMA + real chain, one exp per gap, five-step pipeline");
  run<0, 14>("14 MFMA");
  run<3, 14>("14 MFMA + real chain, four-step pipeline");
  run<6, 14>("14 MFMA + real chain, cyclic table A (doubles first)");
  run<7, 14>("14 MFMA + real chain, cyclic table B (doubles mid-block)");
  return 0;
}
