// What does a consumer-style MFMA stream cost per stage on gfx950?  (producer / consumer matvec.)
// 512-thread workgroups, one per CU.  Waves 4-7 run "stages" of 24 v_mfma_f32_32x32x16_f16 with knobs:
//   READS   : ds_read_b128 per stage interleaved behind the MFMAs (0, 16)
//   BARRIER : one s_barrier per stage (waves 0-3 only take part in the barriers)
//   CHAIN   : 1 = three consecutive MFMAs share an accumulator (as in the matvec), 0 = rotate over four accumulators
//   PVALU   : waves 0-3 issue this many VALU instructions per stage (v_exp_f32 : v_fma = 1 : 2) between the barriers
//   CVALU   : the MFMA waves interleave this many VALU instructions per stage between their MFMAs
// Prints shader cycles per stage (s_memtime) and wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int READS, int BARRIER, int CHAIN, int PVALU, int CVALU, int RMODE = 0, int PRIO = 0>
__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int stages, float seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 32768 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = seed * (i & 15);
  __syncthreads();
  if (wid < 4) {
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = seed * (q + 1) + lane;
    for (int st = 0; st < stages; ++st) {
#pragma unroll
      for (int q = 0; q < PVALU; ++q) {
        if (q % 3 == 0) v[q % 8] = __builtin_amdgcn_exp2f(v[q % 8]);
        else v[q % 8] = fmaf(v[q % 8], 1.0001f, 0.5f);
      }
      if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    float s = 0;
    for (int q = 0; q < 8; ++q) s += v[q];
    if (s == 12345.678f) out[0] = s;
    return;
  }
  half8 a[4], b[4];
  for (int q = 0; q < 4; ++q)
    for (int j = 0; j < 8; ++j) { a[q][j] = (_Float16)(seed + j + q); b[q][j] = (_Float16)(seed - j - q); }
  floatx16 c[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
  float v[8];
  for (int q = 0; q < 8; ++q) v[q] = seed * (q + 1) + lane;
  half8 spare[16];
  for (int q = 0; q < 16; ++q) spare[q] = a[q % 4];
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
  const char* base = smem + lane * 16;
  const long long t0 = __builtin_readcyclecounter();
  for (int st = 0; st < stages; ++st) {
    const char* sb = base + (st & 7) * 1024;
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      const int acc = CHAIN ? (m / 3) % 4 : m % 4;
      __builtin_amdgcn_sched_barrier(0);
      c[acc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m % 4], b[(m / 2) % 4], c[acc], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (READS && m < 16) {
        const half8 f = *reinterpret_cast<const half8*>(sb + ((m * 5) & 15) * 1024);
        if (RMODE == 0) {         // the fragment lands in an operand that is used about four MFMAs later
          if (m & 1) a[(m + 3) % 4] = f; else b[((m + 8) / 2) % 4] = f;
        } else if (RMODE == 1) {  // lands in a spare register set, consumed (copied into the operands) 12 MFMAs later
          spare[m] = f;
        } else {                  // never consumed by an MFMA: summed at the end
          spare[m & 3] = f;
        }
      }
      if (READS && RMODE == 1 && m >= 12 && m < 16) { a[m % 4] = spare[m - 12]; }
#pragma unroll
      for (int q = 0; q < CVALU; ++q)
        if (q * 24 / CVALU == m) v[q % 8] = fmaf(v[q % 8], 1.0001f, 0.5f);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (BARRIER) __builtin_amdgcn_s_barrier();
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && wid == 4) cyc[blockIdx.x] = t1 - t0;
  float s = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += c[q][r];
  for (int q = 0; q < 8; ++q) s += v[q];
  for (int q = 0; q < 16; ++q) s += (float)spare[q][0];
  if (s == 12345.678f) out[0] = s;
}

template <int READS, int BARRIER, int CHAIN, int PVALU, int CVALU, int RMODE = 0, int PRIO = 0>
void run(const char* name) {
  float* d; long long* cyc;
  hipMalloc(&d, 64); hipMalloc(&cyc, 256 * 8);
  const int stages = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<READS, BARRIER, CHAIN, PVALU, CVALU, RMODE, PRIO><<<256, 512, 32768>>>(d, cyc, 200, 1.f);
  hipEventRecord(e0);
  k<READS, BARRIER, CHAIN, PVALU, CVALU, RMODE, PRIO><<<256, 512, 32768>>>(d, cyc, stages, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < 256; ++i) mean += (double)h[i]; mean /= 256;
  printf("%-58s %7.1f cycles/stage (s_memtime)  wall %.3f ms -> %.2f GHz\n", name, mean / stages, ms, mean / (ms * 1e-3) * 1e-9);
  hipFree(d); hipFree(cyc);
}

int main() {
  run<0, 0, 0, 0, 0>("24 MFMA, four accumulators");
  run<0, 0, 1, 0, 0>("24 MFMA, chains of three");
  run<16, 0, 1, 0, 0>("  + 16 ds_read_b128");
  run<0, 1, 1, 0, 0>("  + barrier per stage");
  run<16, 1, 1, 0, 0>("  + 16 ds_read_b128 + barrier");
  run<16, 1, 1, 0, 64>("  + 16 reads + barrier + 64 VALU in the MFMA waves");
  run<16, 1, 1, 48, 64>("  + ... + 48 VALU (16 exp) in the other waves");
  run<16, 1, 1, 96, 0>("  16 reads + barrier, 96 VALU (32 exp) in the other waves");
  run<16, 0, 1, 96, 0>("  16 reads, NO barrier, 96 VALU in the other waves");
  run<0, 0, 1, 96, 0>("  no reads, no barrier, 96 VALU in the other waves");
  run<0, 0, 1, 0, 96>("  no reads, no barrier, 96 VALU in the MFMA waves");
  run<16, 0, 1, 0, 0, 1>("16 reads consumed 12 MFMAs later");
  run<16, 0, 1, 0, 0, 2>("16 reads never consumed by an MFMA");
  run<16, 1, 1, 48, 64, 0, 1>("16 reads + barrier + 64 VALU in MFMA waves (prio 1) + 48 VALU others");
  run<16, 1, 1, 48, 64, 0, 3>("16 reads + barrier + 64 VALU in MFMA waves (prio 3) + 48 VALU others");
  run<0, 1, 1, 48, 64, 0, 0>("no reads, barrier + 64 VALU in MFMA waves + 48 VALU others");
  run<0, 1, 1, 48, 64, 0, 3>("no reads, barrier + 64 VALU in MFMA waves (prio 3) + 48 VALU others");
  return 0;
}
