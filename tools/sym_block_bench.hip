// What would the symmetric form of the few-right-hand-sides Gram matvec buy?    With <= 32 vectors the
// fat-wave kernel is VALU-bound: per 32 x 32 block 16 v_exp + 8 v_cvt_pk + 16 v_fma_mix* (~360 issue cycles) against 9 MFMAs (288 cycles of
// matrix pipe).  K is symmetric: the exp / split of block (I, J) could serve W_I += K_IJ V_J AND W_J += K_IJ^T V_I -- the second product needs
// the block with lanes along j: a transpose through LDS (4 ds_write_b128 of the hi / lo A fragments, 8 ds_read_b64_tr_b16 back).
// Instruction-stream and data-toggling model on random data, one wave per SIMD -- NOT a correct matvec:
//   MODE 0: the shipped block loop for <= 32 vectors and d = 9 .. 12 (9 slots: c0 c1 d0 c2 d1 c3 d2 c4 c5, the 40-instruction chain by kFatSplit1)
//   MODE 1: a block PAIR per iteration: 3 distance + 6 forward + 6 transposed contraction MFMAs, the chain ONCE, the LDS transpose, and
//           the cross-wave sum of the W_J partials through LDS once per column block and 4 global atomics per wave for it
// Prints ns and cycles per BLOCK of K entries served (MODE 1: per half iteration).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void chain_op(float (&w)[16], half8 (&ah)[2], half8 (&al)[2], unsigned (&lp)[8], int t, int i) {
  if (t == 0) {
    w[i] = __builtin_amdgcn_exp2f(w[i]);
  } else if (t == 1) {
    const half2v h = {(_Float16)w[2 * i], (_Float16)w[2 * i + 1]};
    ah[i >> 2][(i & 3) * 2] = h[0];
    ah[i >> 2][(i & 3) * 2 + 1] = h[1];
  } else {
    const int pr = i >> 1;
    const half2v h = {ah[pr >> 2][(pr & 3) * 2], ah[pr >> 2][(pr & 3) * 2 + 1]};
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    if ((i & 1) == 0) {
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lp[pr]) : "v"(hb), "v"(w[i]));
    } else {
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp[pr]) : "v"(hb), "v"(w[i]));
      const half2v l = __builtin_bit_cast(half2v, lp[pr]);
      al[pr >> 2][(pr & 3) * 2] = l[0];
      al[pr >> 2][(pr & 3) * 2 + 1] = l[1];
    }
  }
}

template <int MODE, int FEAT>
__global__ __launch_bounds__(256, 1) void k(const _Float16* __restrict__ rnd, float* out, long long* cyc, int blocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 65536 / 2; i += 256) {
    const _Float16 v = rnd[(blockIdx.x * 977 + i) % (1 << 20)];
    const int piece = (i / 512) % 12;
    reinterpret_cast<_Float16*>(smem)[i] = piece >= 8 ? (_Float16)((float)v * (1.f / 12000.f)) : v;
  }
  __syncthreads();
  float agpr_seed = 0.f;
  asm volatile("; agpr" : "+a"(agpr_seed));
  floatx16 acc[4], acct[2];   // W_I blocks of the wave's four row blocks; W_J of the tile's two column blocks
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  for (int q = 0; q < 2; ++q)
    for (int r = 0; r < 16; ++r) acct[q][r] = 0.f;
  acc[0][0] = agpr_seed;
  half8 vf[4], vfi[4], aj[3], bi[3];      // probe fragments of the column block (hi / lo x 2 k-steps) and of the row block; distance operands
  half8 ahc[2], alc[2], ahn[2], aln[2], aht[2], alt[2], ahtn[2], altn[2];
  unsigned lp[8];
  float wn[16];
  floatx16 wd16;
  const char* base = smem + lane * 16;
  for (int q = 0; q < 4; ++q) {
    vf[q] = *reinterpret_cast<const half8*>(base + q * 1024);
    vfi[q] = *reinterpret_cast<const half8*>(base + (4 + q) * 1024);
  }
  for (int q = 0; q < 3; ++q) {
    aj[q] = *reinterpret_cast<const half8*>(base + (8 + (q & 1)) * 1024);
    bi[q] = *reinterpret_cast<const half8*>(base + (10 + (q & 1)) * 1024);
  }
  for (int q = 0; q < 2; ++q) {
    ahc[q] = vf[q]; alc[q] = vf[q + 2]; ahn[q] = vf[q]; aln[q] = vf[q + 2]; aht[q] = vf[q]; alt[q] = vf[q + 2]; ahtn[q] = vf[q]; altn[q] = vf[q + 2];
  }
  for (int q = 0; q < 16; ++q) wn[q] = -0.5f * (float)(q + 1) - 1e-3f * lane;
  for (int q = 0; q < 8; ++q) lp[q] = 0;
  // transpose staging area of this wave: [hi | lo] images of one 32 x 32 block, 2 KB each, behind the 48 KB of tile images
  char* tr = smem + 49152 + wid * 4096;
  char* wj = smem + 49152 + 16384;          // the four waves' W_J partials of one column block: 4 x 4 KB
  char* wjw = wj + wid * 4096;
  float* wout = out + 1024 + (size_t)blockIdx.x * 1024 * 1024;   // 4 MB of W per workgroup for the atomics to land in
  constexpr int NSLOT = MODE == 0 ? 9 : 15;
  // MODE 0: the shipped table (kFatSplit1): pair p: both exps behind slot p, hi p + 1, mixlo p + 2, mixhi p + 3 (mod 9 into the next block)
  // MODE 1: the same 40 instructions spread evenly over 15 slots (tables e1 / h1 / m1 below)
  const long long t0 = __builtin_readcyclecounter();
  for (int b4 = 0; b4 < blocks / 4; ++b4) {
    const char* vb = base + (b4 & 3) * 12288;
    auto body = [&](auto bc) {
      constexpr int b = decltype(bc)::value;
#pragma unroll
      for (int slot = 0; slot < NSLOT; ++slot) {
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) {
          if (slot == 2 || slot == 4 || slot == 6) {
            const int q = (slot - 2) / 2;
            if (q == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(wd16) : "a"(aj[q]), "a"(bi[q]));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(wd16) : "a"(aj[q]), "a"(bi[q]));
          } else {
            const int m = slot < 2 ? slot : (slot == 3 ? 2 : (slot == 5 ? 3 : slot - 3));
            const int s = m / 3, w = m % 3;
            acc[b & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w == 2 ? alc[s] : ahc[s], w == 1 ? vf[s * 2 + 1] : vf[s * 2], acc[b & 3], 0, 0, 0);
          }
        } else {
          if (slot == 8 || slot == 10 || slot == 12) {
            const int q = (slot - 8) / 2;
            if (q == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(wd16) : "a"(aj[q]), "a"(bi[q]));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(wd16) : "a"(aj[q]), "a"(bi[q]));
          } else {
            const int m = slot < 8 ? slot : (slot == 9 ? 8 : (slot == 11 ? 9 : slot - 3));   // 12 contraction MFMAs: 6 forward, 6 transposed
            if (m < 6) {
              const int s = m / 3, w = m % 3;
              acc[b & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w == 2 ? alc[s] : ahc[s], w == 1 ? vf[s * 2 + 1] : vf[s * 2], acc[b & 3], 0, 0, 0);
            } else {
              const int s = (m - 6) / 3, w = (m - 6) % 3;
              if (FEAT & 4) acct[b & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w == 2 ? alt[s] : aht[s], w == 1 ? vfi[s * 2 + 1] : vfi[s * 2], acct[b & 1], 0, 0, 0);
              else acct[b & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w == 2 ? alc[s] : ahc[s], w == 1 ? vf[s * 2 + 1] : vf[s * 2], acct[b & 1], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the chain of the next block
        if (MODE == 0) {
          constexpr int e0[16] = {0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7};
          constexpr int h0[8] = {1, 2, 3, 4, 5, 6, 7, 8};
          constexpr int m0[16] = {2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10};
#pragma unroll
          for (int i = 0; i < 16; ++i) if (e0[i] % 9 == slot) chain_op(wn, ahn, aln, lp, 0, i);
#pragma unroll
          for (int i = 0; i < 8; ++i) if (h0[i] % 9 == slot) chain_op(wn, ahn, aln, lp, 1, i);
#pragma unroll
          for (int i = 0; i < 16; ++i) if (m0[i] % 9 == slot) chain_op(wn, ahn, aln, lp, 2, i);
        } else {
          // the 40 chain instructions in dependency order, cut into 15 slots of ~24 issue cycles each (v_exp 11, v_cvt_pk 5.5, v_fma_mix 8.6)
          constexpr int e1[16] = {0, 0, 1, 1, 2, 2, 4, 4, 6, 6, 7, 8, 9, 10, 11, 12};
          constexpr int h1[8] = {1, 3, 4, 6, 8, 10, 12, 13};
          constexpr int m1[16] = {3, 3, 5, 5, 7, 7, 9, 9, 10, 11, 12, 13, 13, 14, 14, 14};
#pragma unroll
          for (int i = 0; i < 16; ++i) if (e1[i] == slot) chain_op(wn, ahn, aln, lp, 0, i);
#pragma unroll
          for (int i = 0; i < 8; ++i) if (h1[i] == slot) chain_op(wn, ahn, aln, lp, 1, i);
#pragma unroll
          for (int i = 0; i < 16; ++i) if (m1[i] == slot) chain_op(wn, ahn, aln, lp, 2, i);
          // the transpose of the CURRENT block's A fragments through LDS: 4 ds_write_b128 (slots 0-3), 8 ds_read_b64_tr_b16 (slots 4-7, two each)
          if ((FEAT & 1) && slot < 4) *reinterpret_cast<half8*>(tr + (slot & 1) * 2048 + (slot >> 1) * 1024 + lane * 16) = (slot & 1) ? alc[slot >> 1] : ahc[slot >> 1];
          if ((FEAT & 1) && slot >= 4 && slot < 8) {
            const int s = (slot - 4) >> 1, hl = (slot - 4) & 1;
            const char* src = tr + hl * 2048 + s * 1024 + (lane & 15) * 64 + (lane >> 4) * 8;
            const half4 a = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src)));
            const half4 c = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(src + 32)));
            half8& dst = hl ? altn[s] : ahtn[s];   // consumed by the transposed products one block later: no LDS latency on the MFMA's path
            dst = half8{a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
          }
          // the W_J partials of the four waves meet in LDS, once per column block (= every fourth block of a wave): each wave writes its 16
          // registers, a workgroup barrier, each wave sums one quarter (4 registers x 4 waves) and sends it to W with 4 global atomics
          // (ds_add_f32 instead -- 4 per block -- costs ~1400 cycles per block: LDS float atomics serialise; first version of this file)
          if ((FEAT & 2) && (b & 3) == 3 && slot >= 9 && slot < 13) {
            const int q4 = slot - 9;
            *reinterpret_cast<float4*>(wjw + q4 * 1024 + lane * 16) = float4{acct[0][q4 * 4], acct[0][q4 * 4 + 1], acct[0][q4 * 4 + 2], acct[0][q4 * 4 + 3]};
          }
          if ((FEAT & 2) && (b & 3) == 3 && slot == 13) __builtin_amdgcn_s_barrier();
          if ((FEAT & 2) && (b & 3) == 3 && slot == 14) {
            float4 sum = *reinterpret_cast<const float4*>(wj + 0 * 4096 + wid * 1024 + lane * 16);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
              const float4 t = *reinterpret_cast<const float4*>(wj + w * 4096 + wid * 1024 + lane * 16);
              sum.x += t.x; sum.y += t.y; sum.z += t.z; sum.w += t.w;
            }
            float* dst = wout + ((size_t)(b4 & 1023) * 4 + wid) * 256 + lane;
            unsafeAtomicAdd(dst, sum.x); unsafeAtomicAdd(dst + 64, sum.y); unsafeAtomicAdd(dst + 128, sum.z); unsafeAtomicAdd(dst + 192, sum.w);
          }
        }
        if ((b & 3) == 3 && slot < 4) vf[slot] = *reinterpret_cast<const half8*>(vb + slot * 1024 + 1024);
        if (MODE == 1 && (b & 3) == 2 && slot < 4) vfi[slot] = *reinterpret_cast<const half8*>(vb + slot * 1024 + 5120);
        if ((b & 3) == 1 && slot < 3) aj[slot] = *reinterpret_cast<const half8*>(vb + (9 + (slot & 1)) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
      ahc[0] = ahn[0]; ahc[1] = ahn[1];
      alc[0] = aln[0]; alc[1] = aln[1];
      if (MODE == 1) { aht[0] = ahtn[0]; aht[1] = ahtn[1]; alt[0] = altn[0]; alt[1] = altn[1]; }
#pragma unroll
      for (int r = 0; r < 16; ++r) wn[r] = wd16[r];
    };
    body(std::integral_constant<int, 0>{});
    body(std::integral_constant<int, 1>{});
    body(std::integral_constant<int, 2>{});
    body(std::integral_constant<int, 3>{});
  }
  const long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  float s = 0;
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  for (int q = 0; q < 2; ++q)
    for (int r = 0; r < 16; ++r) s += acct[q][r];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE, int FEAT>
void run(const char* name, const _Float16* rnd, float* d, int per) {
  long long* cyc;
  hipMalloc(&cyc, 256 * 8);
  const int blocks = 400000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, FEAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
  k<MODE, FEAT><<<256, 256, 81920>>>(rnd, d, cyc, 20000);
  float best = 1e30f;
  double cycles = 0;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    k<MODE, FEAT><<<256, 256, 81920>>>(rnd, d, cyc, blocks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < 256; ++i) mean += (double)h[i];
    mean /= 256;
    if (ms < best) { best = ms; cycles = mean; }
  }
  printf("%-72s %7.1f ns, %7.1f cycles per block of K served (%.2f GHz)\n", name, best * 1e6 / blocks / per, cycles / blocks / per, cycles / (best * 1e-3) * 1e-9);
  fflush(stdout);
  hipFree(cyc);
}

int main() {
  const size_t n = 1 << 20;
  _Float16* h = (_Float16*)malloc(n * 2);
  srand(3);
  for (size_t i = 0; i < n; ++i) h[i] = (_Float16)((rand() % 4001 - 2000) * 4.0f);
  _Float16* rnd;
  hipMalloc(&rnd, n * 2);
  hipMemcpy(rnd, h, n * 2, hipMemcpyHostToDevice);
  float* d;
  hipMalloc(&d, 4096 + (size_t)256 * 1024 * 1024 * 4);
  hipMemset(d, 0, 4096 + (size_t)256 * 1024 * 1024 * 4);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 0>("shipped loop, <= 32 vectors, d = 9..12 (9 MFMAs + chain per block)", rnd, d, 1);
    run<1, 7>("symmetric pair (15 MFMAs + chain + LDS transpose + W_J flush per 2 blocks)", rnd, d, 2);
    run<1, 5>("  ... without the W_J flush", rnd, d, 2);
    run<1, 6>("  ... without the LDS transpose (stale transposed fragments)", rnd, d, 2);
    run<1, 4>("  ... without either", rnd, d, 2);
    run<1, 0>("  ... without either, the 6 extra products on the forward operands", rnd, d, 2);
  }
  return 0;
}
