// K-loop of the split gradient GEMM  S = L^T R  (hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_f16) as a stand-alone model, to decide the
// round-4 rewrite of k_rbf_mfma_grad_h (csrc/mfx_rbf_mfma.hip) by measurement.  Same packed operands ([kb][column][8] halves, hi and lo
// images), same 256 x 256 workgroup tile, same grid mapping (blockIdx.x = XCD label, 8 x 8 patches, 8 tiles per workgroup) as the product;
// the per-tile epilogue is replaced by a weighted checksum  sum_ij w_i u_j S_ij  (separable weights attached to the row / column LABEL,
// so a wrong block pairing or a lost stage changes it) -- all variants must agree on every workgroup's number.
//   V0  the product's loop: 8 waves (4 x 2 of 64 x 128), two per SIMD half a stage apart (ping-pong), LDS-DMA ring of four 16-row stages
//   V1  ONE wave per SIMD: 4 waves (2 x 2 of 128 x 128), 256 accumulators in AGPRs, LDS-DMA ring of four stages, DMA issued by the wave itself
//   V2  as V1, operands staged through registers (global_load_dwordx4 -> ds_write_b128), two LDS slots
// Build: hipcc -O3 --offload-arch=gfx950 tools/gradk_bench.hip -o tools/gradk_bench.bin ;  run: tools/gradk_bench.bin [rows] [cols] [batch] [reps]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int kTM = 256, kTN = 256, kSub = 8, kSplit = 8;

template <int AUX = 0>
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}

// ---- operands: x ~ U(-1, 1) * 2^14 * 2^-(3 (b % 5)), hi = f16(x), lo = f16(x - hi), packed [kb][col][8] -------------------------------
__global__ void k_fill(_Float16* hi, _Float16* lo, int64_t npad, int64_t nkb, uint32_t seed) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x, kb = blockIdx.y;
  if (col >= npad) return;
  half8 h, l;
  for (int q = 0; q < 8; ++q) {
    uint64_t z = ((uint64_t)(kb * 8 + q) * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)col * 0xBF58476D1CE4E5B9ull) ^ seed;
    z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29; z *= 0xD6E8FEB86659FD93ull; z ^= z >> 32;
    const float u = ((float)(uint32_t)(z & 0xFFFFFF) / 8388608.f - 1.f) * 16384.f * exp2f(-3.f * (float)((kb * 8 + q) % 5));
    const _Float16 fh = (_Float16)u;
    h[q] = fh;
    l[q] = (_Float16)(u - (float)fh);
  }
  *reinterpret_cast<half8*>(hi + (kb * npad + col) * 8) = h;
  *reinterpret_cast<half8*>(lo + (kb * npad + col) * 8) = l;
}

struct Args {
  const _Float16 *Lh, *Ll, *Rh, *Rl;
  int64_t npad_l, npad_r, nkb;
  int tiles_per_block;
  int chunk_major;    // V3 only: 1 = ONE tile per workgroup, workgroups ordered (column chunk of 64 tiles, row block, tile of the chunk): the
                      // 168-MB R chunk all XCDs are working on stays in the Infinity Cache while the row blocks stream past it
  double* out;        // one checksum per workgroup
  long long* stamps;  // per workgroup: shader cycles, 100 MHz ticks
};

// weights of the checksum: w_i = wblk(i / 32) * wlow(i % 8), u_j = wblk(j / 32) * wlow(j % 8) (labels relative to the tile)
__device__ __forceinline__ float wblk(int blk) { return 1.f + 0.375f * (float)blk; }
__device__ __forceinline__ float wlow(int x) { return 1.f + 0.125f * (float)(x & 7); }
// one 32 x 32 accumulator block (register r <-> row (r & 3) + 8 (r >> 2) + 4 lhi, lane <-> column l31) times the weights
__device__ __forceinline__ double block_sum(const floatx16& c, int rowblk, int colblk, int l31, int lhi) {
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) t = fmaf(c[r], wlow((r & 3) + 4 * lhi), t);
  return (double)t * (double)(wblk(rowblk) * wblk(colblk) * wlow(l31));
}

// ================================================================================================ V0: the product's loop
struct SmemV0 {
  _Float16 a_hi[4][2][kTM][8], a_lo[4][2][kTM][8], b_hi[4][2][kTN][8], b_lo[4][2][kTN][8];
};
__global__ __launch_bounds__(512, 1) void k_v0(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  SmemV0& sm = *reinterpret_cast<SmemV0*>(smem_raw);
  constexpr int NBW = 4, TN = kTN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int64_t i0 = (int64_t)(blockIdx.y / kSub) * kTM;
  const int64_t ntj = g.npad_r / TN;
  const int64_t tj_begin = ((int64_t)blockIdx.x * kSub + blockIdx.y % kSub) * g.tiles_per_block;
  int64_t tj_end = tj_begin + g.tiles_per_block;
  if (tj_end > ntj) tj_end = ntj;
  const int64_t nstage = g.nkb / 2;
  const uint32_t stage_bytes_l = (uint32_t)(2 * g.npad_l * 16), stage_bytes_r = (uint32_t)(2 * g.npad_r * 16);
  const int cwa = wid * 64 + lane, cwb = wid * 64 + lane;
  const uint32_t offL = (uint32_t)((((int64_t)(cwa >> 8) * g.npad_l) + i0 + (cwa & 255)) * 16);
  const uint32_t offR0 = (uint32_t)((((int64_t)(cwb / TN) * g.npad_r) + (cwb % TN)) * 16);
  const char *Lhb = (const char*)g.Lh, *Llb = (const char*)g.Ll, *Rhb = (const char*)g.Rh, *Rlb = (const char*)g.Rl;
  const int ca = wid * 64, cb = wid * 64;
  auto issue_stage = [&](uint32_t l_off, uint32_t r_off, int slot) {
    glds16(Lhb + l_off, &sm.a_hi[slot][ca >> 8][ca & 255][0]);
    glds16(Llb + l_off, &sm.a_lo[slot][ca >> 8][ca & 255][0]);
    glds16(Rhb + r_off, &sm.b_hi[slot][cb / TN][cb % TN][0]);
    glds16(Rlb + r_off, &sm.b_lo[slot][cb / TN][cb % TN][0]);
  };
  double cs = 0.0;
  long long c0 = 0, r0 = 0;
  if (tid == 0) { c0 = clock64(); r0 = wall_clock64(); }
  for (int64_t tj = tj_begin; tj < tj_end; ++tj) {
    const int64_t j0 = tj * TN;
    floatx16 acc[2][NBW];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    uint32_t ol = offL, orr = offR0 + (uint32_t)(j0 * 16);
    if (tj == tj_begin) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (q < nstage) issue_stage(ol, orr, q);
        ol += stage_bytes_l;
        orr += stage_bytes_r;
      }
    } else {
      ol += 3 * stage_bytes_l;
      orr += 3 * stage_bytes_r;
    }
    {
      const int64_t left0 = nstage - 1;
      if (left0 >= 2) __builtin_amdgcn_s_waitcnt(0x0F78);
      else if (left0 == 1) __builtin_amdgcn_s_waitcnt(0x0F74);
      else __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    if (wid >= 4) __builtin_amdgcn_s_barrier();
    for (int64_t st = 0; st < nstage; ++st) {
      const int slot = (int)(st & 3);
      __builtin_amdgcn_s_barrier();  // B1
      if (st + 3 < nstage) issue_stage(ol, orr, (int)((st + 3) & 3));
      ol += stage_bytes_l;
      orr += stage_bytes_r;
      half8 ah[2], al[2], bh[NBW], bl[NBW];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        ah[a] = *reinterpret_cast<const half8*>(&sm.a_hi[slot][lhi][wm * 64 + a * 32 + l31][0]);
        al[a] = *reinterpret_cast<const half8*>(&sm.a_lo[slot][lhi][wm * 64 + a * 32 + l31][0]);
      }
#pragma unroll
      for (int b = 0; b < NBW; ++b) {
        bh[b] = *reinterpret_cast<const half8*>(&sm.b_hi[slot][lhi][wn * (NBW * 32) + b * 32 + l31][0]);
        bl[b] = *reinterpret_cast<const half8*>(&sm.b_lo[slot][lhi][wn * (NBW * 32) + b * 32 + l31][0]);
      }
      const int64_t later = nstage - 2 - st;
      if (st + 1 < nstage) {
        if (later >= 2) __builtin_amdgcn_s_waitcnt(0x0F78);
        else if (later == 1) __builtin_amdgcn_s_waitcnt(0x0F74);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();  // B2
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (wid < 4) __builtin_amdgcn_s_barrier();
    // "epilogue": the ring is free, the next tile's first stages go out first (as the product's register epilogue does)
    __builtin_amdgcn_s_barrier();
    if (tj + 1 < tj_end) {
      uint32_t nl = offL, nr = offR0 + (uint32_t)((j0 + TN) * 16);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (q < nstage) issue_stage(nl, nr, q);
        nl += stage_bytes_l;
        nr += stage_bytes_r;
      }
    }
    double t = 0.0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < NBW; ++b) t += block_sum(acc[a][b], wm * 2 + a, wn * NBW + b, l31, lhi);
    cs += t * (double)(1 + (tj % 3));
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const int64_t wg = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  atomicAdd(&g.out[wg], cs);
  if (tid == 0) {
    g.stamps[2 * wg] = clock64() - c0;
    g.stamps[2 * wg + 1] = wall_clock64() - r0;
  }
}

// ================================================================================================ V3: V0's structure on 16x16x32
// 8 waves (4 x 2 of 64 x 128 = 4 x 8 blocks of 16 x 16), two per SIMD half a stage apart; a stage is 32 batch rows (one K = 32 MFMA step):
// 24 fragment reads and 96 MFMAs of 16 cycles per wave.  LDS: two stages of [array 4][kb-group 4][column 256][8 halves] = 64 KB each; the
// next stage is requested when the other one's last readers are done (one stage = 3072 SIMD cycles to land).
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int kSlot3 = 4 * 4 * 256 * 16;
__device__ __forceinline__ double block_sum16(const floatx4& c, int rowlabel0, int collabel0, int l15, int lq) {
  // rows rowlabel0 + 4 lq + r (r = 0..3), column collabel0 + l15; rowlabel0, collabel0 multiples of 16
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) t = fmaf(c[r], wlow(4 * lq + r), t);
  return (double)t * (double)(wblk(rowlabel0 >> 5) * wblk(collabel0 >> 5) * wlow(l15));
}
// VAR: 0 as built into the product; 1 column blocks outer; 2 no s_setprio; 3 products grouped (all hh, all hl, all lh);
//      timing diagnostics with WRONG sums: 4 no LDS-DMA in the loop, 5 no fragment reads in the loop, 6 neither, 7 neither and no barriers;
//      8 / 9 (round 5, on VAR 2's loop): only 6 / 7 of a wave's 8 LDS-DMA pieces per stage -- the L2 -> LDS bytes per flop of a 3-byte operand
//      (f16 hi + 8-bit lo: 3/4) and, bracketing it from above, of a 256 x 384 workgroup tile (5/6 of the 256 x 256 tile's; 7/8 here)
template <int VAR, int AUX = 0>
__global__ __launch_bounds__(512, 1) void k_v3(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  int64_t i0 = (int64_t)(blockIdx.y / kSub) * kTM;
  const int64_t ntj = g.npad_r / kTN;
  int64_t tj_begin = ((int64_t)blockIdx.x * kSub + blockIdx.y % kSub) * g.tiles_per_block;
  int64_t tj_end = tj_begin + g.tiles_per_block;
  if (g.chunk_major) {
    const int64_t per_chunk = (g.npad_l / kTM) * kSub;  // workgroups (per XCD label) of one column chunk
    const int64_t chunk = blockIdx.y / per_chunk, rem = blockIdx.y % per_chunk;
    i0 = (rem / kSub) * kTM;
    tj_begin = chunk * (kSplit * kSub) + (int64_t)blockIdx.x * kSub + rem % kSub;
    tj_end = tj_begin + 1;
  }
  if (tj_end > ntj) tj_end = ntj;
  const int nstage = (int)(g.nkb / 4);
  const char* gbase[4] = {(const char*)g.Lh, (const char*)g.Ll, (const char*)g.Rh, (const char*)g.Rl};
  const uint32_t stage_bytes_l = (uint32_t)(4 * g.npad_l * 16), stage_bytes_r = (uint32_t)(4 * g.npad_r * 16);
  // my 8 pieces of a stage: piece i -> array i >> 1, kb-group 2 (i & 1) + (wid >> 2), columns 64 (wid & 3) + lane
  const int kb0 = wid >> 2, chunk = wid & 3;
  const uint32_t lane16 = (uint32_t)lane * 16;
  const uint32_t offL = (uint32_t)((kb0 * g.npad_l + i0 + chunk * 64) * 16), offR = (uint32_t)((kb0 * g.npad_r + chunk * 64) * 16);
  const uint32_t kb2L = (uint32_t)(2 * g.npad_l * 16), kb2R = (uint32_t)(2 * g.npad_r * 16);
  auto issue_stage = [&](uint32_t l_off, uint32_t r_off, int slot) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (VAR == 8 && (i == 3 || i == 7)) continue;  // 3/4 of the stream (the lo images at half size)
      if (VAR == 9 && i == 7) continue;              // 7/8 of the stream
      const int arr = i >> 1, hi2 = i & 1;
      const uint32_t so = arr < 2 ? l_off + (hi2 ? kb2L : 0u) : r_off + (hi2 ? kb2R : 0u);
      glds16<AUX>(gbase[arr] + (so + lane16), smem + slot * kSlot3 + ((arr * 4 + 2 * hi2 + kb0) * 256 + chunk * 64) * 16);
    }
  };
  auto a_ptr = [&](int slot, int hl, int a) {
    return reinterpret_cast<const half8*>(smem + slot * kSlot3 + (((0 + hl) * 4 + lq) * 256 + wm * 64 + a * 16 + l15) * 16);
  };
  auto b_ptr = [&](int slot, int hl, int b) {
    return reinterpret_cast<const half8*>(smem + slot * kSlot3 + (((2 + hl) * 4 + lq) * 256 + wn * 128 + b * 16 + l15) * 16);
  };
  double cs = 0.0;
  long long c0 = 0, r0 = 0;
  if (tid == 0) { c0 = clock64(); r0 = wall_clock64(); }
  for (int64_t tj = tj_begin; tj < tj_end; ++tj) {
    const int64_t j0 = tj * kTN;
    floatx4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
    uint32_t ol = offL, orr = offR + (uint32_t)(j0 * 16);
    if (tj == tj_begin) {
      __syncthreads();
      issue_stage(ol, orr, 0);
    }
    ol += stage_bytes_l;
    orr += stage_bytes_r;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // stage 0 (requested here or before the previous tile's checksum)
    if (wid >= 4) __builtin_amdgcn_s_barrier();
    half8 ah[4], al[4], bh[8], bl[8];
    if constexpr (VAR >= 5) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        ah[a] = *a_ptr(0, 0, a);
        al[a] = *a_ptr(0, 1, a);
      }
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        bh[b] = *b_ptr(0, 0, b);
        bl[b] = *b_ptr(0, 1, b);
      }
    }
    for (int st = 0; st < nstage; ++st) {
      const int slot = st & 1;
      if (wid < 4) __builtin_amdgcn_s_waitcnt(0x0F70);  // lower half: my pieces of this stage, requested a stage ago
      if constexpr (VAR != 7) __builtin_amdgcn_s_barrier();  // B1
      // the other slot's last readers (the upper half, stage st - 1) passed this barrier: refill it with stage st + 1
      if constexpr (VAR != 4 && VAR != 6 && VAR != 7)
        if (st + 1 < nstage) issue_stage(ol, orr, slot ^ 1);
      ol += stage_bytes_l;
      orr += stage_bytes_r;
      if constexpr (VAR < 5) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          ah[a] = *a_ptr(slot, 0, a);
          al[a] = *a_ptr(slot, 1, a);
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          bh[b] = *b_ptr(slot, 0, b);
          bl[b] = *b_ptr(slot, 1, b);
        }
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a) asm volatile("" : "+v"(ah[a]), "+v"(al[a]));
#pragma unroll
        for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(bh[b]), "+v"(bl[b]));
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      if (wid >= 4) __builtin_amdgcn_s_waitcnt(0x0F70);  // upper half: my pieces of stage st + 1 before the barrier the lower half reads it behind
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (VAR != 7) __builtin_amdgcn_s_barrier();  // B2
      if constexpr (VAR != 2 && VAR < 8) __builtin_amdgcn_s_setprio(1);
      if constexpr (VAR == 1) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
          }
      } else if constexpr (VAR == 3) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w == 2 ? al[a] : ah[a], w == 1 ? bl[b] : bh[b], acc[a][b], 0, 0, 0);
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 8; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
          }
      }
      if constexpr (VAR != 2 && VAR < 8) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (wid < 4) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();  // both slots are free
    if (tj + 1 < tj_end) issue_stage(offL, offR + (uint32_t)((j0 + kTN) * 16), 0);
    double t = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) t += block_sum16(acc[a][b], wm * 64 + a * 16, wn * 128 + b * 16, l15, lq);
    cs += t * (double)(1 + (tj % 3));
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const int64_t wg = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  atomicAdd(&g.out[wg], cs);
  if (tid == 0) {
    g.stamps[2 * wg] = clock64() - c0;
    g.stamps[2 * wg + 1] = wall_clock64() - r0;
  }
}

// ================================================================================================ V1 / V2: one wave per SIMD
// LDS slot: [array a_hi a_lo b_hi b_lo][kb-group 2][column 256][8 halves] = 32 KB.  Piece q (1 KiB) = array q >> 3, kb-group (q >> 2) & 1,
// 64-column chunk q & 3; wave w moves the pieces w + 4 i (i = 0..7): array i >> 1, kb-group i & 1, chunk w.
constexpr int kSlotBytes = 4 * 2 * 256 * 16;

// MFMA m of a stage (m = 0..47): row block a = m / 12, column block b = (m / 3) % 4, product m % 3 (hh, hl, lh)
// gap actions of a stage: fragment f (0..15) of the NEXT stage is read behind MFMA kRead[f]; transfer piece i (0..7) behind kMove[i]
//   fragment f: 0-3 ah[a], 4-7 al[a], 8-11 bh[b], 12-15 bl[b]
template <int MODE>  // 1: LDS-DMA (four slots), 2: register staging (two slots)
__global__ __launch_bounds__(256, 1) void k_fat(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NSLOT = MODE == 1 ? 4 : 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int64_t i0 = (int64_t)(blockIdx.y / kSub) * kTM;
  const int64_t ntj = g.npad_r / kTN;
  const int64_t tj_begin = ((int64_t)blockIdx.x * kSub + blockIdx.y % kSub) * g.tiles_per_block;
  int64_t tj_end = tj_begin + g.tiles_per_block;
  if (tj_end > ntj) tj_end = ntj;
  const int nstage = (int)(g.nkb / 2);
  const int ntile = (int)(tj_end - tj_begin);
  const int total = ntile * nstage;  // stages of the whole sweep, one continuous pipeline across tiles

  float agpr_seed = 0.f;
  asm volatile("; accumulators in AGPRs" : "+a"(agpr_seed));

  // my transfer pieces of a stage: piece i -> array i >> 1 (L hi, L lo, R hi, R lo), kb-group i & 1, columns 64 wid + lane of the tile's
  // 256.  Source = array base (SGPR pair) + 32-bit offset = scalar cursor + 16 lane: the cursor is wave-uniform (SALU only).
  const char* gbase[4] = {(const char*)g.Lh, (const char*)g.Ll, (const char*)g.Rh, (const char*)g.Rl};
  const uint32_t stage_bytes_l = (uint32_t)(2 * g.npad_l * 16), stage_bytes_r = (uint32_t)(2 * g.npad_r * 16);
  const uint32_t offL = (uint32_t)((i0 + wid * 64) * 16), offR = (uint32_t)((tj_begin * kTN + wid * 64) * 16);
  const uint32_t kbL = (uint32_t)(g.npad_l * 16), kbR = (uint32_t)(g.npad_r * 16);
  const uint32_t lane16 = (uint32_t)lane * 16;
  int t_st = 0, t_tile = 0;      // stage within the tile / tile of the transfer cursor
  uint32_t t_l = offL, t_r = offR;
  auto advance_transfer = [&]() {
    ++t_st;
    t_l += stage_bytes_l;
    t_r += stage_bytes_r;
    if (t_st == nstage) {  // next tile: L restarts, R moves to the next 256 columns; past the last tile the cursor wraps to the
      t_st = 0;            // first one (requests that nobody reads, but always inside the operands: no branch around a transfer)
      t_l = offL;
      t_r += (uint32_t)(kTN * 16) - (uint32_t)nstage * stage_bytes_r;
      if (++t_tile == ntile) {
        t_tile = 0;
        t_r = offR;
      }
    }
  };
  auto piece_src = [&](int i) -> const char* {
    const int arr = i >> 1, kbg = i & 1;
    const uint32_t so = arr < 2 ? t_l + (kbg ? kbL : 0u) : t_r + (kbg ? kbR : 0u);
    return gbase[arr] + (so + lane16);
  };
  auto piece_dst = [&](int slot, int i) -> char* {  // wave-uniform base of the piece; lane adds 16 lane
    const int arr = i >> 1, kbg = i & 1;
    return smem + slot * kSlotBytes + ((arr * 2 + kbg) * 256 + wid * 64) * 16;
  };
  auto frag_ptr = [&](int slot, int f) -> const half8* {
    const int arr = f >> 2, blk = f & 3;  // arr: 0 a_hi, 1 a_lo, 2 b_hi, 3 b_lo
    const int col = (arr < 2 ? wm : wn) * 128 + blk * 32 + l31;
    return reinterpret_cast<const half8*>(smem + slot * kSlotBytes + ((arr * 2 + lhi) * 256 + col) * 16);
  };

  floatx16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  acc[0][0][0] = agpr_seed;

  half8 fr[2][16];  // fragments by stage parity
  uintx4 gr[8];     // MODE 2: staging registers
  double cs = 0.0;
  long long c0 = 0, r0 = 0;
  if (tid == 0) { c0 = clock64(); r0 = wall_clock64(); }

  // ---- prologue: stages 0 (and, for the DMA ring, 1..3) in LDS, fragments of stage 0 in registers --------------------------------------
  if constexpr (MODE == 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int i = 0; i < 8; ++i) glds16(piece_src(i), piece_dst(q, i));
      advance_transfer();
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
  } else {
    // stages 0 and 1 into slots 0 and 1, stage 2 into the staging registers
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int i = 0; i < 8; ++i) gr[i] = *reinterpret_cast<const uintx4*>(piece_src(i));
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<uintx4*>(piece_dst(q, i) + lane * 16) = gr[i];
      advance_transfer();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) gr[i] = *reinterpret_cast<const uintx4*>(piece_src(i));
    advance_transfer();
    __syncthreads();
  }
#pragma unroll
  for (int f = 0; f < 16; ++f) fr[0][f] = *frag_ptr(0, f);

  // one stage; PAR = parity of the stage (fragment set), SLOT = its LDS slot
  auto stage = [&](auto par_c, auto slot_c, int S) {
    constexpr int PAR = decltype(par_c)::value, SLOT = decltype(slot_c)::value;
    // everybody's data of stage S + 1 is in LDS, everybody's reads of slot(S) (made during stage S - 1) are done
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int m = 0; m < 48; ++m) {
      const int a = m / 12, b = (m / 3) & 3, w = m % 3;
      __builtin_amdgcn_sched_barrier(0);
      // asm: the accumulators are PINNED to the accumulation registers and the fragments to VGPRs (with the intrinsic the allocator
      // splits the 256 accumulators between both files and spills 300 registers); in-place accumulation needs no wait states
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[a][b]) : "v"(fr[PAR][(w == 2 ? 4 : 0) + a]), "v"(fr[PAR][(w == 1 ? 12 : 8) + b]));
      __builtin_amdgcn_sched_barrier(0);
      // fragment reads of stage S + 1 (slot SLOT + 1): two gaps out of three, f = 0..15 behind MFMAs 0, 1, 3, 4, 6, 7, ...
      if (m % 3 != 2 && (m / 3) * 2 + (m % 3) < 16) {
        const int f = (m / 3) * 2 + (m % 3);
        fr[PAR ^ 1][f] = *frag_ptr((SLOT + 1) % NSLOT, f);
      }
      // transfers: one piece behind every third MFMA from the 24th on (gaps 26, 29, ..., 47)
      if (m >= 24 && m % 3 == 2) {
        const int i = (m - 24) / 3;
        if constexpr (MODE == 1) {
          // stage S + 4 into slot SLOT (its last readers finished before this stage's barrier)
          glds16(piece_src(i), piece_dst(SLOT, i));
        } else {
          // stage S + 2 (in gr since the previous stage) into slot SLOT; then stage S + 3 into gr
          *reinterpret_cast<uintx4*>(piece_dst(SLOT, i) + lane * 16) = gr[i];
          gr[i] = *reinterpret_cast<const uintx4*>(piece_src(i));
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    advance_transfer();
  };

  int S = 0;
  for (int64_t tj = tj_begin; tj < tj_end; ++tj) {
    for (int q = 0; q < nstage; q += 4, S += 4) {  // (nstage is a multiple of 4: batch % 64 == 0)
      stage(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, S);
      stage(std::integral_constant<int, 1>{}, std::integral_constant<int, 1 % NSLOT>{}, S + 1);
      stage(std::integral_constant<int, 0>{}, std::integral_constant<int, 2 % NSLOT>{}, S + 2);
      stage(std::integral_constant<int, 1>{}, std::integral_constant<int, 3 % NSLOT>{}, S + 3);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the compiler does not see the MFMAs: their results are read by VALU below
    double t = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        t += block_sum(acc[a][b], wm * 4 + a, wn * 4 + b, l31, lhi);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
      }
    cs += t * (double)(1 + (tj % 3));
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const int64_t wg = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  atomicAdd(&g.out[wg], cs);
  if (tid == 0) {
    g.stamps[2 * wg] = clock64() - c0;
    g.stamps[2 * wg + 1] = wall_clock64() - r0;
  }
}

int main(int argc, char** argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 32768, cols = argc > 2 ? atoll(argv[2]) : 131072;
  const int64_t batch = argc > 3 ? atoll(argv[3]) : 2560;
  const int reps = argc > 4 ? atoi(argv[4]) : 3;
  const int64_t nkb = batch / 8;
  if (rows % kTM || cols % (kTN * kSplit * kSub) || batch % 64) { fprintf(stderr, "rows %% 256, cols %% 16384, batch %% 64\n"); return 1; }
  _Float16 *Lh, *Ll, *Rh, *Rl;
  CK(hipMalloc(&Lh, batch * rows * 2)); CK(hipMalloc(&Ll, batch * rows * 2));
  CK(hipMalloc(&Rh, batch * cols * 2)); CK(hipMalloc(&Rl, batch * cols * 2));
  k_fill<<<dim3((unsigned)(rows / 256), (unsigned)nkb), 256>>>(Lh, Ll, rows, nkb, 1u);
  k_fill<<<dim3((unsigned)(cols / 256), (unsigned)nkb), 256>>>(Rh, Rl, cols, nkb, 2u);
  CK(hipDeviceSynchronize());
  const int64_t nti = rows / kTM, ntj = cols / kTN;
  const int tiles_per_block = (int)(ntj / (kSplit * kSub));
  const dim3 grid(kSplit, (unsigned)(nti * kSub));
  const int64_t nwg = (int64_t)grid.x * grid.y;
  double* out[19];
  long long* stamps;
  for (int v = 0; v < 19; ++v) CK(hipMalloc(&out[v], nwg * 8 * tiles_per_block));
  CK(hipMalloc(&stamps, nwg * 16 * tiles_per_block));
  Args a{Lh, Ll, Rh, Rl, rows, cols, nkb, tiles_per_block, 0, nullptr, stamps};
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_v0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmemV0)));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fat<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kSlotBytes));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fat<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kSlotBytes));
#define V3ATTR(V) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_v3<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kSlot3))
#define V3ATTRA(V, A) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_v3<V, A>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kSlot3))
  V3ATTRA(2, 1); V3ATTRA(2, 2); V3ATTRA(2, 16); V3ATTRA(2, 3);
  V3ATTR(0); V3ATTR(1); V3ATTR(2); V3ATTR(3); V3ATTR(4); V3ATTR(5); V3ATTR(6); V3ATTR(7); V3ATTR(8); V3ATTR(9);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double mfma_cycles_per_simd_per_tile = (double)(nkb / 2) * 48.0 * 32.0;  // both layouts: 48 MFMAs per SIMD and 16-row stage
  const char* names[18] = {"V0 round-3 loop (32x32x16, 8 waves, ping-pong)", "V1 fat waves, LDS-DMA", "V2 fat waves, register staging",
                           "V3 as V0 on 16x16x32 (K = 32 stages)", "V3.1 column blocks outer", "V3.2 no s_setprio", "V3.3 products grouped",
                           "V3.4 [diag] no LDS-DMA", "V3.5 [diag] no fragment reads", "V3.6 [diag] neither", "V3.7 [diag] neither, no barriers",
                           "V3.2 + sc0 loads", "V3.2 + nt loads", "V3.2 + sc1 loads", "V3.2 + sc0 nt loads", "V3.2 chunk-major, one tile per workgroup",
                           "V3.8 [diag] 3/4 of the L2->LDS stream (3-byte operand)", "V3.9 [diag] 7/8 of the stream (~ 256 x 384 tile: 5/6)"};
  const unsigned vmask = argc > 5 ? (unsigned)strtoul(argv[5], nullptr, 0) : 0xFu;
  std::vector<double> ref, cur(nwg);
  std::vector<long long> st(2 * nwg);
  for (int rep = 0; rep < reps; ++rep)
    for (int v = 0; v < 18; ++v) {
      if (!((vmask >> v) & 1)) continue;
      a.out = out[v];
      const int64_t nwg_v = v == 15 ? nwg * tiles_per_block : nwg;
      CK(hipMemset(out[v], 0, nwg_v * 8));
      CK(hipEventRecord(e0));
      if (v == 0) k_v0<<<grid, 512, sizeof(SmemV0)>>>(a);
      else if (v == 1) k_fat<1><<<grid, 256, 4 * kSlotBytes>>>(a);
      else if (v == 2) k_fat<2><<<grid, 256, 2 * kSlotBytes>>>(a);
      else if (v == 3) k_v3<0><<<grid, 512, 2 * kSlot3>>>(a);
      else if (v == 4) k_v3<1><<<grid, 512, 2 * kSlot3>>>(a);
      else if (v == 5) k_v3<2><<<grid, 512, 2 * kSlot3>>>(a);
      else if (v == 6) k_v3<3><<<grid, 512, 2 * kSlot3>>>(a);
      else if (v == 7) k_v3<4><<<grid, 512, 2 * kSlot3>>>(a);
      else if (v == 8) k_v3<5><<<grid, 512, 2 * kSlot3>>>(a);
      else if (v == 9) k_v3<6><<<grid, 512, 2 * kSlot3>>>(a);
      else if (v == 10) k_v3<7><<<grid, 512, 2 * kSlot3>>>(a);
      else if (v == 11) k_v3<2, 1><<<grid, 512, 2 * kSlot3>>>(a);
      else if (v == 12) k_v3<2, 2><<<grid, 512, 2 * kSlot3>>>(a);
      else if (v == 13) k_v3<2, 16><<<grid, 512, 2 * kSlot3>>>(a);
      else if (v == 14) k_v3<2, 3><<<grid, 512, 2 * kSlot3>>>(a);
      else if (v == 16) k_v3<8><<<grid, 512, 2 * kSlot3>>>(a);
      else if (v == 17) k_v3<9><<<grid, 512, 2 * kSlot3>>>(a);
      else {
        a.chunk_major = 1;
        k_v3<2><<<dim3(grid.x, grid.y * tiles_per_block), 512, 2 * kSlot3>>>(a);
        a.chunk_major = 0;
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      cur.resize(nwg_v);
      st.resize(2 * nwg_v);
      CK(hipMemcpy(cur.data(), out[v], nwg_v * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(st.data(), stamps, nwg_v * 16, hipMemcpyDeviceToHost));
      double worst = 0.0, big = 0.0;  // largest difference relative to the largest checksum (single checksums cancel to ~0)
      if (v == 15) {  // other workgroup numbering: compare the sum over all tiles
        double tot = 0.0, tot_ref = 0.0, babs = 0.0;
        for (int64_t i = 0; i < nwg_v; ++i) tot += cur[i];
        for (size_t i = 0; i < ref.size(); ++i) { tot_ref += ref[i]; babs = std::max(babs, fabs(ref[i])); }
        worst = fabs(tot - tot_ref) / (babs + 1e-300);
      } else {
        if (ref.empty() || (v == 0)) ref = cur;
        for (int64_t i = 0; i < nwg; ++i) { worst = std::max(worst, fabs(cur[i] - ref[i])); big = std::max(big, fabs(ref[i])); }
        worst /= big + 1e-300;
      }
      const int tpb = v == 15 ? 1 : tiles_per_block;
      std::vector<double> clk(nwg_v), util(nwg_v);
      for (int64_t i = 0; i < nwg_v; ++i) {
        clk[i] = (double)st[2 * i] / (double)st[2 * i + 1] * 0.1;  // GHz
        util[i] = mfma_cycles_per_simd_per_tile * tpb / (double)st[2 * i];
      }
      std::sort(clk.begin(), clk.end());
      std::sort(util.begin(), util.end());
      const double flops = 2.0 * rows * cols * batch;
      printf("rep %d  %-48s %8.2f ms  %6.1f TFLOP/s algorithmic  clock %.3f GHz  matrix pipe %.1f %% of the workgroup's cycles  max rel diff vs V0 %.2e\n",
             rep, names[v], ms, flops / ms * 1e-9, clk[nwg_v / 2], 100.0 * util[nwg_v / 2], worst);
      fflush(stdout);
    }
  return 0;
}
