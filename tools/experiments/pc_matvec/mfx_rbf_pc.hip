// EXPERIMENT, not part of libmfx: producer / consumer form of the matrix-core Gram matvec (3 x f16 split), gfx950.
//   W[i][b] = s * sum_j K(x_i, x_j) V[j][b] + noise V[i][b]   (util/gp_util.py:160-176,225-226,536-541 of the reference)
// Built, parity-tested through libmfx (round 3, commits c85ebd9 .. ) and measured SLOWER than the kernel it was meant to replace
// (7.05 vs 5.99 ms per launch at the C4 shape; 51 % vs 69 % matrix-pipe share): see README.md next to this file and DESIGN.md §3.2.
// What replaced the idea is csrc/mfx_rbf_fat.hip.  The file is kept so that the measurement can be repeated (build.sh, pc_model.hip).
#include <type_traits>

#include "mfx_internal.h"
#include "mfx_rbf_common.h"

namespace mfx {

int64_t rbf_pc_smem_bytes(int dpad);
int rbf_pc_launch(int dpad, int kind, bool vec4, dim3 grid, hipStream_t stream, const float* xs, const float* sq, int64_t n,
                  const float* outputscale, const float* noise, const float* vscale, const float* x, int64_t ldx, float* y,
                  int64_t ldy, int64_t p, const void* pkv, const void* pka, float* part, const int* rangeflag, int64_t ldpart,
                  int64_t row0, int64_t rend);

// ================================================================================================
// Producer / consumer form of the pipelined 3 x f16 matvec ("pc"): k_rbf_pc_apply.
//
// What the h3 kernel above could not fix (PMC, profiles/r02n_*): every wave alternates distance MFMAs -> v_exp -> hi/lo split
// -> contraction MFMAs on its own blocks, 43 % of the wave cycles wait on an instruction dependency, and the matrix pipe is
// busy 68 % of the time although neither it nor the vector issue is saturated.  Here the two waves of a SIMD (w and w + 4 of
// an 8-wave workgroup) have different jobs:
//   * waves 0-3, PRODUCERS: for their pair's 64 rows and the stage's 32 columns, the distance MFMAs, v_exp, hi/lo split of
//     the two 32 x 32 blocks of K, written as ready-made MFMA A fragments (lane-contiguous 16-B packs: conflict-free
//     ds_write_b128 / ds_read_b128) into an LDS ring of three stages;
//   * waves 4-7, CONSUMERS: nothing but the 24 contraction MFMAs of a stage and the 16 fragment reads that feed them
//     (K fragments two groups ahead, probe fragments one group ahead, rolling through the registers their last user freed):
//     a dependency-free MFMA stream beside a dependency-free VALU stream.
// One s_barrier per stage (32 columns) and NO other synchronisation: at time ts the producers write stage ts + 2, the
// consumers contract stage ts and prefetch the first fragments of stage ts + 1; a ring slot is rewritten one full stage
// after its last reader.  Tile images (LDS-DMA, as before): probe image V(t) in three buffers, column operand A(t) in two,
// tile t + 2 requested at the first stage of tile t (consumer time) and waited for before that stage's barrier.
//   who reads what, by time stage (b = stage index = 2 tile + column block):
//     K(b)   written at time b - 2;  read (fragments) at times b - 1 and b;     slot b % 3 rewritten at time b + 1
//     A(t)   requested at time 2t - 5, landed by the end of 2t - 4;  read at times 2t - 3, 2t - 2;    buffer t % 2 refilled from 2t - 1
//     V(t)   requested at time 2t - 4, landed by the end of 2t - 2;  read at times 2t - 1 .. 2t + 1;  buffer t % 3 refilled from 2t + 2
// 256 rows per workgroup (4 pairs x 64 rows): the consumer keeps 64 accumulators + 64 chain masters (kChainTiles) in registers.
// ================================================================================================
#ifndef MFX_PC_PRIO
#define MFX_PC_PRIO 0  // A/B builds: 1 = consumers at s_setprio 1, 2 = producers at s_setprio 1
#endif
#ifndef MFX_PC_DIAG
#define MFX_PC_DIAG 0  // timing diagnostics (WRONG results): 1 = the consumers skip their MFMAs, 2 = the producers skip exp / split / store,
                       // 3 = the producers only take part in the barriers and the DMA, 4 = 3 without the per-stage barriers,
                       // 5 / 6 / 7 = 3 and the consumers read no fragments / only K / only probe fragments
#endif
#ifndef MFX_PC_STAMP
#define MFX_PC_STAMP 0  // tools/pc_model.hip: every wave writes the shader cycles of its main loop to `part` (as int64, [workgroup][wave])
#endif
#ifndef MFX_PC_SWAP
#define MFX_PC_SWAP 0  // A/B builds: 1 = the consumers are waves 0-3 (the older half of the workgroup), the producers waves 4-7
#endif

// entries (2 pr, 2 pr + 1) of a distance block: K' = exp2(arg) (RBF) or the Matern factor, and its hi / lo f16 pieces
// (the same arithmetic as exp_split_pair of the h3 kernel)
template <int KIND>
__device__ __forceinline__ void pc_exp_split_pair(const floatx16& kd, const int pr, const bool diag_blk, const bool neg,
                                                  const int l31, const int lhi, half8 (&ah)[2], half8 (&al)[2]) {
  const int s = pr >> 2, q = (pr & 3) * 2;
  float k0, k1;
  const float d0 = neg ? -kd[8 * s + q] : kd[8 * s + q];
  const float d1 = neg ? -kd[8 * s + q + 1] : kd[8 * s + q + 1];
  if constexpr (KIND == MFX_KERNEL_RBF) {
    if constexpr (kClampRbf) {
      k0 = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(d0, -3.0e38f, kKShift));
      k1 = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(d1, -3.0e38f, kKShift));
    } else {
      k0 = __builtin_amdgcn_exp2f(d0);
      k1 = __builtin_amdgcn_exp2f(d1);
    }
  } else {
    // register r <-> column (r & 3) + 8 (r >> 2) + 4 lhi of the block: zero self-distance on the diagonal block
    const int r0 = 8 * s + q, r1 = r0 + 1;
    const float t0 = (diag_blk && l31 == (r0 & 3) + 8 * (r0 >> 2) + 4 * lhi) ? kEpsC : d0;
    const float t1 = (diag_blk && l31 == (r1 & 3) + 8 * (r1 >> 2) + 4 * lhi) ? kEpsC : d1;
    k0 = matern_from_te<KIND>(t0, 32768.f);
    k1 = matern_from_te<KIND>(t1, 32768.f);
  }
  const half2v h = {(_Float16)k0, (_Float16)k1};  // one v_cvt_pk_f16_f32 (round to nearest)
  float l0, l1;
  const unsigned hb = __builtin_bit_cast(unsigned, h);
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hb), "v"(k0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hb), "v"(k1));
  const half2v l = {(_Float16)l0, (_Float16)l1};
  ah[s][q] = h[0]; ah[s][q + 1] = h[1];
  al[s][q] = l[0]; al[s][q + 1] = l[1];
}

template <int DPAD>
struct PcSmem {
  using Tile = RbfTileH3<DPAD, 2, 64, false>;
  static constexpr int kVBytes = 2 * 8 * Tile::P * 16;   // probe image of a tile: 8 hi rows, 8 lo rows of 64 packs (16 KiB)
  static constexpr int kABytes = 64 * Tile::AROW * 2;    // column operand of a tile
  static constexpr int kPairBytes = 2 * 4 * 1024;        // a pair's share of a stage: 2 row blocks x (hi s0, hi s1, lo s0, lo s1) x 1 KiB
  static constexpr int kStageBytes = 4 * kPairBytes;
  static constexpr int kOffA = 3 * kVBytes, kOffRing = kOffA + 2 * kABytes, kTotal = kOffRing + 3 * kStageBytes;
  static_assert(kVBytes % 1024 == 0 && kABytes % 1024 == 0, "tile images are whole 1-KiB DMA pieces");
};

template <int DPAD, bool VEC4, int KIND>
__global__ __launch_bounds__(512, 1) void k_rbf_pc_apply(const float* __restrict__ xs, const float* __restrict__ sq, int64_t n,
                                                         const float* __restrict__ outputscale, const float* __restrict__ noise,
                                                         const float* __restrict__ vscale, const float* __restrict__ x,
                                                         int64_t ldx, float* __restrict__ y, int64_t ldy, int64_t p,
                                                         const uintx4* __restrict__ pkv, const uintx4* __restrict__ pka,
                                                         float* __restrict__ part, const int* __restrict__ rangeflag,
                                                         int64_t ldpart, int64_t row0, int64_t rend) {
  if (rangeflag && *rangeflag != 0) return;  // f16 range guard: the fp32-distance launch queued behind this one does the work
  using S = PcSmem<DPAD>;
  using Tile = typename S::Tile;
  constexpr int KD = Tile::KD, NKD = Tile::NKD, AROW = Tile::AROW;
  constexpr int kVB = S::kVBytes, kAB = S::kABytes;
  extern __shared__ __attribute__((aligned(16))) char pc_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int pw = wid & 3;  // the pair: producer wave pw and consumer wave pw + 4 sit on one SIMD
  const int64_t i_pair = row0 + (int64_t)blockIdx.x * 256 + pw * 64;
  const int64_t b0 = (int64_t)blockIdx.y * 64;
  const int64_t ntile_all = (n + 63) / 64;
  // gridDim.z > 1: column split (see rbf_split_count): this workgroup sweeps tiles [t_first, t_first + ntl)
  const int64_t t_first = ntile_all * blockIdx.z / gridDim.z;
  const int ntl = (int)(ntile_all * (blockIdx.z + 1) / gridDim.z - t_first);

  // LDS-DMA of the images of local tile tl: wave w copies the 1-KiB pieces w, w + 8, ...  A request must not be waited for in
  // the stage that makes it (a stage is ~900 cycles, an L2 miss more): the column operand A(t + 2) is requested at time 2t - 1
  // and the probe image V(t + 2) at time 2t; at the end of every EVEN time 2t each wave waits for all of its pieces but the two
  // of V(t + 2) it has just requested (vmcnt(2): the counter retires in order) -- that covers A(t + 2), first read at 2t + 1, and
  // V(t + 1), first read at 2t + 1.
  static_assert(kVB == 16 * 1024, "two probe-image pieces per wave and tile: the counted waits below rely on it");
  auto issue_v = [&](int tl) {
    const char* vsrc = reinterpret_cast<const char*>(pkv) + ((int64_t)blockIdx.y * ntile_all + t_first + tl) * kVB;
    char* vdst = pc_smem + (tl % 3) * kVB;
#pragma unroll
    for (int c = 0; c < 2; ++c) glds16(vsrc + (wid + 8 * c) * 1024 + lane * 16, vdst + (wid + 8 * c) * 1024);
  };
  auto issue_a = [&](int tl) {
    const char* asrc = reinterpret_cast<const char*>(pka) + (t_first + tl) * kAB;
    char* adst = pc_smem + S::kOffA + (tl & 1) * kAB;
#pragma unroll
    for (int c = 0; c < (kAB / 1024 + 7) / 8; ++c) {
      const int piece = wid + 8 * c;
      if (piece < kAB / 1024) glds16(asrc + piece * 1024 + lane * 16, adst + piece * 1024);
    }
  };
  for (int tl = 0; tl < 2; ++tl)
    if (tl < ntl) {
      issue_a(tl);
      issue_v(tl);
    }

  if ((wid < 4) != (MFX_PC_SWAP != 0)) {
    // ------------------------------------------------------------------------------------------ producer
    if (MFX_PC_PRIO == 2) __builtin_amdgcn_s_setprio(1);
    // B operand of the distance product, resident: the f16 image [Bh | Bl | Bh | 0] of [x_i, 1, |x_i|^2], negated for the
    // odd row block (with the columns' sign the distance of block (jb, mi) comes out as (-1)^(jb + mi) t: DESIGN.md §3.2)
    half8 bih[2][NKD];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      int64_t i = i_pair + mi * 32 + l31;
      if (i >= rend) i = rend - 1;
#pragma unroll
      for (int q = 0; q < NKD; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int slot = q * 16 + lhi * 8 + e;
          const int comp = slot / KD, kk = slot % KD;
          float v = 0.f;
          if (comp < 3) v = (kk < DPAD) ? xs[i * DPAD + kk] : (kk == DPAD ? 1.f : sq[i]);
          float hi, lo;
          split_hi_lo(v, hi, lo);
          const float w = comp == 1 ? lo : hi;
          bih[mi][q][e] = (_Float16)((mi & 1) ? -w : w);
        }
    }
    auto load_ajs = [&](int abuf, int jb, half8 (&a)[NKD]) {
      const _Float16* img = reinterpret_cast<const _Float16*>(pc_smem + S::kOffA + abuf * kAB);
#pragma unroll
      for (int q = 0; q < NKD; ++q) a[q] = *reinterpret_cast<const half8*>(img + (jb * 32 + l31) * AROW + q * 16 + lhi * 8);
    };
    auto dist = [&](const half8 (&a)[NKD], int mi, floatx16& kd) {
#pragma unroll
      for (int r = 0; r < 16; ++r) kd[r] = 0.f;
#pragma unroll
      for (int q = 0; q < NKD; ++q) kd = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q], bih[mi][q], kd, 0, 0, 0);
    };
    auto produce = [&](const floatx16& kd, char* dst, const bool neg, const bool diag) {
      if (MFX_PC_DIAG >= 2) {
        asm volatile("" ::"v"(kd));
        return;
      }
      half8 ah[2], al[2];
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) pc_exp_split_pair<KIND>(kd, pr, diag, neg, l31, lhi, ah, al);
      *reinterpret_cast<half8*>(dst) = ah[0];
      *reinterpret_cast<half8*>(dst + 1024) = ah[1];
      *reinterpret_cast<half8*>(dst + 2048) = al[0];
      *reinterpret_cast<half8*>(dst + 3072) = al[1];
    };
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): my pieces of tiles 0 and 1
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    floatx16 kd0, kd1;
    {
      half8 a0[NKD];
      load_ajs(0, 0, a0);
      dist(a0, 0, kd0);
      dist(a0, 1, kd1);
    }
    char* const ring = pc_smem + S::kOffRing + pw * S::kPairBytes + lane * 16;
    int slot = 0;  // production stage % 3
    long long stamp0 = 0;
    if (MFX_PC_STAMP) stamp0 = __builtin_readcyclecounter();
    for (int tl = 0; tl < ntl; ++tl) {
      const bool diag = KIND != MFX_KERNEL_RBF && (t_first + tl) * 64 == i_pair;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        // time stage ts = 2 tl + jb - 2; the consumers are one tile behind.  Requests: V(t + 2) at time 2t, A(t + 2) at time 2t - 1
        const bool req_v = jb == 0 && tl >= 1 && tl + 1 < ntl;
        if (req_v) issue_v(tl + 1);
        if (jb == 1 && tl + 2 < ntl) issue_a(tl + 2);
        half8 an[NKD];  // column operand of the next production stage (a stage past the end reads valid, unused LDS)
        load_ajs((jb == 0 ? tl : tl + 1) & 1, 1 - jb, an);
        char* dst = ring + slot * S::kStageBytes;
        __builtin_amdgcn_sched_barrier(0);
        produce(kd0, dst, (jb & 1) != 0, diag && jb == 0);
        __builtin_amdgcn_sched_barrier(0);
        floatx16 kn0, kn1;
        if (MFX_PC_DIAG >= 3) {
          kn0 = kd0;
          kn1 = kd1;
        } else {
          dist(an, 0, kn0);
          dist(an, 1, kn1);
        }
        __builtin_amdgcn_sched_barrier(0);
        produce(kd1, dst + 4096, ((jb + 1) & 1) != 0, diag && jb == 1);
        __builtin_amdgcn_sched_barrier(0);
        kd0 = kn0;
        kd1 = kn1;
        slot = slot == 2 ? 0 : slot + 1;
        // my K fragments are in LDS (lgkmcnt(0)); even times: my DMA pieces but the two just requested have landed (vmcnt(2))
        // (no request made in this stage -- first tile, tail: the counter holds nothing that may stay in flight)
        if (jb == 0) {
          if (req_v) __builtin_amdgcn_s_waitcnt(0x0072); else __builtin_amdgcn_s_waitcnt(0x0070);
        } else {
          __builtin_amdgcn_s_waitcnt(0xC07F);
        }
        if (MFX_PC_DIAG != 4) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    if (MFX_PC_STAMP && lane == 0)
      reinterpret_cast<long long*>(part)[((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 8 + wid] = __builtin_readcyclecounter() - stamp0;
    __builtin_amdgcn_s_barrier();  // the consumers' last two stages
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // no LDS-DMA of mine is left in flight
    return;
  }
  // -------------------------------------------------------------------------------------------- consumer
  if (MFX_PC_PRIO == 1) __builtin_amdgcn_s_setprio(1);
  floatx16 acc[2][2], mst[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[mi][nb][r] = 0.f;
        mst[mi][nb][r] = 0.f;
      }
  const char* const ringc = pc_smem + S::kOffRing + pw * S::kPairBytes + lane * 16;
  // fragments: kq[g] = (hi, lo) A fragments of the K block that MFMA group g of a stage uses; vq[s] = probe fragments of k-step s
  half8 kq[4][2], vq[2][2][2];
  auto read_k = [&](half8 (&k)[2], const char* stage_base, int mi, int s) {
    k[0] = *reinterpret_cast<const half8*>(stage_base + mi * 4096 + s * 1024);
    k[1] = *reinterpret_cast<const half8*>(stage_base + mi * 4096 + 2048 + s * 1024);
  };
  auto read_v = [&](half8 (&v)[2][2], const char* vimg, int jb, int s) {
    const int row = (jb * 2 + s) * 2 + lhi;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      v[nb][0] = *reinterpret_cast<const half8*>(vimg + row * 1024 + (nb * 32 + l31) * 16);
      v[nb][1] = *reinterpret_cast<const half8*>(vimg + 8192 + row * 1024 + (nb * 32 + l31) * 16);
    }
  };
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): my pieces of tiles 0 and 1
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();  // time -2: the producers write stage 0
  asm volatile("" ::: "memory");
  if (2 < ntl) issue_a(2);     // time -1
  read_k(kq[0], ringc, 0, 0);  // the first fragments of stage 0 (its group order is s = 0, 1, 1, 0)
  read_k(kq[1], ringc, 0, 1);
  read_v(vq[0], pc_smem, 0, 0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // one stage = four groups of six MFMAs: (mi 0, s a), (mi 0, s b), (mi 1, s b), (mi 1, s a) with a = jb, b = 1 - jb: the probe
  // fragments of k-step b are free after group 2, those of k-step a after group 3, and the NEXT stage (a' = b) starts on k-step b
  // the fragment reads of a group go out one or two per MFMA gap (a burst of six behind the first MFMA kept the next MFMA from
  // issuing while the LDS queue took them: PMC, SQ_WAIT_INST_LDS 20 % of the wave cycles, matrix pipe 50 % busy)
  auto group = [&](const int mi, const half8 (&k)[2], const half8 (&v)[2][2], auto&& reads) {
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      const int nb = m / 3, w = m % 3;
      __builtin_amdgcn_sched_barrier(0);
      if (MFX_PC_DIAG != 1)
        acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w == 2 ? k[1] : k[0], w == 1 ? v[nb][1] : v[nb][0], acc[mi][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      reads(m);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto read_k1 = [&](half8& k, const char* stage_base, int mi, int s, int hl) {
    if (MFX_PC_DIAG == 5 || MFX_PC_DIAG == 7) return;
    k = *reinterpret_cast<const half8*>(stage_base + mi * 4096 + hl * 2048 + s * 1024);
  };
  auto read_v1 = [&](half8& v, const char* vimg, int jb, int s, int nb, int hl) {
    if (MFX_PC_DIAG == 5 || MFX_PC_DIAG == 6) return;
    const int row = (jb * 2 + s) * 2 + lhi;
    v = *reinterpret_cast<const half8*>(vimg + hl * 8192 + row * 1024 + (nb * 32 + l31) * 16);
  };
  auto stage = [&](auto jb_tag, const char* kthis, const char* knext, const char* vthis, const char* vnext) {
    constexpr int a = decltype(jb_tag)::value, b = 1 - a;
    // group 1 consumes the probe fragments in the order (nb 0 hi, nb 0 lo, nb 1 hi, nb 1 lo): the order they are requested in
    group(0, kq[0], vq[a], [&](int m) {
      if (m == 0) { read_v1(vq[b][0][0], vthis, a, b, 0, 0); read_v1(vq[b][0][1], vthis, a, b, 0, 1); }
      if (m == 1) { read_v1(vq[b][1][0], vthis, a, b, 1, 0); read_v1(vq[b][1][1], vthis, a, b, 1, 1); }
      if (m == 2) read_k1(kq[2][0], kthis, 1, b, 0);
      if (m == 3) read_k1(kq[2][1], kthis, 1, b, 1);
    });
    group(0, kq[1], vq[b], [&](int m) {
      if (m == 0) read_k1(kq[3][0], kthis, 1, a, 0);
      if (m == 1) read_k1(kq[3][1], kthis, 1, a, 1);
    });
    group(1, kq[2], vq[b], [&](int m) {
      if (m == 0) read_k1(kq[0][0], knext, 0, b, 0);
      if (m == 1) read_k1(kq[0][1], knext, 0, b, 1);
    });
    group(1, kq[3], vq[a], [&](int m) {
      if (m == 0) { read_k1(kq[1][0], knext, 0, a, 0); read_k1(kq[1][1], knext, 0, a, 1); }
      if (m == 1) { read_v1(vq[b][0][0], vnext, b, b, 0, 0); read_v1(vq[b][0][1], vnext, b, b, 0, 1); }
      if (m == 2) read_v1(vq[b][1][0], vnext, b, b, 1, 0);
      if (m == 3) read_v1(vq[b][1][1], vnext, b, b, 1, 1);
    });
  };
  int slot = 0;  // stage % 3
  long long stamp0 = 0;
  if (MFX_PC_STAMP) stamp0 = __builtin_readcyclecounter();
  for (int tc = 0; tc < ntl; ++tc) {
    if (tc > 0 && (tc % kChainTiles) == 0) {  // chain fold: masters += accumulators, accumulators restart from zero
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            mst[mi][nb][r] += acc[mi][nb][r];
            acc[mi][nb][r] = 0.f;
          }
    }
    const bool req_v = tc + 2 < ntl;
    if (req_v) issue_v(tc + 2);
    const int s1 = slot == 2 ? 0 : slot + 1, s2 = s1 == 2 ? 0 : s1 + 1;
    const char* const vt = pc_smem + (tc % 3) * kVB;
    const char* const vt1 = pc_smem + ((tc + 1) % 3) * kVB;
    stage(std::integral_constant<int, 0>{}, ringc + slot * S::kStageBytes, ringc + s1 * S::kStageBytes, vt, vt);
    // vmcnt(2): my pieces of A(tc + 2) and V(tc + 1); those of V(tc + 2) stay in flight (tail, no request: all of them)
    if (req_v) __builtin_amdgcn_s_waitcnt(0x0F72); else __builtin_amdgcn_s_waitcnt(0x0F70);
    if (MFX_PC_DIAG != 4) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tc + 3 < ntl) issue_a(tc + 3);
    stage(std::integral_constant<int, 1>{}, ringc + s1 * S::kStageBytes, ringc + s2 * S::kStageBytes, vt, vt1);
    // every read of this stage was requested at least two MFMAs ago: waiting here is free, and the loop header then sees an empty
    // LDS queue on both of its edges, so the compiler's waits inside the stages are counted ones (it emitted lgkmcnt(0) otherwise)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (MFX_PC_DIAG != 4) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    slot = s2;
  }
  if (MFX_PC_STAMP && lane == 0)
    reinterpret_cast<long long*>(part)[((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 8 + wid] = __builtin_readcyclecounter() - stamp0;
  __builtin_amdgcn_s_waitcnt(0x0F70);  // no LDS-DMA of mine is left in flight
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nb][r] += mst[mi][nb][r];
  const float s = outputscale[0], nz = gridDim.z > 1 ? 0.f : noise[0];
  float* yout = gridDim.z > 1 ? part + (int64_t)blockIdx.z * p * ldpart : y;
  const int64_t ldo = gridDim.z > 1 ? ldpart : ldy;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int64_t b = b0 + nb * 32 + l31;
      if (b >= p) continue;
      const float sb = s * vscale[2 * b + 1] * (1.f / 32768.f);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t i = i_pair + mi * 32 + 8 * g + 4 * lhi;
        if (VEC4 && i + 3 < rend) {
          const float4 xv = *reinterpret_cast<const float4*>(x + b * ldx + i);
          float4 o;
          o.x = fmaf(sb, acc[mi][nb][4 * g + 0], nz * xv.x);
          o.y = fmaf(sb, acc[mi][nb][4 * g + 1], nz * xv.y);
          o.z = fmaf(sb, acc[mi][nb][4 * g + 2], nz * xv.z);
          o.w = fmaf(sb, acc[mi][nb][4 * g + 3], nz * xv.w);
          *reinterpret_cast<float4*>(yout + b * ldo + (i - row0)) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (i + e < rend) yout[b * ldo + (i - row0) + e] = fmaf(sb, acc[mi][nb][4 * g + e], nz * x[b * ldx + i + e]);
        }
      }
    }
}


int64_t rbf_pc_smem_bytes(int dpad) {
  switch (dpad) {
    case 4: return PcSmem<4>::kTotal;
    case 8: return PcSmem<8>::kTotal;
    case 12: return PcSmem<12>::kTotal;
    default: return -1;
  }
}

template <int DPAD, int KIND>
static int pc_launch_dk(bool vec4, dim3 grid, hipStream_t stream, const float* xs, const float* sq, int64_t n,
                        const float* outputscale, const float* noise, const float* vscale, const float* x, int64_t ldx,
                        float* y, int64_t ldy, int64_t p, const void* pkv, const void* pka, float* part,
                        const int* rangeflag, int64_t ldpart, int64_t row0, int64_t rend) {
  constexpr int kSm = PcSmem<DPAD>::kTotal;
  static_assert(kSm <= 160 * 1024, "LDS budget of the producer / consumer kernel");
#define MFX_PC_LAUNCH(V4)                                                                                                  \
  {                                                                                                                        \
    MFX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rbf_pc_apply<DPAD, V4, KIND>),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kSm));                                   \
    k_rbf_pc_apply<DPAD, V4, KIND><<<grid, 512, kSm, stream>>>(xs, sq, n, outputscale, noise, vscale, x, ldx, y, ldy, p,   \
                                                               static_cast<const uintx4*>(pkv), static_cast<const uintx4*>(pka), \
                                                               part, rangeflag, ldpart, row0, rend);                       \
  }
  if (vec4) MFX_PC_LAUNCH(true) else MFX_PC_LAUNCH(false)
#undef MFX_PC_LAUNCH
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

template <int DPAD>
static int pc_launch_d(int kind, bool vec4, dim3 grid, hipStream_t stream, const float* xs, const float* sq, int64_t n,
                       const float* outputscale, const float* noise, const float* vscale, const float* x, int64_t ldx,
                       float* y, int64_t ldy, int64_t p, const void* pkv, const void* pka, float* part, const int* rangeflag,
                       int64_t ldpart, int64_t row0, int64_t rend) {
#define MFX_PC_ARGS vec4, grid, stream, xs, sq, n, outputscale, noise, vscale, x, ldx, y, ldy, p, pkv, pka, part, rangeflag, ldpart, row0, rend
  switch (kind) {
    case MFX_KERNEL_RBF: return pc_launch_dk<DPAD, MFX_KERNEL_RBF>(MFX_PC_ARGS);
    case MFX_KERNEL_MATERN12: return pc_launch_dk<DPAD, MFX_KERNEL_MATERN12>(MFX_PC_ARGS);
    case MFX_KERNEL_MATERN32: return pc_launch_dk<DPAD, MFX_KERNEL_MATERN32>(MFX_PC_ARGS);
    default: set_error("unknown kernel_fn %d", kind); return MFX_ERR_INVALID;
  }
}

int rbf_pc_launch(int dpad, int kind, bool vec4, dim3 grid, hipStream_t stream, const float* xs, const float* sq, int64_t n,
                  const float* outputscale, const float* noise, const float* vscale, const float* x, int64_t ldx, float* y,
                  int64_t ldy, int64_t p, const void* pkv, const void* pka, float* part, const int* rangeflag, int64_t ldpart,
                  int64_t row0, int64_t rend) {
  switch (dpad) {
    case 4: return pc_launch_d<4>(kind, MFX_PC_ARGS);
    case 8: return pc_launch_d<8>(kind, MFX_PC_ARGS);
    case 12: return pc_launch_d<12>(kind, MFX_PC_ARGS);
    default: set_error("producer / consumer Gram matvec supports d <= 12"); return MFX_ERR_UNSUPPORTED;
  }
#undef MFX_PC_ARGS
}

}  // namespace mfx
