// libmfx: the matrix-core Gram matvec (3 x f16 split) as ONE wave per SIMD -- "fat waves", gfx950.
//   W[i][b] = s * sum_j K(x_i, x_j) V[j][b] + noise V[i][b]   (util/gp_util.py:160-176,225-226,536-541 of the reference)
//
// Why (round-3 measurements): with two waves per SIMD -- the same-program kernel k_rbf_mfma_apply_h3 as well as
// the producer / consumer split (round 3; tools/experiments/pc_matvec/) -- the matrix pipe idles a third of the time: a wave's MFMA and another wave's
// VALU instruction compete for the SIMD's one vector issue port, and neither wave can see the other's schedule.  ONE wave that
// owns the SIMD can: tools/valu_mix_bench.hip runs 14 MFMAs plus the whole exp / hi-lo split chain of a 32 x 32 block in 470
// cycles (452 for the MFMAs alone) when every VALU instruction is placed behind a chosen MFMA so that
//   * no MFMA gap carries more than ~24 issue cycles of VALU work (v_exp_f32 = 8, the others 4) and at most two v_exp,
//   * no instruction follows its producer within one gap (exp -> cvt hi -> fma_mix -> cvt lo, one gap apart each),
// against 532 for the "pair by pair" order of the h3 kernel.  That placement is the table kSplit below.
//
// Shapes: RBF, d <= 16 (two distance MFMAs per block for d <= 8, three for d = 9 .. 12 -- BASELINE config 2 has d = 9 --, four for d = 13 .. 16),
// any number of vectors.
// Geometry: 256-thread workgroups = 4 waves x 128 rows (four 32-row blocks mi) x 64 probes; 512 registers per lane: 128
// accumulators + 128 chain masters (kChainTiles) live in the accumulation registers, the rest in VGPRs.  A 64-column tile is
// 8 blocks (jb, mi); per block 12 contraction MFMAs, 2 distance MFMAs (for the block after next) and 48 VALU instructions.
// The block pipeline runs ACROSS tiles (no per-tile prologue): during block b the wave contracts K_b, splits K_{b+1} (its last
// two pairs finish in the first gaps of block b + 1, before the MFMAs that read them) and computes the distances of K_{b+2}.
// Tile images (pre-packed by k_pack_tiles, LDS-DMA): two buffers; everything a tile contributes is in registers after its
// fifth block, so ONE barrier per tile (behind block 4: "tile t + 1 has landed, tile t's buffer is free") is all the
// synchronisation there is, and the request for tile t + 2 (made in block 5) has a whole tile to land.
#include <type_traits>

#include "mfx_internal.h"
#include "mfx_rbf_common.h"

namespace mfx {

// MFMA slots of a block and the placement of the split chain behind them, by the number NB of 32-probe blocks of a chunk.
//   NB = 2 (33..64 vectors): 14 slots -- 0-9 contraction c0-c9, 10 distance d0, 11 c10, 12 d1, 13 c11;
//   NB = 1 (<= 32 vectors):   8 slots -- c0 c1 c2 d0 c3 d1 c4 c5 (the distances two contraction MFMAs before the block ends: their
//                             first reader, a v_exp behind slot 0 of the next block, then follows three more MFMAs).
// Table entry = slot + kSlots * lag: lag 1 = in the NEXT block (only pairs of the second k-step, whose consumers come later in
// that block).  Steps of the chain of a pair of entries (2p, 2p + 1):  e: v_exp_f32 of one entry;  h: hi pair = v_cvt_pk_f16_f32;
// m: lo half = f16(k - hi) written straight into its half of the packed lo register (v_fma_mixlo_f16 for the even entry, then
// v_fma_mixhi_f16 for the odd one: 40 VALU instructions per block instead of 48 with v_fma_mix_f32 x 2 + v_cvt_pk; bit-identical).
//   NB = 2: issue cycles per slot (exp 8, others 4) 16 12 16 12 16 20 16 16 16 16 20 16 16 16 -- under the ~24 that hide behind an
//           MFMA, and no step follows its producer within one slot;
//   NB = 1: the same 40 instructions behind 8 MFMAs: 28 issue cycles in EVERY slot (pair p: both exps behind slot p, hi p + 1,
//           mixlo p + 2, mixhi p + 3) -- the block is VALU-bound (~290 cycles against 256 of matrix pipe), evenly.
struct FatSplit {
  int e[16], h[8], m[16];
};
// The chain tables by chunk width; the slot ORDER by chunk width and by the number NKD of distance MFMAs of a block
// (NKD = 2: d <= 8; NKD = 3: d = 9 .. 12 -- BASELINE config 2 has d = 9; NKD = 4: d = 13 .. 16).  With every further distance MFMA a block has one more slot and the
// tables stay as they are: an entry >= kSlots still means "in the next block", one slot later than before -- every consumer
// (the contraction MFMAs of k-step 1) still comes later.
constexpr FatSplit kFatSplit2 = {{0, 1, 2, 3, 4, 5, 5, 6, 7, 8, 9, 10, 10, 11, 12, 13},
                                 {2, 4, 6, 7, 9, 11, 12, 14},
                                 {3, 4, 5, 6, 7, 8, 8, 9, 10, 11, 12, 13, 13, 14, 15, 16}};
constexpr FatSplit kFatSplit1 = {{0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7},
                                 {1, 2, 3, 4, 5, 6, 7, 8},
                                 {2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10}};
template <int NB, int NKD>
struct FatPlan;
template <>
struct FatPlan<2, 2> {
  static constexpr int kSlots = 14;
  static constexpr FatSplit kSplit = kFatSplit2;
  // slot -> distance step q (or -1), slot -> contraction MFMA m (or -1), contraction MFMA -> slot
  static constexpr int dist_q(int slot) { return slot == 10 ? 0 : (slot == 12 ? 1 : -1); }
  static constexpr int contr_m(int slot) { return slot < 10 ? slot : (slot == 11 ? 10 : (slot == 13 ? 11 : -1)); }
  static constexpr int slot_of(int m) { return m < 10 ? m : (m == 10 ? 11 : 13); }
};
template <>
struct FatPlan<1, 2> {
  static constexpr int kSlots = 8;
  static constexpr FatSplit kSplit = kFatSplit1;
  static constexpr int dist_q(int slot) { return slot == 3 ? 0 : (slot == 5 ? 1 : -1); }
  static constexpr int contr_m(int slot) { return slot < 3 ? slot : (slot == 4 ? 3 : (slot >= 6 ? slot - 2 : -1)); }
  static constexpr int slot_of(int m) { return m < 3 ? m : (m == 3 ? 4 : m + 2); }
};
// NKD = 3, 64 vectors: 15 slots -- c0-c8, d0, c9, d1, c10, d2, c11 (the last distance MFMA again two MFMAs before the first v_exp on its block)
template <>
struct FatPlan<2, 3> {
  static constexpr int kSlots = 15;
  static constexpr FatSplit kSplit = kFatSplit2;
  static constexpr int dist_q(int slot) { return slot == 9 ? 0 : (slot == 11 ? 1 : (slot == 13 ? 2 : -1)); }
  static constexpr int contr_m(int slot) { return slot < 9 ? slot : (slot == 10 ? 9 : (slot == 12 ? 10 : (slot == 14 ? 11 : -1))); }
  static constexpr int slot_of(int m) { return m < 9 ? m : (m == 9 ? 10 : (m == 10 ? 12 : 14)); }
};
// NKD = 3, <= 32 vectors: 9 slots -- c0 c1 d0 c2 d1 c3 d2 c4 c5
template <>
struct FatPlan<1, 3> {
  static constexpr int kSlots = 9;
  static constexpr FatSplit kSplit = kFatSplit1;
  static constexpr int dist_q(int slot) { return slot == 2 ? 0 : (slot == 4 ? 1 : (slot == 6 ? 2 : -1)); }
  static constexpr int contr_m(int slot) { return slot < 2 ? slot : (slot == 3 ? 2 : (slot == 5 ? 3 : (slot >= 7 ? slot - 3 : -1))); }
  static constexpr int slot_of(int m) { return m < 2 ? m : (m == 2 ? 3 : (m == 3 ? 5 : m + 3)); }
};

// NKD = 4 (d = 13 .. 16), 64 vectors: 16 slots -- c0-c7, d0, c8, d1, c9, d2, c10, d3, c11;  <= 32 vectors: 10 slots -- c0 d0 c1 d1 c2 d2 c3 d3 c4 c5
template <>
struct FatPlan<2, 4> {
  static constexpr int kSlots = 16;
  static constexpr FatSplit kSplit = kFatSplit2;
  static constexpr int dist_q(int slot) { return (slot >= 8 && slot <= 14 && (slot & 1) == 0) ? (slot - 8) / 2 : -1; }
  static constexpr int contr_m(int slot) { return slot < 8 ? slot : ((slot & 1) ? 8 + (slot - 9) / 2 : -1); }
  static constexpr int slot_of(int m) { return m < 8 ? m : 9 + 2 * (m - 8); }
};
template <>
struct FatPlan<1, 4> {
  static constexpr int kSlots = 10;
  static constexpr FatSplit kSplit = kFatSplit1;
  static constexpr int dist_q(int slot) { return (slot >= 1 && slot <= 7 && (slot & 1) == 1) ? (slot - 1) / 2 : -1; }
  static constexpr int contr_m(int slot) { return slot == 0 ? 0 : (slot <= 6 && (slot & 1) == 0 ? slot / 2 : (slot >= 8 ? slot - 4 : -1)); }
  static constexpr int slot_of(int m) { return m == 0 ? 0 : (m <= 3 ? 2 * m : m + 4); }
};

template <int DPAD, int NB>
struct FatSmem {
  static constexpr int KD = DPAD + 2;
  static constexpr int NKD = (3 * KD + 15) / 16;
  static constexpr int AROW = NKD * 16 + 8;
  static constexpr int kABytes = 64 * AROW * 2;  // column operand of a tile (k_pack_tiles' pka)
  static constexpr int kVRow = NB * 32 * 16;      // one image row: the 16-B packs of the chunk's NB * 32 probes
  static constexpr int kVBytes = 2 * 8 * kVRow;   // probe image of a tile: 8 hi rows, then 8 lo rows
  static constexpr int kTile = kABytes + kVBytes;
  static constexpr int kTotal = 2 * kTile;
  static_assert(kABytes % 1024 == 0, "tile images are whole 1-KiB DMA pieces");
};

template <int DPAD, bool VEC4, int NB>
__global__ __launch_bounds__(256, 1) void k_rbf_fat_apply(const float* __restrict__ xs, const float* __restrict__ sq, int64_t n,
                                                          const float* __restrict__ outputscale, const float* __restrict__ noise,
                                                          const float* __restrict__ vscale, const float* __restrict__ x,
                                                          int64_t ldx, float* __restrict__ y, int64_t ldy, int64_t p,
                                                          const uintx4* __restrict__ pkv, const uintx4* __restrict__ pka,
                                                          float* __restrict__ part, const int* __restrict__ rangeflag,
                                                          int64_t ldpart, int64_t row0, int64_t rend) {
  if (rangeflag && *rangeflag != 0) return;  // f16 range guard: the fp32-distance launch queued behind this one does the work
  using S = FatSmem<DPAD, NB>;
  constexpr int KD = S::KD, NKD = S::NKD, AROW = S::AROW;
  static_assert(NKD >= 2 && NKD <= 4, "two to four distance MFMAs per block (d <= 16)");
  using Plan = FatPlan<NB, NKD>;
  constexpr int kSlots = Plan::kSlots;
  extern __shared__ __attribute__((aligned(16))) char fat_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int64_t i_wave = row0 + (int64_t)blockIdx.x * 512 + wid * 128;
  const int64_t b0 = (int64_t)blockIdx.y * (NB * 32);
  const int64_t ntile_all = (n + 63) / 64;
  // gridDim.z > 1: column split (rbf_split_count): this workgroup sweeps tiles [t_first, t_first + ntl)
  const int64_t t_first = ntile_all * blockIdx.z / gridDim.z;
  const int ntl = (int)(ntile_all * (blockIdx.z + 1) / gridDim.z - t_first);

  // an "a"-constrained operand keeps the function from being marked amdgpu-no-agpr: the contraction MFMAs are then selected in
  // their AGPR form and the 256 accumulator / master registers do not compete with the VGPR working set
  float agpr_seed = 0.f;
  asm volatile("; accumulators in AGPRs" : "+a"(agpr_seed));

  // LDS-DMA of local tile tl into buffer tl & 1: wave w copies the 1-KiB pieces w, w + 4, ... ([A image | V image])
  auto issue_tile_dma = [&](int tl) {
    const char* asrc = reinterpret_cast<const char*>(pka) + (t_first + tl) * (int64_t)S::kABytes;
    const char* vsrc = reinterpret_cast<const char*>(pkv) + ((int64_t)blockIdx.y * ntile_all + t_first + tl) * S::kVBytes;
    char* dst = fat_smem + (tl & 1) * S::kTile;
#pragma unroll
    for (int c = 0; c < (S::kABytes / 1024 + 3) / 4; ++c) {
      const int piece = wid + 4 * c;
      if (piece < S::kABytes / 1024) glds16(asrc + piece * 1024 + lane * 16, dst + piece * 1024);
    }
#pragma unroll
    for (int c = 0; c < S::kVBytes / 4096; ++c) glds16(vsrc + (wid + 4 * c) * 1024 + lane * 16, dst + S::kABytes + (wid + 4 * c) * 1024);
    static_assert(S::kVBytes % 4096 == 0, "a whole number of probe-image pieces per wave");
  };
  if (0 < ntl) issue_tile_dma(0);
  if (1 < ntl) issue_tile_dma(1);

  // B operand of the distance product, resident: the f16 image [Bh | Bl | Bh | 0] of [x_i, 1, |x_i|^2], negated for the odd row
  // blocks (with the columns' sign -- odd column blocks are packed negated -- block (jb, mi) yields (-1)^(jb + mi) t: the f16 MFMA's
  // rounding bias enters K with alternating sign)
  half8 bih[4][NKD];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    int64_t i = i_wave + mi * 32 + l31;
    if (i >= rend) i = rend - 1;
    if (i < row0) i = row0;
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
  // each fragment becomes ONE 128-bit accumulation-register tuple here, once: the distance MFMAs below take their operands as "a"
  // tuples, and a fragment the allocator keeps in four scattered registers is copied into place in front of every one of them
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int q = 0; q < NKD; ++q) asm volatile("" : "+a"(bih[mi][q]));
  floatx16 acc[4][NB], mst[4][NB];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[mi][nb][r] = 0.f;
        mst[mi][nb][r] = 0.f;
      }
  acc[0][0][0] = agpr_seed;

  // ---- fragment reads ---------------------------------------------------------------------------------------------------------
  auto read_a = [&](half8 (&a)[NKD], int buf, int jb) {  // column operand of a 32-column block: the distance MFMAs' A fragments
    const _Float16* img = reinterpret_cast<const _Float16*>(fat_smem + buf * S::kTile);
#pragma unroll
    for (int q = 0; q < NKD; ++q) a[q] = *reinterpret_cast<const half8*>(img + (jb * 32 + l31) * AROW + q * 16 + lhi * 8);
  };
  auto read_v1 = [&](half8& v, int buf, int jb, int s, int nb, int hl) {  // one probe fragment (k-step s, probe block nb, hi / lo)
    const int row = (jb * 2 + s) * 2 + lhi;
    v = *reinterpret_cast<const half8*>(fat_smem + buf * S::kTile + S::kABytes + hl * (8 * S::kVRow) + row * S::kVRow + (nb * 32 + l31) * 16);
  };
  // ---- the distance MFMAs: asm, because their 16 results must land in VGPRs with a literal-zero addend while the function's
  //      intrinsic MFMAs are in AGPR form.  The compiler does not see an MFMA there, so the wait states between the MFMA's
  //      write and the first VALU read are kept by construction: the first v_exp on a distance block follows at least two more
  //      MFMAs of this wave (it owns the SIMD's matrix pipe: >= 64 cycles). -----------------------------------------------------
  auto dist_step = [&](floatx16& kd, const half8& a, const half8& b, const bool first) {
    // (A / B operands in accumulation registers: the row operand lives there for good and the column operand is read from LDS
    //  straight into them, so neither takes part in the VGPR working set)
    // No wait states are inserted for an MFMA the compiler cannot see: the operands must be IN PLACE, not copied there (v_accvgpr_mov)
    // in front of the asm -- measured as wrong K blocks when the allocator kept bih scattered.  bih is pinned as tuples above, the
    // column operand arrives by ds_read_b128 (s_waitcnt is data-flow, the compiler keeps that); tests/test_gpu_matvec_kernels.py
    // checks every block position of a tile against the oracle.
    if (first) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(kd) : "a"(a), "a"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(kd) : "a"(a), "a"(b));
  };
  // ---- one micro-step of the split chain of a distance block, in place: t = 0 exp2 of entry i; 1 hi = f16 of pair i; 2 entry i
  //      minus its hi; 3 lo = f16 of pair i.  Register r of the block <-> column (r & 3) + 8 (r >> 2) + 4 lhi; the pair (2p, 2p + 1)
  //      is one packed register of the A fragment of k-step p >> 2. --------------------------------------------------------------
  auto split_op = [&](floatx16& w, half8 (&ah)[2], half8 (&al)[2], unsigned (&lopk)[8], const int t, const int i, const bool neg) {
    if (t == 0) {
      w[i] = __builtin_amdgcn_exp2f(neg ? -w[i] : w[i]);
    } else if (t == 1) {
      const half2v h = {(_Float16)w[2 * i], (_Float16)w[2 * i + 1]};  // one v_cvt_pk_f16_f32, round to nearest
      ah[i >> 2][(i & 3) * 2] = h[0];
      ah[i >> 2][(i & 3) * 2 + 1] = h[1];
    } else {
      const int pr = i >> 1;
      const half2v h = {ah[pr >> 2][(pr & 3) * 2], ah[pr >> 2][(pr & 3) * 2 + 1]};
      const unsigned hb = __builtin_bit_cast(unsigned, h);
      if ((i & 1) == 0) {
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lopk[pr]) : "v"(hb), "v"(w[i]));
      } else {
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lopk[pr]) : "v"(hb), "v"(w[i]));
        const half2v l = __builtin_bit_cast(half2v, lopk[pr]);
        al[pr >> 2][(pr & 3) * 2] = l[0];
        al[pr >> 2][(pr & 3) * 2 + 1] = l[1];
      }
    }
  };
  // all micro-steps of the table that fall behind MFMA slot `slot` (lag 0: on the next block's data, lag 1: on this block's)
  auto split_slot = [&](const int slot, const int lag, floatx16& w, half8 (&ah)[2], half8 (&al)[2], unsigned (&lopk)[8], const bool neg) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (Plan::kSplit.e[i] == slot + kSlots * lag) split_op(w, ah, al, lopk, 0, i, neg);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (Plan::kSplit.h[i] == slot + kSlots * lag) split_op(w, ah, al, lopk, 1, i, neg);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (Plan::kSplit.m[i] == slot + kSlots * lag) split_op(w, ah, al, lopk, 2, i, neg);
  };

  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): my pieces of tiles 0 and 1
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- pipeline state.  Blocks are numbered along the sweep: block = 8 tile + 4 jb + mi. --------------------------------------
  // distance blocks: current (its last pairs still being split), next (being split), next but one.  Three roles, eight blocks per tile:
  // with three buffers that change roles by moves, the tile loop's back edge carries a rotation by 8 mod 3 = 2 positions (~6 v_mov per
  // block, 12 % of the VALU instructions of the VALU-bound <= 32-vector form).  There (registers to spare: 304 of 512) the buffers are a ring
  // of FOUR indexed by the block number -- 4 divides 8, every index is static, no move; with 64 vectors (all 512 registers in use, VALU
  // hidden behind the MFMAs) the three buffers stay.
  constexpr bool kRing4 = NB == 1;
  floatx16 wr[kRing4 ? 4 : 3];
#define MFX_WC(blk) wr[kRing4 ? ((blk) & 3) : 0]
#define MFX_WN(blk) wr[kRing4 ? (((blk) + 1) & 3) : 1]
#define MFX_WD(blk) wr[kRing4 ? (((blk) + 2) & 3) : 2]
  half8 ahc[2], alc[2], ahn[2], aln[2];   // A fragments (hi, lo) x k-step of the current and of the next block
  unsigned lpc[8], lpn[8];                 // packed lo pairs in the making (between the mixlo and the mixhi step)
  half8 vf[2][NB][2];           // probe fragments of the current column block [k-step][probe block][hi / lo]
  half8 ajs[2][NKD];            // column operand of a column block, by parity
  // prologue (once per sweep, not per tile): blocks 0 and 1 by hand
  read_a(ajs[0], 0, 0);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) read_v1(vf[s][nb][hl], 0, 0, s, nb, hl);
#pragma unroll
  for (int q = 0; q < NKD; ++q) dist_step(MFX_WC(0), ajs[0][q], bih[0][q], q == 0);
#pragma unroll
  for (int q = 0; q < NKD; ++q) dist_step(MFX_WN(0), ajs[0][q], bih[1][q], q == 0);
  asm volatile("" : "+v"(MFX_WC(0)));  // ties the first reader of wc behind the last of these MFMAs (two MFMAs after the one that wrote wc)
  // block 0's split as far as the table places it before a block boundary (its lag-1 steps run in the loop, like every block's)
#pragma unroll
  for (int slot = 0; slot < kSlots; ++slot) split_slot(slot, 0, MFX_WC(0), ahc, alc, lpc, false);

  const float sc = outputscale[0];
  int tl = 0;
  for (; tl < ntl; ++tl) {
    const int buf = tl & 1;
    if (tl > 0 && (tl % kChainTiles) == 0) {  // chain fold: masters += accumulators, accumulators restart from zero
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            mst[mi][nb][r] += acc[mi][nb][r];
            acc[mi][nb][r] = 0.f;
          }
    }
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
      const int jb = blk >> 2, mi = blk & 3;
      // block blk + 1 (being split: wn -> ahn / aln) and block blk + 2 (distances: wd); both may lie in the next tile
      const int blk1 = (blk + 1) & 7, blk2 = (blk + 2) & 7;
      const int jb2 = blk2 >> 2, mi2 = blk2 & 3;
      const bool neg_c = ((jb + mi) & 1) != 0, neg_n = (((blk1 >> 2) + (blk1 & 3)) & 1) != 0;
      if (blk == 5 && tl + 2 < ntl) issue_tile_dma(tl + 2);  // (this tile's buffer has been dead since the barrier behind block 4)
      // Invariant at this point: (ahc, alc) hold K_blk except for the table's lag-1 steps (still to run on wc); wn holds the
      // distances of block blk + 1, untouched; ajs[jb2 & 1] holds the column operand that block blk + 2 needs from its first
      // distance slot on.
#pragma unroll
      for (int slot = 0; slot < kSlots; ++slot) {
        __builtin_amdgcn_sched_barrier(0);
        if (Plan::dist_q(slot) >= 0) {
          const int q = Plan::dist_q(slot);
          dist_step(MFX_WD(blk), ajs[jb2 & 1][q], bih[mi2][q], q == 0);
        } else {
          const int m = Plan::contr_m(slot);
          const int s = m / (3 * NB), nb = (m / 3) % NB, w = m % 3;
          acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w == 2 ? alc[s] : ahc[s], w == 1 ? vf[s][nb][1] : vf[s][nb][0],
                                                               acc[mi][nb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        split_slot(slot, 1, MFX_WC(blk), ahc, alc, lpc, neg_c);  // this block's late pairs first (their consumers are a few slots away)
        split_slot(slot, 0, MFX_WN(blk), ahn, aln, lpn, neg_n);
        // fragment reads, one per gap: the probe fragments of the NEXT column block during this one (blocks mi = 1, 2: sixteen
        // reads... eight per block), the column operand of the column block after that in block mi = 1
        if (mi == 1 && slot < NKD) {
          // ajs of column block c + 1 is needed by the distances of block (c + 1, mi 0), issued in block (c, mi 2)
          ajs[(jb + 1) & 1][slot] = *reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(fat_smem + (jb == 1 ? buf ^ 1 : buf) * S::kTile) +
                                                                   (((jb + 1) & 1) * 32 + l31) * AROW + slot * 16 + lhi * 8);
        }
        if (mi == 3) {
          // the probe fragments of the next column block roll in behind the last MFMA of this column block that reads the
          // register they replace: fragment (s, nb, hi) is read by MFMAs 3 NB s + 3 nb and + 2, (s, nb, lo) by + 1
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
              for (int hl = 0; hl < 2; ++hl) {
                const int mlast = 3 * NB * s + 3 * nb + (hl ? 1 : 2);
                if (Plan::slot_of(mlast) == slot) read_v1(vf[s][nb][hl], jb == 1 ? buf ^ 1 : buf, (jb + 1) & 1, s, nb, hl);
              }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (blk == 4) {
        // tile tl + 1 (requested a tile ago) has landed for everybody, and everybody is done with this tile's buffer: its last reads
        // -- the probe fragments of column block 1, rolled in during block 3 -- were consumed by the MFMAs of this block
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      // rotate: next -> current, next-but-one -> next
      ahc[0] = ahn[0]; ahc[1] = ahn[1];
      alc[0] = aln[0]; alc[1] = aln[1];
      if constexpr (!kRing4) {
        wr[0] = wr[1];
        wr[1] = wr[2];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) lpc[q] = lpn[q];
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // no LDS-DMA of mine is left in flight
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nb][r] += mst[mi][nb][r];
  const float nz = gridDim.z > 1 ? 0.f : noise[0];
  float* yout = gridDim.z > 1 ? part + (int64_t)blockIdx.z * p * ldpart : y;
  const int64_t ldo = gridDim.z > 1 ? ldpart : ldy;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int64_t b = b0 + nb * 32 + l31;
      if (b >= p) continue;
      const float sb = sc * vscale[2 * b + 1] * (1.f / 32768.f);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t i = i_wave + mi * 32 + 8 * g + 4 * lhi;
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

#undef MFX_WC
#undef MFX_WN
#undef MFX_WD

template <int DPAD, int NB>
static int fat_launch_dn(bool vec4, dim3 grid, hipStream_t stream, const float* xs, const float* sq, int64_t n,
                         const float* outputscale, const float* noise, const float* vscale, const float* x, int64_t ldx, float* y,
                         int64_t ldy, int64_t p, const void* pkv, const void* pka, float* part, const int* rangeflag,
                         int64_t ldpart, int64_t row0, int64_t rend) {
  constexpr int kSm = FatSmem<DPAD, NB>::kTotal;
#define MFX_FAT_LAUNCH(V4)                                                                                                   \
  k_rbf_fat_apply<DPAD, V4, NB><<<grid, 256, kSm, stream>>>(xs, sq, n, outputscale, noise, vscale, x, ldx, y, ldy, p,         \
                                                            static_cast<const uintx4*>(pkv), static_cast<const uintx4*>(pka), \
                                                            part, rangeflag, ldpart, row0, rend)
  if (vec4) MFX_FAT_LAUNCH(true); else MFX_FAT_LAUNCH(false);
#undef MFX_FAT_LAUNCH
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

int rbf_fat_launch(int dpad, int nb, bool vec4, dim3 grid, hipStream_t stream, const float* xs, const float* sq, int64_t n,
                   const float* outputscale, const float* noise, const float* vscale, const float* x, int64_t ldx, float* y,
                   int64_t ldy, int64_t p, const void* pkv, const void* pka, float* part, const int* rangeflag, int64_t ldpart,
                   int64_t row0, int64_t rend) {
#define MFX_FAT_ARGS vec4, grid, stream, xs, sq, n, outputscale, noise, vscale, x, ldx, y, ldy, p, pkv, pka, part, rangeflag, ldpart, row0, rend
  if (dpad == 4 && nb == 1) return fat_launch_dn<4, 1>(MFX_FAT_ARGS);
  if (dpad == 4 && nb == 2) return fat_launch_dn<4, 2>(MFX_FAT_ARGS);
  if (dpad == 8 && nb == 1) return fat_launch_dn<8, 1>(MFX_FAT_ARGS);
  if (dpad == 8 && nb == 2) return fat_launch_dn<8, 2>(MFX_FAT_ARGS);
  if (dpad == 12 && nb == 1) return fat_launch_dn<12, 1>(MFX_FAT_ARGS);
  if (dpad == 12 && nb == 2) return fat_launch_dn<12, 2>(MFX_FAT_ARGS);
  if (dpad == 16 && nb == 1) return fat_launch_dn<16, 1>(MFX_FAT_ARGS);
  if (dpad == 16 && nb == 2) return fat_launch_dn<16, 2>(MFX_FAT_ARGS);
#undef MFX_FAT_ARGS
  set_error("fat-wave Gram matvec supports d <= 16 and chunks of 32 or 64 vectors");
  return MFX_ERR_UNSUPPORTED;
}

}  // namespace mfx
