// libmfx: the fat-wave Gram matvec (mfx_rbf_fat.hip) with the hi(K) lo(V) product on the block-scaled FP8 matrix pipe -- chunks of
// 33..64 vectors, RBF, d <= 8: BASELINE config 4's matvec.
//   W[i][b] = s * sum_j K(x_i, x_j) V[j][b] + noise V[i][b]   (util/gp_util.py:160-176,225-226,536-541 of the reference)
//
// Why (round-3 measurements): of the three products that emulate fp32 -- hi(K) hi(V), lo(K) hi(V), hi(K) lo(V) -- the
// last one only repairs the probes' own f16 rounding, and with BOTH its operands rounded to 3 mantissa bits the C4 gradient stays where
// it was (four probe sets, worst component 2.6e-5 .. 6.0e-5 against 1.2e-5 .. 6.0e-5; profiles/r03j_*), while the same rounding in
// lo(K) hi(V) breaks the 1e-4 gate.  v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands runs k = 64 in 64 cycles (the f16 MFMA: k = 16
// in 32), and the chip holds 1.85-1.9 GHz under it against 1.5-1.6 under the f16 stream (tools/fp8_mfma_rate.hip): per pair of
// blocks 20 f16 MFMAs + 2 FP8 MFMAs instead of 28 f16 MFMAs.
//
// What changes against k_rbf_fat_apply (everything else -- geometry, block pipeline across tiles, LDS-DMA, one barrier per tile, the
// asm distance MFMAs, the alternating block sign -- is the same):
//   * a block has 10 slots c0 c1 c2 c3 c4 c5 d0 c6 d1 c7 (8 contraction MFMAs (k-step, probe block, hi hi | lo hi) and 2 distance
//     MFMAs); the blocks of the second column block (jb = 1) add 2 slots: one FP8 MFMA per probe block with k = 64 = BOTH column
//     blocks of the tile.  Its A operand -- hi(K) as e4m3, K / 128 -- is made by 8 more VALU instructions per block
//     (v_cvt_scalef32_pk_fp8_f32 of a pair, table q) and kept per row block (32 registers); its B operand -- lo(V) x 32 as e4m3 -- comes
//     pre-packed (k_pack_tiles<..., LO8>) in a 4-KiB image per tile instead of the 8-KiB f16 one.  Fixed scales (2^7, 2^-5 as E8M0
//     bytes): e4m3 spans 17 binades, entries of K below 2^-17 of the outputscale and lo(V) of probe entries below 2^-17 of their
//     row maximum flush to zero -- in this correction product only;
//   * the chain masters (kChainTiles) live in memory (one coalesced 256-byte row per accumulator register and wave, touched once
//     per 128 tiles): their 128 registers are what the FP8 operands and the column / probe fragments now occupy.
#include <type_traits>

#include "mfx_internal.h"
#include "mfx_rbf_common.h"

#ifndef MFX_FAT_DIAG
#define MFX_FAT_DIAG 0  // timing diagnostics (WRONG results): 1 = no LDS-DMA in the tile loop, 2 = also no barrier, 3 = also no fragment reads
#endif

namespace mfx {

typedef int intx8 __attribute__((ext_vector_type(8)));

// Placement of the split chain of the NEXT block behind the MFMA slots of the block that executes it, two tables: T10 for the blocks
// of the first column block (10 slots of 32 cycles: these blocks are bound by vector ISSUE, ~31 cycles of VALU per slot + 8 for the MFMA)
// and T12 for those of the second (12 slots; most of the work sits behind the two 64-cycle FP8 MFMAs, 20 cycles per slot elsewhere: bound
// by the matrix pipe).  Entry = slot + S * lag (S = 10 or 12; lag 1 = in the following block, whatever its kind).  Steps of the pair
// (2p, 2p + 1): e: v_exp_f32 of one entry (8 cycles; the others 4); h: hi pair = v_cvt_pk_f16_f32; m: lo half = f16(k - hi)
// (v_fma_mixlo_f16, then v_fma_mixhi_f16); entries 4 d .. 4 d + 3 as ONE e4m3 dword, from the two f16 hi pairs by integer arithmetic
// (v_cvt_scalef32_pk_fp8_f32 measured at ~15 cycles, tools/valu_cost_bench.hip): qa: v_pk_sub_u16 ... clamp x 2 (round + re-bias the
// exponent: f16 -> e4m3 of K / 128 is a shift of the bit pattern), qb: v_lshrrev_b32 x 2, qc: v_perm_b32.  No step follows its producer
// within a slot.  Deadlines: hi / lo pairs of k-step 0 inside the executing block, hi pairs of k-step 1 by slot 3 and lo pairs by slot 4
// of the next one (readers c4, c5), the e4m3 dwords by its slot 9 (reader: the FP8 MFMA in slot 10).  Found by annealing the per-slot
// loads (fat8_tables.py in this directory).
struct Fat8Table {
  int e[16], h[8], m[16], qa[4], qb[4], qc[4];
};
constexpr Fat8Table kT10 = {{6, 2, 5, 5, 1, 3, 0, 6, 3, 4, 7, 7, 0, 2, 1, 4},
                            {7, 6, 5, 7, 7, 8, 3, 5},
                            {8, 9, 7, 8, 6, 8, 8, 9, 9, 12, 11, 14, 6, 8, 11, 12},
                            {9, 11, 10, 9},
                            {12, 13, 15, 10},
                            {14, 18, 19, 14}};
constexpr Fat8Table kT12 = {{1, 7, 7, 6, 6, 8, 3, 5, 2, 0, 4, 5, 0, 2, 3, 4},
                            {9, 9, 9, 7, 9, 8, 6, 8},
                            {10, 11, 10, 11, 10, 11, 9, 11, 11, 13, 10, 11, 13, 16, 10, 15},
                            {10, 10, 10, 10},
                            {11, 11, 11, 11},
                            {17, 12, 14, 13}};
struct Fat8Plan {
  static constexpr int kSlots = 10;  // f16 slots of every block: c0 c1 c2 c3 c4 c5 d0 c6 d1 c7 (the jb = 1 blocks add two FP8 slots)
  static constexpr int dist_q(int slot) { return slot == 6 ? 0 : (slot == 8 ? 1 : -1); }
  static constexpr int contr_m(int slot) { return slot < 6 ? slot : (slot == 7 ? 6 : (slot == 9 ? 7 : -1)); }
  static constexpr int slot_of(int m) { return m < 6 ? m : (m == 6 ? 7 : 9); }
};

template <int DPAD>
struct Fat8Smem {
  static constexpr int KD = DPAD + 2;
  static constexpr int NKD = (3 * KD + 15) / 16;
  static constexpr int AROW = NKD * 16 + 8;
  static constexpr int kABytes = 64 * AROW * 2;   // column operand of a tile (k_pack_tiles' pka)
  static constexpr int kVRow = 2 * 32 * 16;        // one hi image row: the 16-B packs of the chunk's 64 probes
  static constexpr int kVHiBytes = 8 * kVRow;      // 8 hi rows (f16)
  static constexpr int kVBytes = kVHiBytes + 4096; // + lo as e4m3: [column block][probe block][k half][probe] x 16 B
  static constexpr int kVStride = 2 * kVHiBytes;   // distance of two tiles in pkv (the layout of the f16-lo image is kept)
  static constexpr int kTile = kABytes + kVBytes;
  static constexpr int kTotal = 2 * kTile;
  static_assert(kABytes % 1024 == 0, "tile images are whole 1-KiB DMA pieces");
};

template <int DPAD, bool VEC4>
__global__ __launch_bounds__(256, 1) void k_rbf_fat8_apply(const float* __restrict__ xs, const float* __restrict__ sq, int64_t n,
                                                           const float* __restrict__ outputscale, const float* __restrict__ noise,
                                                           const float* __restrict__ vscale, const float* __restrict__ x,
                                                           int64_t ldx, float* __restrict__ y, int64_t ldy, int64_t p,
                                                           const uintx4* __restrict__ pkv, const uintx4* __restrict__ pka,
                                                           float* __restrict__ part, const int* __restrict__ rangeflag,
                                                           int64_t ldpart, int64_t row0, int64_t rend, float* __restrict__ mstbuf) {
  if (rangeflag && *rangeflag != 0) return;  // f16 range guard: the fp32-distance launch queued behind this one does the work
  constexpr int NB = 2;
  using S = Fat8Smem<DPAD>;
  using Plan = Fat8Plan;
  constexpr int KD = S::KD, NKD = S::NKD, AROW = S::AROW, kSlots = Plan::kSlots;
  static_assert(NKD == 2, "two distance MFMAs per block (d <= 8)");
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
    const char* vsrc = reinterpret_cast<const char*>(pkv) + ((int64_t)blockIdx.y * ntile_all + t_first + tl) * S::kVStride;
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
  floatx16 acc[4][NB];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nb][r] = 0.f;
  acc[0][0][0] = agpr_seed;
  // chain masters (kChainTiles): in memory, one 256-byte row per accumulator register of this wave -- touched once per 128 tiles
  float* const mst = mstbuf + ((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wid) * (128 * 64) + lane;

  // ---- fragment reads ---------------------------------------------------------------------------------------------------------
  auto read_a = [&](half8 (&a)[NKD], int buf, int jb) {  // column operand of a 32-column block: the distance MFMAs' A fragments
    const _Float16* img = reinterpret_cast<const _Float16*>(fat_smem + buf * S::kTile);
#pragma unroll
    for (int q = 0; q < NKD; ++q) a[q] = *reinterpret_cast<const half8*>(img + (jb * 32 + l31) * AROW + q * 16 + lhi * 8);
  };
  auto read_v1 = [&](half8& v, int buf, int jb, int s, int nb) {  // one hi probe fragment (k-step s, probe block nb)
    const int row = (jb * 2 + s) * 2 + lhi;
    v = *reinterpret_cast<const half8*>(fat_smem + buf * S::kTile + S::kABytes + row * S::kVRow + (nb * 32 + l31) * 16);
  };
  // lo(V) of a whole tile as e4m3, the B operand of the FP8 MFMA: lane (probe l31, k half lhi) holds 32 bytes = its 16 columns of each
  // column block (byte 16 jb + r <-> column 32 jb + (r & 3) + 8 (r >> 2) + 4 lhi: the A fragment's own order); two 16-byte planes
  auto read_vq = [&](uintx4& v, int buf, int jb, int nb) {
    v = *reinterpret_cast<const uintx4*>(fat_smem + buf * S::kTile + S::kABytes + S::kVHiBytes + (((jb * NB + nb) * 2 + lhi) * 32 + l31) * 16);
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
    // column operand arrives by ds_read_b128 (s_waitcnt is data-flow, the compiler keeps that); tests/test_gpu_pc_matvec.py
    // checks every block position of a tile against the oracle.
    if (first) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(kd) : "a"(a), "a"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(kd) : "a"(a), "a"(b));
  };
  // ---- one micro-step of the split chain of a distance block, in place: t = 0 exp2 of entry i; 1 hi = f16 of pair i; 2 entry i
  //      minus its hi; 3 lo = f16 of pair i.  Register r of the block <-> column (r & 3) + 8 (r >> 2) + 4 lhi; the pair (2p, 2p + 1)
  //      is one packed register of the A fragment of k-step p >> 2. --------------------------------------------------------------
  auto split_op = [&](floatx16& w, half8 (&ah)[2], half8 (&al)[2], unsigned (&lopk)[8], uintx4& kq, unsigned (&tq)[8], const int t, const int i, const bool neg) {
    if (t == 0) {
      w[i] = __builtin_amdgcn_exp2f(neg ? -w[i] : w[i]);
    } else if (t == 3) {
      // hi(K) of entries 4 i .. 4 i + 3 as e4m3 of K / 128, from the f16 hi pairs 2 i and 2 i + 1 (K > 0): the e4m3 byte of an f16 pattern
      // h is ((h + 0x40) - (15 << 10)) >> 7 -- round the mantissa to 3 bits, exponent field E8 = E16 - 15, saturating at 0 for
      // K < 2^-14 of the outputscale (2^15 K <= 2^15 exactly: the exp is clamped, so E8 <= 15 with mantissa 0 = 256)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int pr = 2 * i + q;
        const unsigned hpair = __builtin_bit_cast(uintx4, ah[pr >> 2])[pr & 3];
        asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(tq[2 * i + q]) : "v"(hpair), "v"(0x3BC03BC0u));
      }
    } else if (t == 4) {
      tq[2 * i] >>= 7;
      tq[2 * i + 1] >>= 7;
    } else if (t == 5) {
      kq[i] = __builtin_amdgcn_perm(tq[2 * i + 1], tq[2 * i], 0x06040200u);  // bytes 0, 2 of each: entries 4 i .. 4 i + 3 in order
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
  // `twelve`: the table of the block that EXECUTES (lag 0) / executed (lag 1) the chain's in-block part
  auto split_slot_t = [&](auto twelve_c, const int slot, const int lag, floatx16& w, half8 (&ah)[2], half8 (&al)[2], unsigned (&lopk)[8],
                          uintx4& kq, unsigned (&tq)[8], const bool neg) {
    constexpr bool twelve = decltype(twelve_c)::value;
    constexpr Fat8Table T = twelve ? kT12 : kT10;
    const int at = slot + (twelve ? 12 : 10) * lag;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (T.e[i] == at) split_op(w, ah, al, lopk, kq, tq, 0, i, neg);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (T.h[i] == at) split_op(w, ah, al, lopk, kq, tq, 1, i, neg);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (T.qa[i] == at) split_op(w, ah, al, lopk, kq, tq, 3, i, neg);
      if (T.qb[i] == at) split_op(w, ah, al, lopk, kq, tq, 4, i, neg);
      if (T.qc[i] == at) split_op(w, ah, al, lopk, kq, tq, 5, i, neg);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (T.m[i] == at) split_op(w, ah, al, lopk, kq, tq, 2, i, neg);
  };
  auto split_slot = [&](const bool twelve, const int slot, const int lag, floatx16& w, half8 (&ah)[2], half8 (&al)[2], unsigned (&lopk)[8],
                        uintx4& kq, unsigned (&tq)[8], const bool neg) {
    if (twelve) split_slot_t(std::true_type{}, slot, lag, w, ah, al, lopk, kq, tq, neg);
    else split_slot_t(std::false_type{}, slot, lag, w, ah, al, lopk, kq, tq, neg);
  };

  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): my pieces of tiles 0 and 1
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- pipeline state.  Blocks are numbered along the sweep: block = 8 tile + 4 jb + mi. --------------------------------------
  floatx16 wc, wn, wd;          // distance blocks: current (its last pairs still being split), next (being split), next but one
  half8 ahc[2], alc[2], ahn[2], aln[2];   // A fragments (hi, lo) x k-step of the current and of the next block
  unsigned lpc[8], lpn[8];                 // packed lo pairs in the making (between the mixlo and the mixhi step)
  unsigned tqc[8], tqn[8];                 // e4m3 dwords in the making (between the re-bias, the shift and the byte gather)
  half8 vf[2][NB];              // hi probe fragments of the current column block [k-step][probe block]
  uintx4 vq[NB][2];             // lo probe fragments of the current TILE as e4m3 [probe block][column block = operand half]
  uintx4 kq[4][2];              // hi(K) of the current tile as e4m3 [row block][column block = operand half]
  half8 ajs[2][NKD];            // column operand of a column block, by parity
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) kq[mi][hb] = uintx4{0u, 0u, 0u, 0u};
  // prologue (once per sweep, not per tile): blocks 0 and 1 by hand
  read_a(ajs[0], 0, 0);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) read_v1(vf[s][nb], 0, 0, s, nb);
#pragma unroll
  for (int q = 0; q < NKD; ++q) dist_step(wc, ajs[0][q], bih[0][q], q == 0);
#pragma unroll
  for (int q = 0; q < NKD; ++q) dist_step(wn, ajs[0][q], bih[1][q], q == 0);
  asm volatile("" : "+v"(wc));  // ties the first reader of wc behind the last of these MFMAs (two MFMAs after the one that wrote wc)
  // block 0's split as far as the table places it before a block boundary (its lag-1 steps run in the loop, like every block's)
#pragma unroll
  for (int slot = 0; slot < 12; ++slot) split_slot(true, slot, 0, wc, ahc, alc, lpc, kq[0][0], tqc, false);  // (block 0 follows a 12-slot block)

  if (ntl > kChainTiles) {  // the masters start from zero
    float* mrow = mst;
    for (int e = 0; e < 128; ++e, mrow += 64) *mrow = 0.f;
  }
  const float sc = outputscale[0];
  int tl = 0;
  for (; tl < ntl; ++tl) {
    const int buf = tl & 1;
    if (tl > 0 && (tl % kChainTiles) == 0) {  // chain fold: masters += accumulators, accumulators restart from zero
      float* mrow = mst;
      asm volatile("" : "+v"(mrow));  // (opaque: the 128 row addresses are formed here, not hoisted out of the tile loop and spilled)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            mrow[r * 64] += acc[mi][nb][r];  // 256-byte rows: immediate offsets 0 .. 3840
            acc[mi][nb][r] = 0.f;
          }
          mrow += 16 * 64;
          asm volatile("" : "+v"(mrow) : : "memory");  // one accumulator block at a time: 16 values in flight, not 128
        }
    }
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
      const int jb = blk >> 2, mi = blk & 3;
      // block blk + 1 (being split: wn -> ahn / aln) and block blk + 2 (distances: wd); both may lie in the next tile
      const int blk1 = (blk + 1) & 7, blk2 = (blk + 2) & 7;
      const int jb2 = blk2 >> 2, mi2 = blk2 & 3;
      const bool neg_c = ((jb + mi) & 1) != 0, neg_n = (((blk1 >> 2) + (blk1 & 3)) & 1) != 0;
      if (MFX_FAT_DIAG == 0 && blk == 5 && tl + 2 < ntl) issue_tile_dma(tl + 2);  // (this tile's buffer has been dead since the barrier behind block 4)
      // Invariant at this point: (ahc, alc) hold K_blk except for the table's lag-1 steps (still to run on wc); wn holds the
      // distances of block blk + 1, untouched; ajs[jb2 & 1] holds the column operand that block blk + 2 needs from its first
      // distance slot on.
      // where the split chain of block blk + 1 puts its e4m3 pairs: that block's half of its row block's operand
      const int jb1 = blk1 >> 2, mi1 = blk1 & 3;
      const int jbp = ((blk + 7) & 7) >> 2;  // kind of the block before this one: its table placed this block's lag-1 steps
#pragma unroll
      for (int slot = 0; slot < kSlots + NB; ++slot) {
        if (slot >= kSlots && jb == 0) continue;  // (compile-time after unrolling: only the jb = 1 blocks have the two FP8 slots)
        __builtin_amdgcn_sched_barrier(0);
        if (slot >= kSlots) {
          // hi(K) lo(V) of the whole tile for this row block: ONE block-scaled FP8 MFMA per probe block, k = 64 (both column blocks);
          // scales 2^7 (K was stored / 128) and 2^-5 (lo(V) was stored x 32) as E8M0 bytes
          const int nb = slot - kSlots;
          intx8 a8, b8;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a8[e] = (int)kq[mi][0][e]; a8[4 + e] = (int)kq[mi][1][e];
            b8[e] = (int)vq[nb][0][e]; b8[4 + e] = (int)vq[nb][1][e];
          }
          acc[mi][nb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[mi][nb], 0, 0, 0, 0x86868686, 0, 0x7a7a7a7a);
        } else if (Plan::dist_q(slot) >= 0) {
          const int q = Plan::dist_q(slot);
          dist_step(wd, ajs[jb2 & 1][q], bih[mi2][q], q == 0);
        } else {
          const int m = Plan::contr_m(slot);
          const int s = m / (2 * NB), nb = (m / 2) % NB, w = m % 2;  // w = 0: hi(K) hi(V), 1: lo(K) hi(V)
          acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w == 1 ? alc[s] : ahc[s], vf[s][nb], acc[mi][nb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        split_slot(jbp == 1, slot, 1, wc, ahc, alc, lpc, kq[mi][jb], tqc, neg_c);  // this block's late steps first (their consumers are a few slots away)
        split_slot(jb == 1, slot, 0, wn, ahn, aln, lpn, kq[mi1][jb1], tqn, neg_n);
        // fragment reads, one per gap: the probe fragments of the NEXT column block during this one (blocks mi = 1, 2: sixteen
        // reads... eight per block), the column operand of the column block after that in block mi = 1
        if (MFX_FAT_DIAG < 3 && mi == 1 && slot < NKD) {
          // ajs of column block c + 1 is needed by the distances of block (c + 1, mi 0), issued in block (c, mi 2)
          ajs[(jb + 1) & 1][slot] = *reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(fat_smem + (jb == 1 ? buf ^ 1 : buf) * S::kTile) +
                                                                   (((jb + 1) & 1) * 32 + l31) * AROW + slot * 16 + lhi * 8);
        }
        if (MFX_FAT_DIAG < 3 && mi == 3) {
          // the hi probe fragments of the next column block roll in behind the last MFMA of this column block that reads the
          // register they replace: fragment (s, nb) is read by MFMAs 2 NB s + 2 nb and + 1
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              if (Plan::slot_of(2 * NB * s + 2 * nb + 1) == slot) read_v1(vf[s][nb], jb == 1 ? buf ^ 1 : buf, (jb + 1) & 1, s, nb);
        }
        // the e4m3 lo fragments of THIS tile (read by the FP8 MFMAs of blocks 4-7; the previous tile's were last read in its block 7)
        if (MFX_FAT_DIAG < 3 && blk == 2 && slot < 2 * NB) read_vq(vq[slot >> 1][slot & 1], buf, slot & 1, slot >> 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (blk == 4) {
        // tile tl + 1 (requested a tile ago) has landed for everybody, and everybody is done with this tile's buffer: its last reads
        // -- the probe fragments of column block 1, rolled in during block 3 -- were consumed by the MFMAs of this block
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (MFX_FAT_DIAG < 2) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      // rotate: next -> current, next-but-one -> next
      ahc[0] = ahn[0]; ahc[1] = ahn[1];
      alc[0] = aln[0]; alc[1] = aln[1];
      wc = wn;
      wn = wd;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        lpc[q] = lpn[q];
        tqc[q] = tqn[q];
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // no LDS-DMA of mine is left in flight
  if (ntl > kChainTiles) {
    const float* mrow = mst;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][nb][r] += mrow[r * 64];
        mrow += 16 * 64;
        asm volatile("" : "+v"(mrow) : : "memory");
      }
  }
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

int rbf_fat8_launch(int dpad, bool vec4, dim3 grid, hipStream_t stream, const float* xs, const float* sq, int64_t n,
                    const float* outputscale, const float* noise, const float* vscale, const float* x, int64_t ldx, float* y,
                    int64_t ldy, int64_t p, const void* pkv, const void* pka, float* part, const int* rangeflag, int64_t ldpart,
                    int64_t row0, int64_t rend, float* mstbuf) {
#define MFX_FAT8_LAUNCH(D, V4)                                                                                                 \
  k_rbf_fat8_apply<D, V4><<<grid, 256, Fat8Smem<D>::kTotal, stream>>>(xs, sq, n, outputscale, noise, vscale, x, ldx, y, ldy, p, \
                                                                      static_cast<const uintx4*>(pkv), static_cast<const uintx4*>(pka), \
                                                                      part, rangeflag, ldpart, row0, rend, mstbuf)
  if (dpad == 4) {
    if (vec4) MFX_FAT8_LAUNCH(4, true); else MFX_FAT8_LAUNCH(4, false);
  } else if (dpad == 8) {
    if (vec4) MFX_FAT8_LAUNCH(8, true); else MFX_FAT8_LAUNCH(8, false);
  } else {
    set_error("FP8-assisted fat-wave Gram matvec supports d <= 8");
    return MFX_ERR_UNSUPPORTED;
  }
#undef MFX_FAT8_LAUNCH
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

}  // namespace mfx
