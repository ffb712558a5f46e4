// Pieces shared by the matrix-core RBF / Matern Gram kernels (mfx_rbf_mfma.hip, mfx_rbf_fat.hip): vector types, the kernel
// family's constants, the hi / lo f16 split, the LDS-DMA copy and the layout of the pre-packed tile images.
#pragma once
#include "mfx_internal.h"

namespace mfx {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));


// -log2(e)/2: exp(-dist/2) = exp2(kNegHalfLog2e * dist)
constexpr float kNegHalfLog2e = -0.72134752044448170368f;

// Matern kernels: the distance product is scaled by log2(e)^2 (x 3 for nu = 3/2), so that r' = sqrt(t) is already
// log2(e) * r:  K = exp2(-r') [nu = 1/2],  (1 + r'/log2(e)) exp2(-r') [nu = 3/2]   (util/gp_util.py:69-148)
constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;
constexpr float kEpsF32 = 1.1920928955078125e-7f;
// K / outputscale from t = factor * dist (un-clamped), shift = log2 of an optional power-of-two scale of K
// sqrt as ONE v_sqrt_f32 (1 ulp): __builtin_sqrtf expands to a 16-instruction correctly-rounded sequence, which made
// the Matern Gram matvec 2.5x the RBF one (every VALU instruction per kernel entry is paid in full here).
template <int KIND>
__device__ __forceinline__ float matern_from_t(float t, float shift) {
  const float rp = __builtin_amdgcn_sqrtf(fmaxf(t, 0.f) + kEpsF32 * kLog2e * kLog2e);
  const float e = __builtin_amdgcn_exp2f(shift - rp);
  return KIND == MFX_KERNEL_MATERN32 ? fmaf(e * rp, kLn2, e) : e;
}
// the same from te = t + eps log2(e)^2 (the eps rides in the distance product): max(t, 0) + eps' == max(t + eps', eps'),
// so clamp and offset are one v_med3; `scale` = 2^shift is folded into the polynomial factor (no v_sub before the exp2)
constexpr float kEpsC = kEpsF32 * kLog2e * kLog2e;
template <int KIND>
__device__ __forceinline__ float matern_from_te(float te, float scale) {
  const float rp = __builtin_amdgcn_sqrtf(__builtin_amdgcn_fmed3f(te, kEpsC, 3.0e38f));
  const float e = __builtin_amdgcn_exp2f(-rp);
  return KIND == MFX_KERNEL_MATERN32 ? e * fmaf(rp, kLn2 * scale, scale) : e * scale;
}


constexpr float kKShift = 15.f;
// Chains of the pipelined matvec.  The f16 MFMA aligns what it adds to the accumulator with a few guard bits and truncates
// (tools/mfma_f16_trunc.hip): harmless for sign-mixed sums, but Krylov vectors are dominated by the smooth leading eigenvectors
// of an all-positive kernel matrix -- the accumulator of a row then grows monotonically over its 24576 MFMAs and the loss grows
// with the size of the accumulator relative to the products (-8.7e-5 relative on a constant vector at n = 131072; the SLQ
// gradient was 4.4e-4 off; tools/diag_matvec_bias.py).  What helps is a SMALL accumulator: every kChainTiles tiles the
// accumulators are folded (fp32 VALU adds, round to nearest) into master accumulators and restart from zero.  The masters of
// all but one 32 x 32 block live in LDS (lane-private slots: no barrier, no bank conflict), the last block's in 16 of the
// spare registers: no partial sums through HBM.  (Round 2 first cut the sweep into 16 column splits with partial sums in HBM:
// same accuracy, +7.6 % per matvec and 1.1 GB of extra traffic per launch; negating the accumulators in registers instead of
// restarting them did nothing.)
#ifndef MFX_CHAIN_TILES
#define MFX_CHAIN_TILES 128
#endif
constexpr int kChainTiles = MFX_CHAIN_TILES;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));  // 16-B pack as a native vector (HIP's uint4 struct went to scratch)
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// hi = x rounded to 11 significant bits (round-half-up on the magnitude: add half an f16 ulp to the
// bit pattern, clear the low 13 mantissa bits -- a carry ripples into the exponent correctly), so hi is
// exactly an f16 and |lo| = |x - hi| <= 2^-12 |x| (exact in fp32).  Rounding instead of truncating
// keeps the dropped lo*lo term at 2^-24 relative AND sign-random (truncation made it a coherent bias).
__device__ __forceinline__ void split_hi_lo(float x, float& hi, float& lo) {
  hi = __uint_as_float((__float_as_uint(x) + 0x1000u) & 0xFFFFE000u);
  lo = x - hi;
}


// The reference clamps the squared distance at 0 before the exponential (util/gp_util.py:173).  In fp32 a computed squared
// distance is negative only by round-off (|t| <~ 1e-6 of the operands' squares), so the clamp changes K_ij by at most that
// round-off -- the same size as the error of every other entry -- while costing one VALU instruction per entry (7 % of the
// RBF matvec).  MFX_RBF_CLAMP=1 at build time restores it; 2^15 K cannot overflow f16 either way (arg <= 15 + 1e-5).
#ifndef MFX_RBF_CLAMP
#define MFX_RBF_CLAMP 0
#endif
constexpr bool kClampRbf = MFX_RBF_CLAMP != 0;

// async global -> LDS copy of 16 bytes per lane (global_load_lds_dwordx4): the destination is
// wave-uniform base + lane * 16 (1 KiB per wave-instruction), no staging registers
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// tile of the pipelined kernel: RbfTileH plus the f16 hi/lo image of the distance operand (DH variant).
// PADROW: the in-kernel split stores 8-byte pieces from many rows at once and needs the row pad; the pre-packed path copies the image
// by LDS-DMA in 1 KiB pieces and wants [ajh | vhi | vlo] contiguous (the b128 fragment reads are conflict-free either way).
template <int DPAD, int NB, int kTJ, bool PADROW = true>
struct RbfTileH3 {
  static constexpr int KD = DPAD + 2;
  static constexpr int NKD = (3 * KD + 15) / 16;  // f16 MFMAs (K = 16) per block for hi.hi + hi.lo + lo.hi
  static constexpr int AROW = NKD * 16 + 8;       // halves per column, padded: conflict-free ds_read_b128
  static constexpr int P = NB * 32;
  static constexpr int ROW = P * 8 + (PADROW ? 8 : 0);
  float aj[KD][kTJ];
  _Float16 ajh[kTJ][AROW];
  _Float16 vhi[(kTJ / 32) * 4][ROW];
  _Float16 vlo[(kTJ / 32) * 4][ROW];
};


// launcher of the fat-wave matvec kernel (mfx_rbf_fat.hip): RBF, d <= 8, chunks of nb * 32 vectors (nb = 1, 2);
// grid = (ceil(rows / 512), chunks, splits)
int rbf_fat_launch(int dpad, int nb, bool vec4, dim3 grid, hipStream_t stream, const float* xs, const float* sq, int64_t n,
                   const float* outputscale, const float* noise, const float* vscale, const float* x, int64_t ldx, float* y,
                   int64_t ldy, int64_t p, const void* pkv, const void* pka, float* part, const int* rangeflag, int64_t ldpart,
                   int64_t row0, int64_t rend);

}  // namespace mfx
