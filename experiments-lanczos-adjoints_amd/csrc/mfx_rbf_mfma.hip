// libmfx: matrix-free RBF Gram matvec on the CDNA4 matrix cores (fp32-exact MFMA, gfx950).
//
//   W[i][b] = s * sum_j exp(-max(0, |x_i|^2 + |x_j|^2 - 2 x_i.x_j) / 2) V[j][b] + noise V[i][b]
//   (x already divided by the lengthscale; util/gp_util.py:160-176,225-226,536-541)
//
// The Gram matrix is never materialised (the reference materialises (n/num x n) tiles,
// util/gp_util.py:496-509).  GEMM view: M = n rows i, N = probes, K = n columns j.
//   * v_mfma_f32_32x32x2_f32: A = K-tile (32 i x 2 j), B = V-tile (2 j x 32 probes), D = 32 i x 32 probes.
//   * The squared distances are themselves a small MFMA product (inner dimension d + 2, augmented
//     vectors), computed TRANSPOSED so that its accumulator registers, after v_min + v_exp in place,
//     are exactly the A operand of the contraction MFMA (lane = row i, register = column j) -- no
//     shuffle, no LDS round trip for K, ~2 VALU instructions per kernel entry.
//   * B comes from an LDS image Vt[j][probe] (probe-contiguous, conflict-free ds_read_b32), filled
//     from the (p, n) probe-major global layout with 256-B coalesced segments and a transposing store
//     (row pad 1 -> at most 2-way ds_write conflicts, which are free on gfx950).
//   * one wave owns MI x 32 rows and all NB x 32 probes of its chunk: MI*NB*16 accumulator registers.
// Bound: fp32 MFMA (2 n^2 p flop per matvec) -- the exp/distance VALU work co-issues underneath.
#include <stdlib.h>

#include <type_traits>

#include "mfx_internal.h"
#include "mfx_rbf_common.h"

namespace mfx {

// geometry of the parameter-gradient GEMM (both the fp32 and the 3 x f16 variant)
constexpr int kGM = 128, kGN = 128, kGK = 32, kGSplit = 8;
// L2 sharing inside an XCD (split gradient GEMM): the 64 workgroups resident on an XCD form an 8 x 8 patch -- 8 row blocks
// x 8 column sub-ranges -- so every L and every R operand slice is read by 8 workgroups at about the same time (1 fabric
// read + 7 L2 hits).  With one row block per workgroup and a private 1.3 MB L panel re-read for each of its 128 tiles,
// the launch pulled 2.2 TB through the fabric (FETCH_SIZE, L2 hit rate 19 %): it was fabric-bound at 7 TB/s, not MFMA-bound.
constexpr int kGSub = 8;
constexpr int kGLd = kGM + 4;  // padded row of the transposed S tile: conflict-free ds_write_b128


__device__ __forceinline__ float dist_factor(int kind) {  // multiplies the squared distance inside the MFMA product
  return kind == MFX_KERNEL_RBF ? -0.72134752044448170368f : (kind == MFX_KERNEL_MATERN32 ? 3.f : 1.f) * kLog2e * kLog2e;
}

// epilogue of the gradient GEMMs: K_ij / outputscale and the lengthscale weight from the clamped squared distance
__device__ __forceinline__ void grad_weights(int kind, float dist, float& kv, float& wl) {
  if (kind == MFX_KERNEL_RBF) {
    kv = __builtin_amdgcn_exp2f(-0.72134752044448170368f * dist);
    wl = kv;
  } else if (kind == MFX_KERNEL_MATERN32) {
    const float r = __builtin_amdgcn_sqrtf(3.f * dist + kEpsF32);
    const float e = __builtin_amdgcn_exp2f(-kLog2e * r);
    kv = (1.f + r) * e;
    wl = 3.f * e;
  } else {
    const float r = __builtin_amdgcn_sqrtf(dist + kEpsF32);
    const float e = __builtin_amdgcn_exp2f(-kLog2e * r);
    kv = e;
    wl = dist > 0.f ? e / r : 0.f;
  }
}

// exp2(min(x, 0)) in ONE VALU instruction: v_exp_f32 with the clamp output modifier ([0, 1]); exp2 is
// monotone, so clamping the result equals clamping the argument (the reference clamps the squared
// distance at 0, util/gp_util.py:173).  fp32 MFMA shares the FP32 lanes with the VALU on gfx950
// (measured: removing the min/exp shortens the kernel by exactly their issue cycles), so every VALU
// instruction per kernel entry is paid in full.  Written so that the COMPILER emits the instruction (fmed3(x, 0, 1) folds into the
// clamp modifier of v_exp_f32): x is an MFMA result and the result feeds an MFMA operand, and the wait states on both sides are the
// hazard recogniser's job -- it does not look inside inline asm (rounds 1-4 had asm("v_exp_f32_e64 ... clamp; s_nop 1") here, which
// covered the second hazard by hand and the first one not at all; see the register epilogue of k_rbf_mfma_grad_h for what that cost).
__device__ __forceinline__ float exp2_clamped(float x) { return __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x), 0.f, 1.f); }

template <int DPAD, int NB, int kTJ>
struct RbfTile {
  static constexpr int KD = DPAD + 2;        // augmented inner dimension of the distance product
  static constexpr int LDV = NB * 32 + 1;    // padded probe row
  float aj[KD][kTJ];   // A operand of the distance MFMA, k-major: c * [-2 x_j, |x_j|^2, 1]
  float vt[kTJ][LDV];  // V^T tile [j][probe]
};

// One workgroup = 4 waves x 64 rows; grid.y = chunks of NB*32 probes.  Per 64-column tile and per
// (32-row, 32-column) block the wave runs
//   (1) KD/2 distance MFMAs  D^T[j][i] = sum_k A_j[k] B_i[k] = c (|x_i|^2 + |x_j|^2 - 2 x_i.x_j)   (c = -log2(e)/2)
//       A_j from LDS (k-major, conflict-free), B_i = [x_i, 1, |x_i|^2] resident in registers;
//   (2) 16 x { v_min(., 0), v_exp }  in place: the accumulator registers ARE the A operand of step (3):
//       register r of lane (i = lane & 31, h = lane >> 5) holds column j_r(h) = (r & 3) + 8 (r >> 2) + 4 h;
//   (3) 16 x NB contraction MFMAs  W[i][probe] += K[i][j_r(h)] V[j_r(h)][probe]  with B = Vt rows j_r(h).
// K never exists outside VGPRs.  Global -> LDS staging is register-prefetched and double-buffered: one
// barrier per tile.
template <int DPAD, int NB, bool VEC4, int kMI, int kTJ>
__global__ __launch_bounds__(256, 2) void k_rbf_mfma_apply(const float* __restrict__ xs, const float* __restrict__ sq,
                                                           int64_t n, const float* __restrict__ outputscale,
                                                           const float* __restrict__ noise,
                                                           const float* __restrict__ x, int64_t ldx,
                                                           float* __restrict__ y, int64_t ldy, int64_t p, int kind,
                                                           int64_t row0, int64_t rend) {
  // rows [row0, rend) of the operator (a row shard, or all of it): x has the full length n, y is indexed from row0
  using Tile = RbfTile<DPAD, NB, kTJ>;
  const float cfac = dist_factor(kind);
  constexpr int KD = Tile::KD, KS = KD / 2;
  __shared__ __attribute__((aligned(16))) Tile tile[2];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int64_t i_wave = row0 + (int64_t)blockIdx.x * (4 * kMI * 32) + (int64_t)wid * (kMI * 32);
  const int64_t b0 = (int64_t)blockIdx.y * (NB * 32);

  // B operand of the distance product: B_i[k], k = 2 s + lhi
  float bi[kMI][KS];
#pragma unroll
  for (int mi = 0; mi < kMI; ++mi) {
    int64_t i = i_wave + mi * 32 + l31;
    if (i >= rend) i = rend - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 2 * s + lhi;
      bi[mi][s] = (k < DPAD) ? xs[i * DPAD + k] : (k == DPAD ? 1.f : sq[i]);
    }
  }
  floatx16 acc[kMI][NB];
#pragma unroll
  for (int mi = 0; mi < kMI; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nb][r] = 0.f;

  // ---- staging: registers <- global, LDS <- registers -----------------------------------------
  constexpr int kF4 = NB * 32 * (kTJ / 4);   // float4 chunks of the V tile
  constexpr int kVPT = (kF4 + 255) / 256;    // per thread
  constexpr int kXPT = (kTJ * DPAD + 255) / 256;
  float4 rv[kVPT];
  float rx[kXPT], rsq = 0.f;

  auto load_tile = [&](int64_t j0) {
#pragma unroll
    for (int u = 0; u < kVPT; ++u) {
      const int f = tid + 256 * u;
      const int bq = f / (kTJ / 4), j4 = (f % (kTJ / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < kF4 && b0 + bq < p) {
        const float* src = x + (b0 + bq) * ldx + j0 + j4;
        if (VEC4 && j0 + j4 + 3 < n) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          if (j0 + j4 + 0 < n) v.x = src[0];
          if (j0 + j4 + 1 < n) v.y = src[1];
          if (j0 + j4 + 2 < n) v.z = src[2];
          if (j0 + j4 + 3 < n) v.w = src[3];
        }
      }
      rv[u] = v;
    }
#pragma unroll
    for (int u = 0; u < kXPT; ++u) {
      const int t = tid + 256 * u;
      const int64_t g = j0 * DPAD + t;
      rx[u] = (t < kTJ * DPAD && g < n * DPAD) ? xs[g] : 0.f;
    }
    if (tid < kTJ) rsq = (j0 + tid < n) ? sq[j0 + tid] : 0.f;
  };
  auto store_tile = [&](Tile& tl) {
#pragma unroll
    for (int u = 0; u < kVPT; ++u) {
      const int f = tid + 256 * u;
      if (f < kF4) {
        const int bq = f / (kTJ / 4), j4 = (f % (kTJ / 4)) * 4;
        tl.vt[j4 + 0][bq] = rv[u].x;
        tl.vt[j4 + 1][bq] = rv[u].y;
        tl.vt[j4 + 2][bq] = rv[u].z;
        tl.vt[j4 + 3][bq] = rv[u].w;
      }
    }
#pragma unroll
    for (int u = 0; u < kXPT; ++u) {
      const int t = tid + 256 * u;
      if (t < kTJ * DPAD) tl.aj[t % DPAD][t / DPAD] = -2.f * cfac * rx[u];
    }
    if (tid < kTJ) {
      tl.aj[DPAD][tid] = cfac * rsq;
      tl.aj[DPAD + 1][tid] = cfac;
    }
  };

  load_tile(0);
  store_tile(tile[0]);
  __syncthreads();
  const int64_t ntile = (n + kTJ - 1) / kTJ;
  for (int64_t t = 0; t < ntile; ++t) {
    const Tile& tl = tile[t & 1];
    if (t + 1 < ntile) load_tile((t + 1) * kTJ);
#pragma unroll
    for (int jb = 0; jb < kTJ / 32; ++jb) {
      float aj[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) aj[s] = tl.aj[2 * s + lhi][jb * 32 + l31];
#pragma unroll
      for (int mi = 0; mi < kMI; ++mi) {
        floatx16 kd;
#pragma unroll
        for (int r = 0; r < 16; ++r) kd[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) kd = __builtin_amdgcn_mfma_f32_32x32x2f32(aj[s], bi[mi][s], kd, 0, 0, 0);
        if (kind == MFX_KERNEL_RBF) {
#pragma unroll
          for (int r = 0; r < 16; ++r) kd[r] = exp2_clamped(kd[r]);
        } else {
          // the diagonal 32 x 32 block: a point's distance to itself is exactly 0
          const bool diag_blk = (t * kTJ + jb * 32) == (i_wave + mi * 32);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float tv = kd[r];
            if (diag_blk && l31 == (r & 3) + 8 * (r >> 2) + 4 * lhi) tv = 0.f;
            kd[r] = kind == MFX_KERNEL_MATERN32 ? matern_from_t<MFX_KERNEL_MATERN32>(tv, 0.f)
                                                : matern_from_t<MFX_KERNEL_MATERN12>(tv, 0.f);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jr = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kd[r], tl.vt[jr][nb * 32 + l31], acc[mi][nb], 0, 0, 0);
        }
      }
    }
    if (t + 1 < ntile) store_tile(tile[(t + 1) & 1]);
    __syncthreads();
  }
  // ---- epilogue: y[b][i] = s * acc + noise * x[b][i];  D layout: col = lane & 31 (probe),
  //      row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)  -> 4 consecutive rows per register quad
  const float s = outputscale[0], nz = noise[0];
#pragma unroll
  for (int mi = 0; mi < kMI; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int64_t b = b0 + nb * 32 + l31;
      if (b >= p) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t i = i_wave + mi * 32 + 8 * g + 4 * lhi;
        if (VEC4 && i + 3 < rend) {
          const float4 xv = *reinterpret_cast<const float4*>(x + b * ldx + i);
          float4 o;
          o.x = fmaf(s, acc[mi][nb][4 * g + 0], nz * xv.x);
          o.y = fmaf(s, acc[mi][nb][4 * g + 1], nz * xv.y);
          o.z = fmaf(s, acc[mi][nb][4 * g + 2], nz * xv.z);
          o.w = fmaf(s, acc[mi][nb][4 * g + 3], nz * xv.w);
          *reinterpret_cast<float4*>(y + b * ldy + (i - row0)) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (i + e < rend) y[b * ldy + (i - row0) + e] = fmaf(s, acc[mi][nb][4 * g + e], nz * x[b * ldx + i + e]);
        }
      }
    }
}

// ================================================================================================
// 3 x f16 split path ("fp32 emulation on the f16 matrix pipe").
//
// Why: on gfx950 the fp32-input MFMA runs at the fp32 VALU rate and does not co-execute with VALU
// work (ablation in DESIGN.md §3.2), so the fp32-exact kernel above is capped at ~70 % of 157 TFLOP/s.
// The f16 MFMA (v_mfma_f32_32x32x16_f16) is a separate pipe with 16x the rate.  Every fp32 operand is
// split into hi + lo with hi = top 11 significant bits (exactly an f16), lo = the remainder rounded to
// f16 (11 more bits), and the product is accumulated in fp32 as  hi*hi + hi*lo + lo*hi  (the dropped
// lo*lo term is 2^-24 relative, random sign).  K in [0, 1] needs no scaling; each probe vector is scaled by a
// power of two so that its largest entry is ~2^14 (scale undone, exactly, in the epilogue).
// Accuracy: |error| <= ~4 eps_fp32 per product, fp32 accumulation -- validated against the fp64 oracle
// at the same tolerances as the fp32-exact path (tests/test_gpu_parity.py).
// ================================================================================================
// per-row power-of-two scale: 2^(14 - e) for |max| = f 2^e  (1 for an all-zero row).  Two tiny kernels instead of
// one workgroup per row (98 us at n = 131072): slice maxima -> atomicMax on the bit pattern of the non-negative
// float (monotone as an integer), then the scales.
__global__ __launch_bounds__(256) void k_row_amax_bits(const float* __restrict__ x, int64_t ldx, int64_t n,
                                                       unsigned* __restrict__ amax_bits) {
  __shared__ float sm[4];
  const int64_t b = blockIdx.y;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    m = fmaxf(m, fabsf(x[b * ldx + i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(amax_bits + b, __float_as_uint(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]))));
}

__global__ void k_row_scale_from_bits(const unsigned* __restrict__ amax_bits, int64_t rows, float* __restrict__ scale) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= rows) return;
  const float m = __uint_as_float(amax_bits[b]);
  int e = 0;
  float s = 1.f;
  if (m > 0.f && m < 3.0e38f) {
    frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)
    s = ldexpf(1.f, 14 - e < 126 ? 14 - e : 126);  // tiny / denormal rows (|max| < 2^-112): keep s and 1/s finite and normal
  }
  scale[2 * b] = s;
  scale[2 * b + 1] = 1.f / s;
}

// 2^(14 - e) for |max| = f 2^e (1 for an all-zero row); tiny / denormal rows (|max| < 2^-112): s and 1/s stay finite and normal
__device__ __forceinline__ float row_scale_of(float m) {
  int e = 0;
  float s = 1.f;
  if (m > 0.f && m < 3.0e38f) {
    frexpf(m, &e);
    s = ldexpf(1.f, 14 - e < 126 ? 14 - e : 126);
  }
  return s;
}

// Pre-packed path: ONE launch in front of k_pack_tiles instead of three (zero + atomic maxima + scales).  Slice maxima of |x| per
// row, no atomics: part[b gx + slice]; k_pack_tiles folds a row's <= 64 slice maxima into its scale itself.  Block (0, 0) also
// clears the f16 range flag, which k_pack_tiles (behind this launch in stream order) raises.
__global__ __launch_bounds__(256) void k_row_amax_part(const float* __restrict__ x, int64_t ldx, int64_t n, float* __restrict__ part,
                                                       int* __restrict__ flag) {
  __shared__ float sm[4];
  const int64_t b = blockIdx.y;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    m = fmaxf(m, fabsf(x[b * ldx + i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[b * gridDim.x + blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    if (blockIdx.x == 0 && blockIdx.y == 0) *flag = 0;
  }
}

// scale (rows, 2) = [s, 1/s]; the amax bit patterns live right behind it in the caller's buffer
static int row_scales(const float* x, int64_t ldx, int64_t n, int64_t rows, float* scale, hipStream_t stream) {
  unsigned* bits = reinterpret_cast<unsigned*>(scale + 2 * rows);
  MFX_TRY(zero_async(bits, sizeof(unsigned) * (rows + 1), stream));  // + the f16 range flag behind them
  int64_t gx = (n + 8191) / 8192;
  if (gx > 64) gx = 64;
  k_row_amax_bits<<<dim3((unsigned)gx, (unsigned)rows), 256, 0, stream>>>(x, ldx, n, bits);
  k_row_scale_from_bits<<<(unsigned)((rows + 255) / 256), 256, 0, stream>>>(bits, rows, scale);
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

template <int DPAD, int NB, int kTJ>
struct RbfTileH {
  static constexpr int KD = DPAD + 2;
  static constexpr int P = NB * 32;
  static constexpr int ROW = P * 8 + 8;  // halves per (jb, s, h) row of 8-packs, padded by one pack
  float aj[KD][kTJ];
  _Float16 vhi[(kTJ / 32) * 4][ROW];  // [(jb*2 + s)*2 + h][probe][8]
  _Float16 vlo[(kTJ / 32) * 4][ROW];
};


// ================================================================================================
// Pipelined 3 x f16 kernel ("h3"): the production fp32 RBF Gram matvec.
//
// Measured on gfx950 (ablations): in the un-pipelined split kernel the fp32 distance MFMAs,
// the exp/split VALU work and the f16 contraction MFMAs simply ADD UP -- the two waves of a SIMD run the
// same program between the same barriers and fall into lock-step, and the fp32 MFMA shares the FP32 lanes
// with the VALU.  So the overlap is built into ONE wave's instruction stream, pinned with sched_barrier:
//   phase 1:  the 5 fp32 distance MFMAs of block b+1 (FP32 lanes)  ||  the first half of block b's f16
//             contraction MFMAs (matrix pipe)
//   phase 2:  the exp / hi-lo split of block b+1 (VALU, FP32 lanes)  ||  the second half of block b's MFMAs
// The distances stay on the fp32 MFMA on purpose: it is a round-to-nearest fmaf chain, whereas the f16 MFMA
// truncates its internal sum (tools/mfma_f16_rounding.hip: up to -1.75 ulp, biased), which is harmless in
// the sign-mixed contraction but showed up as a 5x larger gradient error when used for the exponent.
// ================================================================================================
// waves per workgroup: the pre-packed variant runs 8 waves (512 rows) on ONE staged tile -- the same two waves per SIMD as two
// 4-wave workgroups, but half the L2 -> LDS traffic and half the LDS tile copies
template <bool PK>
struct H3Waves {
  static constexpr int value = PK ? 8 : 4;
};

template <int DPAD, int NB, bool VEC4, int KIND, bool DH, bool PK>
__global__ __launch_bounds__(64 * H3Waves<PK>::value, PK ? 1 : 2) void k_rbf_mfma_apply_h3(const float* __restrict__ xs, const float* __restrict__ sq,
                                                              int64_t n, const float* __restrict__ outputscale,
                                                              const float* __restrict__ noise,
                                                              const float* __restrict__ vscale,
                                                              const float* __restrict__ x, int64_t ldx,
                                                              float* __restrict__ y, int64_t ldy, int64_t p,
                                                              const uintx4* __restrict__ pkv, const uintx4* __restrict__ pka,
                                                              float* __restrict__ part, const int* __restrict__ rangeflag,
                                                              int64_t ldpart, int64_t row0, int64_t rend) {
  // rows [row0, rend) of the operator (a row shard, or all of it; row0 % 64 == 0): x and the packed tile images cover all n
  // columns, y / part are indexed from row0
  static_assert(!PK || DH, "pre-packed operands exist for the f16-distance variant only");
  // f16 range guard: when a scaled input is too large for the f16 image of the distance operands (|x/l|^2 beyond ~6e4 / |c|),
  // k_pack_tiles raises the flag; the f16-distance launch then returns at once and the fp32-distance launch behind it does
  // the work (and vice versa) -- correct for any input, no host round trip, one empty launch per matvec.
  if (rangeflag && (*rangeflag != 0) == DH) return;
  // gridDim.z > 1: column split for small n (too few 256-row blocks to fill 256 CUs): workgroup z sweeps its share of the
  // 64-column tiles and writes a partial result to part[z][probe][row]; k_split_reduce adds them in a fixed order.
  constexpr int kMI = 2, kTJ = 64;
  using Tile = RbfTileH3<DPAD, NB, kTJ, !PK>;
  constexpr int KD = Tile::KD, KS = KD / 2, NKD = Tile::NKD;
  constexpr float cfac = KIND == MFX_KERNEL_RBF ? kNegHalfLog2e : (KIND == MFX_KERNEL_MATERN32 ? 3.f : 1.f) * kLog2e * kLog2e;
  extern __shared__ __attribute__((aligned(16))) char h3_smem[];
  Tile* const tile = reinterpret_cast<Tile*>(h3_smem);  // [2]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  constexpr int WV = H3Waves<PK>::value;
  // master accumulators of the chain folds (PK variant): blocks 0 .. kMI NB - 2 in LDS behind the tiles, the last one in registers
  constexpr int kBlocks = kMI * NB;
  float* const master = reinterpret_cast<float*>(h3_smem + 2 * sizeof(Tile)) + ((size_t)wid * (kBlocks - 1) * 16) * 64 + lane;
  float mreg[16];
  const int64_t i_wave = row0 + (int64_t)blockIdx.x * (WV * kMI * 32) + (int64_t)wid * (kMI * 32);
  const int64_t b0 = (int64_t)blockIdx.y * (NB * 32);

  // B operand of the distance product, resident in registers: [x_i, 1, |x_i|^2].
  //   fp32 variant: one float per k-step.   DH variant: the f16 image [Bh | Bl | Bh | 0] of the 3-product split
  //   (slot c*KD + k pairs with [Ah | Ah | Al] of the column operand), NEGATED for the odd row block: together with the
  //   columns' sign (odd column block negated in store_tile) the distance comes out as (-1)^(jb+mi) t, so the f16 MFMA's
  //   round-toward-minus-infinity bias enters K with alternating sign over the 32x32 blocks instead of coherently.
  float bi[kMI][DH ? 1 : KS];
  half8 bih[kMI][DH ? NKD : 1];
#pragma unroll
  for (int mi = 0; mi < kMI; ++mi) {
    int64_t i = i_wave + mi * 32 + l31;
    if (i >= rend) i = rend - 1;
    if constexpr (DH) {
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
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + lhi;
        bi[mi][s] = (k < DPAD) ? xs[i * DPAD + k] : (k == DPAD ? 1.f : sq[i]);
      }
    }
  }
  floatx16 acc[kMI][NB];
#pragma unroll
  for (int mi = 0; mi < kMI; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nb][r] = 0.f;
  if constexpr (PK) {
#pragma unroll
    for (int r = 0; r < 16; ++r) mreg[r] = 0.f;
#pragma unroll
    for (int q = 0; q < (kBlocks - 1) * 16; ++q) master[q * 64] = 0.f;
  }
  auto fold_chain = [&]() {  // masters += accumulators; accumulators restart from zero
#pragma unroll
    for (int mi = 0; mi < kMI; ++mi)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int idx = mi * NB + nb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (idx < kBlocks - 1) {
            float* m = master + (idx * 16 + r) * 64;
            *m += acc[mi][nb][r];
          } else {
            mreg[r] += acc[mi][nb][r];
          }
          acc[mi][nb][r] = 0.f;
        }
      }
  };

  // ---- PK: both LDS images come PRE-PACKED from k_pack_tiles (once per matvec instead of once per workgroup and tile: the
  //      hi/lo split of the probe tile was 40 of the ~93 VALU instructions per 32x32 block, and VALU time adds to MFMA time).
  //      A tile's probe image is 2 x 8 rows x P packs of 16 B (hi rows, then lo rows), its column operand 64 x AROW halves;
  //      staging = plain 16-B copies global -> registers -> LDS, no arithmetic.
  constexpr int kVPK = 2 * 8 * Tile::P;                 // 16-B packs of the probe image per tile
  constexpr int kAPK = kTJ * Tile::AROW * 2 / 16;       // 16-B packs of the column operand per tile
  static_assert(!PK || (kVPK % 64 == 0 && kAPK % 64 == 0), "tile images are whole 1-KiB DMA pieces");
  const int64_t ntile_all = (n + kTJ - 1) / kTJ;
  // LDS-DMA of tile t's two images straight into the tile buffer: wave w copies the 1-KiB pieces w, w + WV, ...; no staging
  // registers, no ds_write (less data movement is what still buys time on this power-limited kernel)
  auto issue_tile_dma = [&](int64_t t, Tile& tl) {
    const char* vsrc = reinterpret_cast<const char*>(pkv + ((int64_t)blockIdx.y * ntile_all + t) * kVPK);
    const char* asrc = reinterpret_cast<const char*>(pka + t * kAPK);
    char* vdst = reinterpret_cast<char*>(&tl.vhi[0][0]);  // vhi and vlo are contiguous (no row pad in this layout)
    char* adst = reinterpret_cast<char*>(&tl.ajh[0][0]);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int c = 0; c < (kVPK / 64 + WV - 1) / WV; ++c) {
      const int piece = w + WV * c;
      if (piece < kVPK / 64) glds16(vsrc + piece * 1024 + lane * 16, vdst + piece * 1024);
    }
#pragma unroll
    for (int c = 0; c < (kAPK / 64 + WV - 1) / WV; ++c) {
      const int piece = w + WV * c;
      if (piece < kAPK / 64) glds16(asrc + piece * 1024 + lane * 16, adst + piece * 1024);
    }
  };

  constexpr int kF4 = NB * 32 * (kTJ / 4);
  constexpr int kVPT = (kF4 + 255) / 256;
  constexpr int kXPT = (kTJ * DPAD + 255) / 256;
  float4 rv[kVPT];
  float rx[kXPT], rsq = 0.f;
  float vs[kVPT];
#pragma unroll
  for (int u = 0; u < kVPT; ++u) {
    const int f = tid + 256 * u;
    const int64_t b = b0 + f / (kTJ / 4);
    vs[u] = (f < kF4 && b < p) ? vscale[2 * b] : 0.f;
  }

  auto load_tile = [&](int64_t j0) {
#pragma unroll
    for (int u = 0; u < kVPT; ++u) {
      const int f = tid + 256 * u;
      const int bq = f / (kTJ / 4), j4 = (f % (kTJ / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < kF4 && b0 + bq < p) {
        const float* src = x + (b0 + bq) * ldx + j0 + j4;
        if (VEC4 && j0 + j4 + 3 < n) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          if (j0 + j4 + 0 < n) v.x = src[0];
          if (j0 + j4 + 1 < n) v.y = src[1];
          if (j0 + j4 + 2 < n) v.z = src[2];
          if (j0 + j4 + 3 < n) v.w = src[3];
        }
      }
      rv[u] = v;
    }
#pragma unroll
    for (int u = 0; u < kXPT; ++u) {
      const int t = tid + 256 * u;
      const int64_t g = j0 * DPAD + t;
      rx[u] = (t < kTJ * DPAD && g < n * DPAD) ? xs[g] : 0.f;
    }
    if (tid < kTJ) rsq = (j0 + tid < n) ? sq[j0 + tid] : 0.f;
  };
  auto store_tile = [&](Tile& tl) {
#pragma unroll
    for (int u = 0; u < kVPT; ++u) {
      const int f = tid + 256 * u;
      if (f < kF4) {
        const int bq = f / (kTJ / 4), j4 = (f % (kTJ / 4)) * 4;
        const int row = ((j4 >> 5) * 2 + ((j4 >> 4) & 1)) * 2 + ((j4 >> 2) & 1);
        const int col = bq * 8 + 4 * ((j4 >> 3) & 1);
        float h0, h1, h2, h3, l0, l1, l2, l3;
        split_hi_lo(rv[u].x * vs[u], h0, l0);
        split_hi_lo(rv[u].y * vs[u], h1, l1);
        split_hi_lo(rv[u].z * vs[u], h2, l2);
        split_hi_lo(rv[u].w * vs[u], h3, l3);
        half4 hh = {(_Float16)h0, (_Float16)h1, (_Float16)h2, (_Float16)h3};
        half4 ll = {(_Float16)l0, (_Float16)l1, (_Float16)l2, (_Float16)l3};
        *reinterpret_cast<half4*>(&tl.vhi[row][col]) = hh;
        *reinterpret_cast<half4*>(&tl.vlo[row][col]) = ll;
      }
    }
    auto put_a = [&](int j, int kk, float v) {  // DH: column operand [Ah | Ah | Al], odd column block negated
      float hi, lo;
      split_hi_lo(((j >> 5) & 1) ? -v : v, hi, lo);
      tl.ajh[j][kk] = (_Float16)hi;
      tl.ajh[j][KD + kk] = (_Float16)hi;
      tl.ajh[j][2 * KD + kk] = (_Float16)lo;
    };
#pragma unroll
    for (int u = 0; u < kXPT; ++u) {
      const int t = tid + 256 * u;
      if (t < kTJ * DPAD) {
        if constexpr (DH) {
          put_a(t / DPAD, t % DPAD, -2.f * cfac * rx[u]);
        } else {
          tl.aj[t % DPAD][t / DPAD] = -2.f * cfac * rx[u];
        }
      }
    }
    if (tid < kTJ) {
      // K' = 2^15 K: lo(K') stays a NORMAL f16 for K >= 4e-6 (RBF: shift folded into the product; Matern: into exp2)
      const float a_sq = cfac * rsq + (KIND == MFX_KERNEL_RBF ? kKShift : kEpsC);
      if constexpr (DH) {
        put_a(tid, DPAD, a_sq);
        put_a(tid, DPAD + 1, cfac);
      } else {
        tl.aj[DPAD][tid] = a_sq;
        tl.aj[DPAD + 1][tid] = cfac;
      }
    }
  };
  if constexpr (DH && !PK) {  // the zero padding of the f16 column operand is written once
    for (int t = tid; t < 2 * kTJ * (Tile::AROW - 3 * KD); t += 256) {
      const int w = Tile::AROW - 3 * KD;
      tile[t / (kTJ * w)].ajh[(t / w) % kTJ][3 * KD + t % w] = (_Float16)0.f;
    }
  }
  // entries (2 pr, 2 pr + 1): K' = exp2(min(arg, 15)) and its hi/lo f16 pieces.  min as ONE compiler-visible
  // v_med3_f32 (fminf adds a canonicalising v_max; inline asm would hide the MFMA-result hazard from hipcc).
  // `neg`: this block's distances were produced negated (DH sign alternation); the sign is a free source modifier
  auto exp_split_pair = [&](const floatx16& kd, int pr, const bool diag_blk, const bool neg, half8 (&ah)[2],
                            half8 (&al)[2]) {
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
    // lo = k - hi in ONE instruction each: v_fma_mix_f32 reads the f16 half directly (no v_cvt_f32_f16 + v_sub).
    // Inputs/outputs are VALU values only, so no MFMA hazard hides inside the asm.
    float l0, l1;
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hb), "v"(k0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hb), "v"(k1));
    const half2v l = {(_Float16)l0, (_Float16)l1};
    ah[s][q] = h[0]; ah[s][q + 1] = h[1];
    al[s][q] = l[0]; al[s][q + 1] = l[1];
  };

  const int64_t ntile_tot = (n + kTJ - 1) / kTJ;
  const int64_t t_first = ntile_tot * blockIdx.z / gridDim.z, ntile = ntile_tot * (blockIdx.z + 1) / gridDim.z;
  if constexpr (PK) {
    issue_tile_dma(t_first, tile[t_first & 1]);
  } else {
    load_tile(t_first * kTJ);
    store_tile(tile[t_first & 1]);
  }
  __syncthreads();
  for (int64_t t = t_first; t < ntile; ++t) {
    const Tile& tl = tile[t & 1];
    if constexpr (PK) {
      if (t > t_first && ((t - t_first) % kChainTiles) == 0) fold_chain();
    }
    if (t + 1 < ntile) {
      // (PK) the other buffer was last read during tile t - 1, and every wave has passed the barrier that ended it
      if constexpr (PK) issue_tile_dma(t + 1, tile[(t + 1) & 1]); else load_tile((t + 1) * kTJ);
    }
    // the one tile whose 64 columns are this wave's 64 rows holds the diagonal: only there (and only for the Matern
    // kernels) the per-entry self-distance fix is compiled in -- two copies of the tile body, chosen wave-uniformly
    auto do_tile = [&](auto diag_tag) {
    constexpr bool kDiag = decltype(diag_tag)::value;
    if constexpr (DH) {
      // ---- wide schedule: the exp / split of block b+1 is spread over ALL contraction MFMAs of block b (its distances were
      //      issued during block b-1), so the VALU stream runs beside the matrix pipe for the whole block instead of half of it
      //      (PMC: 29 % of the MFMA-busy cycles co-executed with VALU under the two-phase schedule) ----------------------
      half8 ah[2], al[2];
      floatx16 kdn;
      // column-operand fragments of one 32-column block (shared by its two row blocks: one LDS read per column block)
      half8 ajs[NKD];
      auto load_a = [&](int jbx) {
#pragma unroll
        for (int q = 0; q < NKD; ++q) ajs[q] = *reinterpret_cast<const half8*>(&tl.ajh[jbx * 32 + l31][q * 16 + lhi * 8]);
      };
      auto dist = [&](floatx16& kd, int mix) {
#pragma unroll
        for (int r = 0; r < 16; ++r) kd[r] = 0.f;
#pragma unroll
        for (int q = 0; q < NKD; ++q) kd = __builtin_amdgcn_mfma_f32_32x32x16_f16(ajs[q], bih[mix][q], kd, 0, 0, 0);
      };
      // B fragments (probe tile) of a block are fetched from LDS one block ahead: the ds_read latency in front of the first MFMA
      // of every block was ~10 % of the tile
      half8 bhb[2][2][NB], blb[2][2][NB];
      auto load_b = [&](int buf, int jbx) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int row = (jbx * 2 + s) * 2 + lhi;
            bhb[buf][s][nb] = *reinterpret_cast<const half8*>(&tl.vhi[row][(nb * 32 + l31) * 8]);
            blb[buf][s][nb] = *reinterpret_cast<const half8*>(&tl.vlo[row][(nb * 32 + l31) * 8]);
          }
      };
      constexpr bool kPrefB = PK && DPAD <= 16 && (NB == 1 || DPAD <= 8);  // elsewhere the second fragment set does not fit in 256 VGPRs
      if (kPrefB) load_b(0, 0);
      {
        static_assert(kMI == 2, "block order (jb, mi) = (0,0), (0,1), (1,0), (1,1)");
        floatx16 kd;
        load_a(0);
        dist(kd, 0);
        dist(kdn, 1);
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) exp_split_pair(kd, pr, kDiag, false, ah, al);
      }
#pragma unroll
      for (int blk = 0; blk < 2 * kMI; ++blk) {
        const int jb = blk / kMI, mi = blk % kMI;
        constexpr int NM = 6 * NB;
        constexpr int MD = NM - NKD - 1;  // the distance MFMAs of block b+2 go behind contraction MFMAs MD .. MD+NKD-1
        const bool has_next = blk + 1 < 2 * kMI, has_next2 = blk + 2 < 2 * kMI;
        const int jbn = (blk + 1) / kMI, min_ = (blk + 1) % kMI;
        const int jb2 = (blk + 2) / kMI, mi2 = (blk + 2) % kMI;
        // the two row blocks of a column block share its probe fragments: one LDS read per column block, not per block
        const int cur = kPrefB ? (jb & 1) : 0;
        if (!kPrefB && (mi == 0 || !PK)) load_b(0, jb);  // (the in-kernel-split variants have no registers to keep them)
        if (has_next2 && mi2 == 0) load_a(jb2);  // blocks (jb2, 0) and (jb2, 1) are issued from blocks blk and blk + 1
        const bool negn = ((jbn + min_) & 1) != 0;
        floatx16 kdn2;
#pragma unroll
        for (int r = 0; r < 16; ++r) kdn2[r] = 0.f;
        half8 ahn[2], aln[2];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          const int s = m / (3 * NB), nb = (m / 3) % NB, w = m % 3;
          __builtin_amdgcn_sched_barrier(0);
          acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w == 2 ? al[s] : ah[s], w == 1 ? blb[cur][s][nb] : bhb[cur][s][nb],
                                                               acc[mi][nb], 0, 0, 0);
          if constexpr (MD >= 0) {
            if (has_next2 && m >= MD && m - MD < NKD)
              kdn2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ajs[m - MD], bih[mi2][m - MD], kdn2, 0, 0, 0);
          } else {  // more distance MFMAs than contraction slots behind which to hide them (DPAD 32, one probe block: 7 against 6): spread evenly
            if (has_next2) {
#pragma unroll
              for (int q = 0; q < NKD; ++q)
                if (q >= NKD * m / NM && q < NKD * (m + 1) / NM)
                  kdn2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ajs[q], bih[mi2][q], kdn2, 0, 0, 0);
            }
          }
          if (kPrefB && m == 0 && blk == 0 && 2 * kMI > kMI) load_b(1, 1);  // the second column block's fragments, two blocks ahead
          __builtin_amdgcn_sched_barrier(0);
          if (has_next) {
#pragma unroll
            for (int pr = 0; pr < 8; ++pr)
              if (pr >= 8 * m / NM && pr < 8 * (m + 1) / NM) exp_split_pair(kdn, pr, kDiag && jbn == min_, negn, ahn, aln);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) {
          ah[0] = ahn[0]; ah[1] = ahn[1];
          al[0] = aln[0]; al[1] = aln[1];
          kdn = kdn2;
        }
      }
      return;
    }
    half8 ah[2], al[2];
    {  // pipeline prologue: fragments of block 0 of this tile
      floatx16 kd;
#pragma unroll
      for (int r = 0; r < 16; ++r) kd[r] = 0.f;
      if constexpr (DH) {
#pragma unroll
        for (int q = 0; q < NKD; ++q)
          kd = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(&tl.ajh[l31][q * 16 + lhi * 8]),
                                                      bih[0][q], kd, 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) kd = __builtin_amdgcn_mfma_f32_32x32x2f32(tl.aj[2 * s + lhi][l31], bi[0][s], kd, 0, 0, 0);
      }
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) exp_split_pair(kd, pr, kDiag, false, ah, al);  // block (jb 0, mi 0): diagonal in that tile, sign +
    }
#pragma unroll
    for (int blk = 0; blk < 2 * kMI; ++blk) {
      const int jb = blk / kMI, mi = blk % kMI;
      constexpr int NM = 6 * NB;       // contraction MFMAs of this block
      constexpr int NM1 = NM / 2;      // issued in phase 1, next to the distance MFMAs
      const bool has_next = blk + 1 < 2 * kMI;
      const int jbn = (blk + 1) / kMI, min_ = (blk + 1) % kMI;
      half8 bh[2][NB], bl[2][NB];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int row = (jb * 2 + s) * 2 + lhi;
          bh[s][nb] = *reinterpret_cast<const half8*>(&tl.vhi[row][(nb * 32 + l31) * 8]);
          bl[s][nb] = *reinterpret_cast<const half8*>(&tl.vlo[row][(nb * 32 + l31) * 8]);
        }
      float ajn[DH ? 1 : KS];
      half8 ajhn[DH ? NKD : 1];
      if (has_next) {
        if constexpr (DH) {
#pragma unroll
          for (int q = 0; q < NKD; ++q) ajhn[q] = *reinterpret_cast<const half8*>(&tl.ajh[jbn * 32 + l31][q * 16 + lhi * 8]);
        } else {
#pragma unroll
          for (int s = 0; s < KS; ++s) ajn[s] = tl.aj[2 * s + lhi][jbn * 32 + l31];
        }
      }
      const bool negn = DH && (((jbn + min_) & 1) != 0);
      floatx16 kdn;
#pragma unroll
      for (int r = 0; r < 16; ++r) kdn[r] = 0.f;
      half8 ahn[2], aln[2];
      auto contraction = [&](int m) {
        const int s = m / (3 * NB), nb = (m / 3) % NB, w = m % 3;
        acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w == 2 ? al[s] : ah[s], w == 1 ? bl[s][nb] : bh[s][nb],
                                                             acc[mi][nb], 0, 0, 0);
      };
      // ---- phase 1: the distance MFMAs of the next block, issued as early as possible (two per
      //      contraction MFMA) so that their result latency is covered by the rest of phase 1 -------
#pragma unroll
      for (int m = 0; m < NM1; ++m) {
        __builtin_amdgcn_sched_barrier(0);
        contraction(m);
        if (has_next) {
          if constexpr (DH) {
            if (m < NKD) kdn = __builtin_amdgcn_mfma_f32_32x32x16_f16(ajhn[m], bih[min_][m], kdn, 0, 0, 0);
          } else {
#pragma unroll
            for (int s = 0; s < KS; ++s)
              if (s >= 2 * m && s < 2 * (m + 1))
                kdn = __builtin_amdgcn_mfma_f32_32x32x2f32(ajn[s], bi[min_][s], kdn, 0, 0, 0);
          }
        }
      }
      if (has_next) {  // distance steps that did not fit next to the NM1 contraction MFMAs (d > 8 with one probe block)
        if constexpr (DH) {
#pragma unroll
          for (int q = NM1; q < NKD; ++q) kdn = __builtin_amdgcn_mfma_f32_32x32x16_f16(ajhn[q], bih[min_][q], kdn, 0, 0, 0);
        } else {
#pragma unroll
          for (int s = 2 * NM1; s < KS; ++s) kdn = __builtin_amdgcn_mfma_f32_32x32x2f32(ajn[s], bi[min_][s], kdn, 0, 0, 0);
        }
      }
      // ---- phase 2: exp / split of the next block between the remaining contraction MFMAs -------
#pragma unroll
      for (int m = NM1; m < NM; ++m) {
        __builtin_amdgcn_sched_barrier(0);
        contraction(m);
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) {
#pragma unroll
          for (int pr = 0; pr < 8; ++pr)
            if (pr >= 8 * (m - NM1) / (NM - NM1) && pr < 8 * (m - NM1 + 1) / (NM - NM1))
              exp_split_pair(kdn, pr, kDiag && jbn == min_, negn, ahn, aln);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) {
        ah[0] = ahn[0]; ah[1] = ahn[1];
        al[0] = aln[0]; al[1] = aln[1];
      }
    }
    };
    if (KIND != MFX_KERNEL_RBF && t * kTJ == i_wave) {
      do_tile(std::true_type{});
    } else {
      do_tile(std::false_type{});
    }
    if (t + 1 < ntile) {
      if constexpr (!PK) store_tile(tile[(t + 1) & 1]);
    }
    __syncthreads();  // (also drains the LDS-DMA: vmcnt(0))
  }
  if constexpr (PK) {
#pragma unroll
    for (int mi = 0; mi < kMI; ++mi)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int idx = mi * NB + nb;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][nb][r] += idx < kBlocks - 1 ? master[(idx * 16 + r) * 64] : mreg[r];
      }
  }
  const float s = outputscale[0], nz = gridDim.z > 1 ? 0.f : noise[0];
  float* yout = gridDim.z > 1 ? part + (int64_t)blockIdx.z * p * ldpart : y;
  const int64_t ldo = gridDim.z > 1 ? ldpart : ldy;
#pragma unroll
  for (int mi = 0; mi < kMI; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int64_t b = b0 + nb * 32 + l31;
      if (b >= p) continue;
      const float sb = s * vscale[2 * b + 1] * (1.f / 32768.f);
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

// y = sum_z part[z] + noise x   (fixed summation order: deterministic)
__global__ __launch_bounds__(256) void k_split_reduce(const float* __restrict__ part, int64_t ldpart, int nsplit, int64_t p,
                                                      int64_t nrow, int64_t ldy, const float* __restrict__ noise,
                                                      const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                      int64_t row0) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;  // local row: the point is row0 + i
  if (i >= nrow) return;
  float acc = 0.f;
  for (int z = 0; z < nsplit; ++z) acc += part[((int64_t)z * p + b) * ldpart + i];
  y[b * ldy + i] = fmaf(noise[0], x[b * ldx + row0 + i], acc);
}

// ------------------------------------------------------------------------------------------------
// Pre-pass of the pipelined matvec: the LDS images of every 64-column tile, written ONCE per matvec.
//   probe image   pkv[chunk][tile][lo?][row (jb, s, h)][probe][8]  : 16-B packs = the MFMA B fragments (hi / lo f16 of the scaled
//                 probe values of columns j = 32 jb + 16 s + 8 (e >> 2) + 4 h + (e & 3), e = 0..7)
//   column operand pka[tile][j][AROW] : [Ah | Ah | Al | 0] of c [-2 x_j, |x_j|^2 (+ shift or eps), 1], odd 32-column block negated
// grid (ntile, chunks + 1): blockIdx.y < chunks packs that probe chunk, the last one the column operand.
// ------------------------------------------------------------------------------------------------
template <int DPAD, int NB, int KIND>
__global__ __launch_bounds__(256) void k_pack_tiles(const float* __restrict__ xs, const float* __restrict__ sq, int64_t n,
                                                    float* __restrict__ vscale, const float* __restrict__ amax_part, int gx,
                                                    const float* __restrict__ x, int64_t ldx, int64_t p,
                                                    uintx4* __restrict__ pkv, uintx4* __restrict__ pka, int* __restrict__ rangeflag) {
  constexpr int kTJ = 64;
  using Tile = RbfTileH3<DPAD, NB, kTJ>;
  constexpr int KD = Tile::KD, P = Tile::P, AROW = Tile::AROW;
  constexpr float cfac = KIND == MFX_KERNEL_RBF ? kNegHalfLog2e : (KIND == MFX_KERNEL_MATERN32 ? 3.f : 1.f) * kLog2e * kLog2e;
  const int tid = threadIdx.x;
  const int64_t t = blockIdx.x, j0 = t * kTJ, ntile = gridDim.x;
  if ((int)blockIdx.y < (int)gridDim.y - 1) {
    const int64_t b0 = (int64_t)blockIdx.y * P;
    uintx4* dst = pkv + ((int64_t)blockIdx.y * ntile + t) * (2 * 8 * P);
    // the power-of-two scales of this chunk's vectors from their slice maxima (k_row_amax_part); tile 0 publishes [s, 1/s] for the
    // epilogue of the matvec kernel
    __shared__ float svs[P];
    if (tid < P) {
      const int64_t b = b0 + tid;
      float sc = 0.f;
      if (b < p) {
        float m = 0.f;
        for (int g = 0; g < gx; ++g) m = fmaxf(m, amax_part[b * gx + g]);
        sc = row_scale_of(m);
        if (t == 0) {
          vscale[2 * b] = sc;
          vscale[2 * b + 1] = 1.f / sc;
        }
      }
      svs[tid] = sc;
    }
    __syncthreads();
    for (int f0 = tid; f0 < 8 * P; f0 += 256) {
      // 8 consecutive lanes read the 64 consecutive columns (256 B) of ONE probe row; the 16-B stores of a lane group land in
      // 8 different image rows, 8 consecutive probes each (a first version with lanes along the probes read 4 B per 512-KB
      // stride and cost a millisecond)
      const int row = f0 & 7, bq = f0 >> 3, f = row * P + bq;
      const int jb = row >> 2, s = (row >> 1) & 1, h = row & 1;
      const int64_t b = b0 + bq;
      half8 hh, ll;
      const float vs = svs[bq];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int64_t j = j0 + 32 * jb + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3);
        const float v = (b < p && j < n) ? x[b * ldx + j] * vs : 0.f;
        float fh, fl;
        split_hi_lo(v, fh, fl);
        hh[e] = (_Float16)fh;
        ll[e] = (_Float16)fl;
      }
      dst[f] = __builtin_bit_cast(uintx4, hh);
      dst[8 * P + f] = __builtin_bit_cast(uintx4, ll);
    }
  } else {
    __shared__ __attribute__((aligned(16))) _Float16 img[kTJ][AROW];
    for (int e = tid; e < kTJ * AROW; e += 256) (&img[0][0])[e] = (_Float16)0.f;
    __syncthreads();
    for (int e = tid; e < kTJ * KD; e += 256) {
      const int j = e / KD, kk = e % KD;
      const int64_t jg = j0 + j;
      float v;
      if (kk < DPAD) v = jg < n ? -2.f * cfac * xs[jg * DPAD + kk] : 0.f;
      else if (kk == DPAD) v = cfac * (jg < n ? sq[jg] : 0.f) + (KIND == MFX_KERNEL_RBF ? kKShift : kEpsC);
      else v = cfac;
      // largest magnitudes of the f16 images: this column-operand entry, and |x_j|^2 itself as a row-operand entry
      if (fabsf(v) > 6.0e4f || (kk == DPAD && jg < n && sq[jg] > 6.0e4f)) atomicOr(rangeflag, 1);
      float hi, lo;
      split_hi_lo(((j >> 5) & 1) ? -v : v, hi, lo);
      img[j][kk] = (_Float16)hi;
      img[j][KD + kk] = (_Float16)hi;
      img[j][2 * KD + kk] = (_Float16)lo;
    }
    __syncthreads();
    uintx4* dst = pka + t * (kTJ * AROW * 2 / 16);
    for (int f = tid; f < kTJ * AROW * 2 / 16; f += 256) dst[f] = reinterpret_cast<const uintx4*>(&img[0][0])[f];
  }
}

// column splits of the pipelined matvec (grid.z), for OCCUPANCY: too few 512-row workgroups to fill 256 CUs (small n, or a row
// shard of a large n).  512-row workgroups of 8 waves, one per CU and round: with s column splits the launch takes
// ceil(wgs s / 256) rounds of 1/s of the columns each; pick the s (at most 16, at least 8 tiles per split) that minimises
// rounds / s, plus a small charge per split for the prologue and the partial sums.  (n = 45 730: 90 row blocks -> s = 8, 720
// workgroups in 3 rounds = 0.375 of an unsplit launch, where "fill one round" (s = 2) gives 0.5.)
// Which kernel runs the chunks (of 64 vectors, or one of at most 32) of an RBF operator with d <= 8 (BASELINE config 4's matvec,
// config 2's): the fat-wave kernel (mfx_rbf_fat.hip: one wave per SIMD; at 64 vectors 82 % matrix-pipe share, -16 % cycles
// against h3).  MFX_RBF_FAT=0 runs the same-program
// kernel k_rbf_mfma_apply_h3 (two waves per SIMD) instead -- the one A/B switch kept, because it reproduces the comparison of
// DESIGN.md §3.2 and the bit-identity test of the two kernels; h3 also takes every other shape.
static bool rbf_fat() {
  static const int v = [] {
    const char* e = getenv("MFX_RBF_FAT");
    return e ? atoi(e) : 1;
  }();
  return v != 0;
}

static int rbf_split_count(int64_t nrow, int64_t n, int64_t p, int dpad) {
  static const int forced = [] {
    const char* e = getenv("MFX_RBF_SPLIT");  // A/B: force the split count
    return e ? atoi(e) : 0;
  }();
  const int64_t pw = (p <= 32 || dpad > 16) ? 32 : 64;  // vectors per chunk (DPAD 32: one probe block per chunk)
  const int64_t wgs = ((nrow + 511) / 512) * ((p + pw - 1) / pw);
  const int64_t ntile = (n + 63) / 64;
  int64_t smax = ntile / 8;
  if (forced > 0) return forced <= 16 && forced <= ntile ? forced : 1;
  if (smax > 16) smax = 16;
  if (smax < 1 || wgs >= 2048) return 1;
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= (int)smax; ++s) {
    const double rounds = (double)((wgs * s + 255) / 256);
    const double cost = rounds / s + 0.004 * s;
    if (cost < best_cost - 1e-12) {
      best_cost = cost;
      best = s;
    }
  }
  return best;
}

static int64_t rbf_pack_bytes_v(int64_t n, int64_t p, int dpad) {
  const int64_t P = (p <= 32 || dpad > 16) ? 32 : 64, chunks = (p + P - 1) / P, ntile = (n + 63) / 64;
  return chunks * ntile * 2 * 8 * P * 16;
}
int64_t rbf_pack_ws_bytes(const mfx_operator* op, int64_t p) {
  if (op->dtype != MFX_F32 || op->d > 32 || p < 1) return 0;
  const int64_t ntile = (op->n + 63) / 64;
  const int dpad = op->d <= 4 ? 4 : op->d <= 8 ? 8 : op->d <= 12 ? 12 : op->d <= 16 ? 16 : 32;
  const int64_t arow = ((3 * (dpad + 2) + 15) / 16) * 16 + 8;
  return align_up(rbf_pack_bytes_v(op->n, p, dpad), 256) + align_up(ntile * 64 * arow * 2, 256) +
         align_up((int64_t)rbf_split_count(op_nrows(op), op->n, p, dpad) * p * align_up(op_nrows(op), 4) * 4, 256);
}

template <int DPAD, int NB, int KIND>
static int launch_apply_h3k(const mfx_operator* op, const float* xs, const float* sq, const float* x, int64_t ldx,
                            float* y, int64_t ldy, int64_t p, float* vscale, void* pk, hipStream_t stream) {
  const int64_t n = op->n;
  const int64_t row0 = op_row0(op), nrow = op_nrows(op), rend = row0 + nrow;
  MFX_REQUIRE(row0 % 64 == 0, MFX_ERR_INVALID, "matrix-core Gram matvec: row0 = %lld must be a multiple of 64", (long long)row0);
  // the pre-packed tile images need the caller's pack workspace (mfx_workspace_bytes sizes it); DPAD = 32 (16 < d <= 32, round 5): only the
  // in-kernel-split form with fp32-MFMA distances is built -- the packed images and the fat-wave kernel keep 3 KD / 16 f16 distance operands
  // per block resident or in the 160 KB of LDS next to the chain masters, which stops at DPAD = 16
  const bool pack = pk != nullptr && (DPAD <= 16 || NB == 1);
  // vscale region (65536 x 3 floats): [0, 2p) scales, [2p, 3p) |max| bit patterns (in-kernel-split path only), [3p] f16 range flag,
  // [3p + 64, ...) slice maxima of the pre-packed path
  int* rangeflag = reinterpret_cast<int*>(vscale + 3 * p);
  float* amax_part = vscale + 3 * p + 64;
  int64_t gx = (n + 8191) / 8192;
  if (gx > 64) gx = 64;
  while (gx > 1 && 3 * p + 64 + p * gx > (int64_t)65536 * 3) --gx;
  // (one slice per vector must fit behind the scales and the flag: p <= 49136 -- far beyond any probe count; unguarded before round 5)
  MFX_REQUIRE(3 * p + 64 + p * gx <= (int64_t)65536 * 3, MFX_ERR_UNSUPPORTED, "matrix-core Gram matvec: %lld vectors exceed the scale / slice-maximum region of the workspace (at most 49136)", (long long)p);
  if (pack) {
    k_row_amax_part<<<dim3((unsigned)gx, (unsigned)p), 256, 0, stream>>>(x, ldx, n, amax_part, rangeflag);
    MFX_CHECK_LAUNCH();
  } else {
    MFX_TRY(row_scales(x, ldx, n, p, vscale, stream));
  }
  const unsigned chunks = (unsigned)((p + NB * 32 - 1) / (NB * 32));
  const dim3 grid((unsigned)((nrow + 255) / 256), chunks);  // 4-wave workgroups (256 rows); the pre-packed variant uses grid_pk
  const bool vec4 = (n % 4 == 0) && (nrow % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                    (reinterpret_cast<uintptr_t>(y) % 16 == 0);
  uintx4* pkv = nullptr;
  uintx4* pka = nullptr;
  const int64_t ntile = (n + 63) / 64;
  const int64_t off_a = align_up((int64_t)chunks * ntile * 2 * 8 * NB * 32 * 16, 256);
  const int64_t arow = ((3 * (DPAD + 2) + 15) / 16) * 16 + 8;
  float* part = pk ? reinterpret_cast<float*>(static_cast<char*>(pk) + off_a + align_up(ntile * 64 * arow * 2, 256)) : nullptr;
  const int64_t ldpart = align_up(nrow, 4);  // the partials have their own stride (any n, any ldy)
  const int nsplit = part ? rbf_split_count(nrow, n, p, DPAD) : 1;
  const dim3 grid3(grid.x, grid.y, (unsigned)nsplit);
  const dim3 grid_pk((unsigned)((nrow + 511) / 512), grid.y, (unsigned)nsplit);
  if constexpr (DPAD <= 16 || NB == 1) {
    if (pack) {
      pkv = static_cast<uintx4*>(pk);
      pka = reinterpret_cast<uintx4*>(static_cast<char*>(pk) + off_a);
      k_pack_tiles<DPAD, NB, KIND><<<dim3((unsigned)ntile, chunks + 1), 256, 0, stream>>>(xs, sq, n, vscale, amax_part, (int)gx, x, ldx, p, pkv, pka, rangeflag);
      MFX_CHECK_LAUNCH();
    }
  }
  // LDS: the two tile buffers + (pre-packed variant) the chain masters of 8 waves x (2 NB - 1) blocks x 16 registers x 64 lanes
#define MFX_H3_LAUNCH(V4, DHV, PKV, FLAG)                                                                            \
  {                                                                                                                  \
    constexpr size_t kSm = 2 * sizeof(RbfTileH3<DPAD, NB, 64, !(PKV)>) + ((PKV) ? (size_t)8 * (2 * NB - 1) * 16 * 64 * 4 : 0); \
    if (kSm > 64 * 1024)                                                                                             \
      MFX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rbf_mfma_apply_h3<DPAD, NB, V4, KIND, DHV, PKV>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSm));                       \
    k_rbf_mfma_apply_h3<DPAD, NB, V4, KIND, DHV, PKV><<<(PKV) ? grid_pk : grid3, 64 * H3Waves<PKV>::value, kSm, stream>>>( \
        xs, sq, n, (const float*)op->outputscale, (const float*)op->noise, vscale, x, ldx, y, ldy, p, pkv, pka, part, FLAG, ldpart, row0, rend); \
  }
  if constexpr (DPAD > 16 && NB > 1) {
    if (vec4) MFX_H3_LAUNCH(true, false, false, nullptr) else MFX_H3_LAUNCH(false, false, false, nullptr)
  } else if (pack) {
    bool done = false;
    if constexpr (KIND == MFX_KERNEL_RBF && DPAD <= 16) {
      if (rbf_fat()) {  // fat waves (mfx_rbf_fat.hip): four waves of 128 rows, one per SIMD
        MFX_TRY(rbf_fat_launch(DPAD, NB, vec4, grid_pk, stream, xs, sq, n, (const float*)op->outputscale, (const float*)op->noise, vscale,
                               x, ldx, y, ldy, p, pkv, pka, part, rangeflag, ldpart, row0, rend));
        done = true;
      }
    }
    if (!done) {
      if (vec4) MFX_H3_LAUNCH(true, true, true, rangeflag) else MFX_H3_LAUNCH(false, true, true, rangeflag)
    }
    // f16 range guard: this launch returns at once unless k_pack_tiles raised the flag, in which case the launches above did
    if (vec4) MFX_H3_LAUNCH(true, false, false, rangeflag) else MFX_H3_LAUNCH(false, false, false, rangeflag)
  } else {  // no pack workspace: fp32-MFMA distances, in-kernel split of the probe tiles
    if (vec4) MFX_H3_LAUNCH(true, false, false, nullptr) else MFX_H3_LAUNCH(false, false, false, nullptr)
  }
#undef MFX_H3_LAUNCH
  MFX_CHECK_LAUNCH();
  if (nsplit > 1) {
    k_split_reduce<<<dim3((unsigned)((nrow + 255) / 256), (unsigned)p), 256, 0, stream>>>(part, ldpart, nsplit, p, nrow, ldy,
                                                                                        (const float*)op->noise, x, ldx, y, row0);
    MFX_CHECK_LAUNCH();
  }
  return MFX_OK;
}

template <int DPAD, int NB>
static int launch_apply_h3(const mfx_operator* op, const float* xs, const float* sq, const float* x, int64_t ldx,
                           float* y, int64_t ldy, int64_t p, float* vscale, void* pk, hipStream_t stream) {
  switch (op->kernel_fn) {
    case MFX_KERNEL_RBF: return launch_apply_h3k<DPAD, NB, MFX_KERNEL_RBF>(op, xs, sq, x, ldx, y, ldy, p, vscale, pk, stream);
    case MFX_KERNEL_MATERN12: return launch_apply_h3k<DPAD, NB, MFX_KERNEL_MATERN12>(op, xs, sq, x, ldx, y, ldy, p, vscale, pk, stream);
    case MFX_KERNEL_MATERN32: return launch_apply_h3k<DPAD, NB, MFX_KERNEL_MATERN32>(op, xs, sq, x, ldx, y, ldy, p, vscale, pk, stream);
    default: set_error("unknown kernel_fn %d", op->kernel_fn); return MFX_ERR_INVALID;
  }
}

int rbf_mfma_apply_h3(const mfx_operator* op, const float* xs, const float* sq, int dpad, const float* x, int64_t ldx,
                      float* y, int64_t ldy, int64_t p, float* vscale, void* pk, hipStream_t stream) {
#define MFX_H3_CASE(D)                                                                             \
  case D:                                                                                          \
    return p <= 32 ? launch_apply_h3<D, 1>(op, xs, sq, x, ldx, y, ldy, p, vscale, pk, stream)      \
                   : launch_apply_h3<D, 2>(op, xs, sq, x, ldx, y, ldy, p, vscale, pk, stream)
  switch (dpad) {
    MFX_H3_CASE(4);
    MFX_H3_CASE(8);
    MFX_H3_CASE(12);
    MFX_H3_CASE(16);
    case 32:  // the pre-packed form exists with one probe block per chunk only (LDS); without the pack workspace: in-kernel split
      return (p <= 32 || pk) ? launch_apply_h3<32, 1>(op, xs, sq, x, ldx, y, ldy, p, vscale, pk, stream)
                             : launch_apply_h3<32, 2>(op, xs, sq, x, ldx, y, ldy, p, vscale, pk, stream);
    default: set_error("split matrix-core Gram matvec supports d <= 32"); return MFX_ERR_UNSUPPORTED;
  }
#undef MFX_H3_CASE
}

// ================================================================================================
// Parameter-gradient sweep, 3 x f16 split: S = L^T R on the f16 matrix pipe.
//   pre-pass  k_pack_f16: (batch, n) fp32 rows -> hi/lo f16 packs [kb][i][8] (8 consecutive batch rows of
//             column i = one 16-byte MFMA A/B fragment), scaled by one power of two per operand
//             (global |max| -> 2^14), zero padded to batch % 32 == 0 and n % 128 == 0: the GEMM main loop
//             has no conversions, no bounds checks, fragment loads are plain ds_read_b128;
//   GEMM      same 128 x 128 tiling / epilogue as k_rbf_mfma_grad, 24 f16 MFMAs per 32-row stage instead
//             of 64 fp32 MFMAs at twice the cycles.
// ================================================================================================
__global__ __launch_bounds__(256) void k_row_amax(const float* __restrict__ x, int64_t ldx, int64_t n,
                                                  float* __restrict__ amax) {
  __shared__ float sm[4];
  const int64_t b = blockIdx.x;
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(x[b * ldx + i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) amax[b] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// scale[0] = 2^(14 - e), scale[1] = 1 / scale[0] with max_b amax[b] = f 2^e
__global__ void k_global_scale(const float* __restrict__ amax, int64_t rows, float* __restrict__ scale) {
  __shared__ float sm[256];
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < rows; i += 256) m = fmaxf(m, amax[i]);
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int e = 0;
    float s = 1.f;
    if (sm[0] > 0.f && sm[0] < 3.0e38f) {
      frexpf(sm[0], &e);
      s = ldexpf(1.f, 14 - e < 126 ? 14 - e : 126);
    }
    scale[0] = s;
    scale[1] = 1.f / s;
  }
}

// Pseudo-random sign of column i of a packed gradient-GEMM operand (salt: which operand).  Column i of L carries sigma_i,
// column j of R carries tau_j, so the accumulator of S_ij holds sigma_i tau_j S_ij and the epilogue undoes the sign.
// Why: the f16 MFMA's floor bias (see the note above k_rbf_mfma_grad_h) is the SAME sign in every accumulator; with the operands' columns signed
// at random it enters S_ij as -sigma_i tau_j beta ulp -- sign-random over (i, j), uncorrelated with dK_ij/dtheta -- and
// sums like noise (~ |dK|_F) instead of coherently (~ sum_ij dK_ij, n times larger).  Measured: profiles/r02a_*.
__host__ __device__ __forceinline__ bool grad_col_sign(int64_t i, uint32_t salt) {
  uint32_t h = (uint32_t)i * 0x9E3779B1u + salt;
  h ^= h >> 15;
  h *= 0x85EBCA77u;
  h ^= h >> 13;
  h *= 0xC2B2AE3Du;
  return (h >> 31) != 0;
}
constexpr uint32_t kSaltL = 0x51ED270Bu, kSaltR = 0xB5297A4Du;

// ONE product instead of three for the rows that cannot matter, and the rows ORDERED BY SIZE.
// S = sum_b L_b^T R_b does not care in which order the batch rows are summed, and the adjoint states of the last Krylov steps are
// orders of magnitude smaller than those of the first (config 4: row maxima 8.8e3 at step 0, ~2e2 at steps 1-13, < 8 from step 30 on,
// 2e-3 at step 39; the basis rows are unit vectors).  The pre-pass therefore sorts the rows by the bound m_b = |L_b|max |R_b|max,
// largest first (a bitonic sort of <= 8192 keys in LDS, ties by the (step, probe) index: deterministic), and packs them in that order:
//   * the 32 rows that meet in one K = 32 MFMA step have similar magnitudes (what the (step, probe) order of round 2 was for: the
//     f16 MFMA aligns its products to the largest of them and truncates the rest);
//   * the cross products hi lo + lo hi of a row are 2^-11 of its hi hi product.  The TAIL of the sorted order -- the longest run of
//     the smallest rows whose bounds add up to at most 2^-MFX_GRAD_ONE_PRODUCT_LOG2 (default 2^-10) of the sum of ALL bounds -- is
//     multiplied hi hi only: with |lo| <= 2^-11 |x| the two cross terms of a row are at most 2 x 2^-11 m_b = 2^-10 m_b, so what that drops from
//     any S_ij is at most 2^-(10 + 10) = 2^-20 of sum_b m_b even if it all had one sign -- a bound relative to the summed row bounds
//     sum_b |L_b|max |R_b|max, NOT to |S_ij|; what protects the gradient (which cancels to 1e-4 .. 1e-5 of its terms) is measured: the
//     16-probe-set table of profiles/r05b_*.  The three-product sum itself drops terms of that size (lo lo, the MFMA's truncation).  Those stages (from stage n3 on) stage and
//     multiply their hi images only: a third of the MFMAs, half of the L2 -> LDS bytes.  (A per-row cut relative to the largest row
//     would let MANY small rows through that together carry the gradient; measured at config 4: per-row cut 2^-6 -> 4e-4 off.)
// Decided on the device per launch; no row qualifies when the rows are of similar size, and then nothing changes but the order.
// On row shards every rank picks its n3 from the maxima of ITS rows of L: the ranks may run different arithmetic on their tails
// (harmless: each rank's partial sums obey the bound above; the ranks' results are added in fp64 by the estimator's all-reduce).
// Knob: MFX_GRAD_ONE_PRODUCT_LOG2 at build time sets the default (10); the environment variable of the same name, read once per process,
// overrides it at run time -- 0 switches the tail off (every stage three products): the A/B without a rebuild.
#ifndef MFX_GRAD_ONE_PRODUCT_LOG2
#define MFX_GRAD_ONE_PRODUCT_LOG2 10
#endif
static int grad_one_product_log2() {
  static const int v = [] {
    const char* e = getenv("MFX_GRAD_ONE_PRODUCT_LOG2");
    const int x = e ? atoi(e) : MFX_GRAD_ONE_PRODUCT_LOG2;
    return x < 0 ? 0 : (x > 60 ? 60 : x);
  }();
  return v;
}
constexpr int kSortMax = 8192;  // rows the LDS sort takes (64 KB of 64-bit keys); larger batches keep the (step, probe) order, three products
// perm[bt] = source row of packed row bt (-1: a zero row of the padding); tail[0] = n3, the first one-product stage
__global__ __launch_bounds__(1024) void k_order_rows(const float* __restrict__ amaxL, const float* __restrict__ amaxR, int64_t batch,
                                                     int64_t inner, int64_t bpad, int npow2, int one_log2, int* __restrict__ perm,
                                                     int* __restrict__ tail) {
  extern __shared__ unsigned long long keys[];  // (bound bits << 32) | (0xFFFFFFFF - default position): sorted DESCENDING
  const int64_t outer = batch / inner;
  for (int i = threadIdx.x; i < npow2; i += 1024) {
    unsigned long long kv = 0;  // padding: after every real row
    if (i < batch) {
      const int64_t src = inner > 1 ? ((int64_t)i % outer) * inner + (int64_t)i / outer : i;  // default order: (step, probe)
      float m = amaxL[src] * amaxR[src];
      if (!(m >= 0.f) || m > 3.0e38f) m = 3.0e38f;  // NaN / inf rows first: never in the one-product tail
      kv = ((unsigned long long)__float_as_uint(m) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    }
    keys[i] = kv;
  }
  __syncthreads();
  for (int size = 2; size <= npow2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < npow2 / 2; t += 1024) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const unsigned long long a = keys[lo], b = keys[hi];
        if (desc ? a < b : a > b) {
          keys[lo] = b;
          keys[hi] = a;
        }
      }
      __syncthreads();
    }
  for (int64_t bt = threadIdx.x; bt < bpad; bt += 1024) {
    int src = -1;
    if (bt < batch) {
      const int64_t i = 0xFFFFFFFFu - (unsigned)(keys[bt] & 0xFFFFFFFFull);
      src = (int)(inner > 1 ? (i % outer) * inner + i / outer : i);
    }
    perm[bt] = src;
  }
  // the one-product tail: the longest run of the SMALLEST rows whose bounds add up to at most 2^-one_log2 of the sum of all bounds
  // (a row with a NaN / inf bound makes the sum infinite: no tail).  Block-wide: thread t owns `per` consecutive sorted rows; a suffix
  // scan over the threads' partial sums (wave shuffles + 16 wave totals in LDS) gives every thread the mass of the rows behind its
  // own, and the smallest row index whose suffix mass stays within the cut is found with one LDS atomicMin.  (Round 4 had thread 0 sum
  // and scan all <= 8192 keys alone, in front of both pack launches of every sweep.)
  __shared__ double wave_sum[16];
  __shared__ int first_row;
  const int nstage = (int)(bpad / 32);
  if (one_log2 <= 0) {
    if (threadIdx.x == 0) tail[0] = nstage;
    return;
  }
  const int per = npow2 >= 1024 ? npow2 / 1024 : 1;
  const int64_t r0 = (int64_t)threadIdx.x * per, r1 = r0 + per < batch ? r0 + per : batch;
  double mine = 0.0;
  for (int64_t bt = r1 - 1; bt >= r0; --bt) mine += (double)__uint_as_float((unsigned)(keys[bt] >> 32));
  // inclusive suffix sum over the lanes of a wave, then over the waves
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double suf = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_down(suf, off, 64);
    if (lane + off < 64) suf += o;
  }
  if (lane == 0) wave_sum[wv] = suf;
  if (threadIdx.x == 0) first_row = (int)batch;
  __syncthreads();
  double behind = 0.0, total = 0.0;  // mass of the waves behind mine; mass of all rows
  for (int w = 0; w < 16; ++w) {
    total += wave_sum[w];
    if (w > wv) behind += wave_sum[w];
  }
  const double cut = total < 1.0e38 ? ldexp(total, -one_log2) : -1.0;
  double mass = behind + (suf - mine);  // everything behind this thread's rows
  int64_t first = r1;
  for (int64_t bt = r1 - 1; bt >= r0; --bt) {
    const double m = (double)__uint_as_float((unsigned)(keys[bt] >> 32));
    if (mass + m > cut) break;
    mass += m;
    first = bt;
  }
  // rows are sorted by decreasing bound, so "suffix mass <= cut" holds from some row on: the smallest such row over all threads
  if (first < r1) atomicMin(&first_row, (int)first);
  __syncthreads();
  if (threadIdx.x == 0) tail[0] = (int)(((int64_t)first_row + 31) / 32);  // whole stages only
}
__global__ void k_order_default(int64_t batch, int64_t inner, int64_t bpad, int* __restrict__ perm, int* __restrict__ tail) {
  const int64_t bt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t outer = batch / inner;
  if (bt < bpad) perm[bt] = bt < batch ? (int)(inner > 1 ? (bt % outer) * inner + bt / outer : bt) : -1;
  if (bt == 0) tail[0] = (int)(bpad / 32);
}

// hi/lo packs: out[(kb * npad + i) * 8 + q] = piece of (+-) scale * x[perm[8 kb + q]][i], sign per column i = grad_col_sign(i, salt).
// The packed row order is k_order_rows' (by decreasing size).  Why the order matters beyond the one-product tail: the f16 MFMA aligns
// the products of an output element to the largest of them and truncates the others TOWARDS ZERO (tools/mfma_f16_trunc.hip), and the
// adjoint states of different Krylov steps differ by orders of magnitude -- with the caller's (probe, step) order every small product
// lost bits, always in the direction that shrinks its contribution, a sign-symmetric bias that no sign alternation can cancel
// (1.6e-4 / 4.3e-4 gradient error at n = 65536; round 2 packed (step, probe) for that reason, round 4 sorts outright).
__global__ __launch_bounds__(256) void k_pack_f16(const float* __restrict__ x, int64_t ldx, int64_t batch, int64_t n,
                                                  int64_t npad, const float* __restrict__ scale, uint32_t salt,
                                                  const int* __restrict__ perm, _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t kb = blockIdx.y;
  if (i >= npad) return;
  float s = scale[0];
  if (grad_col_sign(i, salt)) s = -s;
  half8 h, l;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int64_t src = perm[kb * 8 + q];  // k_order_rows: rows by decreasing size; -1 = a zero row of the padding
    const float v = (src >= 0 && i < n) ? x[src * ldx + i] * s : 0.f;
    float fh, fl;
    split_hi_lo(v, fh, fl);
    h[q] = (_Float16)fh;
    l[q] = (_Float16)fl;
  }
  *reinterpret_cast<half8*>(hi + (kb * npad + i) * 8) = h;
  *reinterpret_cast<half8*>(lo + (kb * npad + i) * 8) = l;
}

// rows per workgroup tile of the split gradient GEMM: 8 waves (4 x 2 of 64 x 64) on ONE staged tile pair -- the same two waves per
// SIMD as two 128 x 128 workgroups, but (256 + 128) instead of 2 x (128 + 128) operand columns through L2 -> LDS per stage
constexpr int kHM = 256;
constexpr int kHLd = kHM + 4;

// NBW = 32-column groups per wave: the workgroup tile is 256 rows x (2 NBW 32) columns, 8 waves as 4 x 2 of 64 x (NBW 32).
//   NBW = 2: 256 x 128 (round 1);  NBW = 4: 256 x 256 -- (256 + 256) instead of 2 x (256 + 128) operand columns through L2 -> LDS
//   per stage and per 256 x 256 of S, i.e. a third less of that traffic.  Its 128 accumulator registers
//   fit since the fp32 master accumulators are gone (see the note on the floor bias below).
// Round 4: the contraction runs on v_mfma_f32_16x16x32_f16 (a wave's 64 x (NBW 32) sub-tile = 4 x 2 NBW blocks of 16 x 16, the same 32 NBW
// accumulator registers): same flops and the same matrix-pipe cycles as on 32x32x16, but the chip holds a HIGHER CLOCK under this shape
// (tools/gradk_bench.hip, the K-loop alone on one box, alternating: 49.1 ms / 1.45 GHz on 32x32x16, 44.6 ms / 1.61 GHz on 16x16x32, both 90 %
// matrix pipe; MI355X_MICROARCH.md 'DVFS give-back' item 7).  One wave per SIMD on 128 x 128 (the round-3 plan) was measured there too and
// is SLOWER (51.9 ms: 88 % pipe with LDS-DMA issued by the computing wave, 82 % with register staging) -- a third fewer fragment bytes from LDS
// buy no clock.  A stage is now 32 batch rows (one K = 32 MFMA step): two 64-KB LDS images instead of a ring of four 16-row ones.
template <int DPAD, int NBW, bool REGEPI = false>
struct GradSmemH {
  static constexpr int TN = 2 * NBW * 32;  // columns of the workgroup tile
  // columns per pass of the LDS epilogue: 128; 64 for DPAD = 32 (16 < d <= 32, round 5), where x_i and x_j of the tile are twice as wide and a
  // 128-column S^T overlay would push the workgroup over the 160 KB of LDS
  static constexpr int GN = DPAD > 16 ? 64 : kGN;
  static constexpr int KQ = (DPAD + 2 + 3) / 4;  // k-steps of the epilogue's distance product (v_mfma_f32_16x16x4_f32)
  union {
    struct {
      _Float16 a_hi[2][4][kHM][8];  // [slot][kb-group within the 32-row stage][i][8]
      _Float16 a_lo[2][4][kHM][8];
      _Float16 b_hi[2][4][TN][8];
      _Float16 b_lo[2][4][TN][8];
    } st;
    float s_t[GN][kHLd];  // S^T of ONE pass of the epilogue
  } u;
  float xi[kHM][DPAD];
  float sqi[kHM];
  float xj[GN][DPAD];
  float sqj[GN];
  float sgj[GN];  // tau_j of the staged columns (+-1)
  float bq[REGEPI ? 4 * KQ : 1][REGEPI ? TN : 1];  // register epilogue: B' = [x_j, 1, |x_j|^2, 0..] of the tile's columns, k-major
  double red[8][DPAD + 2];
};


// The f16 MFMA truncates its internal sum towards -infinity: a bias of a fraction of an ulp per MFMA, invisible in
// any single accumulator but COHERENT across all n^2 accumulators of S, and the gradient sum_ij S_ij dK_ij/dtheta
// cancels to ~1e-4..1e-5 of its terms -- measured as a 0.1-1 % gradient error.  Round 1 cured it with alternating-sign
// K-chunks folded into fp32 master accumulators (64 more registers, 128 VALU instructions per chunk).  The pseudo-random
// column signs of the packed operands (grad_col_sign) do the same job for free: with them and WITHOUT the chunks the C4
// gradient is 9.8e-6 / 3.0e-5 off fp64, without either 2.2e-2 / 5.6e-2 (profiles/r02h_*) -- so the chunks and the master
// accumulators are gone, which is what makes room for the 256 x 256 tile.

template <int DPAD, int NBW, bool REGEPI>
__global__ __launch_bounds__(512, 1) void k_rbf_mfma_grad_h(const float* __restrict__ xs, const float* __restrict__ sq,
                                                            int64_t n, int64_t npad_l, int64_t npad_r, int ard, int kind,
                                                            const _Float16* __restrict__ Lh, const _Float16* __restrict__ Ll,
                                                            const _Float16* __restrict__ Rh, const _Float16* __restrict__ Rl,
                                                            int64_t nkb /* batch_pad / 8 */, int tiles_per_block,
                                                            uint32_t salt_l, uint32_t salt_r, double* __restrict__ partial,
                                                            int64_t row0, int64_t nrow, const int* __restrict__ one_product) {
  // rows: the nrow points row0 .. of X that the L operand covers (a row shard, or all n); columns: all n points
  using Smem = GradSmemH<DPAD, NBW, REGEPI>;
  constexpr int TN = Smem::TN;
  constexpr int NB16 = 2 * NBW;  // 16-column blocks of a wave's sub-tile
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  // blockIdx.x = XCD label (workgroups are dealt to XCDs round-robin in linear order); within an XCD, blockIdx.y
  // enumerates (row block, column sub-range) with the sub-range fastest, see kGSub
  const int64_t i0 = (int64_t)(blockIdx.y / kGSub) * kHM;
  const int64_t ntj = (n + TN - 1) / TN;
  const int64_t tj_begin = ((int64_t)blockIdx.x * kGSub + blockIdx.y % kGSub) * tiles_per_block;
  int64_t tj_end = tj_begin + tiles_per_block;
  if (tj_end > ntj) tj_end = ntj;

  for (int t = tid; t < kHM * DPAD; t += 512) {
    const int64_t g = i0 * DPAD + t;
    (&sm.xi[0][0])[t] = g < nrow * DPAD ? xs[row0 * DPAD + g] : 0.f;
  }
  if (tid < kHM) sm.sqi[tid] = (i0 + tid < nrow) ? sq[row0 + i0 + tid] : 0.f;

  constexpr int NG = REGEPI ? 3 : DPAD + 2;  // register epilogue: (lengthscale, outputscale, noise)
  double gsum[NG];
#pragma unroll
  for (int c = 0; c < NG; ++c) gsum[c] = 0.0;

  // Staging by LDS-DMA, no staging registers: a stage = 4 kb-groups (32 batch rows) x (256 | TN) columns x 16 B per operand image,
  // 1-KiB pieces (one wave-instruction = 64 columns of one kb-group).  Wave w moves, per stage:
  //   L hi and lo: columns 64 (w & 3) .., kb-groups (w >> 2) and (w >> 2) + 2                                  -> 4 pieces
  //   R hi and lo: TN = 256 the same -> 4 pieces;  TN = 128: columns 64 (w & 1) .., kb-group w >> 1             -> 2 pieces
  // Addresses are 32-bit byte offsets from the (scalar) image bases, advanced by one constant per stage.  TWO slots: the stage
  // being read and the next one in flight (requested when the last readers of its slot are done, a whole stage = 3072 matrix-pipe
  // cycles of the SIMD earlier for the lower half of the workgroup, half that for the upper half).
  const int nstage = (int)(nkb / 4);
  const uint32_t stage_bytes_l = (uint32_t)(4 * npad_l * 16), stage_bytes_r = (uint32_t)(4 * npad_r * 16);
  const int lcol = (wid & 3) * 64, lkb = wid >> 2;                          // wave-uniform
  const int rcol = TN == 256 ? (wid & 3) * 64 : (wid & 1) * 64, rkb = TN == 256 ? (wid >> 2) : (wid >> 1);
  const uint32_t offL = (uint32_t)(((int64_t)lkb * npad_l + i0 + lcol + lane) * 16);
  const uint32_t offR0 = (uint32_t)(((int64_t)rkb * npad_r + rcol + lane) * 16);
  const uint32_t kb2_l = (uint32_t)(2 * npad_l * 16), kb2_r = (uint32_t)(2 * npad_r * 16);
  const char* Lhb = reinterpret_cast<const char*>(Lh);
  const char* Llb = reinterpret_cast<const char*>(Ll);
  const char* Rhb = reinterpret_cast<const char*>(Rh);
  const char* Rlb = reinterpret_cast<const char*>(Rl);
  auto issue_stage = [&](uint32_t l_off, uint32_t r_off, int slot, int hi_only) {  // hi_only: a one-product stage (from n3 on: k_order_rows)
    glds16(Lhb + l_off, &sm.u.st.a_hi[slot][lkb][lcol][0]);
    glds16(Lhb + l_off + kb2_l, &sm.u.st.a_hi[slot][lkb + 2][lcol][0]);
    glds16(Rhb + r_off, &sm.u.st.b_hi[slot][rkb][rcol][0]);
    if constexpr (TN == 256) glds16(Rhb + r_off + kb2_r, &sm.u.st.b_hi[slot][rkb + 2][rcol][0]);
    if (!hi_only) {
      glds16(Llb + l_off, &sm.u.st.a_lo[slot][lkb][lcol][0]);
      glds16(Llb + l_off + kb2_l, &sm.u.st.a_lo[slot][lkb + 2][lcol][0]);
      glds16(Rlb + r_off, &sm.u.st.b_lo[slot][rkb][rcol][0]);
      if constexpr (TN == 256) glds16(Rlb + r_off + kb2_r, &sm.u.st.b_lo[slot][rkb + 2][rcol][0]);
    }
  };
  // stages from n3 on are one-product stages (a scalar of the pre-pass, k_order_rows; the same for every workgroup)
  int n3 = one_product[0];
  n3 = n3 < 0 ? 0 : (n3 > nstage ? nstage : n3);
  const int one_first = n3 == 0 ? 1 : 0;

  for (int64_t tj = tj_begin; tj < tj_end; ++tj) {
    const int64_t j0 = tj * TN;
    floatx4 acc[4][NB16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < NB16; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

    uint32_t ol = offL, orr = offR0 + (uint32_t)(j0 * 16);
    // (opaque per tile: otherwise the four 64-bit L addresses of a stage, which do not depend on the tile, are hoisted out of the
    //  tile loop and sit in -- or spill from -- registers the K-loop needs)
    asm volatile("" : "+v"(ol));
    if (!REGEPI || tj == tj_begin) {
      __syncthreads();  // previous tile's epilogue reads of the overlaid S^T tile are done
      issue_stage(ol, orr, 0, one_first);
    }  // (REGEPI: stage 0 of this tile was requested before the previous tile's epilogue)
    ol += stage_bytes_l;
    orr += stage_bytes_r;
    // PING-PONG: waves w and w + 4 share a SIMD.  With one barrier per stage all eight waves read their fragments from LDS
    // together (192 KB per stage with the matrix pipe idle) and then queue their MFMAs together.  Here the two
    // halves of the workgroup run half a stage apart: two barriers per stage -- B1 before the fragment reads, B2 before the
    // MFMAs -- and the upper half takes one extra barrier first, so that while one wave of a SIMD is in its MFMA cluster
    // the other one is in its read cluster.  Who needs what by when (A = waves 0-3, B = waves 4-7; barrier
    // numbers in A's count: A.B1(st) = 2 st, A.B2(st) = B.B1(st) = 2 st + 1, B.B2(st) = 2 st + 2):
    //   * A reads stage st between barriers 2 st and 2 st + 1, B between 2 st + 1 and 2 st + 2 (each waits for its reads,
    //     lgkmcnt(0), before its B2);
    //   * the slot of stage st - 1 is therefore free after barrier 2 st: every wave requests stage st + 1 into it after its
    //     B1(st) (A behind barrier 2 st, B behind 2 st + 1);
    //   * stage st + 1 is first read behind barrier 2 st + 2: every wave waits for ITS pieces (vmcnt(0): nothing younger is in
    //     flight) before it ARRIVES there -- A before its B1(st + 1), B before its B2(st);  stage 0 before the first barrier.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (wid >= 4) __builtin_amdgcn_s_barrier();  // the half-stage offset of the upper half
    // stages [0, n3): three products; stages [n3, nstage): hi hi only (n3 from k_order_rows).  TWO loops, each with its own straight-line
    // body: with the choice as a branch inside one loop the allocator spilled 140-190 registers around the joins
    auto run_stages = [&](auto one_c, int st_begin, int st_end) {
      constexpr bool ONE = decltype(one_c)::value;
      for (int st = st_begin; st < st_end; ++st) {
        const int slot = st & 1;
        if (wid < 4) __builtin_amdgcn_s_waitcnt(0x0F70);  // (A) my pieces of this stage, requested a stage ago
        __builtin_amdgcn_s_barrier();  // B1
        if (st + 1 < nstage) issue_stage(ol, orr, slot ^ 1, ONE ? 1 : (st + 1 >= n3 ? 1 : 0));
        ol += stage_bytes_l;
        orr += stage_bytes_r;
        half8 ah[4], al[ONE ? 1 : 4], bh[NB16], bl[ONE ? 1 : NB16];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          ah[a] = *reinterpret_cast<const half8*>(&sm.u.st.a_hi[slot][lq][wm * 64 + a * 16 + l15][0]);
          if constexpr (!ONE) al[a] = *reinterpret_cast<const half8*>(&sm.u.st.a_lo[slot][lq][wm * 64 + a * 16 + l15][0]);
        }
#pragma unroll
        for (int b = 0; b < NB16; ++b) {
          bh[b] = *reinterpret_cast<const half8*>(&sm.u.st.b_hi[slot][lq][wn * (NBW * 32) + b * 16 + l15][0]);
          if constexpr (!ONE) bl[b] = *reinterpret_cast<const half8*>(&sm.u.st.b_lo[slot][lq][wn * (NBW * 32) + b * 16 + l15][0]);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the fragments are in registers
        if (wid >= 4) __builtin_amdgcn_s_waitcnt(0x0F70);  // (B) my pieces of stage st + 1
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // B2  (no s_setprio around the MFMA cluster: measured 0.5 % faster without, tools/gradk_bench.hip)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < NB16; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
            if constexpr (!ONE) {
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    run_stages(std::false_type{}, 0, n3);
    run_stages(std::true_type{}, n3, nstage);
    if (wid < 4) __builtin_amdgcn_s_barrier();  // the lower half catches up: both halves have taken 2 nstage + 1 barriers
    if constexpr (REGEPI) {
      // Everything below that depends only on the workgroup's rows is invariant over the tile loop; hoisted out of it, it
      // would sit in registers through the K-loop, which has none to spare (measured: 380 B of spills, one reload per stage,
      // +14 % run time; adding an opaque zero does not help -- the compiler hoists the rest of the sum).  The row index itself goes
      // through opaque per-tile copies of the two lane coordinates: nothing that depends on them can leave the tile loop, and
      // only the coordinates themselves (which the K-loop needs anyway) stay live across it.
      int l15_t = l15, lq4_t = 4 * lq;
      asm volatile("" : "+v"(l15_t), "+v"(lq4_t));
      // ---- register epilogue (one lengthscale): the kernel entries are formed IN THE ACCUMULATOR LAYOUT -- the distance block
      //      D = A' B'^T with A' = [-2 x_i, |x_i|^2, 1], B' = [x_j, 1, |x_j|^2] on the fp32 MFMA (v_mfma_f32_16x16x4_f32: lane =
      //      column j, register r <-> row 4 (lane >> 4) + r, exactly like acc) -- so S o dK is 4 elementwise products per block:
      //      no S^T through LDS, no barriers, and the slots are free, so the next tile's first stage is in flight while this runs.
      constexpr int KQ = Smem::KQ;
      // (bq of the previous tile: every wave has passed at least one K-loop barrier since it read it)
      uint32_t taub[NB16];  // tau_j as a sign-bit mask
#pragma unroll
      for (int b = 0; b < NB16; ++b) taub[b] = grad_col_sign(j0 + wn * (NBW * 32) + b * 16 + l15, salt_r) ? 0x80000000u : 0u;
      for (int t = tid; t < TN * 4 * KQ; t += 512) {  // B' of the tile's columns (xs is a few MB: L2), BEFORE the prefetch
        const int col = t % TN, c = t / TN;
        const int64_t j = j0 + col;
        float v = 0.f;
        if (c < DPAD) v = j < n ? xs[j * DPAD + c] : 0.f;
        else if (c == DPAD) v = 1.f;
        else if (c == DPAD + 1) v = j < n ? sq[j] : 0.f;
        sm.bq[c][col] = v;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): my bq writes have landed (the DMA counter is left alone)
      __builtin_amdgcn_s_barrier();        // every wave is done reading the last stage of this tile, and bq is complete
      if (tj + 1 < tj_end) issue_stage(ol - (uint32_t)(nstage + 1) * stage_bytes_l, offR0 + (uint32_t)((j0 + TN) * 16), 0, one_first);
      const bool diag_tile = (row0 + i0 < j0 + TN) && (j0 < row0 + i0 + kHM);
      float gl = 0.f, gs = 0.f, gn = 0.f;
      // one straight-line copy of the body per (tile meets the diagonal or not): with the diagonal test as a run-time branch inside
      // the 128 unrolled entries the compiler spills.  Per entry FIVE VALU instructions (the two waves of a SIMD run their epilogues
      // side by side, VALU-bound): the distance block comes out of the MFMA already scaled, t = c dist with c = -log2(e) / 2 folded
      // into A'; k = v_exp_f32(t) with the clamp modifier (k <= 1: the reference's max(dist, 0), util/gp_util.py:173, where it matters);
      // sigma_i tau_j undone by ONE v_xor3 with two sign masks; u = s k; sum u; sum u t (= c sum s k dist, unscaled at the end --
      // a distance that is negative by round-off enters as it is, ~1e-7 of one entry).
      auto body = [&](auto diag_c) {
        constexpr bool DIAG = decltype(diag_c)::value;
        int dj[NB16];  // column of block b's lane, relative to the first row of the workgroup's tile
#pragma unroll
        for (int b = 0; b < NB16; ++b) dj[b] = (int)(j0 - (row0 + i0)) + wn * (NBW * 32) + b * 16 + l15_t;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int il = wm * 64 + a * 16 + l15_t;
          float ai[KQ];
#pragma unroll
          for (int sI = 0; sI < KQ; ++sI) {
            const int c = 4 * sI + lq;
            const float v = c < DPAD ? -2.f * sm.xi[il][c < DPAD ? c : 0] : (c == DPAD ? sm.sqi[il] : (c == DPAD + 1 ? 1.f : 0.f));
            ai[sI] = kNegHalfLog2e * v;
          }
          uint32_t sgm[4];  // sigma of the row of register r as a sign-bit mask
          const int row_r0 = (int)i0 + wm * 64 + a * 16 + lq4_t;  // (local rows: < 2^31)
#pragma unroll
          for (int r = 0; r < 4; ++r) sgm[r] = grad_col_sign(row_r0 + r, salt_l) ? 0x80000000u : 0u;
#pragma unroll
          for (int b = 0; b < NB16; ++b) {
            floatx4 D = {0.f, 0.f, 0.f, 0.f};
            // a few distance blocks in flight, not all of them: the first operand of every other block waits (through an opaque
            // move) for the sums of the previous ones -- otherwise the optimiser gathers the MFMAs of all blocks up front
            float a0 = ai[0];
            if ((b & 1) == 0) asm volatile("" : "+v"(a0), "+v"(gs), "+v"(gl));
#pragma unroll
            for (int sI = 0; sI < KQ; ++sI)
              D = __builtin_amdgcn_mfma_f32_16x16x4f32(sI == 0 ? a0 : ai[sI], sm.bq[4 * sI + lq][wn * (NBW * 32) + b * 16 + l15], D, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float t = D[r];
              // sigma_i tau_j S_ij -> S_ij: flip the sign bit
              const float s_ij = __uint_as_float(__float_as_uint(acc[a][b][r]) ^ sgm[r] ^ taub[b]);
              if constexpr (DIAG) {
                const bool on = dj[b] == wm * 64 + a * 16 + lq4_t + r;
                t = on ? 0.f : t;
                gn += on ? s_ij : 0.f;
              }
              // exp2 with the clamp modifier, as an instruction the COMPILER emits (fmed3(x, 0, 1) folds into v_exp_f32 ... clamp): t comes
              // straight out of an MFMA, and the wait states between an MFMA's write and a VALU read are inserted by the compiler's hazard
              // recogniser -- which does not look inside inline asm.  Round 4 had this as asm("v_exp_f32_e64 ... clamp"): with the
              // 2-MFMA distance chain of d <= 4 it read the accumulator before the MFMA had written it (S o dK off by O(1) for every
              // non-ARD RBF operator with d = 2 .. 4; found by tools/fuzz_matvec.py in round 5), with the 3-MFMA chain of d = 5 .. 8
              // the same read happened to land late enough.
              const float kv = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(t), 0.f, 1.f);
              const float u = s_ij * kv;
              gs += u;
              gl = fmaf(u, t, gl);
            }
          }
        }
      };
      if (diag_tile) body(std::true_type{}); else body(std::false_type{});
      gsum[0] += (double)gl * (1.0 / (double)kNegHalfLog2e);
      gsum[1] += (double)gs;
      gsum[2] += (double)gn;
    } else
    // ---- epilogue, in passes of 128 columns (the S^T overlay holds one pass): the waves whose blocks lie in the pass
    //      dump them, then thread = row i walks 64 of the pass's columns (as in k_rbf_mfma_grad) -------------------------
#pragma unroll
    for (int pass = 0; pass < TN / Smem::GN; ++pass) {
      __syncthreads();  // all waves are done with the stage images (pass 0) / with the previous pass's S^T
#pragma unroll
      for (int b = 0; b < NB16; ++b) {
        const int cb = wn * (NBW * 32) + b * 16;  // first column of the block within the tile
        if (cb / Smem::GN != pass) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float4 q;
          q.x = acc[a][b][0]; q.y = acc[a][b][1]; q.z = acc[a][b][2]; q.w = acc[a][b][3];
          *reinterpret_cast<float4*>(&sm.u.s_t[(cb % Smem::GN) + l15][wm * 64 + a * 16 + 4 * lq]) = q;
        }
      }
      const int64_t jp = j0 + pass * Smem::GN;
      for (int t = tid; t < Smem::GN * DPAD; t += 512) {
        const int64_t g = jp * DPAD + t;
        (&sm.xj[0][0])[t] = g < n * DPAD ? xs[g] : 0.f;
      }
      if (tid < Smem::GN) {
        sm.sqj[tid] = (jp + tid < n) ? sq[jp + tid] : 0.f;
        sm.sgj[tid] = grad_col_sign(jp + tid, salt_r) ? -1.f : 1.f;
      }
      __syncthreads();
      // (an opaque copy of the thread index per pass: the row quantities below are invariant over the tile loop, and hoisted out of
      //  it they spill around the K-loop, which has no register to spare)
      int tid_t = tid;
      asm volatile("" : "+v"(tid_t));
      const int il = tid_t & (kHM - 1), jh = (tid_t >> 8) * (Smem::GN / 2);
      const int64_t i = row0 + i0 + il;  // the point; L / the packs are indexed by the local row i0 + il
      const float sgi = grad_col_sign(i0 + il, salt_l) ? -1.f : 1.f;  // sigma_i: the accumulators hold sigma_i tau_j S_ij
      float xiv[DPAD];
#pragma unroll
      for (int c = 0; c < DPAD; c += 4) {
        const float4 q = *reinterpret_cast<const float4*>(&sm.xi[il][c]);
        xiv[c] = q.x; xiv[c + 1] = q.y; xiv[c + 2] = q.z; xiv[c + 3] = q.w;
      }
      const float si = sm.sqi[il];
      float gt[DPAD + 2];
#pragma unroll
      for (int c = 0; c < DPAD + 2; ++c) gt[c] = 0.f;
#pragma unroll(DPAD > 16 ? 1 : 2)  // (DPAD 32: x_j of two columns in flight is 64 registers -- with one, no spills)
      for (int jj = 0; jj < Smem::GN / 2; ++jj) {
        const int jl = jh + jj;
        const int64_t j = jp + jl;
        float xjv[DPAD];
#pragma unroll
        for (int c = 0; c < DPAD; c += 4) {
          const float4 q = *reinterpret_cast<const float4*>(&sm.xj[jl][c]);
          xjv[c] = q.x; xjv[c + 1] = q.y; xjv[c + 2] = q.z; xjv[c + 3] = q.w;
        }
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < DPAD; ++c) dot = fmaf(xiv[c], xjv[c], dot);
        float dist = fmaf(-2.f, dot, si + sm.sqj[jl]);
        dist = fmaxf(dist, 0.f);
        const bool live = (i0 + il < nrow) && (j < n);
        const float s_ij = live ? sm.u.s_t[jl][il] * (sgi * sm.sgj[jl]) : 0.f;
        float kv, wl;
        grad_weights(kind, i == j ? 0.f : dist, kv, wl);
        gt[DPAD] = fmaf(s_ij, kv, gt[DPAD]);
        const float w = s_ij * wl;
        if (ard) {
#pragma unroll
          for (int c = 0; c < DPAD; c += 2) {  // packed fp32 (v_pk_add / v_pk_mul / v_pk_fma): two dimensions per instruction
            const floatx2 xi2 = {xiv[c], xiv[c + 1]}, xj2 = {xjv[c], xjv[c + 1]};
            const floatx2 df = xi2 - xj2;
            floatx2 g2 = {gt[c], gt[c + 1]};
            g2 = __builtin_elementwise_fma(df * w, df, g2);
            gt[c] = g2[0];
            gt[c + 1] = g2[1];
          }
        } else {
          gt[0] = fmaf(w, dist, gt[0]);
        }
        if (i == j) gt[DPAD + 1] += s_ij;
      }
#pragma unroll
      for (int c = 0; c < DPAD + 2; ++c) gsum[c] += (double)gt[c];
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // no LDS-DMA of mine is left in flight
  if (tid < 8 * (DPAD + 2)) (&sm.red[0][0])[tid] = 0.0;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NG; ++c) {
    const double v = wave_sum(gsum[c]);
    // register epilogue: slots (0, DPAD, DPAD + 1) of the (DPAD + 2)-wide partial row, as the LDS epilogue fills them
    const int slot = REGEPI ? (c == 0 ? 0 : DPAD - 1 + c) : c;
    if (lane == 0) sm.red[wid][slot] = v;
  }
  __syncthreads();
  if (tid < DPAD + 2) {
    const int64_t blk = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    partial[blk * (DPAD + 2) + tid] = ((sm.red[0][tid] + sm.red[1][tid]) + (sm.red[2][tid] + sm.red[3][tid])) +
                                      ((sm.red[4][tid] + sm.red[5][tid]) + (sm.red[6][tid] + sm.red[7][tid]));
  }
}

int64_t rbf_grad_h_ws_bytes(int64_t n, int64_t batch) {
  const int64_t npad = (n + kHM - 1) / kHM * kHM, bpad = (batch + 31) / 32 * 32;
  return 4 * bpad * npad * (int64_t)sizeof(_Float16) + 3 * bpad * (int64_t)sizeof(float) + 1536;
}

template <int DPAD, int NBW>
static int launch_grad_h_t(const mfx_operator* op, const float* xs, const float* sq, int64_t n, int64_t npad_l, int64_t npad,
                           const _Float16* Lh, const _Float16* Ll, const _Float16* Rh, const _Float16* Rl, int64_t bpad,
                           uint32_t salt_l, uint32_t salt_r, double* partial, int64_t* nblocks_out, const int* one_product,
                           hipStream_t stream) {
  constexpr int TN = GradSmemH<DPAD, NBW>::TN;
  const int64_t row0 = op_row0(op), nrow = op_nrows(op);
  const int64_t nti = (nrow + kHM - 1) / kHM, ntj = (n + TN - 1) / TN;
  const int tiles_per_block = (int)((ntj + kGSplit * kGSub - 1) / (kGSplit * kGSub));
  const dim3 grid(kGSplit, (unsigned)(nti * kGSub));
  // RBF kernel, one lengthscale, d <= 8: the register epilogue; Matern / ARD / d > 8: the LDS epilogue
  bool launched = false;
  if constexpr (DPAD <= 8) {
    if (!op->ard && op->kernel_fn == MFX_KERNEL_RBF) {
      const size_t sh = sizeof(GradSmemH<DPAD, NBW, true>);
      MFX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rbf_mfma_grad_h<DPAD, NBW, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
      k_rbf_mfma_grad_h<DPAD, NBW, true><<<grid, 512, sh, stream>>>(xs, sq, n, npad_l, npad, op->ard, op->kernel_fn, Lh, Ll, Rh, Rl,
                                                                    bpad / 8, tiles_per_block, salt_l, salt_r, partial, row0, nrow, one_product);
      launched = true;
    }
  }
  if (!launched) {
    const size_t sh = sizeof(GradSmemH<DPAD, NBW, false>);
    MFX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rbf_mfma_grad_h<DPAD, NBW, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    k_rbf_mfma_grad_h<DPAD, NBW, false><<<grid, 512, sh, stream>>>(xs, sq, n, npad_l, npad, op->ard, op->kernel_fn, Lh, Ll, Rh, Rl,
                                                                   bpad / 8, tiles_per_block, salt_l, salt_r, partial, row0, nrow, one_product);
  }
  MFX_CHECK_LAUNCH();
  *nblocks_out = nti * kGSub * kGSplit;
  return MFX_OK;
}

template <int DPAD>
static int launch_grad_h(const mfx_operator* op, const float* xs, const float* sq, const float* L, int64_t ldl,
                         const float* R, int64_t ldr, int64_t batch, int64_t inner, double* partial, int64_t* nblocks_out,
                         void* hws, hipStream_t stream) {
  const int64_t n = op->n, nrow = op_nrows(op);
  const int64_t npad = (n + kHM - 1) / kHM * kHM, npad_l = (nrow + kHM - 1) / kHM * kHM, bpad = (batch + 31) / 32 * 32;
  if (inner < 1 || batch % inner != 0) inner = 1;
  char* base = static_cast<char*>(hws);
  float* amaxL = reinterpret_cast<float*>(base);
  float* amaxR = amaxL + bpad;
  float* scl = amaxR + bpad;  // [sL, 1/sL, sR, 1/sR]
  int* tail = reinterpret_cast<int*>(base + 2 * bpad * 4 + 64);  // n3: the first one-product stage (k_order_rows)
  int* perm = tail + 16;                                          // packed row -> source row
  _Float16* Lh = reinterpret_cast<_Float16*>(base + align_up(2 * bpad * 4 + 128 + bpad * 4, 256));
  _Float16* Ll = Lh + bpad * npad_l;
  _Float16* Rh = Ll + bpad * npad_l;
  _Float16* Rl = Rh + bpad * npad;
  k_row_amax<<<(unsigned)batch, 256, 0, stream>>>(L, ldl, nrow, amaxL);
  k_row_amax<<<(unsigned)batch, 256, 0, stream>>>(R, ldr, n, amaxR);
  k_global_scale<<<1, 256, 0, stream>>>(amaxL, batch, scl);
  k_global_scale<<<1, 256, 0, stream>>>(amaxR, batch, scl + 2);
  if (bpad <= kSortMax) {
    int npow2 = 64;
    while (npow2 < bpad) npow2 <<= 1;
    if (npow2 * 8 > 48 * 1024)
      MFX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_order_rows), hipFuncAttributeMaxDynamicSharedMemorySize, kSortMax * 8));
    k_order_rows<<<1, 1024, (size_t)npow2 * 8, stream>>>(amaxL, amaxR, batch, inner, bpad, npow2, grad_one_product_log2(), perm, tail);
  } else {
    k_order_default<<<(unsigned)((bpad + 255) / 256), 256, 0, stream>>>(batch, inner, bpad, perm, tail);
  }
  const dim3 pgrid((unsigned)((npad + 255) / 256), (unsigned)(bpad / 8));
  const dim3 pgrid_l((unsigned)((npad_l + 255) / 256), (unsigned)(bpad / 8));
  const uint32_t salt_l = kSaltL, salt_r = kSaltR;
  k_pack_f16<<<pgrid_l, 256, 0, stream>>>(L, ldl, batch, nrow, npad_l, scl, salt_l, perm, Lh, Ll);
  k_pack_f16<<<pgrid, 256, 0, stream>>>(R, ldr, batch, n, npad, scl + 2, salt_r, perm, Rh, Rl);
  MFX_CHECK_LAUNCH();
  // 256 x 256 workgroup tile (NBW = 4) for d <= 8; d = 9..16 keeps the 256 x 128 tile (the epilogue registers on top of 128
  // accumulators spill there)
  if constexpr (DPAD <= 8)
    return launch_grad_h_t<DPAD, 4>(op, xs, sq, n, npad_l, npad, Lh, Ll, Rh, Rl, bpad, salt_l, salt_r, partial, nblocks_out, tail, stream);
  return launch_grad_h_t<DPAD, 2>(op, xs, sq, n, npad_l, npad, Lh, Ll, Rh, Rl, bpad, salt_l, salt_r, partial, nblocks_out, tail, stream);
}

// returns the device pointer holding [sL, 1/sL, sR, 1/sR] through scales_out
int rbf_mfma_grad_h(const mfx_operator* op, const float* xs, const float* sq, int dpad, const float* L, int64_t ldl,
                    const float* R, int64_t ldr, int64_t batch, int64_t inner, double* partial, int64_t* nblocks_out, void* hws,
                    const float** scales_out, hipStream_t stream) {
  const int64_t bpad = (batch + 31) / 32 * 32;
  *scales_out = reinterpret_cast<const float*>(hws) + 2 * bpad;
  switch (dpad) {
    case 4: return launch_grad_h<4>(op, xs, sq, L, ldl, R, ldr, batch, inner, partial, nblocks_out, hws, stream);
    case 8: return launch_grad_h<8>(op, xs, sq, L, ldl, R, ldr, batch, inner, partial, nblocks_out, hws, stream);
    case 12: return launch_grad_h<12>(op, xs, sq, L, ldl, R, ldr, batch, inner, partial, nblocks_out, hws, stream);
    case 16: return launch_grad_h<16>(op, xs, sq, L, ldl, R, ldr, batch, inner, partial, nblocks_out, hws, stream);
    case 32: return launch_grad_h<32>(op, xs, sq, L, ldl, R, ldr, batch, inner, partial, nblocks_out, hws, stream);
    default: set_error("split parameter sweep supports d <= 32"); return MFX_ERR_UNSUPPORTED;
  }
}

// 0: exact fp32 MFMA, 1: 3 x f16 matvec (fat-wave / pipelined kernels), 2: also the gradient GEMM split
int rbf_mode(const mfx_operator* op) { return op->rbf_mode; }

bool rbf_mfma_supported(const mfx_operator* op, int64_t p) {
  // from 4 probes on, padding the probe dimension to 32 already beats the VALU kernel (C2-like: 12 ms -> ~1 ms)
  return op->dtype == MFX_F32 && p >= 4 && op->d <= 16;
}

// 16 < d <= 64 (round 5): the split kernels keep their distance operands resident and stop at d = 16, the exact-fp32 kernels of this
// file are generic in the padded dimension (the distance product is KD / 2 = 17 or 33 fp32 MFMAs per block) -- every arithmetic mode
// runs them there.  About half of the datasets the reference's UCI loaders fetch have 17 .. 27 input columns (util/uci_util.py:68-316);
// the VALU kernel they fell to is 18 x slower per matvec than d = 16 on the matrix cores (profiles/r05k_*).
// 16 < d <= 32 in the split modes: the h3 kernel's in-kernel-split form -- fp32-MFMA distances (17 per block), the CONTRACTION on the f16 pipe
bool rbf_mfma_h3_wide_supported(const mfx_operator* op, int64_t p) {
  return op->dtype == MFX_F32 && op->d > 16 && op->d <= 32 && (p >= 4 || op->n >= 2048);
}
bool rbf_mfma_exact_wide_supported(const mfx_operator* op, int64_t p) {
  return op->dtype == MFX_F32 && op->d > 16 && op->d <= 128 && (p >= 4 || op->n >= 2048);
}

template <int DPAD, int NB, int MI, int TJ>
static int launch_apply_mi(const mfx_operator* op, const float* xs, const float* sq, const float* x, int64_t ldx,
                           float* y, int64_t ldy, int64_t p, hipStream_t stream) {
  const int64_t n = op->n;
  const int64_t row0 = op_row0(op), nrow = op_nrows(op), rend = row0 + nrow;
  MFX_REQUIRE(row0 % 64 == 0, MFX_ERR_INVALID, "matrix-core Gram matvec: row0 = %lld must be a multiple of 64", (long long)row0);
  const dim3 grid((unsigned)((nrow + 4 * MI * 32 - 1) / (4 * MI * 32)), (unsigned)((p + NB * 32 - 1) / (NB * 32)));
  const bool vec4 = (n % 4 == 0) && (nrow % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                    (reinterpret_cast<uintptr_t>(y) % 16 == 0);
  if (vec4) {
    k_rbf_mfma_apply<DPAD, NB, true, MI, TJ><<<grid, 256, 0, stream>>>(xs, sq, n, (const float*)op->outputscale,
                                                                   (const float*)op->noise, x, ldx, y, ldy, p, op->kernel_fn, row0, rend);
  } else {
    k_rbf_mfma_apply<DPAD, NB, false, MI, TJ><<<grid, 256, 0, stream>>>(xs, sq, n, (const float*)op->outputscale,
                                                                    (const float*)op->noise, x, ldx, y, ldy, p, op->kernel_fn, row0, rend);
  }
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

template <int DPAD, int NB>
static int launch_apply(const mfx_operator* op, const float* xs, const float* sq, const float* x, int64_t ldx,
                        float* y, int64_t ldy, int64_t p, hipStream_t stream) {
  // 64 rows per wave (2 workgroups per CU at n = 131072) unless the problem is too small to fill the chip
  // (DPAD = 64: the resident row operand of the distance product is 33 registers per 32 rows -- one row block per wave)
  const bool small = (op_nrows(op) + 255) / 256 < 512;
  if (small || DPAD > 32) return launch_apply_mi<DPAD, NB, 1, 64>(op, xs, sq, x, ldx, y, ldy, p, stream);
  if constexpr (DPAD <= 32) return launch_apply_mi<DPAD, NB, 2, 64>(op, xs, sq, x, ldx, y, ldy, p, stream);
  return MFX_ERR_UNSUPPORTED;
}

template <int DPAD>
static int launch_apply_d(const mfx_operator* op, const float* xs, const float* sq, const float* x, int64_t ldx,
                          float* y, int64_t ldy, int64_t p, hipStream_t stream) {
  if (p <= 32) return launch_apply<DPAD, 1>(op, xs, sq, x, ldx, y, ldy, p, stream);
  return launch_apply<DPAD, 2>(op, xs, sq, x, ldx, y, ldy, p, stream);  // chunks of 64 probes in grid.y
}

int rbf_mfma_apply(const mfx_operator* op, const float* xs, const float* sq, int dpad, const float* x, int64_t ldx,
                   float* y, int64_t ldy, int64_t p, hipStream_t stream) {
  switch (dpad) {
    case 4: return launch_apply_d<4>(op, xs, sq, x, ldx, y, ldy, p, stream);
    case 8: return launch_apply_d<8>(op, xs, sq, x, ldx, y, ldy, p, stream);
    case 12: return launch_apply_d<12>(op, xs, sq, x, ldx, y, ldy, p, stream);
    case 16: return launch_apply_d<16>(op, xs, sq, x, ldx, y, ldy, p, stream);
    case 32: return launch_apply_d<32>(op, xs, sq, x, ldx, y, ldy, p, stream);
    case 64: return launch_apply_d<64>(op, xs, sq, x, ldx, y, ldy, p, stream);
    case 96: return launch_apply_d<96>(op, xs, sq, x, ldx, y, ldy, p, stream);
    case 128: return launch_apply_d<128>(op, xs, sq, x, ldx, y, ldy, p, stream);
    default: set_error("exact-fp32 matrix-core Gram matvec supports d <= 128"); return MFX_ERR_UNSUPPORTED;
  }
}

// ================================================================================================
// Parameter-gradient sweep on the matrix cores.
//
//   d/dtheta sum_bt L_bt^T K(theta) R_bt = sum_ij S_ij dK_ij/dtheta,   S = L^T R  (n x n, rank <= batch)
//
// GEMM view: M = n (i), N = n (j), K = batch (all (probe, Lanczos-step) pairs of the adjoint at once:
// 2560 for BASELINE config 4).  Both operands are K-major in HBM ((batch, n) rows, n contiguous), which
// is exactly the A[i][k] / B[k][j] lane layout of v_mfma_f32_32x32x2_f32 -- LDS tiles are straight row
// copies (ds_write_b128), fragments are conflict-free ds_read_b32.  S never leaves the accumulators:
// the epilogue evaluates K_ij (distance from LDS-broadcast x_i and register x_j, v_exp_f32), multiplies
// and reduces to (d + 2) numbers per workgroup.  128 x 128 tile per workgroup (4 waves, 64 x 64 each),
// BK = 32 batch rows per stage, register-prefetched double buffering.
// grid = (8 column ranges, n/128 row tiles): blockIdx.x is the XCD label (round-robin dispatch), so the
// 64 resident workgroups of one XCD walk the SAME R panel (1.3 MB, L2-resident) while their L panels
// come from the Infinity Cache.
// ================================================================================================


template <int DPAD>
struct GradSmem {
  union {
    struct {
      float a[2][kGK][kGM];
      float b[2][kGK][kGN];
    } st;
    float s_t[kGN][kGLd];  // S^T tile (column j major) for the epilogue
  } u;
  float xi[kGM][DPAD];
  float sqi[kGM];
  float xj[kGN][DPAD];
  float sqj[kGN];
  double red[4][DPAD + 2];
};

template <int DPAD, bool VEC4>
__device__ __forceinline__ void grad_load_stage(float4 (&ra)[4], float4 (&rb)[4], const float* __restrict__ L,
                                                int64_t ldl, const float* __restrict__ R, int64_t ldr,
                                                int64_t batch, int64_t nrow, int64_t n, int64_t bt0, int64_t i0, int64_t j0,
                                                int tid) {
  // 32 rows x 128 floats = 1024 float4 per operand; thread t takes float4 (row = f / 32, col4 = f % 32)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int f = tid + 256 * u;
    const int row = f >> 5, c4 = (f & 31) * 4;
    const int64_t bt = bt0 + row;
    float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
    if (bt < batch) {
      const float* pa = L + bt * ldl + i0 + c4;
      const float* pb = R + bt * ldr + j0 + c4;
      if (VEC4 && i0 + c4 + 3 < nrow) {
        va = *reinterpret_cast<const float4*>(pa);
      } else {
        if (i0 + c4 + 0 < nrow) va.x = pa[0];
        if (i0 + c4 + 1 < nrow) va.y = pa[1];
        if (i0 + c4 + 2 < nrow) va.z = pa[2];
        if (i0 + c4 + 3 < nrow) va.w = pa[3];
      }
      if (VEC4 && j0 + c4 + 3 < n) {
        vb = *reinterpret_cast<const float4*>(pb);
      } else {
        if (j0 + c4 + 0 < n) vb.x = pb[0];
        if (j0 + c4 + 1 < n) vb.y = pb[1];
        if (j0 + c4 + 2 < n) vb.z = pb[2];
        if (j0 + c4 + 3 < n) vb.w = pb[3];
      }
    }
    ra[u] = va;
    rb[u] = vb;
  }
}

template <int DPAD, bool VEC4>
__global__ __launch_bounds__(256, DPAD > 16 ? 1 : 2) /* DPAD = 32: 97 KB of LDS, one workgroup per CU anyway */ void k_rbf_mfma_grad(const float* __restrict__ xs, const float* __restrict__ sq,
                                                          int64_t n, int ard, int kind, const float* __restrict__ L,
                                                          int64_t ldl, const float* __restrict__ R, int64_t ldr,
                                                          int64_t batch, int tiles_per_block,
                                                          double* __restrict__ partial, int64_t row0, int64_t nrow) {
  // rows: the nrow points row0 .. of X that L covers (a row shard, or all n), L indexed locally; columns: all n points
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  GradSmem<DPAD>& sm = *reinterpret_cast<GradSmem<DPAD>*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;  // wave -> 64 x 64 quadrant of the 128 x 128 tile
  const int64_t i0 = (int64_t)(blockIdx.y / kGSub) * kGM;  // 8 x 8 L2-sharing patches per XCD, see kGSub
  const int64_t ntj = (n + kGN - 1) / kGN;
  const int64_t tj_begin = ((int64_t)blockIdx.x * kGSub + blockIdx.y % kGSub) * tiles_per_block;
  int64_t tj_end = tj_begin + tiles_per_block;
  if (tj_end > ntj) tj_end = ntj;

  // x_i, |x_i|^2 of this row tile (fixed for the workgroup)
  for (int t = tid; t < kGM * DPAD; t += 256) {
    const int64_t g = i0 * DPAD + t;
    (&sm.xi[0][0])[t] = g < nrow * DPAD ? xs[row0 * DPAD + g] : 0.f;
  }
  if (tid < kGM) sm.sqi[tid] = (i0 + tid < nrow) ? sq[row0 + i0 + tid] : 0.f;

  double gsum[DPAD + 2];
#pragma unroll
  for (int c = 0; c < DPAD + 2; ++c) gsum[c] = 0.0;

  const int64_t nstage = (batch + kGK - 1) / kGK;
  for (int64_t tj = tj_begin; tj < tj_end; ++tj) {
    const int64_t j0 = tj * kGN;
    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[4], rb[4];
    grad_load_stage<DPAD, VEC4>(ra, rb, L, ldl, R, ldr, batch, nrow, n, 0, i0, j0, tid);
    __syncthreads();  // previous tile's epilogue / LDS reads are done
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = tid + 256 * u;
      *reinterpret_cast<float4*>(&sm.u.st.a[0][f >> 5][(f & 31) * 4]) = ra[u];
      *reinterpret_cast<float4*>(&sm.u.st.b[0][f >> 5][(f & 31) * 4]) = rb[u];
    }
    __syncthreads();
    for (int64_t st = 0; st < nstage; ++st) {
      const int cur = (int)(st & 1);
      if (st + 1 < nstage) grad_load_stage<DPAD, VEC4>(ra, rb, L, ldl, R, ldr, batch, nrow, n, (st + 1) * kGK, i0, j0, tid);
#pragma unroll
      for (int kk = 0; kk < kGK; kk += 2) {
        float av[2], bv[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) av[a] = sm.u.st.a[cur][kk + lhi][wm * 64 + a * 32 + l31];
#pragma unroll
        for (int b = 0; b < 2; ++b) bv[b] = sm.u.st.b[cur][kk + lhi][wn * 64 + b * 32 + l31];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
      }
      if (st + 1 < nstage) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int f = tid + 256 * u;
          *reinterpret_cast<float4*>(&sm.u.st.a[cur ^ 1][f >> 5][(f & 31) * 4]) = ra[u];
          *reinterpret_cast<float4*>(&sm.u.st.b[cur ^ 1][f >> 5][(f & 31) * 4]) = rb[u];
        }
      }
      __syncthreads();
    }
    // ---- epilogue: S^T tile -> LDS (overlays the staging buffers; the K-loop's last barrier has
    //      passed), then thread = row i walks 64 columns j: W_ij = S_ij exp(-dist_ij / 2) ------------
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 q;
          q.x = acc[a][b][4 * g + 0]; q.y = acc[a][b][4 * g + 1]; q.z = acc[a][b][4 * g + 2]; q.w = acc[a][b][4 * g + 3];
          *reinterpret_cast<float4*>(&sm.u.s_t[wn * 64 + b * 32 + l31][wm * 64 + a * 32 + 8 * g + 4 * lhi]) = q;
        }
    for (int t = tid; t < kGN * DPAD; t += 256) {
      const int64_t g = j0 * DPAD + t;
      (&sm.xj[0][0])[t] = g < n * DPAD ? xs[g] : 0.f;
    }
    if (tid < kGN) sm.sqj[tid] = (j0 + tid < n) ? sq[j0 + tid] : 0.f;
    __syncthreads();
    {
      const int il = tid & (kGM - 1), jh = (tid >> 7) * 64;
      const int64_t i = row0 + i0 + il;  // the point; L is indexed by the local row i0 + il
      float xiv[DPAD];
#pragma unroll
      for (int c = 0; c < DPAD; c += 4) {
        const float4 q = *reinterpret_cast<const float4*>(&sm.xi[il][c]);
        xiv[c] = q.x; xiv[c + 1] = q.y; xiv[c + 2] = q.z; xiv[c + 3] = q.w;
      }
      const float si = sm.sqi[il];
      float gt[DPAD + 2];
#pragma unroll
      for (int c = 0; c < DPAD + 2; ++c) gt[c] = 0.f;
#pragma unroll 2
      for (int jj = 0; jj < 64; ++jj) {
        const int jl = jh + jj;
        const int64_t j = j0 + jl;
        float xjv[DPAD];
#pragma unroll
        for (int c = 0; c < DPAD; c += 4) {
          const float4 q = *reinterpret_cast<const float4*>(&sm.xj[jl][c]);
          xjv[c] = q.x; xjv[c + 1] = q.y; xjv[c + 2] = q.z; xjv[c + 3] = q.w;
        }
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < DPAD; ++c) dot = fmaf(xiv[c], xjv[c], dot);
        float dist = fmaf(-2.f, dot, si + sm.sqj[jl]);
        dist = fmaxf(dist, 0.f);
        const bool live = (i0 + il < nrow) && (j < n);
        const float s_ij = live ? sm.u.s_t[jl][il] : 0.f;
        float kv, wl;
        grad_weights(kind, i == j ? 0.f : dist, kv, wl);
        gt[DPAD] = fmaf(s_ij, kv, gt[DPAD]);
        const float w = s_ij * wl;
        if (ard) {
#pragma unroll
          for (int c = 0; c < DPAD; c += 2) {  // packed fp32 (v_pk_add / v_pk_mul / v_pk_fma): two dimensions per instruction
            const floatx2 xi2 = {xiv[c], xiv[c + 1]}, xj2 = {xjv[c], xjv[c + 1]};
            const floatx2 df = xi2 - xj2;
            floatx2 g2 = {gt[c], gt[c + 1]};
            g2 = __builtin_elementwise_fma(df * w, df, g2);
            gt[c] = g2[0];
            gt[c + 1] = g2[1];
          }
        } else {
          gt[0] = fmaf(w, dist, gt[0]);
        }
        if (i == j) gt[DPAD + 1] += s_ij;
      }
#pragma unroll
      for (int c = 0; c < DPAD + 2; ++c) gsum[c] += (double)gt[c];
    }
  }
  // ---- workgroup reduction -> one partial row per workgroup ------------------------------------
#pragma unroll
  for (int c = 0; c < DPAD + 2; ++c) {
    const double v = wave_sum(gsum[c]);
    if (lane == 0) sm.red[wid][c] = v;
  }
  __syncthreads();
  if (tid < DPAD + 2) {
    const int64_t blk = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    partial[blk * (DPAD + 2) + tid] = sm.red[0][tid] + sm.red[1][tid] + sm.red[2][tid] + sm.red[3][tid];
  }
}

bool rbf_mfma_grad_supported(const mfx_operator* op, int64_t batch) {
  // from n = 2048 on even ONE (lambda, x) pair (the PCG backward of the log-marginal likelihood) is cheaper here: the cost is
  // the n^2 epilogue, which the VALU sweep pays at a quarter of the rate (16.8 vs ~4 ms at n = 36 584)
  return op->dtype == MFX_F32 && (batch >= 16 || op->n >= 2048) && op->d <= 16 && op->n >= 256;
}

bool rbf_mfma_grad_exact_wide_supported(const mfx_operator* op, int64_t batch) {  // 16 < d <= 64: see rbf_mfma_exact_wide_supported
  // (beyond: x_i and x_j of a 128 x 128 tile, 2 x 128 x DPAD floats, no longer fit the LDS next to the operand stages)
  return op->dtype == MFX_F32 && (batch >= 16 || op->n >= 2048) && op->d > 16 && op->d <= 64 && op->n >= 256;
}

int64_t rbf_mfma_grad_partial_rows(int64_t n) { return ((n + kGM - 1) / kGM) * kGSplit * kGSub; }

template <int DPAD>
static int launch_grad(const mfx_operator* op, const float* xs, const float* sq, const float* L, int64_t ldl,
                       const float* R, int64_t ldr, int64_t batch, double* partial, int64_t* nblocks_out,
                       hipStream_t stream) {
  const int64_t n = op->n, row0 = op_row0(op), nrow = op_nrows(op);
  const int64_t nti = (nrow + kGM - 1) / kGM, ntj = (n + kGN - 1) / kGN;
  const int tiles_per_block = (int)((ntj + kGSplit * kGSub - 1) / (kGSplit * kGSub));
  const dim3 grid(kGSplit, (unsigned)(nti * kGSub));
  const size_t sh = sizeof(GradSmem<DPAD>);
  const bool vec4 = (n % 4 == 0) && (nrow % 4 == 0) && (ldl % 4 == 0) && (ldr % 4 == 0) && (reinterpret_cast<uintptr_t>(L) % 16 == 0) &&
                    (reinterpret_cast<uintptr_t>(R) % 16 == 0);
  if (vec4) {
    MFX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rbf_mfma_grad<DPAD, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    k_rbf_mfma_grad<DPAD, true><<<grid, 256, sh, stream>>>(xs, sq, n, op->ard, op->kernel_fn, L, ldl, R, ldr, batch, tiles_per_block, partial, row0, nrow);
  } else {
    MFX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rbf_mfma_grad<DPAD, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    k_rbf_mfma_grad<DPAD, false><<<grid, 256, sh, stream>>>(xs, sq, n, op->ard, op->kernel_fn, L, ldl, R, ldr, batch, tiles_per_block, partial, row0, nrow);
  }
  MFX_CHECK_LAUNCH();
  *nblocks_out = nti * kGSub * kGSplit;
  return MFX_OK;
}

int rbf_mfma_grad(const mfx_operator* op, const float* xs, const float* sq, int dpad, const float* L, int64_t ldl,
                  const float* R, int64_t ldr, int64_t batch, double* partial, int64_t* nblocks_out,
                  hipStream_t stream) {
  switch (dpad) {
    case 4: return launch_grad<4>(op, xs, sq, L, ldl, R, ldr, batch, partial, nblocks_out, stream);
    case 8: return launch_grad<8>(op, xs, sq, L, ldl, R, ldr, batch, partial, nblocks_out, stream);
    case 12: return launch_grad<12>(op, xs, sq, L, ldl, R, ldr, batch, partial, nblocks_out, stream);
    case 16: return launch_grad<16>(op, xs, sq, L, ldl, R, ldr, batch, partial, nblocks_out, stream);
    case 32: return launch_grad<32>(op, xs, sq, L, ldl, R, ldr, batch, partial, nblocks_out, stream);
    case 64: return launch_grad<64>(op, xs, sq, L, ldl, R, ldr, batch, partial, nblocks_out, stream);
    default: set_error("exact-fp32 matrix-core parameter sweep supports d <= 64"); return MFX_ERR_UNSUPPORTED;
  }
}

}  // namespace mfx
