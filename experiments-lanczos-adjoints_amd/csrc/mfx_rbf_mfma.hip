// libmfx: matrix-free RBF Gram matvec on the CDNA4 matrix cores (fp32-exact MFMA, gfx950).
//
//   W[i][b] = s * sum_j exp(-max(0, |x_i|^2 + |x_j|^2 - 2 x_i.x_j) / 2) V[j][b] + noise V[i][b]
//   (x already divided by the lengthscale; util/gp_util.py:160-176,225-226,536-541)
//
// The Gram matrix is never materialised (the reference materialises (n/num x n) tiles,
// util/gp_util.py:496-509).  GEMM view: M = n rows i, N = probes, K = n columns j.
//   * v_mfma_f32_32x32x2_f32: A = K-tile (32 i x 2 j), B = V-tile (2 j x 32 probes), D = 32 i x 32 probes.
//   * A-operand layout is lane -> (i = lane & 31, j = lane >> 5): every lane EVALUATES its own kernel
//     entries (distance from its register-resident x_i and an LDS-broadcast x_j, v_exp_f32) straight
//     into the MFMA A register -- no shuffle, no LDS round trip for K.
//   * B comes from an LDS image Vt[j][probe] (probe-contiguous, conflict-free ds_read_b32), filled
//     from the (p, n) probe-major global layout with 256-B coalesced segments and a transposing store
//     (row pad 1 -> at most 2-way ds_write conflicts, which are free on gfx950).
//   * one wave owns MI x 32 rows and all NB x 32 probes of its chunk: MI*NB*16 accumulator registers.
// Bound: fp32 MFMA (2 n^2 p flop per matvec) -- the exp/distance VALU work co-issues underneath.
#include "mfx_internal.h"

namespace mfx {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kTJ = 64;   // columns j per LDS tile
constexpr int kMI = 2;    // 32-row sub-tiles per wave

template <int DPAD, int NB>
struct RbfTile {
  static constexpr int LDV = NB * 32 + 1;
  float xj[kTJ][DPAD];
  float sqj[kTJ];
  float vt[kTJ][LDV];
};

template <int DPAD, int NB, bool VEC4>
__global__ __launch_bounds__(256, 2) void k_rbf_mfma_apply(const float* __restrict__ xs, const float* __restrict__ sq,
                                                           int64_t n, const float* __restrict__ outputscale,
                                                           const float* __restrict__ noise,
                                                           const float* __restrict__ x, int64_t ldx,
                                                           float* __restrict__ y, int64_t ldy, int64_t p) {
  __shared__ __attribute__((aligned(16))) RbfTile<DPAD, NB> tile;
  constexpr int LDV = RbfTile<DPAD, NB>::LDV;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int64_t i_wave = (int64_t)blockIdx.x * (4 * kMI * 32) + (int64_t)wid * (kMI * 32);
  const int64_t b0 = (int64_t)blockIdx.y * (NB * 32);

  float xi[kMI][DPAD], sqi[kMI];
#pragma unroll
  for (int mi = 0; mi < kMI; ++mi) {
    int64_t i = i_wave + mi * 32 + l31;
    if (i >= n) i = n - 1;
#pragma unroll
    for (int c = 0; c < DPAD; ++c) xi[mi][c] = xs[i * DPAD + c];
    sqi[mi] = sq[i];
  }
  floatx16 acc[kMI][NB];
#pragma unroll
  for (int mi = 0; mi < kMI; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nb][r] = 0.f;

  for (int64_t j0 = 0; j0 < n; j0 += kTJ) {
    __syncthreads();
    // ---- stage x_j, |x_j|^2 -------------------------------------------------------------------
    for (int t = tid; t < kTJ * DPAD; t += 256) {
      const int64_t g = j0 * DPAD + t;
      (&tile.xj[0][0])[t] = g < n * DPAD ? xs[g] : 0.f;
    }
    if (tid < kTJ) tile.sqj[tid] = (j0 + tid < n) ? sq[j0 + tid] : 0.f;
    // ---- stage V^T: global (probe, j) 256-B segments -> LDS [j][probe] -------------------------
    constexpr int kF4 = NB * 32 * (kTJ / 4);  // float4 chunks in the tile
    for (int f = tid; f < kF4; f += 256) {
      const int bq = f / (kTJ / 4), j4 = (f % (kTJ / 4)) * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (b0 + bq < p) {
        const float* src = x + (b0 + bq) * ldx + j0 + j4;
        if (VEC4 && j0 + j4 + 3 < n) {
          const float4 q = *reinterpret_cast<const float4*>(src);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j0 + j4 + e < n) v[e] = src[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) tile.vt[j4 + e][bq] = v[e];
    }
    __syncthreads();
    // ---- 32 K-steps of 2 columns ----------------------------------------------------------------
#pragma unroll 4
    for (int ks = 0; ks < kTJ / 2; ++ks) {
      const int jj = 2 * ks + lhi;
      float xjv[DPAD];
#pragma unroll
      for (int c = 0; c < DPAD; c += 4) {
        const float4 q = *reinterpret_cast<const float4*>(&tile.xj[jj][c]);
        xjv[c] = q.x; xjv[c + 1] = q.y; xjv[c + 2] = q.z; xjv[c + 3] = q.w;
      }
      const float sj = tile.sqj[jj];
      float a[kMI];
#pragma unroll
      for (int mi = 0; mi < kMI; ++mi) {
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < DPAD; ++c) dot = fmaf(xi[mi][c], xjv[c], dot);
        const float dist = fmaf(-2.f, dot, sqi[mi] + sj);
        // exp(-max(0, dist) / 2) = exp2(min(0, -log2(e)/2 * dist))
        a[mi] = __builtin_amdgcn_exp2f(fminf(-0.72134752044448170368f * dist, 0.f));
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float bv = tile.vt[jj][nb * 32 + l31];
#pragma unroll
        for (int mi = 0; mi < kMI; ++mi)
          acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], bv, acc[mi][nb], 0, 0, 0);
      }
    }
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
        if (VEC4 && i + 3 < n) {
          const float4 xv = *reinterpret_cast<const float4*>(x + b * ldx + i);
          float4 o;
          o.x = fmaf(s, acc[mi][nb][4 * g + 0], nz * xv.x);
          o.y = fmaf(s, acc[mi][nb][4 * g + 1], nz * xv.y);
          o.z = fmaf(s, acc[mi][nb][4 * g + 2], nz * xv.z);
          o.w = fmaf(s, acc[mi][nb][4 * g + 3], nz * xv.w);
          *reinterpret_cast<float4*>(y + b * ldy + i) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (i + e < n) y[b * ldy + i + e] = fmaf(s, acc[mi][nb][4 * g + e], nz * x[b * ldx + i + e]);
        }
      }
    }
}

bool rbf_mfma_supported(const mfx_operator* op, int64_t p) {
  return op->dtype == MFX_F32 && p >= 16 && op->d <= 16;
}

template <int DPAD, int NB>
static int launch_apply(const mfx_operator* op, const float* xs, const float* sq, const float* x, int64_t ldx,
                        float* y, int64_t ldy, int64_t p, hipStream_t stream) {
  const int64_t n = op->n;
  const dim3 grid((unsigned)((n + 4 * kMI * 32 - 1) / (4 * kMI * 32)), (unsigned)((p + NB * 32 - 1) / (NB * 32)));
  const bool vec4 = (n % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                    (reinterpret_cast<uintptr_t>(y) % 16 == 0);
  if (vec4) {
    k_rbf_mfma_apply<DPAD, NB, true><<<grid, 256, 0, stream>>>(xs, sq, n, (const float*)op->outputscale,
                                                               (const float*)op->noise, x, ldx, y, ldy, p);
  } else {
    k_rbf_mfma_apply<DPAD, NB, false><<<grid, 256, 0, stream>>>(xs, sq, n, (const float*)op->outputscale,
                                                                (const float*)op->noise, x, ldx, y, ldy, p);
  }
  MFX_CHECK_LAUNCH();
  return MFX_OK;
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
    default: set_error("RBF MFMA path supports d <= 16"); return MFX_ERR_UNSUPPORTED;
  }
}

bool rbf_mfma_grad_supported(const mfx_operator* /*op*/, int64_t /*batch*/) { return false; }

int rbf_mfma_grad(const mfx_operator*, const float*, const float*, int, const float*, int64_t, const float*, int64_t,
                  int64_t, double*, int64_t*, hipStream_t) {
  set_error("RBF MFMA gradient sweep not built");
  return MFX_ERR_UNSUPPORTED;
}

}  // namespace mfx
