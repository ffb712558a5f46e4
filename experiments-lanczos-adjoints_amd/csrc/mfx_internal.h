// Internal helpers shared by the libmfx translation units (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <functional>
#include <string>

#include "mfx.h"

namespace mfx {

void set_error(const char* fmt, ...);

#define MFX_CHECK_HIP(expr)                                                          \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      mfx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                     __LINE__);                                                      \
      return MFX_ERR_HIP;                                                            \
    }                                                                                \
  } while (0)

#define MFX_CHECK_LAUNCH() MFX_CHECK_HIP(hipGetLastError())

#define MFX_REQUIRE(cond, code, ...)  \
  do {                                \
    if (!(cond)) {                    \
      mfx::set_error(__VA_ARGS__);    \
      return (code);                  \
    }                                 \
  } while (0)

#define MFX_TRY(expr)          \
  do {                         \
    int _rc = (expr);          \
    if (_rc != MFX_OK) return _rc; \
  } while (0)

// ---- device-side timing of kernel classes (hipEvents on the caller's stream) -------------------
struct ScopedTimer {
  int cls;
  hipStream_t stream;
  void* slot;
  ScopedTimer(int cls, hipStream_t s);
  ~ScopedTimer();
};

// ---- hipGraph replay of launch-bound driver calls (mfx_core.hip) ---------------------------------
// A driver call is a fixed sequence of launches determined by its arguments (no host reads of device data), so for
// small problems -- where the host's ~5 us per launch, not the GPU, sets the pace -- the second call with the same
// arguments is captured into a hipGraph and every later one is a single hipGraphLaunch.  MFX_GRAPHS=0 disables it.
struct GraphKey {
  std::string bytes;
  template <typename X>
  GraphKey& add(const X& x) {
    bytes.append(reinterpret_cast<const char*>(&x), sizeof(X));
    return *this;
  }
};
// eligible = false (or graphs disabled, or kernel timing on): plain fn(stream).  fn must enqueue everything on the stream
// it is given and nothing else (no synchronisation, no allocation).
int run_graphed(bool eligible, const GraphKey& key, hipStream_t stream, const std::function<int(hipStream_t)>& fn);

// ---- device fills / copies as KERNELS -------------------------------------------------------------
// hipMemsetAsync / hipMemcpy2DAsync become memset / memcpy NODES when a driver call is captured into a hipGraph, and the memset
// node of this ROCm release writes garbage on the second launch of an instantiated graph (measured: H of the Arnoldi
// forward came back with 0x1'00000000 patterns below its sub-diagonal; tests/test_gpu_graphs.py).  Kernel nodes replay fine.
int zero_async(void* dst, size_t bytes, hipStream_t stream);                       // bytes % 4 == 0
int copy_rows_async(void* dst, size_t dst_pitch, const void* src, size_t src_pitch,  // `rows` rows of `width` bytes (% 4 == 0)
                    size_t width, size_t rows, hipStream_t stream);

// ---- geometry of the Krylov vector kernels ------------------------------------------------------
constexpr int kBlock = 256;  // 4 waves
constexpr int kEpt = 8;      // elements owned by one thread
constexpr int kSlice = kBlock * kEpt;  // 2048 elements of one vector per workgroup

inline int64_t num_slices(int64_t n) { return (n + kSlice - 1) / kSlice; }
inline size_t dtype_size(int dtype) { return dtype == MFX_F64 ? 8 : 4; }
inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// row block of an operator (mfx_operator.row0 / nrows; nrows == 0 = all rows)
inline int64_t op_row0(const mfx_operator* op) { return op->nrows > 0 ? op->row0 : 0; }
inline int64_t op_nrows(const mfx_operator* op) { return op->nrows > 0 ? op->nrows : op->n; }

// carve helper for caller-provided workspace
struct Carver {
  char* base;
  int64_t off = 0, cap;
  Carver(void* ws, int64_t bytes) : base(static_cast<char*>(ws)), cap(bytes) {}
  void* take(int64_t bytes) {
    void* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  }
  bool ok() const { return off <= cap; }
};

// operator layer (mfx_ops.hip)
int64_t op_workspace_bytes(const mfx_operator* op, int64_t batch_hint, int64_t p_apply);
// Within ONE driver call (a Krylov factorisation, its adjoint, a CG solve) neither the operator's data nor its workspace change:
// the operator's per-call preparation (kernel-Gram: x / lengthscale and |x / lengthscale|^2, k_rbf_prep) then runs once instead
// of once per matvec (the reference's jit hoists the same loop invariant, util/gp_util.py:160-176).  A driver holds one of these
// on its stack; op_apply outside any scope (mfx_op_apply) prepares every time.
struct PrepScope {
  PrepScope();
  ~PrepScope();
  PrepScope(const PrepScope&) = delete;
  PrepScope& operator=(const PrepScope&) = delete;
};

int op_apply(const mfx_operator* op, const void* x, int64_t ldx, void* y, int64_t ldy, int64_t p,
             int transpose, void* ws, int64_t ws_bytes, hipStream_t stream);
int op_apply_cb(const mfx_operator* op, int mode, const void* x, int64_t ldx, const void* aux,
                int64_t ldaux, void* y, int64_t ldy, int64_t p, hipStream_t stream);
// inner > 1: the batch rows come as (batch / inner) groups of `inner` rows whose magnitudes differ widely WITHIN a group
// ((probe, Krylov step) order): lets the split gradient GEMM regroup them, see k_pack_f16
int op_vjp_params(const mfx_operator* op, const void* L, int64_t ldl, const void* R, int64_t ldr,
                  int64_t batch, const mfx_op_grads* grads, void* ws, int64_t ws_bytes,
                  hipStream_t stream, int64_t inner = 1);

int64_t rbf_cross_ws_bytes(const mfx_operator* op, int64_t m);
int op_cross_apply(const mfx_operator* op, const void* xnew, int64_t m, const void* v, int64_t ldv, void* y, int64_t ldy,
                   int64_t p, void* ws, int64_t ws_bytes, hipStream_t stream);

#ifdef __HIPCC__
// ---- wave64 / workgroup reductions -------------------------------------------------------------
// DPP cross-lane moves (no LDS round trip like ds_bpermute behind __shfl_*): quad permutes, mirrors within a row of
// 16 lanes, then row_bcast15 / row_bcast31 carry the row sums across the four rows; lane 63 ends with the total, which
// v_readlane broadcasts.  A wave_sum is ~10 (fp32) / ~20 (fp64) full-rate VALU instructions instead of 6 dependent
// LDS-latency shuffles -- the dots kernels do one per row and slice.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float wave_bcast63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ double wave_bcast63(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
template <typename T>
__device__ __forceinline__ T wave_sum_all(T v) {
  v += dpp_mov<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_mov<0x140, 0xF>(v);  // row_mirror: every lane holds its row's sum
  v += dpp_mov<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v += dpp_mov<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
  return wave_bcast63(v);       // valid in every lane
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  return wave_sum_all(v);
}

// JT (1, 2, 4, 8 or 16) values per lane -> their sums over each ROW of 16 lanes, transposed: the return value of a lane is
// the row total of value number row16_index<JT>(lane).  log2(JT) "halving" steps (a lane keeps half of its values and
// hands the other half to its partner: JT - 1 adds altogether instead of 4 JT), then plain butterfly steps on the one
// value left.  Partners: row_mirror, row_half_mirror, quad xor 2, quad xor 1 -- in THIS order every partner shares the
// side bits used so far (lane bits 3, 2, 1, 0), so both hold the same subset of values.
template <int STEP, typename T>
__device__ __forceinline__ T row16_partner(T v) {
  if constexpr (STEP == 0) return dpp_mov<0x140, 0xF>(v);       // row_mirror       i <-> 15 - i
  else if constexpr (STEP == 1) return dpp_mov<0x141, 0xF>(v);  // row_half_mirror  i <-> 7 - i (within 8)
  else if constexpr (STEP == 2) return dpp_mov<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  else return dpp_mov<0xB1, 0xF>(v);                            // quad_perm [1,0,3,2]
}
template <int N, int STEP, typename T>
__device__ __forceinline__ void row16_halve(T* v, int lane) {
  if constexpr (N > 1) {
    const bool side = (lane >> (3 - STEP)) & 1;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const T keep = side ? v[i + N / 2] : v[i];
      const T send = side ? v[i] : v[i + N / 2];
      v[i] = keep + row16_partner<STEP>(send);
    }
    row16_halve<N / 2, STEP + 1>(v, lane);
  } else if constexpr (STEP < 4) {
    v[0] += row16_partner<STEP>(v[0]);
    row16_halve<1, STEP + 1>(v, lane);
  }
}
template <int JT, typename T>
__device__ __forceinline__ T row16_sums(T (&v)[JT], int lane) {
  static_assert(JT == 1 || JT == 2 || JT == 4 || JT == 8 || JT == 16, "values per lane");
  row16_halve<JT, 0>(v, lane);
  return v[0];
}
// the same over the whole wave: the four row totals are added with two LDS-path shuffles (per JT values, not per value)
template <int JT, typename T>
__device__ __forceinline__ T wave_sums(T (&v)[JT], int lane) {
  T r = row16_sums<JT>(v, lane);
  r += __shfl_xor(r, 16, 64);
  r += __shfl_xor(r, 32, 64);
  return r;  // every lane: the wave total of value number row16_index<JT>(lane)
}
template <int JT>
__device__ __forceinline__ int row16_index(int lane) {  // which of the JT values this lane's row16_sums result belongs to
  return JT == 16 ? (lane & 15) : JT == 8 ? ((lane >> 1) & 7) : JT == 4 ? ((lane >> 2) & 3) : JT == 2 ? ((lane >> 3) & 1) : 0;
}
template <int JT>
__device__ __forceinline__ bool wave_sums_writer(int lane) {  // one lane of the wave per value
  return lane < 16 && (lane & (16 / JT - 1)) == 0;
}

// 16-byte vector access: VEC elements of T (float x4, double x2) or scalar fallback (VEC = 1)
template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Pack {
  T v[VEC];
};
template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_pack(const T* p) {
  return *reinterpret_cast<const Pack<T, VEC>*>(p);
}
template <typename T, int VEC>
__device__ __forceinline__ void store_pack(T* p, const Pack<T, VEC>& v) {
  *reinterpret_cast<Pack<T, VEC>*>(p) = v;
}
template <typename T>
struct VecWidth {
  static constexpr int value = 16 / sizeof(T);
};
#endif

}  // namespace mfx
