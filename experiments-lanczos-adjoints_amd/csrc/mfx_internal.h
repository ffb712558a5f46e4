// Internal helpers shared by the libmfx translation units (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

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
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;  // valid in lane 0
}
template <typename T>
__device__ __forceinline__ T wave_sum_all(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;  // valid in every lane
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
