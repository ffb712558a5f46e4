// libmfx: the stationary kernel families of the Gram operator (util/gp_util.py:69-184; kinds = MFX_KERNEL_*).
#pragma once
#include "mfx_internal.h"

namespace mfx {

__device__ __forceinline__ float exp_neg_half(float d) { return __expf(-0.5f * d); }
__device__ __forceinline__ double exp_neg_half(double d) { return exp(-0.5 * d); }
__device__ __forceinline__ float exp_neg(float r) { return __expf(-r); }
__device__ __forceinline__ double exp_neg(double r) { return exp(-r); }
template <typename T> __device__ __forceinline__ T dtype_eps();
template <> __device__ __forceinline__ float dtype_eps<float>() { return 1.1920928955078125e-7f; }
template <> __device__ __forceinline__ double dtype_eps<double>() { return 2.220446049250313e-16; }

// K_ij / outputscale and the lengthscale weight w (dK_ij/dl_c = outputscale w (x_ic - x_jc)^2 / l_c^3) from the clamped
// scaled squared distance (util/gp_util.py:69-184; kinds = MFX_KERNEL_*)
template <typename T>
__device__ __forceinline__ void kernel_eval(int kind, T dist, T& kv, T& wl) {
  if (kind == MFX_KERNEL_RBF) {
    kv = exp_neg_half(dist);
    wl = kv;
  } else if (kind == MFX_KERNEL_MATERN32) {
    const T r = sqrt(T(3) * dist + dtype_eps<T>());
    const T e = exp_neg(r);
    kv = (T(1) + r) * e;
    wl = T(3) * e;
  } else {
    const T r = sqrt(dist + dtype_eps<T>());
    const T e = exp_neg(r);
    kv = e;
    wl = dist > T(0) ? e / r : T(0);  // d max(0, s)/ds = 0 on the clamped side
  }
}

}  // namespace mfx
