// Cost of a grid-wide barrier on gfx950: cooperative-groups grid.sync() vs a hand-written sense-reversing atomic barrier.
// Build: hipcc --offload-arch=gfx950 -O2 tools/grid_barrier_bench.hip -o tools/grid_barrier_bench.bin
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
namespace cg = cooperative_groups;

__global__ void k_cg(int iters, float* out) {
  cg::grid_group g = cg::this_grid();
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    acc += 1.f;
    g.sync();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}

// one atomic arrive per workgroup, spin on a generation counter (agent scope)
__device__ __forceinline__ void barrier_atomic(unsigned* count, unsigned* gen, unsigned nwg) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    const unsigned arrived = __hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived == nwg - 1) {
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();
  }
  __syncthreads();
}

__global__ void k_atomic(int iters, unsigned* state, float* out) {
  float acc = 0.f;
  const unsigned nwg = gridDim.x;
  for (int i = 0; i < iters; ++i) {
    acc += 1.f;
    barrier_atomic(state, state + 32, nwg);
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}

int main() {
  float* out; unsigned* state;
  hipMalloc(&out, 4); hipMalloc(&state, 256); hipMemset(state, 0, 256);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nwg : {1, 50, 256, 512, 1024}) {
    for (int mode = 0; mode < 2; ++mode) {
      int it = iters; void* args_cg[] = {&it, &out}; void* args_at[] = {&it, &state, &out};
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipError_t e = mode == 0 ? hipLaunchCooperativeKernel((const void*)k_cg, dim3(nwg), dim3(256), args_cg, 0, 0)
                                 : hipLaunchCooperativeKernel((const void*)k_atomic, dim3(nwg), dim3(256), args_at, 0, 0);
        if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); break; }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("%-22s %5d workgroups: %.3f us per barrier\n", mode == 0 ? "cg::grid.sync()" : "atomic sense barrier", nwg, best * 1e3f / iters);
    }
  }
  return 0;
}
