// Microbenchmark (not part of the library): cost and cross-XCD correctness of a software grid barrier on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/microbench/grid_barrier scripts/microbench/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdio.h>
#include <stdlib.h>
namespace cg = cooperative_groups;

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned& epoch, unsigned G) {
  __syncthreads();
  if (threadIdx.x == 0) {
    epoch += G;
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// every round: WG b writes round-tagged values to its slot; after the barrier it checks the slots of 3 other WGs
__global__ void __launch_bounds__(256) sw_kernel(unsigned* bar, float* buf, int rounds, int per, unsigned* errors) {
  unsigned epoch = 0;
  const unsigned G = gridDim.x;
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    float* mine = buf + ((size_t)(r & 1) * G + blockIdx.x) * per;
    for (int i = threadIdx.x; i < per; i += 256) mine[i] = (float)(r * 1000 + blockIdx.x);
    grid_barrier(bar, epoch, G);
    for (int d = 1; d <= 3; ++d) {
      const unsigned o = (blockIdx.x + d * 37) % G;
      const float* theirs = buf + ((size_t)(r & 1) * G + o) * per;
      for (int i = threadIdx.x; i < per; i += 256) bad += (theirs[i] != (float)(r * 1000 + o)) ? 1u : 0u;
    }
  }
  if (bad) atomicAdd(errors, bad);
}

__global__ void __launch_bounds__(256) cg_kernel(float* buf, int rounds, int per, unsigned* errors) {
  cg::grid_group grid = cg::this_grid();
  const unsigned G = gridDim.x;
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    float* mine = buf + ((size_t)(r & 1) * G + blockIdx.x) * per;
    for (int i = threadIdx.x; i < per; i += 256) mine[i] = (float)(r * 1000 + blockIdx.x);
    grid.sync();
    for (int d = 1; d <= 3; ++d) {
      const unsigned o = (blockIdx.x + d * 37) % G;
      const float* theirs = buf + ((size_t)(r & 1) * G + o) * per;
      for (int i = threadIdx.x; i < per; i += 256) bad += (theirs[i] != (float)(r * 1000 + o)) ? 1u : 0u;
    }
  }
  if (bad) atomicAdd(errors, bad);
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256, rounds = argc > 2 ? atoi(argv[2]) : 2000, per = argc > 3 ? atoi(argv[3]) : 256;
  unsigned *bar, *err; float* buf;
  hipMalloc(&bar, 4); hipMalloc(&err, 4); hipMalloc(&buf, (size_t)2 * G * per * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(bar, 0, 4); hipMemset(err, 0, 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL(sw_kernel, dim3(G), dim3(256), 0, 0, bar, buf, rounds, per, err);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned h; hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
    printf("software barrier: G=%d per=%d: %.2f us/round, errors=%u\n", G, per, 1e3 * ms / rounds, h);
  }
  {
    hipMemset(err, 0, 4);
    int r = rounds, p = per;
    void* args[] = {&buf, &r, &p, &err};
    hipEventRecord(e0);
    hipError_t e = hipLaunchCooperativeKernel((void*)cg_kernel, dim3(G), dim3(256), args, 0, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned h; hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
    printf("cooperative grid.sync (%s): %.2f us/round, errors=%u\n", hipGetErrorString(e), 1e3 * ms / rounds, h);
  }
  return 0;
}
