// Microbenchmark: variants of a software grid barrier on gfx950 (cost per round, cross-XCD visibility check).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// V1: one counter, relaxed polling, one acquire fence at the end
__device__ __forceinline__ void bar_v1(unsigned* bar, unsigned& epoch, unsigned G) {
  __syncthreads();
  if (threadIdx.x == 0) {
    epoch += G;
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // agent scope by default in HIP
  }
  __syncthreads();
}
// V2: hierarchical: 8 group counters (group = block % 8, i.e. the XCD), the last arriver of a group bumps the top
// counter, everybody polls the top counter (relaxed) until it reaches 8 * round.
__device__ __forceinline__ void bar_v2(unsigned* ctr, unsigned& round, unsigned G) {
  __syncthreads();
  if (threadIdx.x == 0) {
    ++round;
    const unsigned grp = blockIdx.x & 7u, per = (G + 7u - grp) / 8u;
    const unsigned prev = __hip_atomic_fetch_add(ctr + 32 * (1 + grp), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1 == per * round) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u * round) __builtin_amdgcn_s_sleep(1);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
}

template <int V>
__global__ void __launch_bounds__(256) kern(unsigned* ctr, float* buf, int rounds, int per, unsigned* errors) {
  unsigned epoch = 0;
  const unsigned G = gridDim.x;
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    float* mine = buf + ((size_t)(r & 1) * G + blockIdx.x) * per;
    for (int i = threadIdx.x; i < per; i += 256) mine[i] = (float)(r * 1000 + blockIdx.x);
    if (V == 1) bar_v1(ctr, epoch, G); else bar_v2(ctr, epoch, G);
    for (int d = 1; d <= 3; ++d) {
      const unsigned o = (blockIdx.x + d * 37) % G;
      const float* theirs = buf + ((size_t)(r & 1) * G + o) * per;
      for (int i = threadIdx.x; i < per; i += 256) bad += (theirs[i] != (float)(r * 1000 + o)) ? 1u : 0u;
    }
  }
  if (bad) atomicAdd(errors, bad);
}

int main(int argc, char** argv) {
  const int rounds = 2000;
  unsigned *ctr, *err; float* buf;
  (void)hipMalloc(&ctr, 4 * 32 * 9); (void)hipMalloc(&err, 4); (void)hipMalloc(&buf, (size_t)2 * 512 * 1024 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int Gs[] = {32, 64, 128, 256};
  for (int v = 1; v <= 2; ++v)
    for (int gi = 0; gi < 4; ++gi)
      for (int per = 64; per <= 1024; per *= 16) {
        const int G = Gs[gi];
        (void)hipMemset(ctr, 0, 4 * 32 * 9); (void)hipMemset(err, 0, 4);
        (void)hipEventRecord(e0);
        if (v == 1) hipLaunchKernelGGL(kern<1>, dim3(G), dim3(256), 0, 0, ctr, buf, rounds, per, err);
        else hipLaunchKernelGGL(kern<2>, dim3(G), dim3(256), 0, 0, ctr, buf, rounds, per, err);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned h; (void)hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
        printf("V%d G=%3d per=%4d: %6.2f us/round errors=%u\n", v, G, per, 1e3 * ms / rounds, h);
      }
  return 0;
}
