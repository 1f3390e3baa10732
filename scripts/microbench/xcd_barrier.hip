// Microbenchmark (not part of the library): a persistent kernel confined to ONE XCD (32 CUs, one L2) with a software barrier
// that needs no L2 write-back: relaxed agent-scope atomics (executed in that XCD's L2), stores drained with s_waitcnt,
// L1 invalidated on the acquire side.  Measures (a) where workgroups land (XCC_ID / CU id), (b) the barrier's cost for
// 32 workgroups, (c) correctness of data exchanged through it, (d) the rate at which the 32 CUs stream a weight set.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_barrier scripts/microbench/xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}
__device__ __forceinline__ unsigned hw_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}

__device__ __forceinline__ void xcd_barrier(unsigned* bar, unsigned& epoch, unsigned G) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's stores have reached L2 (L1 is write-through)
  __syncthreads();
  if (threadIdx.x == 0) {
    epoch += G;
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;      // bail-out (sticky flag in bar[2]): a barrier that cannot complete must not hang the GPU
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 4000000u || __hip_atomic_load(bar + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(bar + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // drop stale L1 lines (buffer_inv); no L2 write-back on this side
}

// Flag barrier: every workgroup publishes its epoch in its own word (one store), lane i of wave 0 polls word i - no
// read-modify-write on a shared counter (measured: the counter form costs ~1 us PER WORKGROUP, the atomics serialise).
__device__ __forceinline__ void xcd_flag_barrier(unsigned* flags, unsigned& epoch, unsigned G, unsigned me) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  epoch += 1;
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) __hip_atomic_store(flags + me, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    const unsigned lane = threadIdx.x;
    bool ok = lane >= G;
    while (true) {
      if (!ok) ok = __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
      if (__all(ok)) break;
      if (++spins > 4000000u || __hip_atomic_load(flags + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(flags + 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// L2-scope flag barrier: agent-scope accesses go to the memory side (~1 us each, serialised per line); between workgroups of
// ONE XCD the shared L2 is the coherence point, so the flags are written with plain stores (L1 is write-through) and polled
// with plain loads behind an L1 invalidate.
__device__ __forceinline__ unsigned l2_load(const unsigned* p) {
  unsigned v;
  asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void l2_store(unsigned* p, unsigned v) {
  asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void xcd_flag_nofence(unsigned* flags, unsigned& epoch, unsigned G, unsigned me) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  epoch += 1;
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) __hip_atomic_store(flags + me, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    const unsigned lane = threadIdx.x;
    bool ok = lane >= G;
    while (true) {
      if (!ok) ok = __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
      if (__all(ok)) break;
      if (++spins > 4000000u) break;
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void xcd_l2_barrier(unsigned* flags, unsigned& epoch, unsigned G, unsigned me) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  epoch += 1;
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) l2_store(flags + me, epoch);
    unsigned spins = 0;
    const unsigned lane = threadIdx.x;
    bool ok = lane >= G;
    while (true) {
      if (!ok) ok = l2_load(flags + lane) >= epoch;
      if (__all(ok)) break;
      if (++spins > 4000000u || l2_load(flags + 64)) { l2_store(flags + 64, 1u); break; }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// slot[0]: claim counter, slot[1]: barrier, slot[2]: errors, info[]: per claimed WG {blockIdx, xcc, hw_id}
__global__ void __launch_bounds__(1024) persist_kernel(unsigned* slot, unsigned* info, unsigned target_xcc, unsigned G, int rounds,
                                                       float* buf, int per, const float4* wts, size_t w4_per_wg, float* sink, int mode,
                                                       unsigned* flags, int use_flags) {
  __shared__ float big[20000];      // 80 KB: one workgroup per CU
  __shared__ unsigned my;
  if (threadIdx.x == 0) {
    unsigned m = 0xFFFFFFFFu;
    if (xcc_id() == target_xcc) {
      m = atomicAdd(&slot[0], 1u);
      if (m < G) { info[3 * m] = blockIdx.x; info[3 * m + 1] = xcc_id(); info[3 * m + 2] = hw_id(); }
    }
    my = m;
  }
  big[threadIdx.x] = 0.f;
  __syncthreads();
  const unsigned me = my;
  if (me >= G) return;
  unsigned epoch = 0, bad = 0;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    if (mode >= 1) {      // exchange: write a round-tagged block, read three other workgroups' blocks after the barrier
      float* mine = buf + ((size_t)(r & 1) * G + me) * per;
      for (int i = threadIdx.x; i < per; i += 1024) mine[i] = (float)(r * 1000 + me);
    }
    if (mode >= 2) {      // stream this workgroup's slice of the weight set (float4, contiguous per wave)
      const float4* w = wts + (size_t)me * w4_per_wg;
      for (size_t i = threadIdx.x; i < w4_per_wg; i += 1024) { const float4 v = w[i]; acc += v.x + v.y + v.z + v.w; }
    }
    if (use_flags == 3) xcd_flag_nofence(flags, epoch, G, me);
    else if (use_flags == 2) xcd_l2_barrier(flags, epoch, G, me);
    else if (use_flags) xcd_flag_barrier(flags, epoch, G, me);
    else xcd_barrier(&slot[1], epoch, G);
    if (mode >= 1)
      for (int d = 1; d <= 3; ++d) {
        const unsigned o = (me + d * 7) % G;
        const float* theirs = buf + ((size_t)(r & 1) * G + o) * per;
        for (int i = threadIdx.x; i < per; i += 1024) bad += (theirs[i] != (float)(r * 1000 + o)) ? 1u : 0u;
      }
  }
  if (bad) atomicAdd(&slot[2], bad);
  if (acc == 123.456f) sink[0] = acc;
}

int main(int argc, char** argv) {
  const unsigned G = argc > 1 ? atoi(argv[1]) : 32;
  const int rounds = argc > 2 ? atoi(argv[2]) : 2000, per = argc > 3 ? atoi(argv[3]) : 4096;
  const size_t wbytes = (size_t)(argc > 4 ? atoi(argv[4]) : 23) << 20;
  const int launch_wgs = argc > 5 ? atoi(argv[5]) : 512;
  unsigned *slot, *info; float *buf, *sink; float4* wts;
  hipMalloc(&slot, 64); hipMalloc(&info, 3 * 4 * 256); hipMalloc(&buf, (size_t)2 * G * per * 4); hipMalloc(&sink, 4);
  hipMalloc(&wts, wbytes); hipMemset(wts, 0, wbytes);
  const size_t w4_per_wg = wbytes / 16 / G;
  unsigned* flags; hipMalloc(&flags, 4 * 128);
  const int use_flags = argc > 6 ? atoi(argv[6]) : 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (unsigned xcc = 0; xcc < 2; ++xcc)
    for (int mode = 0; mode <= 2; ++mode)
      for (int rep = 0; rep < 2; ++rep) {
        hipMemset(slot, 0, 64); hipMemset(info, 0xFF, 3 * 4 * 256); hipMemset(flags, 0, 4 * 128);
        hipEventRecord(e0);
        hipLaunchKernelGGL(persist_kernel, dim3(launch_wgs), dim3(1024), 0, 0, slot, info, xcc, G, rounds, buf, per, wts, w4_per_wg, sink, mode, flags, use_flags);
        hipEventRecord(e1);
        hipError_t e = hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned h[16]; hipMemcpy(h, slot, 64, hipMemcpyDeviceToHost);
        if (rep == 1)
          printf("xcc %u mode %d (%s): %s  claimed %u  %.2f us per round  errors %u  bailout %u%s\n", xcc, mode,
                 mode == 0 ? "barrier only" : mode == 1 ? "barrier + exchange" : "barrier + exchange + weight stream",
                 hipGetErrorString(e), h[0], ms * 1000.f / rounds, h[2], h[3],
                 mode == 2 ? "" : "");
        if (rep == 1 && mode == 2) printf("   weight stream: %.1f MB per round -> %.2f TB/s through %u CUs\n", wbytes / 1e6, wbytes / (ms * 1e-3 / rounds) / 1e12, G);
        if (rep == 1 && mode == 0 && xcc == 0) {
          unsigned hi[3 * 64]; hipMemcpy(hi, info, sizeof(hi), hipMemcpyDeviceToHost);
          printf("   placement (blockIdx:xcc:cu_id/se_id):");
          for (unsigned i = 0; i < G && i < 40; ++i) printf(" %u:%u:%u/%u", hi[3 * i], hi[3 * i + 1], (hi[3 * i + 2] >> 8) & 0xF, (hi[3 * i + 2] >> 13) & 0x7);
          printf("\n");
        }
      }
  return 0;
}
