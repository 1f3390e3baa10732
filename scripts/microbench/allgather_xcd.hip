// Microbenchmark (not part of the library): the exchange step of a weight-stationary persistent kernel that spans ALL 8 XCDs.
//
// Every workgroup (one per CU: > 80 KB of LDS each) owns a column slice of a small dense layer whose weights stay in its LDS.
// Per recurrent step each of P producers publishes a (rows x 4) fp32 block of results and every consumer needs ALL blocks
// (the next product's K operand) - an all-gather through L2 / the fabric.  Form measured here (MI355X guide, Guideline 16 R1):
//   producer: 16-byte WRITE-THROUGH stores (sc1) of its block -> s_waitcnt vmcnt(0) -> one relaxed agent-scope flag store
//   consumer: one wave polls all P flags (16-byte sc1 loads, 4 flags per lane) -> __syncthreads -> 16-byte sc1 loads of the
//             payload (L1 bypassed: no buffer_inv anywhere, which costs ~1 us PER WORKGROUP on this part)
// Checked: every payload word of every epoch (values change per epoch, buffers are reused every second epoch, consumers are
// L1-warm, per-workgroup random delays make the load uneven), and the residency / XCD placement census.
// Also probes the lane layout and issue rate of v_mfma_f32_4x4x1_16b_f32 (the 4-column-granular fp32 MFMA the skinny
// products of such a kernel use).
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/microbench/allgather_xcd scripts/microbench/allgather_xcd.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));        \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

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
__device__ __forceinline__ float payload(unsigned e, unsigned p, unsigned row, unsigned c) {
  return (float)((e * 2654435761u + p * 40503u + row * 97u + c) & 0xFFFFFu);
}

struct AgArgs {
  unsigned* flags;      // [256] epoch of each producer's newest block (+ [256] sticky error, [257] give-up count)
  float* xbuf;          // [2][P][64][4]
  unsigned* errs;       // [G] payload words that did not match
  unsigned long long* ticks;      // [4]: sum over epochs of workgroup 0's publish / poll / payload time, total
  unsigned* census;     // [G][2] xcc, hw_id
  int P, G, rows, iters, mode, consumers, jitter;
};

constexpr unsigned SPIN_LIMIT = 1u << 22;

template <int THREADS>
__global__ void __launch_bounds__(THREADS) ag_kernel(const AgArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = THREADS / 64;
  const unsigned me = blockIdx.x;
  if (tid == 0) {
    a.census[2 * me] = xcc_id();
    a.census[2 * me + 1] = hw_id();
    reinterpret_cast<volatile float*>(smem)[0] = 1.f;
  }
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, 2 * a.P * 64 * 16, 0x00020000);
  __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(a.flags, 0, 260 * 4, 0x00020000);
  unsigned bad = 0;
  unsigned long long t_pub = 0, t_poll = 0, t_pay = 0;
  const unsigned long long t_begin = wall_clock64();
  bool dead = false;
  for (unsigned e = 1; e <= (unsigned)a.iters && !dead; ++e) {
    const unsigned slot = e & 1u;
    unsigned long long t0 = wall_clock64();
    if (a.jitter) {      // uneven load: ~1/4 of the workgroups dawdle 0.5-4 us before publishing
      const unsigned h = (me * 2246822519u + e * 3266489917u) >> 24;
      if ((h & 3u) == 0 && tid == 0)
        for (unsigned i = 0; i < 8 + (h >> 2); ++i) __builtin_amdgcn_s_sleep(64);
      __syncthreads();
    }
    // ---- publish
    if ((int)me < a.P && wave == 0) {
      if (lane < a.rows) {
        u32x4 v;
        v.x = __float_as_uint(payload(e, me, lane, 0)); v.y = __float_as_uint(payload(e, me, lane, 1));
        v.z = __float_as_uint(payload(e, me, lane, 2)); v.w = __float_as_uint(payload(e, me, lane, 3));
        __builtin_amdgcn_raw_buffer_store_b128(v, xr, ((slot * a.P + me) * 64 + lane) * 16, 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(a.flags + me, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned long long t1 = wall_clock64();
    // ---- wait for every producer
    if ((int)me < a.consumers) {
      if (wave == 0) {
        unsigned spins = 0;
        while (true) {
          const u32x4 f = __builtin_amdgcn_raw_buffer_load_b128(fr, lane * 16, 0, 16);
          bool ok = true;
          ok = ok && (4 * lane + 0 >= a.P || f.x >= e);
          ok = ok && (4 * lane + 1 >= a.P || f.y >= e);
          ok = ok && (4 * lane + 2 >= a.P || f.z >= e);
          ok = ok && (4 * lane + 3 >= a.P || f.w >= e);
          if (__all(ok)) break;
          if (++spins > SPIN_LIMIT || __hip_atomic_load(a.flags + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(a.flags + 256, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dead = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) reinterpret_cast<volatile unsigned*>(smem)[1] = dead ? 1u : 0u;
      }
      __syncthreads();
      dead = reinterpret_cast<volatile unsigned*>(smem)[1] != 0;
      unsigned long long t2 = wall_clock64();
      // ---- payload: every wave reads its share of the producers' blocks, 4 blocks in flight per lane
      if (a.mode >= 1 && !dead) {
        const int row = lane < a.rows ? lane : 0;
        for (int p0 = wave * 4; p0 < a.P; p0 += NW * 4) {
          u32x4 v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int p = p0 + j < a.P ? p0 + j : a.P - 1;
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, ((slot * a.P + p) * 64 + row) * 16, 0, 16);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int p = p0 + j < a.P ? p0 + j : a.P - 1;
            if (lane < a.rows) {
              bad += __uint_as_float(v[j].x) != payload(e, p, lane, 0);
              bad += __uint_as_float(v[j].y) != payload(e, p, lane, 1);
              bad += __uint_as_float(v[j].z) != payload(e, p, lane, 2);
              bad += __uint_as_float(v[j].w) != payload(e, p, lane, 3);
            }
          }
        }
        __syncthreads();
      }
      unsigned long long t3 = wall_clock64();
      t_pub += t1 - t0; t_poll += t2 - t1; t_pay += t3 - t2;
    }
  }
  if (bad) atomicAdd(a.errs + me, bad);
  if (me == 0 && tid == 0) {
    a.ticks[0] = t_pub; a.ticks[1] = t_poll; a.ticks[2] = t_pay; a.ticks[3] = wall_clock64() - t_begin;
  }
}

// ---- v_mfma_f32_4x4x1_16b_f32: layout + rate
__global__ void mfma_layout_kernel(float* out, int cbsz_mode) {
  const int lane = threadIdx.x;
  const float a = (float)(lane + 1);            // A value of lane l
  const float b = (float)(100 * (lane + 1));    // B value of lane l
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (cbsz_mode == 0) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  else if (cbsz_mode == 1) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 0, 0);
  else c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 2, 0);
  out[lane * 4 + 0] = c.x; out[lane * 4 + 1] = c.y; out[lane * 4 + 2] = c.z; out[lane * 4 + 3] = c.w;
}
template <int NACC>
__global__ void __launch_bounds__(1024) mfma_rate_kernel(float* out, int iters, unsigned long long* ticks) {
  f32x4 c[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) c[j] = {0.f, 0.f, 0.f, 0.f};
  float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)(threadIdx.x & 7) * 1e-3f;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[j], 0, 0, 0);
  }
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j) s += c[j].x + c[j].y + c[j].z + c[j].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int THREADS>
static void run_ag(const char* what, int P, int G, int rows, int iters, int mode, int consumers, int jitter, size_t lds) {
  AgArgs a;
  CK(hipMalloc(&a.flags, 260 * 4));
  CK(hipMalloc(&a.xbuf, (size_t)2 * 256 * 64 * 16));
  CK(hipMalloc(&a.errs, G * 4));
  CK(hipMalloc(&a.ticks, 4 * 8));
  CK(hipMalloc(&a.census, G * 8));
  CK(hipMemset(a.flags, 0, 260 * 4));
  CK(hipMemset(a.xbuf, 0xFF, (size_t)2 * 256 * 64 * 16));
  CK(hipMemset(a.errs, 0, G * 4));
  CK(hipMemset(a.ticks, 0, 32));
  a.P = P; a.G = G; a.rows = rows; a.iters = iters; a.mode = mode; a.consumers = consumers; a.jitter = jitter;
  CK(hipFuncSetAttribute((const void*)ag_kernel<THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((ag_kernel<THREADS>), dim3(G), dim3(THREADS), lds, 0, a);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned> errs(G), census(2 * G), fl(260);
  unsigned long long tk[4];
  CK(hipMemcpy(errs.data(), a.errs, G * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(census.data(), a.census, G * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(fl.data(), a.flags, 260 * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(tk, a.ticks, 32, hipMemcpyDeviceToHost));
  unsigned long long bad = 0;
  for (int i = 0; i < G; ++i) bad += errs[i];
  int per_xcc[16] = {0};
  for (int i = 0; i < G; ++i) per_xcc[census[2 * i] & 15]++;
  // distinct (xcc, se, cu) slots: HW_ID bits [11:8] cu, [15:13] se on gfx9
  std::vector<unsigned> keys;
  for (int i = 0; i < G; ++i) keys.push_back((census[2 * i] << 16) | (census[2 * i + 1] & 0xFF00u));
  std::sort(keys.begin(), keys.end());
  const int distinct = (int)(std::unique(keys.begin(), keys.end()) - keys.begin());
  const double tick_us = 0.01;      // wall_clock64: 100 MHz
  printf("%-44s P=%3d G=%3d thr=%4d rows=%2d cons=%3d jit=%d: %8.2f us/epoch (host)  wg0: publish %.2f poll %.2f payload %.2f total %.2f us  bad words %llu  give-up %u  CUs %d  per-xcc",
         what, P, G, THREADS, rows, consumers, jitter, 1e3 * ms / iters, tk[0] * tick_us / iters, tk[1] * tick_us / iters,
         tk[2] * tick_us / iters, tk[3] * tick_us / iters, bad, fl[256], distinct);
  for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
  printf("\n");
  fflush(stdout);
  CK(hipFree(a.flags)); CK(hipFree(a.xbuf)); CK(hipFree(a.errs)); CK(hipFree(a.ticks)); CK(hipFree(a.census));
}

#include <algorithm>
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  // ---- MFMA 4x4x1 layout
  {
    float* out;
    CK(hipMalloc(&out, 256 * 4));
    std::vector<float> h(256);
    for (int mode = 0; mode < 3; ++mode) {
      hipLaunchKernelGGL(mfma_layout_kernel, dim3(1), dim3(64), 0, 0, out, mode);
      CK(hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost));
      // hypothesis: D[reg i][lane 4b+j] = A(lane 4b'+i) * B(lane 4b+j), b' = b (cbsz 0) or abid (cbsz 4)
      int wrong = 0;
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
          const int b = l >> 2, bsrc = mode == 0 ? b : (mode == 1 ? 0 : 2);
          const float want = (float)(4 * bsrc + i + 1) * (float)(100 * (l + 1));
          wrong += h[l * 4 + i] != want;
        }
      printf("mfma_f32_4x4x1 layout, mode %d (%s): %d of 256 outputs differ from D[reg i][lane] = A[lane 4*blk+i] * B[lane]\n", mode,
             mode == 0 ? "cbsz 0" : (mode == 1 ? "cbsz 4 abid 0" : "cbsz 4 abid 2"), wrong);
      if (wrong) {
        printf("  lane0: %g %g %g %g  lane1: %g %g %g %g  lane5: %g %g %g %g\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[20],
               h[21], h[22], h[23]);
      }
    }
    CK(hipFree(out));
  }
  // ---- MFMA 4x4x1 issue rate: 256 workgroups x 1024 threads (4 waves per SIMD) and x 256 threads (1 wave per SIMD)
  {
    float* out;
    unsigned long long* tk;
    CK(hipMalloc(&out, 256 * 1024 * 4));
    CK(hipMalloc(&tk, 8));
    const int it = 4000;
    for (int thr : {256, 1024}) {
      unsigned long long t[3];
      hipLaunchKernelGGL((mfma_rate_kernel<1>), dim3(256), dim3(thr), 0, 0, out, it, tk);
      CK(hipMemcpy(&t[0], tk, 8, hipMemcpyDeviceToHost));
      hipLaunchKernelGGL((mfma_rate_kernel<3>), dim3(256), dim3(thr), 0, 0, out, it, tk);
      CK(hipMemcpy(&t[1], tk, 8, hipMemcpyDeviceToHost));
      hipLaunchKernelGGL((mfma_rate_kernel<6>), dim3(256), dim3(thr), 0, 0, out, it, tk);
      CK(hipMemcpy(&t[2], tk, 8, hipMemcpyDeviceToHost));
      const int nacc[3] = {1, 3, 6};
      for (int q = 0; q < 3; ++q) {
        const double us = t[q] * 0.01, n = (double)it * nacc[q] * (thr / 64);      // MFMAs per CU
        printf("mfma_f32_4x4x1 rate: %4d threads/CU, %d accumulators per wave: %.1f ns per MFMA per SIMD -> %.1f TFLOP/s chip (peak 157)\n", thr,
               nacc[q], 1e3 * us / (n / 4), n * 512.0 * 256 / us * 1e-6);
      }
    }
    CK(hipFree(out)); CK(hipFree(tk));
  }
  const size_t lds = 96 * 1024;
  // ---- the exchange
  run_ag<1024>("flags only (barrier)", 256, 256, 50, iters, 0, 256, 0, lds);
  run_ag<1024>("flags + payload, everyone reads all", 250, 256, 50, iters, 1, 256, 0, lds);
  run_ag<1024>("same, uneven load", 250, 256, 50, iters, 1, 256, 1, lds);
  run_ag<1024>("150 producers (h block)", 150, 256, 50, iters, 1, 256, 0, lds);
  run_ag<1024>("250 producers, 7 rows (8-GPU shard)", 250, 256, 7, iters, 1, 256, 0, lds);
  run_ag<1024>("same, uneven load", 250, 256, 7, iters, 1, 256, 1, lds);
  run_ag<1024>("32 producers (index leaders)", 32, 256, 50, iters, 1, 256, 0, lds);
  run_ag<512>("512 threads: 250 producers", 250, 256, 50, iters, 1, 256, 0, lds);
  run_ag<256>("256 threads: 250 producers", 250, 256, 50, iters, 1, 256, 0, lds);
  run_ag<1024>("128 workgroups: 125 producers", 125, 128, 50, iters, 1, 128, 0, lds);
  return 0;
}
