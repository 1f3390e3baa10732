// Microbenchmark (not part of the library): how fast can ONE CU / all CUs pull fp32 operands into registers?
//   l2_stream <buffer KiB> <loads in flight per thread U: 4|8|16|32> <threads per WG> <WGs per CU> <pattern 0|1>
// pattern 0: fully coalesced 16-byte loads (thread t of a wave reads 16 B at t*16: 8 full 128-byte lines per instruction)
// pattern 1: MFMA-fragment gather (lane l reads 16 B of row l&15 at column 4*(l>>4): 16 rows x 64 B per instruction)
// pattern 2: GEMM-tile rows (8 lanes per row: 8 rows x 128 B per instruction, rows `row4` float4 apart)
// pattern 3: 16 lanes per row: 4 rows x 256 B per instruction
// Every WG streams the same buffer `reps` times, so after the first pass it is L2 / MALL resident as its size allows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int U, int PAT>
__global__ void __launch_bounds__(1024) stream_kernel(const float4* __restrict__ buf, size_t n4, int reps, int row4, float* out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < reps; ++r) {
    if (PAT == 0) {
      for (size_t i = (size_t)tid; i + (size_t)(U - 1) * blockDim.x < n4; i += (size_t)U * blockDim.x) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = buf[i + (size_t)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
    } else if (PAT == 2 || PAT == 3) {
      constexpr int LPR = PAT == 2 ? 8 : 16;           // lanes per row
      constexpr int RPI = 64 / LPR;                    // rows per instruction
      const size_t rows = n4 / row4;
      for (size_t band = wave; band * (RPI * U) + RPI * U - 1 < rows; band += nw) {
        for (int c = 0; c + LPR <= row4; c += LPR) {
          float4 v[U];
#pragma unroll
          for (int u = 0; u < U; ++u) v[u] = buf[(band * (RPI * U) + u * RPI + lane / LPR) * row4 + c + (lane % LPR)];
#pragma unroll
          for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
      }
    } else {
      // rows of row4 float4; a wave walks 16-row bands: lane -> (row = band*16 + (l&15), col4 = c*4 + (l>>4))
      const size_t rows = n4 / row4;
      for (size_t band = wave; band * 16 + 15 < rows; band += nw) {
        const float4* p = buf + (band * 16 + (lane & 15)) * row4 + (lane >> 4);
        for (int c = 0; c + 4 * (U - 1) < row4 - 3; c += 4 * U) {
          float4 v[U];
#pragma unroll
          for (int u = 0; u < U; ++u) v[u] = p[c + 4 * u];
#pragma unroll
          for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main(int argc, char** argv) {
  const size_t kib = argc > 1 ? atol(argv[1]) : 1024;
  const int U = argc > 2 ? atoi(argv[2]) : 8, threads = argc > 3 ? atoi(argv[3]) : 256, wgs_per_cu = argc > 4 ? atoi(argv[4]) : 1;
  const int pat = argc > 5 ? atoi(argv[5]) : 0;
  const int ncu = argc > 6 ? atoi(argv[6]) : 256;
  const size_t n4 = kib * 1024 / 16;
  float4* buf; float* out;
  hipMalloc(&buf, n4 * 16); hipMalloc(&out, 4);
  hipMemset(buf, 0, n4 * 16);
  const int reps = (int)((size_t)(64 << 20) / (kib * 1024)) + 2;       // ~64 MiB streamed per WG
  const int row4 = 400;                                                  // pattern 1: 1600-float rows
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) {
    hipEventRecord(e0);
#define L(UU) \
    if (pat == 0) hipLaunchKernelGGL((stream_kernel<UU, 0>), dim3(ncu * wgs_per_cu), dim3(threads), 0, 0, buf, n4, reps, row4, out); \
    else if (pat == 2) hipLaunchKernelGGL((stream_kernel<UU, 2>), dim3(ncu * wgs_per_cu), dim3(threads), 0, 0, buf, n4, reps, row4, out); \
    else if (pat == 3) hipLaunchKernelGGL((stream_kernel<UU, 3>), dim3(ncu * wgs_per_cu), dim3(threads), 0, 0, buf, n4, reps, row4, out); \
    else hipLaunchKernelGGL((stream_kernel<UU, 1>), dim3(ncu * wgs_per_cu), dim3(threads), 0, 0, buf, n4, reps, row4, out);
    if (U == 4) { L(4) } else if (U == 8) { L(8) } else if (U == 16) { L(16) } else { L(32) }
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)reps * kib * 1024 * ncu * wgs_per_cu;
  printf("buf %6zu KiB U %2d threads %4d wg/cu %d pat %d cus %3d: %8.1f us  %7.2f TB/s  %6.1f GB/s per CU  %5.1f B/clk/CU @2.4GHz\n", kib, U, threads,
         wgs_per_cu, pat, ncu, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / ncu, bytes / (ms * 1e-3) / ncu / 2.4e9);
  return 0;
}
