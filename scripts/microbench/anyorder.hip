// Does hipExtAnyOrderLaunch give kernel-level concurrency INSIDE one stream on gfx950 / ROCm 7.2?
// (hip_ext.h says the flag "is not supported on AMD GFX9xx boards" for hipExtModuleLaunchKernel; this measures it.)
// Two kernels of 64 workgroups that each spin for `us` microseconds: launched back to back in ONE stream, (a) both ordinary,
// (b) the second with hipExtAnyOrderLaunch.  Concurrent = the pair takes ~1x, serial = ~2x.  Also checks that a THIRD,
// ordinary kernel behind the pair still waits for BOTH (the AQL barrier bit waits for every earlier packet).
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/anyorder.hip -o scripts/microbench/anyorder && scripts/microbench/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(int us, int* flag, int value) {
  const unsigned long long t0 = wall_clock64();                 // 100 MHz
  while (wall_clock64() - t0 < (unsigned long long)us * 100ull) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) *flag = value;
}
__global__ void check(const int* fa, const int* fb, int* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = (*fa == 1 && *fb == 2) ? 1 : 0;
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int *fa, *fb, *ok;
  CK(hipMalloc(&fa, 4)); CK(hipMalloc(&fb, 4)); CK(hipMalloc(&ok, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {64, 256, 1024}) {
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e9f; int allok = 1;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemsetAsync(fa, 0, 4, st)); CK(hipMemsetAsync(fb, 0, 4, st)); CK(hipMemsetAsync(ok, 0, 4, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, st, 200, fa, 1);
        if (mode == 0) hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, st, 200, fb, 2);
        else hipExtLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, 200, fb, 2);
        hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, st, fa, fb, ok);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        int h = 0; CK(hipMemcpy(&h, ok, 4, hipMemcpyDeviceToHost));
        allok &= h;
        if (ms < best) best = ms;
      }
      printf("grid %4d  %s  pair of 200 us kernels + check: %.1f us   third kernel saw both results: %s\n", grid,
             mode ? "second = hipExtAnyOrderLaunch" : "both ordinary               ", best * 1e3f, allok ? "yes" : "NO");
    }
  }
  return 0;
}
