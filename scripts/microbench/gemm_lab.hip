// Laboratory for the fp32 tile main loop (round 5): a direct-to-LDS (global_load_lds) operand pipeline against the
// production register-staged loop, with ablations and an in-kernel clock probe.  Stand-alone (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench/gemm_lab.hip -o scripts/microbench/gemm_lab
//   ./gemm_lab [reps]
// C[m,n] = sum_k A[m,k] B[n,k]  (both operands k-contiguous: the Linear-forward form, common.py:37-65 / rssm.py:138-184).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <vector>
#include <string>
#include <string.h>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct LabArgs {
  const float* A; const float* B; float* C;
  int M, N, K, lda, ldb, ldc;
  int tiles_m, tiles_n, n_tiles;
  unsigned long long* clk;        // [grid][4]: s_memtime begin/end, s_memrealtime begin/end
  const float* zero;              // 64 B of zeros (ragged last k-tile)
};

__device__ __forceinline__ int lab_item_of(int id, int n_items) {      // XCD-chunked order (gemm.hip gemm_item_of)
  const int q = n_items >> 3, r = n_items & 7;
  const int xcd = id & 7, j = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// ABL: 0 full kernel, 1 no global loads (LDS holds garbage), 2 MFMAs only (no LDS reads either)
template <int BM, int BN, int WGM, int WGN, int NS, int ABL>
__global__ void __launch_bounds__(64 * WGM * WGN) glds_kernel(const LabArgs g) {
  constexpr int NW = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MB = WM / 32, NB = WN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;       // 1-KiB pieces (8 rows x 128 B) per wave and operand
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces must divide over the waves");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WGN, wn = wave % WGN;

  unsigned long long t0 = 0, r0 = 0;
  if (tid == 0) { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }

  const int item = lab_item_of(blockIdx.x, g.n_tiles);
  const int m0 = (item % g.tiles_m) * BM, n0 = (item / g.tiles_m) * BN;
  const int nkt = (g.K + 31) / 32;

  // ---- loader state: piece i of this wave covers rows 8*(wave + i*NW) .. +7 of the operand; lane -> (row l>>3, slot l&7),
  // the 16-byte chunk it fetches is slot ^ swz(row) so that the lane-linear LDS image is the swizzled one
  unsigned offA[PA], offB[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = 8 * (wave + i * NW) + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const int gr = min(m0 + row, g.M - 1);
    offA[i] = (unsigned)gr * (unsigned)g.lda + chunk * 4;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int row = 8 * (wave + i * NW) + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const int gr = min(n0 + row, g.N - 1);
    offB[i] = (unsigned)gr * (unsigned)g.ldb + chunk * 4;
  }
  auto issue = [&](int kt, int stage) {
    if (ABL != 0) return;
    const int k0 = kt * 32;
    unsigned char* sA = smem + stage * STAGE;
    unsigned char* sB = sA + A_BYTES;
    const bool ragged = k0 + 32 > g.K;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const float* src = g.A + offA[i] + k0;
      if (ragged) { const int row = 8 * (wave + i * NW) + (lane >> 3); const int chunk = (lane & 7) ^ ((row >> 1) & 7); if (k0 + chunk * 4 >= g.K) src = g.zero; }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sA + (wave + i * NW) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const float* src = g.B + offB[i] + k0;
      if (ragged) { const int row = 8 * (wave + i * NW) + (lane >> 3); const int chunk = (lane & 7) ^ ((row >> 1) & 7); if (k0 + chunk * 4 >= g.K) src = g.zero; }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sB + (wave + i * NW) * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets: row*128 + ((2*kg + half) ^ swz)*16, swz = (row>>1)&7 = (l31>>1)&7 (tile offsets are multiples of 32)
  const int swz = (l31 >> 1) & 7;
  unsigned fa[4], fb[4];
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) {
    const int c = ((2 * kg + half) ^ swz) * 16;
    fa[kg] = (wm * WM + l31) * 128 + c;
    fb[kg] = A_BYTES + (wn * WN + l31) * 128 + c;
  }

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkt) issue(s, s);

  for (int kt = 0; kt < nkt; ++kt) {
    // tile kt has landed for this wave's pieces when at most (NS-2) younger tiles are outstanding
    if (ABL == 0) {
      const int younger = min(NS - 2, nkt - 1 - kt);
      if (NS >= 4 && younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PA + PB)) : "memory");
      else if (NS >= 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt + NS - 1 < nkt) issue(kt + NS - 1, (kt + NS - 1) % NS);
    const unsigned char* st = smem + (kt % NS) * STAGE;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      float4 a[MB], b[NB];
      if (ABL == 2) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) a[mb] = make_float4(1.f + kt, 0.5f, 0.25f, 2.f + lane);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = make_float4(1.f, 0.5f + kg, 0.25f, 2.f);
      } else {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) a[mb] = *reinterpret_cast<const float4*>(st + fa[kg] + mb * 32 * 128);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = *reinterpret_cast<const float4*>(st + fb[kg] + nb * 32 * 128);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const float av = j == 0 ? a[mb].x : j == 1 ? a[mb].y : j == 2 ? a[mb].z : a[mb].w;
            const float bv = j == 0 ? b[nb].x : j == 1 ? b[nb].y : j == 2 ? b[nb].z : b[nb].w;
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mb][nb], 0, 0, 0);
          }
    }
  }

#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int col = n0 + wn * WN + nb * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < g.M && col < g.N) g.C[(size_t)row * g.ldc + col] = acc[mb][nb][r];
      }
    }
  if (tid == 0 && g.clk) {
    unsigned long long* c = g.clk + 4 * (size_t)blockIdx.x;
    c[0] = t0; c[1] = __builtin_readcyclecounter(); c[2] = r0; c[3] = __builtin_amdgcn_s_memrealtime();
  }
}


// ---- the same pipeline with the fragment reads as inline asm: hipcc treats an LDS-DMA in flight as a pending LDS store that
// every ds_read may alias and puts s_waitcnt vmcnt(0) in front of the first fragment read of each k-tile (visible in the .s of
// glds_kernel above) - which serialises the operand stream with the MFMAs.  Here the compiler sees no LDS read at all; the
// lgkmcnt / vmcnt waits are placed by hand (cdna_hip_programming.md rule 18: sched_barrier(0) after each wait).
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_READ128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int BN, int WGM, int WGN, int NS, int PRIO>
__global__ void __launch_bounds__(64 * WGM * WGN) gasm_kernel(const LabArgs g) {
  constexpr int NW = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MB = WM / 32, NB = WN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces must divide over the waves");
  static_assert(MB + NB <= 15, "lgkmcnt is a 4-bit counter");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WGN, wn = wave % WGN;
  unsigned long long t0 = 0, r0 = 0;
  if (tid == 0) { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }

  const int item = lab_item_of(blockIdx.x, g.n_tiles);
  const int m0 = (item % g.tiles_m) * BM, n0 = (item / g.tiles_m) * BN;
  const int nkt = (g.K + 31) / 32;
  if (PRIO >= 2) {      // de-phase the workgroups that share a SIMD: the wave in an odd hardware slot starts half a k-tile late
    const unsigned hw = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID bits 3:0 = wave slot on its SIMD
    if (hw & 1) { if (PRIO == 2) __builtin_amdgcn_s_sleep(40); else if (PRIO == 3) __builtin_amdgcn_s_sleep(20); else __builtin_amdgcn_s_sleep(80); }
  }

  const float* pa[PA]; const float* pb[PB];
  int chk[PA > PB ? PA : PB];
#pragma unroll
  for (int i = 0; i < (PA > PB ? PA : PB); ++i) {
    const int row = 8 * (wave + i * NW) + (lane >> 3);
    chk[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 4;
  }
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = 8 * (wave + i * NW) + (lane >> 3);
    pa[i] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + chk[i];
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int row = 8 * (wave + i * NW) + (lane >> 3);
    pb[i] = g.B + (size_t)min(n0 + row, g.N - 1) * g.ldb + chk[i];
  }
  auto issue = [&](int kt, int stage, bool ragged) {
    const int k0 = kt * 32;
    unsigned char* sA = smem + stage * STAGE;
    unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const float* src = pa[i] + k0;
      if (ragged && k0 + chk[i] >= g.K) src = g.zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sA + (wave + i * NW) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const float* src = pb[i] + k0;
      if (ragged && k0 + chk[i] >= g.K) src = g.zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sB + (wave + i * NW) * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int swz = (l31 >> 1) & 7;
  unsigned fa[4], fb[4];
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) {
    const int c = ((2 * kg + half) ^ swz) * 16;
    fa[kg] = lds0 + (wm * WM + l31) * 128 + c;
    fb[kg] = lds0 + A_BYTES + (wn * WN + l31) * 128 + c;
  }
  const bool k_ragged = (g.K & 31) != 0;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkt) issue(s, s, k_ragged && s == nkt - 1);

  int stage = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    {
      const int younger = min(NS - 2, nkt - 1 - kt);
      if (NS >= 4 && younger == 2) wait_vm<2 * (PA + PB)>();
      else if (NS >= 3 && younger == 1) wait_vm<PA + PB>();
      else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    const unsigned so = stage * STAGE;
    f32x4 a[2][MB], b[2][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) LDS_READ128(a[0][mb], fa[0] + so, mb * 32 * 128);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) LDS_READ128(b[0][nb], fb[0] + so, nb * 32 * 128);
    {
      const int nt = kt + NS - 1;
      int ns = stage + NS - 1; if (ns >= NS) ns -= NS;
      if (nt < nkt) issue(nt, ns, k_ragged && nt == nkt - 1);
    }
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int cur = kg & 1, nxt = cur ^ 1;
      if (kg < 3) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) LDS_READ128(a[nxt][mb], fa[kg + 1] + so, mb * 32 * 128);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) LDS_READ128(b[nxt][nb], fb[kg + 1] + so, nb * 32 * 128);
        wait_lgkm<MB + NB>();
      } else wait_lgkm<0>();
      __builtin_amdgcn_sched_barrier(0);
      if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mb][j], b[cur][nb][j], acc[mb][nb], 0, 0, 0);
      if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++stage == NS) stage = 0;
  }

#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int col = n0 + wn * WN + nb * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < g.M && col < g.N) g.C[(size_t)row * g.ldc + col] = acc[mb][nb][r];
      }
    }
  if (tid == 0 && g.clk) {
    unsigned long long* c = g.clk + 4 * (size_t)blockIdx.x;
    c[0] = t0; c[1] = __builtin_readcyclecounter(); c[2] = r0; c[3] = __builtin_amdgcn_s_memrealtime();
  }
}

__global__ void ref_kernel(const LabArgs g) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), row = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (row >= g.M || col >= g.N) return;
  double s = 0.0;
  for (int k = 0; k < g.K; ++k) s += (double)g.A[(size_t)row * g.lda + k] * (double)g.B[(size_t)col * g.ldb + k];
  g.C[(size_t)row * g.ldc + col] = (float)s;
}

// production library (optional): dm_gemm_f32 through the C-ABI
typedef int (*dm_gemm_f32_t)(int, int, int, int, int, const float*, int, const float*, int, float*, int, const float*, const float*,
                             int, int, void*, size_t, void*);
#include <dlfcn.h>

struct Variant { const char* name; int bm, bn, threads, ns; void (*fn)(const LabArgs); };
#define W(BM, BN, WGM, WGN, NS, PRIO) Variant{"gasm<" #BM "," #BN "," #WGM "x" #WGN ",ns" #NS ",prio" #PRIO ">", BM, BN, 64 * WGM * WGN, NS, gasm_kernel<BM, BN, WGM, WGN, NS, PRIO>}
#define V(BM, BN, WGM, WGN, NS, ABL) Variant{"glds<" #BM "," #BN "," #WGM "x" #WGN ",ns" #NS ",abl" #ABL ">", BM, BN, 64 * WGM * WGN, NS, glds_kernel<BM, BN, WGM, WGN, NS, ABL>}

static double now_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const char* libpath = argc > 2 ? argv[2] : nullptr;
  dm_gemm_f32_t prod = nullptr;
  if (libpath) {
    void* h = dlopen(libpath, RTLD_NOW);
    if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    prod = (dm_gemm_f32_t)dlsym(h, "dm_gemm_f32");
  }
  std::vector<Variant> vars = {
    V(128, 128, 2, 2, 2, 0), V(128, 128, 2, 2, 2, 1), V(128, 128, 2, 2, 2, 2),
    W(128, 128, 2, 2, 2, 0), W(128, 128, 2, 2, 2, 2), W(128, 128, 2, 2, 2, 3), W(128, 128, 2, 2, 2, 4), W(128, 128, 2, 2, 2, 1),
    W(256, 256, 2, 2, 2, 0), W(128, 64, 2, 2, 3, 0), W(128, 64, 2, 2, 3, 2), W(64, 64, 2, 2, 3, 0), W(64, 64, 2, 2, 3, 2), W(64, 64, 2, 2, 3, 3),
  };
  struct Shape { int M, N, K; } shapes[] = {
    {4096, 4096, 4096}, {2500, 1800, 1000}, {2500, 1024, 1000}, {2500, 1800, 600}, {2500, 1000, 600},
    {40000, 400, 400}, {40000, 400, 1624}, {562500, 192, 864}, {62500, 192, 2400}, {2500, 4800, 1536},
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* zero; CK(hipMalloc(&zero, 256)); CK(hipMemset(zero, 0, 256));
  unsigned long long* clk; CK(hipMalloc(&clk, 4 * 8 * (size_t)1 << 20));
  void* ws; const size_t ws_bytes = 256u << 20; CK(hipMalloc(&ws, ws_bytes));
  for (auto& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    float *A, *B, *C, *R;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&R, (size_t)M * N * 4));
    {
      std::vector<float> h((size_t)std::max(M, N) * K);
      uint64_t s = 0x9E3779B97F4A7C15ull ^ (uint64_t)M * 1315423911u ^ (uint64_t)K;
      auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; };
      for (size_t i = 0; i < (size_t)M * K; ++i) h[i] = rnd();
      CK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
      for (size_t i = 0; i < (size_t)N * K; ++i) h[i] = rnd();
      CK(hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
    }
    LabArgs g{A, B, R, M, N, K, K, K, N, 0, 0, 0, nullptr, zero};
    // reference on a sample of rows only for the huge shapes (fp64 accumulate): rows 0..255 and the last 256
    const bool sample = M > 4096;
    const int ref_rows = sample ? 256 : M;
    {
      LabArgs gr = g; gr.M = ref_rows;
      hipLaunchKernelGGL(ref_kernel, dim3((N + 63) / 64, (ref_rows + 3) / 4), dim3(256), 0, 0, gr);
      if (sample) {
        gr.A = A + (size_t)(M - 256) * K; gr.C = R + (size_t)(M - 256) * N;
        hipLaunchKernelGGL(ref_kernel, dim3((N + 63) / 64, 64), dim3(256), 0, 0, gr);
      }
      CK(hipDeviceSynchronize());
    }
    std::vector<float> hr((size_t)M * N), hc((size_t)M * N);
    CK(hipMemcpy(hr.data(), R, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    auto check = [&](const char* name) -> double {
      CK(hipMemcpy(hc.data(), C, (size_t)M * N * 4, hipMemcpyDeviceToHost));
      double worst = 0;
      auto rows = [&](int r0, int r1) {
        for (int r = r0; r < r1; ++r)
          for (int c = 0; c < N; ++c) {
            const double d = fabs((double)hc[(size_t)r * N + c] - (double)hr[(size_t)r * N + c]);
            if (!(d <= worst)) worst = d;      // NaN-propagating
          }
      };
      if (sample) { rows(0, 256); rows(M - 256, M); } else rows(0, M);
      return worst;
    };
    printf("== %d x %d x %d  (%.2f GFLOP)\n", M, N, K, 2.0 * M * N * K * 1e-9);
    const double flops = 2.0 * M * N * K;
    if (prod) {
      CK(hipMemset(C, 0xFF, (size_t)M * N * 4));
      for (int i = 0; i < 3; ++i) prod(0, 0, M, N, K, A, K, B, K, C, N, nullptr, nullptr, 0, 0, ws, ws_bytes, nullptr);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) prod(0, 0, M, N, K, A, K, B, K, C, N, nullptr, nullptr, 0, 0, ws, ws_bytes, nullptr);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      const double ms = now_ms(e0, e1) / reps;
      printf("  %-34s %9.1f us %7.1f TF/s  maxerr %.3g\n", "production dm_gemm_f32", ms * 1e3, flops / ms * 1e-9, check("prod"));
    }
    for (auto& v : vars) {
      const int tm = (M + v.bm - 1) / v.bm, tn = (N + v.bn - 1) / v.bn;
      if ((double)tm * tn < 48) continue;
      LabArgs a = g; a.C = C; a.tiles_m = tm; a.tiles_n = tn; a.n_tiles = tm * tn; a.clk = nullptr;
      const size_t lds = (size_t)v.ns * (v.bm + v.bn) * 128;
      if (lds > 160 * 1024) continue;
      CK(hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      CK(hipMemset(C, 0xFF, (size_t)M * N * 4));
      for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(v.fn, dim3(a.n_tiles), dim3(v.threads), lds, 0, a);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(v.fn, dim3(a.n_tiles), dim3(v.threads), lds, 0, a);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      const double ms = now_ms(e0, e1) / reps;
      const bool abl = strstr(v.name, "abl1") || strstr(v.name, "abl2");
      const double err = abl ? -1.0 : check(v.name);
      // clock probe: one more launch with the per-workgroup stamps
      a.clk = clk;
      hipLaunchKernelGGL(v.fn, dim3(a.n_tiles), dim3(v.threads), lds, 0, a);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> hk(4 * (size_t)a.n_tiles);
      CK(hipMemcpy(hk.data(), clk, hk.size() * 8, hipMemcpyDeviceToHost));
      double cyc = 0, rt = 0; unsigned long long rmin = ~0ull, rmax = 0;
      for (int i = 0; i < a.n_tiles; ++i) { cyc += (double)(hk[4 * i + 1] - hk[4 * i]); rt += (double)(hk[4 * i + 3] - hk[4 * i + 2]); rmin = std::min(rmin, hk[4 * i + 2]); rmax = std::max(rmax, hk[4 * i + 3]); }
      const double ghz = rt > 0 ? cyc / rt * 0.1 : 0;         // s_memrealtime ticks at 100 MHz
      printf("  %-34s %9.1f us %7.1f TF/s  maxerr %.3g  clk %.2f GHz  span %.1f us  wg-cycles %.0f\n", v.name, ms * 1e3, flops / ms * 1e-9, err, ghz,
             (double)(rmax - rmin) * 0.01, cyc / a.n_tiles);
      fflush(stdout);
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(R));
  }
  return 0;
}
