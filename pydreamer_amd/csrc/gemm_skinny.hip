// Skinny fp32 GEMM for the sequential RSSM chains: M <= 64 rows (one batch of recurrent states), any N, any K - and, in
// 64-row chunks, for the <= 512-row products of a data-parallel shard's imagination step.
//
//   C[m,n] = epi( sum_k A[m,k] * B(n,k) )          A row-major (k contiguous); B either [N][K] (weights, forward, and
//                                                  the BPTT loop's transposed weights) or [K][N]
//
// The tiled kernel in gemm.hip needs split-K plus a second (reduce) launch to put more than ~20 workgroups on these
// shapes, and each of the ~500 per-step GEMMs of the T / BPTT loops then costs ~11 us of mostly launch latency.
// Here one launch does the whole product: a workgroup owns a 64 x 16 output strip (grid = N/16 x M/64: 38-113
// workgroups per chunk), its 16 waves split K sixteen ways, each wave accumulates a (16*NRB) x 16 partial (NRB =
// populated 16-row blocks: a 7-row shard loads a quarter of the activation rows) on v_mfma_f32_16x16x4_f32 straight
// from global memory (weights from MALL/HBM and a <= 64-row activation block from L2; no LDS staging, 4 chunks of
// loads in flight per wave), and the 16 partials are summed through LDS in a fixed order (deterministic).
// fp32-input MFMA is an exact fmaf chain, so numerics match the tiled kernel up to summation order.
// skinny_gemm_pair_kernel runs two independent products in one launch (the loops are bound by the launch count).
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SkinnyArgs {
  const float* A; const float* B; float* C;
  const float* bias; const float* add;
  const uint8_t* row_zero;
  int M, N, K, lda, ldb, ldc, ldadd, flags;
};

// Raw loads from clamped (always valid) addresses + a validity mask (bits 0-3: A row blocks, bit 4: B).  The zero-fill
// select is applied where the chunk is CONSUMED, one iteration later, so the loads stay in flight under the MFMAs.
template <int BL, int NRB>
__device__ __forceinline__ void skinny_load(const SkinnyArgs& g, int m0, int n0, int c, int kend, int lane,
                                            float4 (&a)[NRB], float4& b, unsigned& mask) {
  mask = 0u;
  const int l15 = lane & 15, q = lane >> 4;
  const int k = c + 4 * q;
  const bool kin = k < kend;                      // kend % 4 == 0 (host-checked K % 4): a group of 4 is all in or all out
#pragma unroll
  for (int mb = 0; mb < NRB; ++mb) {
    const int m = m0 + mb * 16 + l15;
    const bool ok = kin && m < g.M;
    a[mb] = *reinterpret_cast<const float4*>(g.A + (ok ? (size_t)m * g.lda + k : 0));
    mask |= ok ? (1u << mb) : 0u;
  }
  const int n = n0 + l15;
  const bool okb = kin && n < g.N;
  if (BL == 0) {
    b = *reinterpret_cast<const float4*>(g.B + (okb ? (size_t)n * g.ldb + k : 0));
  } else {
    const float* p = g.B + (okb ? (size_t)k * g.ldb + n : 0);
    const size_t st = okb ? (size_t)g.ldb : 0;
    b = make_float4(p[0], p[st], p[2 * st], p[3 * st]);
  }
  mask |= okb ? 16u : 0u;
}

constexpr int SK_DEPTH = 4;    // 16-k chunks loaded per round (all in flight before the first MFMA)
constexpr int SK_WAVES = 16;   // waves per workgroup = K splits inside it (8 and 4 measured slower, also under contention)

// One (16*NRB) x 16 output strip; NRB = populated 16-row blocks (M <= 16*NRB): a 7-row data-parallel shard reads a
// quarter of the activation traffic of the 50-row case.
template <int BL, int NRB>
__device__ __forceinline__ void skinny_strip(const SkinnyArgs& g, int strip, int m0, float (*part)[64 * 16]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = strip * 16;
  const int chunks = (g.K + 15) / 16;
  const int per = (chunks + SK_WAVES - 1) / SK_WAVES;
  const int c_beg = wave * per * 16;
  int c_end = c_beg + per * 16;
  if (c_end > chunks * 16) c_end = chunks * 16;

  f32x4 acc[NRB];
#pragma unroll
  for (int mb = 0; mb < NRB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // These products are latency bound (weights come from the MALL / HBM, ~1 us a round trip), so a round issues the
  // loads of SK_DEPTH chunks back to back and only then starts consuming them under counted waits; the 16 waves of
  // the workgroup (4 per SIMD) overlap each other's rounds.  With K <= 1024 a wave's whole K range is ONE round.
  // A chunk past the wave's range loads clamped addresses under an all-zero mask and contributes exact zeros.
  int kend = c_end < g.K ? c_end : g.K;
  if (kend < c_beg) kend = c_beg;
  for (int c = c_beg; c < c_end; c += 16 * SK_DEPTH) {
    float4 a[SK_DEPTH][NRB], b[SK_DEPTH];
    unsigned mk[SK_DEPTH];
#pragma unroll
    for (int d = 0; d < SK_DEPTH; ++d) skinny_load<BL, NRB>(g, m0, n0, c + 16 * d, kend, lane, a[d], b[d], mk[d]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < SK_DEPTH; ++d) {
      const bool okb = (mk[d] & 16u) != 0u;
      const float bj[4] = {okb ? b[d].x : 0.f, okb ? b[d].y : 0.f, okb ? b[d].z : 0.f, okb ? b[d].w : 0.f};
#pragma unroll
      for (int mb = 0; mb < NRB; ++mb) {
        const bool oka = ((mk[d] >> mb) & 1u) != 0u;
        const float aj[4] = {oka ? a[d][mb].x : 0.f, oka ? a[d][mb].y : 0.f, oka ? a[d][mb].z : 0.f,
                             oka ? a[d][mb].w : 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[j], bj[j], acc[mb], 0, 0, 0);
      }
    }
  }
  // C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
  for (int mb = 0; mb < NRB; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][(mb * 16 + (lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[mb][r];
  __syncthreads();
  for (int e = tid; e < NRB * 16 * 16; e += SK_WAVES * 64) {
    const int row = m0 + (e >> 4), col = n0 + (e & 15);
    if (row < g.M && col < g.N) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < SK_WAVES; ++w) v += part[w][e];
      if (g.row_zero && g.row_zero[row]) v = 0.f;
      if (g.bias) v += g.bias[col];
      if (g.add) v += g.add[(size_t)row * g.ldadd + col];
      float* cp = g.C + (size_t)row * g.ldc + col;
      if (g.flags & DM_GEMM_ACCUM) v += *cp;
      if (g.flags & DM_GEMM_ELU) v = dm_elu(v);
      *cp = v;
    }
  }
}

template <int BL, int NRB>
__global__ void __launch_bounds__(SK_WAVES * 64) skinny_gemm_kernel(const SkinnyArgs g) {
  __shared__ float part[SK_WAVES][64 * 16];
  skinny_strip<BL, NRB>(g, blockIdx.x, blockIdx.y * 64, part);       // grid.y = 64-row chunks of M
}
// Two independent products in ONE launch (the loops are bound by the number of launches, ~5 us of GPU time and ~6 us of
// host time each): the GRU's input and hidden gate products of a posterior step, the two backward-data products that
// feed step t-1's state gradient.  Workgroups [0, nb0) serve the first product, the rest the second.
struct SkinnyPairArgs { SkinnyArgs g[2]; int nb0; };
template <int NRB>
__global__ void __launch_bounds__(SK_WAVES * 64) skinny_gemm_pair_kernel(const SkinnyPairArgs a) {
  __shared__ float part[SK_WAVES][64 * 16];
  const int b = blockIdx.x;
  if (b < a.nb0) skinny_strip<0, NRB>(a.g[0], b, 0, part);
  else skinny_strip<0, NRB>(a.g[1], b - a.nb0, 0, part);
}

// max_m: 64 for the pair kernel (one chunk); the single-product kernel walks M in 64-row chunks (grid.y) up to
// g_skinny_max_m rows, where it still beats the tiled kernel's <= ~100 workgroups (imagination products of a
// data-parallel shard: M = T*B/8 = 350)
static const int g_skinny_max_m = getenv("DM_SKINNY_MAX_M") ? atoi(getenv("DM_SKINNY_MAX_M")) : 512;
static bool skinny_ok(const DmGemm& q, int max_m) {
  if (q.M > max_m || q.M < 1 || q.N < 1 || q.K < 16) return false;
  if (q.a_layout != 0 || q.a_maj || q.b_maj || q.mulref) return false;
  if ((q.K & 3) || (q.lda & 3) || ((uintptr_t)q.A & 15)) return false;
  if (q.b_layout == 0 && ((q.ldb & 3) || ((uintptr_t)q.B & 15))) return false;
  if ((int64_t)q.N * q.K < (int64_t)64 * 1024) return false;             // tiny products: one tiled workgroup is fine
  return true;
}
static void skinny_fill(const DmGemm& q, SkinnyArgs& a) {
  a.A = q.A; a.B = q.B; a.C = q.C; a.bias = q.bias; a.add = q.add; a.row_zero = q.row_zero;
  a.M = q.M; a.N = q.N; a.K = q.K; a.lda = q.lda; a.ldb = q.ldb; a.ldc = q.ldc; a.ldadd = q.ldadd; a.flags = q.flags;
}
static const int g_skinny_disabled = getenv("DM_GEMM_NO_SKINNY") ? 1 : 0;      // A/B switch for scripts/gemm_bench.py

// Returns 1 if the launch was taken by the skinny kernel, 0 if the shape / alignment does not qualify, < 0 on error.
int dm_gemm_skinny_try(const DmGemm& q, hipStream_t stream) {
  if (g_skinny_disabled || !skinny_ok(q, g_skinny_max_m)) return 0;
  // beyond one chunk it pays only for short reductions over small weight matrices (measured at M = 350: 1000x1024
  // 31.9 -> 24.8 us, 400x400 15.1 -> 8.7 us; 1800x1000 equal; 400x1624 18.2 -> 19.9 us)
  if (q.M > 64 && (q.K > 1024 || (int64_t)q.N * q.K > (int64_t)1100 * 1024)) return 0;
  SkinnyArgs a;
  skinny_fill(q, a);
  const dim3 grid((unsigned)dm_cdiv(q.N, 16), (unsigned)dm_cdiv(q.M, 64));
  const dim3 blk(SK_WAVES * 64);
  if (q.b_layout == 0) {
    if (q.M <= 16) hipLaunchKernelGGL((skinny_gemm_kernel<0, 1>), grid, blk, 0, stream, a);
    else if (q.M <= 32) hipLaunchKernelGGL((skinny_gemm_kernel<0, 2>), grid, blk, 0, stream, a);
    else hipLaunchKernelGGL((skinny_gemm_kernel<0, 4>), grid, blk, 0, stream, a);
  } else {
    if (q.M <= 16) hipLaunchKernelGGL((skinny_gemm_kernel<1, 1>), grid, blk, 0, stream, a);
    else if (q.M <= 32) hipLaunchKernelGGL((skinny_gemm_kernel<1, 2>), grid, blk, 0, stream, a);
    else hipLaunchKernelGGL((skinny_gemm_kernel<1, 4>), grid, blk, 0, stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dm_fail(DM_E_HIP, "skinny gemm: %s", hipGetErrorString(e));
  return 1;
}
// Both products in one launch if both qualify (k-contiguous B); otherwise two ordinary launches.
int dm_gemm_pair_launch(const DmGemm& q0, const DmGemm& q1, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!g_skinny_disabled && skinny_ok(q0, 64) && skinny_ok(q1, 64) && q0.b_layout == 0 && q1.b_layout == 0) {
    SkinnyPairArgs a;
    skinny_fill(q0, a.g[0]);
    skinny_fill(q1, a.g[1]);
    a.nb0 = dm_cdiv(q0.N, 16);
    const dim3 grid((unsigned)(a.nb0 + dm_cdiv(q1.N, 16))), blk(SK_WAVES * 64);
    const int mmax = q0.M > q1.M ? q0.M : q1.M;
    if (mmax <= 16) hipLaunchKernelGGL((skinny_gemm_pair_kernel<1>), grid, blk, 0, stream, a);
    else if (mmax <= 32) hipLaunchKernelGGL((skinny_gemm_pair_kernel<2>), grid, blk, 0, stream, a);
    else hipLaunchKernelGGL((skinny_gemm_pair_kernel<4>), grid, blk, 0, stream, a);
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  DM_TRY(dm_gemm_launch(q0, ws, ws_bytes, stream));
  return dm_gemm_launch(q1, ws, ws_bytes, stream);
}
