// Row-panel Linear kernels for the 400-wide MLP heads (common.py:37-65; a2c.py:37-39; decoders.py:257-319).
//
// A workgroup owns a panel of 64 COMPLETE output rows (N = hidden <= 16*NBLK columns), so everything that needs a whole
// row happens in the epilogue of the GEMM that produced it, with the row still in registers:
//
//   EPI_LN_FWD : y = ELU(LayerNorm(A W^T + b; gamma, beta, eps))         Linear -> LayerNorm -> ELU in ONE launch
//                (+ optionally the MLP's output layer  out = y Wout^T + bout  when out_dim <= 32: a row reduction)
//   EPI_LN_BWD : dx = LayerNorm'ELU' backward of  dy = A W  (the data gradient arriving from the layer above), plus this
//                panel's partial column sums for dbias / dgamma / dbeta
//
// instead of GEMM -> LayerNorm kernel -> (next GEMM) with an HBM round trip of the pre-activation in between, and
// instead of dgrad GEMM -> ln_bwd_dx -> ln_bwd_params (2 launches) -> colsum (2 launches) in backward.
//
// Tiling: 4 waves, wave w owns rows 16w..16w+15 and ALL NBLK 16-column blocks as v_mfma_f32_16x16x4_f32 accumulators
// (4*NBLK = 100 VGPRs at hidden 400); A (64 x 32) and B (N x 32) tiles are staged through LDS with the register
// prefetch of gemm.hip.  LDS row stride 40 floats: the ds_read_b128 fragment reads of the 16x16x4 operand pattern
// (lane l: row l&15, k = 4*(l>>4) .. +3) are conflict-free under gfx950's b128 lane groups (stride 36 is 2-way).
// The k -> (mfma step, lane group) map is the skinny kernel's: step j of a 16-k group feeds k = 16g + 4q + j from lane
// group q, for A and B alike.  fp32-input MFMA = exact fmaf chain, so results are fp32-class like gemm.hip's.
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { PANEL_EPI_LN_FWD = 1, PANEL_EPI_LN_BWD = 2 };
constexpr int PANEL_BM = 64;
constexpr int PANEL_LDA = 40;         // LDS row stride (floats) of the [row][k] images

struct PanelArgs {
  const float* A; const float* B; const float* bias;
  int M, N, K, lda, ldb;
  const float* gamma; const float* beta; float eps;
  // EPI_LN_FWD outputs (each optional)
  float* xpre; float* stats; float* y; int ldy;
  const float* wout; const float* bout; float* out; int out_dim, ldout;
  // EPI_LN_BWD inputs / outputs
  const float* xin; const float* stin; float* dx; int lddx; float* colpart;
  const unsigned short* Bh;      // bf16 copy of B ([n][k], row stride ldb elements; K % 8 == 0), bf16-operand kernel only
  const float* addm;             // EPI_LN_FWD, optional (M x N, dense): added to the product before the LayerNorm
};

__device__ __forceinline__ float panel_red16(float v) {      // sum over the 16 lanes that share lane>>4
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

template <bool AVEC>
__device__ __forceinline__ void panel_load_a(float4 (&r)[2], unsigned& mask, const PanelArgs& g, int m0, int k0, int tid) {
  mask = 0u;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = tid + i * 256;
    const int row = m0 + (f >> 3);
    const int k = k0 + ((f & 7) << 2);
    if (AVEC) {
      const bool ok = row < g.M && k < g.K;
      r[i] = *reinterpret_cast<const float4*>(g.A + (ok ? (size_t)row * g.lda + k : 0));
      mask |= ok ? (0xFu << (4 * i)) : 0u;
    } else {
      float e[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = row < g.M && k + j < g.K;
        e[j] = g.A[ok ? (size_t)row * g.lda + k + j : 0];
        mask |= ok ? (1u << (4 * i + j)) : 0u;
      }
      r[i] = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
}
__device__ __forceinline__ void panel_store_a(const float4 (&r)[2], unsigned mask, float* As, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = tid + i * 256;
    float4 v;
    v.x = (mask >> (4 * i + 0)) & 1u ? r[i].x : 0.f;
    v.y = (mask >> (4 * i + 1)) & 1u ? r[i].y : 0.f;
    v.z = (mask >> (4 * i + 2)) & 1u ? r[i].z : 0.f;
    v.w = (mask >> (4 * i + 3)) & 1u ? r[i].w : 0.f;
    *reinterpret_cast<float4*>(&As[(f >> 3) * PANEL_LDA + ((f & 7) << 2)]) = v;
  }
}

// B tile.  BL 0: B[n][k] (weights, forward) -> LDS [n][k] stride 40.  BL 1: B[k][n] (weights, backward-data) -> LDS [k][n]
// stride 16*NBLK + 4 (ds_read_b32 fragments: lane groups 4 k-rows apart land 16 banks apart).
template <int NBLK, int BL>
struct PanelB {
  static constexpr int NQ = NBLK * 4;                                   // float4 per k-row (BL 1)
  static constexpr int TOTAL = NBLK * 16 * 8;                           // float4 per tile, either layout
  static constexpr int NF4 = (TOTAL + 255) / 256;
  static constexpr int LDM = NBLK * 16 + 4;
  static constexpr int FLOATS = BL == 0 ? NBLK * 16 * PANEL_LDA : 32 * LDM;

  static __device__ __forceinline__ void load(float4 (&r)[NF4], unsigned& mask, const PanelArgs& g, int k0, int tid) {
    mask = 0u;
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int f = tid + i * 256;
      bool ok;
      size_t off;
      if (BL == 0) {
        const int n = f >> 3, k = k0 + ((f & 7) << 2);
        ok = n < NBLK * 16 && k < g.K;
        off = (size_t)n * g.ldb + k;
      } else {
        const int kk = f / NQ, n = (f % NQ) << 2;
        ok = f < TOTAL && k0 + kk < g.K;
        off = (size_t)(k0 + kk) * g.ldb + n;
      }
      r[i] = *reinterpret_cast<const float4*>(g.B + (ok ? off : 0));
      mask |= ok ? (1u << i) : 0u;
    }
  }
  static __device__ __forceinline__ void store(const float4 (&r)[NF4], unsigned mask, float* Bs, int tid) {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int f = tid + i * 256;
      if (f < TOTAL) {
        const bool ok = (mask >> i) & 1u;
        const float4 v = ok ? r[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (BL == 0) *reinterpret_cast<float4*>(&Bs[(f >> 3) * PANEL_LDA + ((f & 7) << 2)]) = v;
        else *reinterpret_cast<float4*>(&Bs[(f / NQ) * LDM + ((f % NQ) << 2)]) = v;
      }
    }
  }
};

// Epilogue shared by the fp32 and the bf16-operand main loops: the workgroup's 64 complete rows are in `acc` in the MFMA
// C/D layout.  `smem` is the kernel's LDS (free after the main loop's trailing barrier; the backward epilogue reduces through it).
template <int NBLK, int EPI>
__device__ __forceinline__ void panel_epilogue(f32x4 (&acc)[NBLK], const PanelArgs& g, float* smem, int m0, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  // C/D map of the 16x16 MFMA: col = lane & 15 (+16*block), row = 4*(lane>>4) + r.  This lane's 4 rows start at rbase.
  // N == 16*NBLK exactly (host-checked), so every column of every block is a real column.
  const int rbase = m0 + wave * 16 + 4 * q;
  const float inv_n = 1.0f / (float)(NBLK * 16);
  bool rok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) rok[r] = rbase + r < g.M;

  if (EPI == PANEL_EPI_LN_FWD) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      const float bi = g.bias ? g.bias[b * 16 + l15] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[b][r] += bi;
        if (g.addm && rok[r]) acc[b][r] += g.addm[(size_t)(rbase + r) * (NBLK * 16) + b * 16 + l15];
        s[r] += acc[b][r];
      }
    }
    float mean[4], rstd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mean[r] = panel_red16(s[r]) * inv_n;
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[b][r] - mean[r];
        ss[r] += d * d;
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) rstd[r] = 1.0f / sqrtf(panel_red16(ss[r]) * inv_n + g.eps);
    if (g.stats && l15 == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (rok[r]) { g.stats[2 * (size_t)(rbase + r)] = mean[r]; g.stats[2 * (size_t)(rbase + r) + 1] = rstd[r]; }
    }
    if (g.xpre) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (rok[r]) {
          float* xr = g.xpre + (size_t)(rbase + r) * (NBLK * 16) + l15;
#pragma unroll
          for (int b = 0; b < NBLK; ++b) xr[b * 16] = acc[b][r];
        }
    }
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      const float ga = g.gamma[b * 16 + l15], be = g.beta[b * 16 + l15];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[b][r] = dm_elu((acc[b][r] - mean[r]) * rstd[r] * ga + be);
    }
    if (g.y) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (rok[r]) {
          float* yr = g.y + (size_t)(rbase + r) * g.ldy + l15;
#pragma unroll
          for (int b = 0; b < NBLK; ++b) yr[b * 16] = acc[b][r];
        }
    }
    if (g.wout) {      // the MLP's output layer as a row reduction over the panel's post-activation rows
      for (int o = 0; o < g.out_dim; ++o) {
        const float* wr = g.wout + (size_t)o * (NBLK * 16) + l15;
        float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          const float w = wr[b * 16];
#pragma unroll
          for (int r = 0; r < 4; ++r) p[r] += acc[b][r] * w;
        }
        const float bo = g.bout ? g.bout[o] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = panel_red16(p[r]);
          if (l15 == 0 && rok[r]) g.out[(size_t)(rbase + r) * g.ldout + o] = t + bo;
        }
      }
    }
  } else {
    // acc = dy.  gl = dy * ELU'(pre), g = gl * gamma, xhat = (x - mean) * rstd;
    //   dx = rstd * (g - mean_row(g) - xhat * mean_row(g xhat));
    // column sums over the panel's rows: dbeta += gl, dgamma += gl * xhat, dbias += dx.
    // Rows past M carry mean = rstd = 0 and a zero A row (dy = 0), so they contribute exact zeros everywhere.
    float mean[4], rstd[4];
    const float* xr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t row = rok[r] ? (size_t)(rbase + r) : 0;
      mean[r] = rok[r] ? g.stin[2 * row] : 0.f;
      rstd[r] = rok[r] ? g.stin[2 * row + 1] : 0.f;
      xr[r] = g.xin + row * (NBLK * 16) + l15;
    }
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sgx[4] = {0.f, 0.f, 0.f, 0.f};
    float* red = smem;              // [4 waves][3][16*NBLK]: the main loop's trailing barrier is behind every wave
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      const float ga = g.gamma[b * 16 + l15], be = g.beta[b * 16 + l15];
      float cg = 0.f, cb = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xh = (xr[r][b * 16] - mean[r]) * rstd[r];
        const float pre = xh * ga + be;
        const float gl = acc[b][r] * (pre > 0.f ? 1.f : expf(pre));
        cb += gl;
        cg += gl * xh;
        const float gg = gl * ga;
        sg[r] += gg;
        sgx[r] += gg * xh;
        acc[b][r] = gg;
      }
      cg += __shfl_xor(cg, 16, 64); cg += __shfl_xor(cg, 32, 64);       // the wave's 16 rows
      cb += __shfl_xor(cb, 16, 64); cb += __shfl_xor(cb, 32, 64);
      if (q == 0) {
        red[(wave * 3 + 1) * (NBLK * 16) + b * 16 + l15] = cg;
        red[(wave * 3 + 2) * (NBLK * 16) + b * 16 + l15] = cb;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sg[r] = panel_red16(sg[r]) * inv_n;
      sgx[r] = panel_red16(sgx[r]) * inv_n;
    }
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      float cx = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xh = (xr[r][b * 16] - mean[r]) * rstd[r];            // second touch of x: L1/L2 hit
        const float d = rstd[r] * (acc[b][r] - sg[r] - xh * sgx[r]);
        acc[b][r] = d;
        cx += d;
      }
      cx += __shfl_xor(cx, 16, 64); cx += __shfl_xor(cx, 32, 64);
      if (q == 0) red[(wave * 3 + 0) * (NBLK * 16) + b * 16 + l15] = cx;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (rok[r]) {
        float* dr = g.dx + (size_t)(rbase + r) * g.lddx + l15;
#pragma unroll
        for (int b = 0; b < NBLK; ++b) dr[b * 16] = acc[b][r];
      }
    __syncthreads();
    float* part = g.colpart + (size_t)blockIdx.x * 3 * (NBLK * 16);
    for (int e = tid; e < 3 * NBLK * 16; e += 256) {       // e = q*N + col: the four waves in fixed order
      const int qq = e / (NBLK * 16), col = e % (NBLK * 16);
      part[e] = (red[(0 * 3 + qq) * (NBLK * 16) + col] + red[(1 * 3 + qq) * (NBLK * 16) + col]) +
                (red[(2 * 3 + qq) * (NBLK * 16) + col] + red[(3 * 3 + qq) * (NBLK * 16) + col]);
    }
  }
}

template <int NBLK, int BL, int EPI, bool AVEC>
__global__ void __launch_bounds__(256, 2) panel_linear_kernel(const PanelArgs g) {
  typedef PanelB<NBLK, BL> PB;
  constexpr int GRP = (NBLK % 5 == 0) ? 5 : (NBLK % 4 == 0) ? 4 : (NBLK % 3 == 0) ? 3 : (NBLK % 2 == 0) ? 2 : 1;
  constexpr int A_FLOATS = PANEL_BM * PANEL_LDA;
  constexpr int RED_FLOATS = (EPI == PANEL_EPI_LN_BWD) ? 4 * 3 * NBLK * 16 : 0;
  constexpr int MAIN_FLOATS = A_FLOATS + PB::FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[MAIN_FLOATS > RED_FLOATS ? MAIN_FLOATS : RED_FLOATS];
  float* As = smem;
  float* Bs = smem + A_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * PANEL_BM;

  f32x4 acc[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 ra[2], rb[PB::NF4];
  unsigned ma = 0u, mb = 0u;
  const int nkt = (g.K + 31) / 32;
  panel_load_a<AVEC>(ra, ma, g, m0, 0, tid);
  PB::load(rb, mb, g, 0, tid);
  for (int kt = 0; kt < nkt; ++kt) {
    panel_store_a(ra, ma, As, tid);
    PB::store(rb, mb, Bs, tid);
    __syncthreads();
    if (kt + 1 < nkt) {                               // register prefetch of the next tile under the MFMAs below
      panel_load_a<AVEC>(ra, ma, g, m0, (kt + 1) * 32, tid);
      PB::load(rb, mb, g, (kt + 1) * 32, tid);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const float4 a4 = *reinterpret_cast<const float4*>(&As[(wave * 16 + l15) * PANEL_LDA + kg * 16 + 4 * q]);
      const float aj[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int gb = 0; gb < NBLK / GRP; ++gb) {
        if (BL == 0) {
          float bj[GRP][4];
#pragma unroll
          for (int u = 0; u < GRP; ++u) {
            const float4 b4 = *reinterpret_cast<const float4*>(&Bs[((gb * GRP + u) * 16 + l15) * PANEL_LDA + kg * 16 + 4 * q]);
            bj[u][0] = b4.x; bj[u][1] = b4.y; bj[u][2] = b4.z; bj[u][3] = b4.w;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < GRP; ++u)
              acc[gb * GRP + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[j], bj[u][j], acc[gb * GRP + u], 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float bv[GRP];
#pragma unroll
            for (int u = 0; u < GRP; ++u) bv[u] = Bs[(kg * 16 + 4 * q + j) * PB::LDM + (gb * GRP + u) * 16 + l15];
#pragma unroll
            for (int u = 0; u < GRP; ++u)
              acc[gb * GRP + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[j], bv[u], acc[gb * GRP + u], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }

  panel_epilogue<NBLK, EPI>(acc, g, smem, m0, tid);
}

// ---- bf16-operand main loop (conf.amp; dm_mlp_params.precision = 1).  Same panels, same epilogues; A and B (weights
// [n][k], the forward layout - the backward caller hands a transposed copy) are loaded as fp32 with the register prefetch
// above, rounded to bf16 (RNE) on their way into LDS and multiplied on v_mfma_f32_16x16x32_bf16 (lane l: row l&15,
// k = 8*(l>>4) .. +7; one ds_read_b128 per fragment) with fp32 accumulation.  25 MFMAs of ~20 cycles replace the 200 fp32
// MFMAs of 32 cycles per 32-k tile, so the loop is no longer matrix-pipe bound: LDS is double-buffered (one barrier per
// tile; 2 x 37 KB, two workgroups per CU) and the kernel runs at the rate its fp32 operands stream in.
constexpr int PANEL_LDH = 40;         // bf16 row stride of the [row][k] images (80 bytes: 16-byte aligned fragments)
// BH: the weights come as a bf16 copy (dm_panel_bf16_weights_launch: one conversion per call instead of one per panel):
// a B tile is 6.25 16-byte loads per thread that go to LDS as they are, instead of 12.5 float4 loads + conversions.
template <int NBLK, int EPI, bool AVEC, bool BH>
__global__ void __launch_bounds__(256, 2) panel_linear_bf16_kernel(const PanelArgs g) {
  typedef PanelB<NBLK, 0> PB;
  constexpr int BH_TOTAL = NBLK * 16 * 4;                   // 16-byte groups per tile: 400 rows x 4
  constexpr int BH_N = (BH_TOTAL + 255) / 256;
  constexpr int A_H = PANEL_BM * PANEL_LDH, B_H = NBLK * 16 * PANEL_LDH;            // bf16 elements per buffer
  constexpr int MAIN_BYTES = 2 * (A_H + B_H) * 2;
  constexpr int RED_BYTES = (EPI == PANEL_EPI_LN_BWD) ? 4 * 3 * NBLK * 16 * 4 : 0;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[MAIN_BYTES > RED_BYTES ? MAIN_BYTES : RED_BYTES];
  unsigned short* const hbase = reinterpret_cast<unsigned short*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * PANEL_BM;

  f32x4 acc[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 ra[2], rb[BH ? 1 : PB::NF4];
  uint4 rh[BH ? BH_N : 1];
  unsigned ma = 0u, mb = 0u;
  const int nkt = (g.K + 31) / 32;
  auto load_b = [&](int k0) {
    if constexpr (BH) {
      mb = 0u;
#pragma unroll
      for (int i = 0; i < BH_N; ++i) {
        const int f = tid + i * 256;
        const int n = f >> 2, k = k0 + ((f & 3) << 3);
        const bool ok = f < BH_TOTAL && k < g.K;            // K % 8 == 0 (host-checked): a group is all in or all out
        rh[i] = *reinterpret_cast<const uint4*>(g.Bh + (ok ? (size_t)n * g.ldb + k : 0));
        mb |= ok ? (1u << i) : 0u;
      }
    } else {
      PB::load(rb, mb, g, k0, tid);
    }
  };
  auto stash = [&](int buf) {         // registers -> bf16 LDS images of buffer `buf`
    unsigned short* Ah = hbase + buf * (A_H + B_H);
    unsigned short* Bh = Ah + A_H;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + i * 256;
      float4 v;
      v.x = (ma >> (4 * i + 0)) & 1u ? ra[i].x : 0.f;
      v.y = (ma >> (4 * i + 1)) & 1u ? ra[i].y : 0.f;
      v.z = (ma >> (4 * i + 2)) & 1u ? ra[i].z : 0.f;
      v.w = (ma >> (4 * i + 3)) & 1u ? ra[i].w : 0.f;
      *reinterpret_cast<uint2*>(&Ah[(f >> 3) * PANEL_LDH + ((f & 7) << 2)]) = dm_pack_bf16x4(v);
    }
    if constexpr (BH) {
#pragma unroll
      for (int i = 0; i < BH_N; ++i) {
        const int f = tid + i * 256;
        if (f < BH_TOTAL)
          *reinterpret_cast<uint4*>(&Bh[(f >> 2) * PANEL_LDH + ((f & 3) << 3)]) = (mb >> i) & 1u ? rh[i] : make_uint4(0u, 0u, 0u, 0u);
      }
    } else {
#pragma unroll
      for (int i = 0; i < PB::NF4; ++i) {
        const int f = tid + i * 256;
        if (f < PB::TOTAL) {
          const float4 v = (mb >> i) & 1u ? rb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<uint2*>(&Bh[(f >> 3) * PANEL_LDH + ((f & 7) << 2)]) = dm_pack_bf16x4(v);
        }
      }
    }
  };
  panel_load_a<AVEC>(ra, ma, g, m0, 0, tid);
  load_b(0);
  stash(0);
  if (nkt > 1) {
    panel_load_a<AVEC>(ra, ma, g, m0, 32, tid);
    load_b(32);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) {
      stash(cur ^ 1);                                   // tile kt+1 (its loads had the whole previous iteration to land)
      if (kt + 2 < nkt) {
        panel_load_a<AVEC>(ra, ma, g, m0, (kt + 2) * 32, tid);
        load_b((kt + 2) * 32);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned short* Ah = hbase + cur * (A_H + B_H);
    const unsigned short* Bh = Ah + A_H;
    const bf16x8 a8 = *reinterpret_cast<const bf16x8*>(&Ah[(wave * 16 + l15) * PANEL_LDH + 8 * q]);
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      const bf16x8 b8 = *reinterpret_cast<const bf16x8*>(&Bh[(b * 16 + l15) * PANEL_LDH + 8 * q]);
      acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[b], 0, 0, 0);
    }
    __syncthreads();
  }
  panel_epilogue<NBLK, EPI>(acc, g, reinterpret_cast<float*>(smem_raw), m0, tid);
}

// bf16 copies of weight matrices for the bf16-operand panels: dst[n][k] = bf16(W[n][k]) (TR = 0) or bf16(W[k][n]) (TR = 1: the
// backward's transposed weights in the same pass).  One launch per MLP call, all layers.
struct PanelCvtArgs {
  const float* w[DM_MAX_MLP_LAYERS + 1];
  unsigned short* dst[DM_MAX_MLP_LAYERS + 1];
  int rows[DM_MAX_MLP_LAYERS + 1], cols[DM_MAX_MLP_LAYERS + 1];      // destination shape (rows x cols, cols contiguous)
  unsigned first[DM_MAX_MLP_LAYERS + 2];
  int count, transpose;
};
__global__ void __launch_bounds__(256) panel_cvt_bf16_kernel(const PanelCvtArgs a) {
  const unsigned total = a.first[a.count];
  for (unsigned e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    int l = 0;
    while (l + 1 < a.count && e >= a.first[l + 1]) ++l;
    const unsigned r = e - a.first[l];
    const int cols = a.cols[l], rows = a.rows[l];
    const int n = r / cols, k = r % cols;
    const float v = a.transpose ? a.w[l][(size_t)k * rows + n] : a.w[l][(size_t)n * cols + k];
    a.dst[l][r] = (unsigned short)dm_f2bf(v);
  }
}
int dm_panel_bf16_weights_launch(int count, const float* const* w, unsigned short* const* dst, const int* rows, const int* cols,
                                 int transpose, hipStream_t st) {
  DM_REQUIRE(count >= 1 && count <= DM_MAX_MLP_LAYERS + 1, DM_E_SHAPE, "panel_bf16_weights: count %d", count);
  PanelCvtArgs a = {};
  unsigned first = 0;
  for (int i = 0; i < count; ++i) {
    a.w[i] = w[i]; a.dst[i] = dst[i]; a.rows[i] = rows[i]; a.cols[i] = cols[i]; a.first[i] = first;
    first += (unsigned)rows[i] * (unsigned)cols[i];
  }
  a.first[count] = first; a.count = count; a.transpose = transpose;
  int blocks = dm_cdiv(first, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(panel_cvt_bf16_kernel, dim3(blocks), dim3(256), 0, st, a);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// out_t[col] = sum_p colpart_t[p][col] for up to 3*DM_MAX_MLP_LAYERS column vectors in ONE launch (fixed order).
struct PanelFinalArgs {
  const float* part[3 * DM_MAX_MLP_LAYERS];   // first panel's vector; panels are `pstride` floats apart
  float* out[3 * DM_MAX_MLP_LAYERS];
  int count, n, npanels, pstride;
};
__global__ void __launch_bounds__(1024) panel_colsum_final_kernel(const PanelFinalArgs a) {
  __shared__ float red[16][64];
  const int cx = threadIdx.x & 63, cy = threadIdx.x >> 6;      // 64 columns x 16 panel lanes, 4 loads in flight per lane
  const int col = blockIdx.x * 64 + cx;
  const float* src = a.part[blockIdx.y] + col;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < a.n) {
    int p = cy;
    for (; p + 48 < a.npanels; p += 64) {
      s0 += src[(size_t)p * a.pstride];
      s1 += src[(size_t)(p + 16) * a.pstride];
      s2 += src[(size_t)(p + 32) * a.pstride];
      s3 += src[(size_t)(p + 48) * a.pstride];
    }
    for (; p < a.npanels; p += 16) s0 += src[(size_t)p * a.pstride];
  }
  red[cy][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (cy == 0 && col < a.n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cx];
    a.out[blockIdx.y][col] = t;
  }
}

// (Round 5 built and removed a 32 x 32-block version of these panels on the LDS-DMA operand pipeline of gemm_dma_kernel - a wave
//  owns 32 rows x 416 columns as 13 blocks of v_mfma_f32_32x32x2_f32, a workgroup 128 complete rows, half the LDS fragment traffic
//  per MAC, same epilogue arithmetic on the 32 x 32 accumulator layout, parity-green on every test_mlp_head_* case.  It is SLOWER:
//  forward 54 vs 67 TF/s, backward 22 vs 50, step 36.4 vs 34.1 ms (profiles/r05_panel32.txt).  The weight tile (416 x 32 floats
//  = 53 KB per stage) forces one 4-wave workgroup per CU, and 40 000 rows are 313 such workgroups on 256 CUs: two rounds at 61 %.
//  The 64-row panels below are 625 workgroups on 512 resident slots.  What bounds these kernels is that quantisation
//  (DESIGN 7), not the fragment traffic.)
// ---------------------------------------------------------------- host side ---------------------
static const int g_panel_min_rows = getenv("DM_PANEL_MIN_ROWS") ? atoi(getenv("DM_PANEL_MIN_ROWS")) : 16384;

// The panel path serves hidden = 400 (the only width pydreamer's heads use: a2c.py:16, decoders.py:259,289).
bool dm_panel_ok(int rows, int hidden) { return hidden == 400 && rows >= g_panel_min_rows; }
int dm_panel_count(int rows) { return dm_cdiv(rows, PANEL_BM); }

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// y = ELU(LN(x W^T + b)) for rows x hidden; xpre / stats / y / fused output layer optional.
// Wh (optional): bf16 copy of W (dm_panel_bf16_weights_launch) for the bf16-operand kernel
int dm_panel_ln_fwd_launch(int rows, int hidden, int kin, const float* x, int ldx, const float* W, const float* b,
                           const float* gamma, const float* beta, float eps, float* xpre, float* stats, float* y,
                           const float* wout, const float* bout, float* out, int out_dim, int ldout, hipStream_t st,
                           const unsigned short* Wh, int ldw, const float* addm) {
  DM_REQUIRE(hidden == 400, DM_E_SHAPE, "panel_ln_fwd: hidden %d (built for 400)", hidden);
  if (ldw <= 0) ldw = kin;
  DM_REQUIRE((kin & 3) == 0 && (ldw & 3) == 0 && ldw >= kin && al16(W), DM_E_SHAPE,
             "panel_ln_fwd: weight rows must be 16-byte aligned (kin %d, ldw %d)", kin, ldw);
  if (rows <= 0) return DM_OK;
  PanelArgs a = {};
  a.A = x; a.lda = ldx; a.B = W; a.ldb = ldw; a.bias = b; a.addm = addm;
  a.M = rows; a.N = hidden; a.K = kin;
  a.gamma = gamma; a.beta = beta; a.eps = eps;
  a.xpre = xpre; a.stats = stats; a.y = y; a.ldy = hidden;
  a.wout = wout; a.bout = bout; a.out = out; a.out_dim = out_dim; a.ldout = ldout;
  const dim3 grid((unsigned)dm_panel_count(rows)), blk(256);
  const int slot = dm_prof_slot_begin(20, 2.0 * rows * hidden * ((double)kin + (wout ? out_dim : 0)),
                                      4.0 * ((double)rows * kin + (double)hidden * kin + (double)rows * hidden * (xpre ? 2 : 1)), st);
  const bool avec = (ldx & 3) == 0 && al16(x);
  if (dm_cur_precision()) {
    a.Bh = Wh;
    if (Wh && avec && (kin & 7) == 0 && (ldw & 7) == 0 && al16(Wh)) hipLaunchKernelGGL((panel_linear_bf16_kernel<25, PANEL_EPI_LN_FWD, true, true>), grid, blk, 0, st, a);
    else if (avec) hipLaunchKernelGGL((panel_linear_bf16_kernel<25, PANEL_EPI_LN_FWD, true, false>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((panel_linear_bf16_kernel<25, PANEL_EPI_LN_FWD, false, false>), grid, blk, 0, st, a);
  } else if (avec) hipLaunchKernelGGL((panel_linear_kernel<25, 0, PANEL_EPI_LN_FWD, true>), grid, blk, 0, st, a);
  else hipLaunchKernelGGL((panel_linear_kernel<25, 0, PANEL_EPI_LN_FWD, false>), grid, blk, 0, st, a);
  dm_prof_slot_end(slot, st);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// dx = LN/ELU backward of dy = dup W (dup: rows x kup, W: kup x hidden row-major), column partials into colpart
// (dm_panel_count(rows) x 3 x hidden floats: [dbias | dgamma | dbeta] per panel).
// wt (optional, hidden x kup floats of scratch): with bf16 operands (dm_cur_precision()) and kup a multiple of 32 the
// weights are transposed into it once and the product runs on the bf16 main loop, which wants them [n][k].
int dm_panel_ln_bwd_launch(int rows, int hidden, int kup, const float* dup, int lddup, const float* W, const float* xpre,
                           const float* stats, const float* gamma, const float* beta, float* dx, float* colpart, float* wt,
                           hipStream_t st, const unsigned short* Wth) {
  DM_REQUIRE(hidden == 400, DM_E_SHAPE, "panel_ln_bwd: hidden %d (built for 400)", hidden);
  DM_REQUIRE(al16(W), DM_E_SHAPE, "panel_ln_bwd: weight must be 16-byte aligned");
  if (rows <= 0) return DM_OK;
  PanelArgs a = {};
  a.A = dup; a.lda = lddup; a.B = W; a.ldb = hidden;
  a.M = rows; a.N = hidden; a.K = kup;
  a.gamma = gamma; a.beta = beta;
  a.xin = xpre; a.stin = stats; a.dx = dx; a.lddx = hidden; a.colpart = colpart;
  const dim3 grid((unsigned)dm_panel_count(rows)), blk(256);
  const int slot = dm_prof_slot_begin(21, 2.0 * rows * hidden * (double)kup,
                                      4.0 * ((double)rows * kup + (double)hidden * kup + 2.0 * rows * hidden), st);
  const bool avec = (kup & 3) == 0 && (lddup & 3) == 0 && al16(dup);
  if (dm_cur_precision() && Wth && avec && kup >= 32 && (kup & 7) == 0 && al16(Wth)) {
    a.Bh = Wth; a.ldb = kup;          // bf16 copy of W^T (hidden x kup), made once per call by the caller
    hipLaunchKernelGGL((panel_linear_bf16_kernel<25, PANEL_EPI_LN_BWD, true, true>), grid, blk, 0, st, a);
  } else if (dm_cur_precision() && wt && avec && kup >= 32 && al16(wt)) {
    DM_TRY(dm_permute4_launch(W, wt, 1, 1, kup, hidden, 0, 1, 3, 2, st));       // W (kup x hidden) -> wt (hidden x kup)
    a.B = wt; a.ldb = kup;
    hipLaunchKernelGGL((panel_linear_bf16_kernel<25, PANEL_EPI_LN_BWD, true, false>), grid, blk, 0, st, a);
  } else if (avec)
    hipLaunchKernelGGL((panel_linear_kernel<25, 1, PANEL_EPI_LN_BWD, true>), grid, blk, 0, st, a);
  else
    hipLaunchKernelGGL((panel_linear_kernel<25, 1, PANEL_EPI_LN_BWD, false>), grid, blk, 0, st, a);
  dm_prof_slot_end(slot, st);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// out[t][0..n) = sum over panels of part[t][p*pstride + 0..n) for `count` vectors, one launch.
int dm_panel_colsum_final_launch(int count, const float* const* part, float* const* out, int n, int npanels, int pstride,
                                 hipStream_t st) {
  DM_REQUIRE(count >= 1 && count <= 3 * DM_MAX_MLP_LAYERS, DM_E_SHAPE, "panel_colsum_final: count %d", count);
  PanelFinalArgs a;
  for (int i = 0; i < count; ++i) { a.part[i] = part[i]; a.out[i] = out[i]; }
  a.count = count; a.n = n; a.npanels = npanels; a.pstride = pstride;
  hipLaunchKernelGGL(panel_colsum_final_kernel, dim3(dm_cdiv(n, 64), count), dim3(1024), 0, st, a);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
