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
  // LNA: A holds PRE-activations x; the product uses ELU(LayerNorm(x; ln_g, ln_b, eps)) (rssm.py:138-146: in_norm /
  // post_norm + ELU feeding the GRU / the posterior head).  Row statistics are computed by the consuming workgroup.
  const float* ln_g; const float* ln_b; float ln_eps;
  // SAMPLE: the strip is one categorical group of 32 logits; the epilogue draws the straight-through sample
  // (rssm.py:147-148) with the sampler contract of elementwise.hip and writes one-hot z, the next step's masked z and idx
  const float* u; const int32_t* forced; float* onehot; int ldo; int32_t* idx; float* z_next; const uint8_t* next_reset;
  // MODE 2 (LayerNorm+ELU BACKWARD prologue): A holds dy, the gradient w.r.t. y = ELU(LN(x)); the product uses
  //   dx = rstd * (g - mean_row(g) - xhat * mean_row(g xhat)),  g = dy * ELU'(pre) * gamma,  pre = xhat * gamma + beta
  // with x (pre-activations, leading dim ldx2) and the saved statistics (mean, rstd per row)
  const float* lnb_x; int lnb_ldx; const float* lnb_stats;
  // the same backward FOLDED and split over two products (common.h DmGemm::eg_x / lnf_ps): the producer's epilogue turns its
  // dy into g = dy ELU'(pre) gamma and leaves the strip's row sums of g and g xhat; the consumer is a PLAIN product on g whose
  // epilogue applies  rstd (P - mean(g) cs - mean(g xhat) rstd (x B^T - mean cs))
  const float* eg_x; int eg_ldx; const float* eg_stats; const float* eg_gamma; const float* eg_beta;
  float* eg_G; int eg_ldg; float* eg_Gf; float* eg_ps;
  const float* lnf_ps; int lnf_nps; const float* lnf_stats; const float* lnf_xw; int lnf_ldxw; const float* lnf_cs;
  // straight-through softmax backward on the completed strip (32-wide strips: one categorical group), see DmGemm::sm_logits
  const float* sm_logits; int sm_ld; float* sm_dlogits; int sm_ldd;
  // EPI 2 (GRU gates backward in the epilogue; the strip holds 16 hidden units of dh' = C): see skinny_gates_bwd
  const float* gb_gi; const float* gb_gh; const float* gb_hin; int gb_ldh, gb_D;
  float* gb_dgi; float* gb_dgh; float* gb_dprev; int gb_ldp; const uint8_t* gb_rz;
  float* gb_dgif; float* gb_dghf;
  // fragment-major operand copies (common.h dm_frag_off; single 64-row chunk only): Af mirrors A, Cf receives C, znf z_next
  const float* Af; float* Cf; float* znf;
};

// Loads that bypass the (per-CU, non-coherent) L1: what a PERSISTENT kernel uses to read what another workgroup of the same
// launch wrote (agent-scope relaxed loads; the compiler tracks their vmcnt like any other load).  See rssm_persist_kernel.
__device__ __forceinline__ float sk_ld1_coh(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int sk_ldi_coh(const int32_t* p) {
  return (int)__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 sk_ld4_coh(const float* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b),
                     __uint_as_float((unsigned)(b >> 32)));
}

// Raw loads from clamped (always valid) addresses + a validity mask (bits 0-3: A row blocks, bit 4: B).  The zero-fill
// select is applied where the chunk is CONSUMED, one iteration later, so the loads stay in flight under the MFMAs.
template <int BL, int NRB, int NCB, bool COH = false>
__device__ __forceinline__ void skinny_load(const SkinnyArgs& g, int m0, int n0, int c, int kend, int lane,
                                            float4 (&a)[NRB], float4 (&b)[NCB], unsigned& mask) {
  mask = 0u;
  const int l15 = lane & 15, q = lane >> 4;
  const int k = c + 4 * q;
  const bool kin = k < kend;                      // kend % 4 == 0 (host-checked K % 4): a group of 4 is all in or all out
  const int nch = (g.K + 15) >> 4;                // fragment-major copy: a chunk past the range re-reads the last one (masked)
  const int cc = (c >> 4) < nch ? (c >> 4) : nch - 1;
#pragma unroll
  for (int mb = 0; mb < NRB; ++mb) {
    const int m = m0 + mb * 16 + l15;
    const bool ok = kin && m < g.M;
    const float* ap = g.Af ? g.Af + ((((size_t)cc * 4 + (m0 >> 4) + mb) * 4 + q) * 16 + l15) * 4      // one contiguous KiB per instruction (rows past M hold stale data: masked like the clamped loads)
                           : g.A + (ok ? (size_t)m * g.lda + k : 0);
    a[mb] = COH ? sk_ld4_coh(ap) : *reinterpret_cast<const float4*>(ap);
    mask |= ok ? (1u << mb) : 0u;
  }
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int n = n0 + cb * 16 + l15;
    const bool okb = kin && n < g.N;
    if (BL == 0) {
      b[cb] = *reinterpret_cast<const float4*>(g.B + (okb ? (size_t)n * g.ldb + k : 0));
    } else {
      const float* p = g.B + (okb ? (size_t)k * g.ldb + n : 0);
      const size_t st = okb ? (size_t)g.ldb : 0;
      b[cb] = make_float4(p[0], p[st], p[2 * st], p[3 * st]);
    }
    mask |= okb ? (16u << cb) : 0u;
  }
}

constexpr int SK_WAVES = 16;   // waves per workgroup = K splits inside it (8 and 4 measured slower, also under contention)
constexpr int SK_LN_MAXK = 1024;

// ELU for the LayerNorm prologue: exp(v) - 1 on the hardware exp (absolute error ~1e-7 near 0, where expm1f's relative
// accuracy buys nothing for an O(1) activation); every workgroup of the launch redoes this transform for its 64 x K
// operand, so it has to be cheap.
__device__ __forceinline__ float skinny_elu(float v) { return v > 0.f ? v : __expf(v) - 1.0f; }

struct __attribute__((aligned(16))) SkinnyShared {
  float lnstat[64][4];      // mean, rstd (+ MODE 2: mean(g), mean(g xhat))
  float lng[SK_LN_MAXK];
  float lnb[SK_LN_MAXK];
  float psum[8][64][2];      // folded LayerNorm backward, consumer side: eight interleaved sub-sums of the producer's strip sums
};

// One (16*NRB) x (16*NCB) output strip; NRB = populated 16-row blocks (M <= 16*NRB): a 7-row data-parallel shard reads
// a quarter of the activation traffic of the 50-row case.  DEPTH = 16-k chunks loaded per round (all in flight before
// the first MFMA).
// COH: the A operand (and the rows the LayerNorm prologue reads) come through L1-bypassing loads (persistent kernel).
// reuse_prologue: sh already holds this product's LayerNorm statistics and parameters (an earlier strip of the same
// workgroup computed them).
template <int BL, int NRB, int NCB, int MODE, int EPI, int DEPTH, bool COH = false>
__device__ __forceinline__ void skinny_strip(const SkinnyArgs& g, int strip, int m0, float* part, SkinnyShared* sh,
                                             float* tile, bool reuse_prologue = false) {
  constexpr bool LNA = MODE == 1, LNB = MODE == 2, SAMPLE = EPI == 1, GATESB = EPI == 2;
  constexpr int PW = 16 * NCB;                    // strip width
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = strip * PW;
  const int chunks = (g.K + 15) / 16;
  const int per = (chunks + SK_WAVES - 1) / SK_WAVES;
  const int c_beg = wave * per * 16;
  int c_end = c_beg + per * 16;
  if (c_end > chunks * 16) c_end = chunks * 16;

  f32x4 acc[NRB][NCB];
#pragma unroll
  for (int mb = 0; mb < NRB; ++mb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[mb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (g.lnf_ps) {
    // thread (part, which, row) adds every 8th strip's sum: one round of independent loads per wave, combined (fixed order) in
    // the epilogue behind the barrier that already separates it from the main loop
    const int r = tid & 63, wq = (tid >> 6) & 1, pt = tid >> 7;
    float sacc = 0.f;
    for (int i = pt; i < g.lnf_nps; i += 8) sacc += g.lnf_ps[((size_t)i * 64 + r) * 2 + wq];
    sh->psum[pt][r][wq] = sacc;
  }
  if (LNA && !reuse_prologue) {
    // row statistics of this strip's A rows (two-pass, the row cached in registers: K <= 1024), gamma / beta staged in LDS
    for (int e = tid; e < g.K; e += SK_WAVES * 64) { sh->lng[e] = g.ln_g[e]; sh->lnb[e] = g.ln_b[e]; }
    for (int rr = wave; rr < 16 * NRB; rr += SK_WAVES) {
      const int row = m0 + rr;
      const float* xr = g.A + (size_t)(row < g.M ? row : 0) * g.lda;
      float xc[SK_LN_MAXK / 64];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < SK_LN_MAXK / 64; ++j) {
        const int cidx = lane + 64 * j;
        xc[j] = cidx < g.K ? (COH ? sk_ld1_coh(xr + cidx) : xr[cidx]) : 0.f;
        s += xc[j];
      }
      const float mean = dm_wave_sum(s) / (float)g.K;
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < SK_LN_MAXK / 64; ++j) {
        const float d = (lane + 64 * j < g.K) ? xc[j] - mean : 0.f;
        v += d * d;
      }
      const float rstd = 1.0f / sqrtf(dm_wave_sum(v) / (float)g.K + g.ln_eps);
      if (lane == 0) { sh->lnstat[rr][0] = mean; sh->lnstat[rr][1] = rstd; }
    }
    __syncthreads();
  }
  if (LNB) {
    // row terms of the LayerNorm backward: mean / rstd from the forward's statistics, mean(g) and mean(g xhat) over the row
    for (int e = tid; e < g.K; e += SK_WAVES * 64) { sh->lng[e] = g.ln_g[e]; sh->lnb[e] = g.ln_b[e]; }
    // A wave's rows (wave, wave + 16, ...; NRB of them) are reduced TOGETHER: the loads of all of them for a column batch
    // are in flight at once and gamma / beta are fetched once per column (one row at a time this prologue was a chain of
    // ~16 dependent load rounds per workgroup: 49 -> 3x us on the BPTT products)
    {
      const float* dr[NRB]; const float* xr[NRB];
      float mean[NRB], rstd[NRB], sg[NRB], sgx[NRB];
#pragma unroll
      for (int i = 0; i < NRB; ++i) {
        const int row = m0 + wave + SK_WAVES * i;
        const size_t rc = row < g.M ? (size_t)row : 0;
        dr[i] = g.A + rc * g.lda;
        xr[i] = g.lnb_x + rc * g.lnb_ldx;
        mean[i] = g.lnb_stats[2 * rc]; rstd[i] = g.lnb_stats[2 * rc + 1];
        sg[i] = 0.f; sgx[i] = 0.f;
      }
#pragma unroll 4
      for (int j = 0; j < SK_LN_MAXK / 64; ++j) {
        const int cidx = lane + 64 * j;
        const bool cin = cidx < g.K;
        const int cc = cin ? cidx : 0;
        const float ga = g.ln_g[cc], be = g.ln_b[cc];
        float dv[NRB], xv[NRB];
#pragma unroll
        for (int i = 0; i < NRB; ++i) { dv[i] = dr[i][cc]; xv[i] = xr[i][cc]; }
#pragma unroll
        for (int i = 0; i < NRB; ++i) {
          const float xh = (xv[i] - mean[i]) * rstd[i];
          const float pre = xh * ga + be;
          const float gg = cin ? dv[i] * (pre > 0.f ? 1.f : __expf(pre)) * ga : 0.f;
          sg[i] += gg;
          sgx[i] += gg * xh;
        }
      }
#pragma unroll
      for (int i = 0; i < NRB; ++i) {
        const float a0 = dm_wave_sum(sg[i]) / (float)g.K, a1 = dm_wave_sum(sgx[i]) / (float)g.K;
        if (lane == 0) {
          const int rr = wave + SK_WAVES * i;
          sh->lnstat[rr][0] = mean[i]; sh->lnstat[rr][1] = rstd[i]; sh->lnstat[rr][2] = a0; sh->lnstat[rr][3] = a1;
        }
      }
    }
    __syncthreads();
  }

  // These products are latency bound (weights come from the MALL / HBM, ~1 us a round trip), so a round issues the
  // loads of DEPTH chunks back to back and only then starts consuming them under counted waits; the 16 waves of
  // the workgroup (4 per SIMD) overlap each other's rounds.  With K <= 1024 a wave's whole K range is ONE round.
  // A chunk past the wave's range loads clamped addresses under an all-zero mask and contributes exact zeros.
  int kend = c_end < g.K ? c_end : g.K;
  if (kend < c_beg) kend = c_beg;
  for (int c = c_beg; c < c_end; c += 16 * DEPTH) {
    float4 a[DEPTH][NRB], b[DEPTH][NCB], ax[LNB ? DEPTH : 1][NRB];
    unsigned mk[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      skinny_load<BL, NRB, NCB, COH>(g, m0, n0, c + 16 * d, kend, lane, a[d], b[d], mk[d]);
      if (LNB) {          // the pre-activations x at the same (row, k) positions as dy
        const int k = c + 16 * d + 4 * (lane >> 4);
#pragma unroll
        for (int mb = 0; mb < NRB; ++mb) {
          const int m = m0 + mb * 16 + (lane & 15);
          const bool ok = k < kend && m < g.M;
          ax[d][mb] = *reinterpret_cast<const float4*>(g.lnb_x + (ok ? (size_t)m * g.lnb_ldx + k : 0));
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      float bj[NCB][4];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const bool okb = ((mk[d] >> (4 + cb)) & 1u) != 0u;
        bj[cb][0] = okb ? b[d][cb].x : 0.f; bj[cb][1] = okb ? b[d][cb].y : 0.f;
        bj[cb][2] = okb ? b[d][cb].z : 0.f; bj[cb][3] = okb ? b[d][cb].w : 0.f;
      }
      float gq[4] = {1.f, 1.f, 1.f, 1.f}, bq[4] = {0.f, 0.f, 0.f, 0.f};
      if (LNA || LNB) {
        const int k = c + 16 * d + 4 * (lane >> 4);
        const int kk = k < SK_LN_MAXK - 3 ? k : 0;          // past-the-range chunks are masked below; keep the read in bounds
        const float4 g4 = *reinterpret_cast<const float4*>(&sh->lng[kk]);
        const float4 b4 = *reinterpret_cast<const float4*>(&sh->lnb[kk]);
        gq[0] = g4.x; gq[1] = g4.y; gq[2] = g4.z; gq[3] = g4.w;
        bq[0] = b4.x; bq[1] = b4.y; bq[2] = b4.z; bq[3] = b4.w;
      }
#pragma unroll
      for (int mb = 0; mb < NRB; ++mb) {
        const bool oka = ((mk[d] >> mb) & 1u) != 0u;
        float aj[4] = {a[d][mb].x, a[d][mb].y, a[d][mb].z, a[d][mb].w};
        float lmean = 0.f, lrstd = 0.f, lc1 = 0.f, lc2 = 0.f;
        if (LNA || LNB) {   // row statistics straight from LDS (keeping them in registers spills the NRB = 4 instance)
          const float4 st4 = *reinterpret_cast<const float4*>(&sh->lnstat[mb * 16 + (lane & 15)][0]);
          lmean = st4.x; lrstd = st4.y; lc1 = st4.z; lc2 = st4.w;
        }
        float xj[4] = {0.f, 0.f, 0.f, 0.f};
        if (LNB) { const float4 x4 = ax[LNB ? d : 0][mb]; xj[0] = x4.x; xj[1] = x4.y; xj[2] = x4.z; xj[3] = x4.w; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (LNA) aj[j] = skinny_elu((aj[j] - lmean) * lrstd * gq[j] + bq[j]);
          if (LNB) {
            const float xh = (xj[j] - lmean) * lrstd;
            const float pre = xh * gq[j] + bq[j];
            const float gg = aj[j] * (pre > 0.f ? 1.f : __expf(pre)) * gq[j];
            aj[j] = lrstd * (gg - lc1 - xh * lc2);
          }
          aj[j] = oka ? aj[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
            acc[mb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[j], bj[cb][j], acc[mb][cb], 0, 0, 0);
      }
    }
  }
  // C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + r
  float* mypart = part + (size_t)wave * (64 * PW);
#pragma unroll
  for (int mb = 0; mb < NRB; ++mb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) mypart[(mb * 16 + (lane >> 4) * 4 + r) * PW + cb * 16 + (lane & 15)] = acc[mb][cb][r];
  __syncthreads();
  for (int e = tid; e < NRB * 16 * PW; e += SK_WAVES * 64) {
    const int lr = e / PW, lc = e % PW;
    const int row = m0 + lr, col = n0 + lc;
    float v = 0.f;
    const bool valid = row < g.M && col < g.N;
    if (valid) {
#pragma unroll
      for (int w = 0; w < SK_WAVES; ++w) v += part[(size_t)w * (64 * PW) + e];
      if (g.lnf_ps) {
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) { p0 += sh->psum[pt][row][0]; p1 += sh->psum[pt][row][1]; }      // (M <= 64: row = the row of the 64)
        const float mg = p0 / (float)g.K, mgx = p1 / (float)g.K;
        const float mean = g.lnf_stats[2 * (size_t)row], rstd = g.lnf_stats[2 * (size_t)row + 1];
        const float cs = g.lnf_cs[col];
        const float xhw = rstd * (g.lnf_xw[(size_t)row * g.lnf_ldxw + col] - mean * cs);
        v = rstd * (v - mg * cs - mgx * xhw);
      }
      if (g.row_zero && g.row_zero[row]) v = 0.f;
      if (g.bias) v += g.bias[col];
      if (g.add) v += g.add[(size_t)row * g.ldadd + col];
      float* cp = g.C + (size_t)row * g.ldc + col;
      if (g.flags & DM_GEMM_ACCUM) v += *cp;
      if (g.flags & DM_GEMM_ELU) v = dm_elu(v);
      *cp = v;
      if (g.Cf) g.Cf[dm_frag_off(row, col)] = v;
      if (GATESB) {
        // v = dh'[row][d] (complete): GRU gates backward for hidden unit d = col (rnn.py:48-49 / nn.GRUCell, gates recomputed
        // from the saved products), the same arithmetic as gru_gates_bwd_kernel (elementwise.hip)
        const int D = g.gb_D;
        const size_t g0 = (size_t)row * 3 * D;
        const float ghn = g.gb_gh[g0 + 2 * D + col];
        const float rg = 1.0f / (1.0f + expf(-(g.gb_gi[g0 + col] + g.gb_gh[g0 + col])));
        const float ug = 1.0f / (1.0f + expf(-(g.gb_gi[g0 + D + col] + g.gb_gh[g0 + D + col])));
        const float ng = tanhf(g.gb_gi[g0 + 2 * D + col] + rg * ghn);
        const float h = g.gb_hin[(size_t)row * g.gb_ldh + col];
        const float dn = v * (1.f - ug);
        const float du = v * (h - ng);
        const float dpn = dn * (1.f - ng * ng);
        const float dpr = dpn * ghn * rg * (1.f - rg);
        const float dpu = du * ug * (1.f - ug);
        g.gb_dgi[g0 + col] = dpr; g.gb_dgi[g0 + D + col] = dpu; g.gb_dgi[g0 + 2 * D + col] = dpn;
        g.gb_dgh[g0 + col] = dpr; g.gb_dgh[g0 + D + col] = dpu; g.gb_dgh[g0 + 2 * D + col] = dpn * rg;
        if (g.gb_dgif) {
          g.gb_dgif[dm_frag_off(row, col)] = dpr; g.gb_dgif[dm_frag_off(row, D + col)] = dpu;
          g.gb_dgif[dm_frag_off(row, 2 * D + col)] = dpn;
          g.gb_dghf[dm_frag_off(row, col)] = dpr; g.gb_dghf[dm_frag_off(row, D + col)] = dpu;
          g.gb_dghf[dm_frag_off(row, 2 * D + col)] = dpn * rg;
        }
        if (g.gb_dprev) {
          const float dv = (g.gb_rz && g.gb_rz[row]) ? 0.f : v * ug;
          g.gb_dprev[(size_t)row * g.gb_ldp + col] += dv;
        }
      }
    }
    if (PW == 16 && g.eg_x) {      // (workgroup-uniform) v = dy[row][col], complete
      float gg = 0.f, gx = 0.f;
      if (valid) {
        const float mean = g.eg_stats[2 * (size_t)row], rstd = g.eg_stats[2 * (size_t)row + 1];
        const float ga = g.eg_gamma[col];
        const float xh = (g.eg_x[(size_t)row * g.eg_ldx + col] - mean) * rstd;
        const float pre = xh * ga + g.eg_beta[col];
        gg = v * (pre > 0.f ? 1.f : __expf(pre)) * ga;
        gx = gg * xh;
        if (g.eg_G) g.eg_G[(size_t)row * g.eg_ldg + col] = gg;
        if (g.eg_Gf) g.eg_Gf[dm_frag_off(row, col)] = gg;
      }
      // the 16 columns of a row sit in 16 consecutive lanes (e = lr * 16 + lc): the strip's row sums, in a fixed order
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { gg += __shfl_xor(gg, o); gx += __shfl_xor(gx, o); }
      if (lc == 0 && row < 64) { g.eg_ps[((size_t)strip * 64 + row) * 2] = gg; g.eg_ps[((size_t)strip * 64 + row) * 2 + 1] = gx; }
    }
    if (PW == 32 && g.sm_logits) {      // (workgroup-uniform) v = dz'[row][group `strip`], complete: same operation order as st_softmax_bwd_kernel<32>
      const float x = valid ? g.sm_logits[(size_t)row * g.sm_ld + col] : -INFINITY;
      const float gz = valid ? v : 0.f;
      float mx = x;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      const float ex = valid ? expf(x - mx) : 0.f;
      float se = ex, sg = ex * gz;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) { se += __shfl_xor(se, o, 64); sg += __shfl_xor(sg, o, 64); }
      if (valid) {
        const float pr = ex / se;
        float* o = g.sm_dlogits + (size_t)row * g.sm_ldd + col;
        *o = *o + pr * (gz - sg / se);
      }
    }
    if (SAMPLE) tile[lr * 33 + lc] = v;
  }
  if (SAMPLE) {
    // One lane per row: the strip IS the row's categorical group `strip` (32 logits).  Same operation order as
    // sample_onehot_lane32_kernel (elementwise.hip), so the drawn indices are bit-identical to the stand-alone sampler.
    __syncthreads();
    const int row = m0 + tid;
    if (tid < 16 * NRB && row < g.M) {
      constexpr int C = 32;
      const int groups = g.N / C;
      const size_t i = (size_t)row * groups + strip;
      int idx;
      if (g.forced) {
        idx = g.forced[i];
      } else {
        float x[C];
#pragma unroll
        for (int k = 0; k < C; ++k) x[k] = tile[tid * 33 + k];
        float mx = x[0];
#pragma unroll
        for (int k = 1; k < C; ++k) mx = fmaxf(mx, x[k]);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < C; ++k) { x[k] = expf(x[k] - mx); sum += x[k]; }
        float total_p = 0.f;
#pragma unroll
        for (int k = 0; k < C; ++k) { x[k] = x[k] / sum; total_p += x[k]; }
        const float target = g.u[i] * total_p;
        float cdf = 0.f;
        idx = 0;
#pragma unroll
        for (int k = 0; k < C; ++k) {
          cdf += x[k];
          idx += (cdf <= target) ? 1 : 0;
        }
        if (idx > C - 1) idx = C - 1;
      }
      float4* dst = reinterpret_cast<float4*>(g.onehot + (size_t)row * g.ldo + (size_t)strip * C);
#pragma unroll
      for (int q4 = 0; q4 < C / 4; ++q4)
        dst[q4] = make_float4(idx == 4 * q4 ? 1.f : 0.f, idx == 4 * q4 + 1 ? 1.f : 0.f, idx == 4 * q4 + 2 ? 1.f : 0.f,
                              idx == 4 * q4 + 3 ? 1.f : 0.f);
      if (g.z_next) {
        const int keep = (g.next_reset && g.next_reset[row]) ? -1 : idx;
        float4* dn = reinterpret_cast<float4*>(g.z_next + ((size_t)row * groups + strip) * C);
#pragma unroll
        for (int q4 = 0; q4 < C / 4; ++q4)
          dn[q4] = make_float4(keep == 4 * q4 ? 1.f : 0.f, keep == 4 * q4 + 1 ? 1.f : 0.f, keep == 4 * q4 + 2 ? 1.f : 0.f,
                               keep == 4 * q4 + 3 ? 1.f : 0.f);
        if (g.znf) {
#pragma unroll
          for (int q4 = 0; q4 < C / 4; ++q4)
            *reinterpret_cast<float4*>(g.znf + dm_frag_off(row, strip * C + 4 * q4)) =
                make_float4(keep == 4 * q4 ? 1.f : 0.f, keep == 4 * q4 + 1 ? 1.f : 0.f, keep == 4 * q4 + 2 ? 1.f : 0.f,
                            keep == 4 * q4 + 3 ? 1.f : 0.f);
        }
      }
      if (g.idx) g.idx[i] = idx;
    }
  }
}

// DM_SKINNY_LDS_PAD (floats, compile-time experiment of round 6): LDS claimed beyond what the strip needs, so that no workgroup of
// the LDS-DMA tile kernels (48 KB each) fits on a CU beside a strip - does a chain kernel run faster with the CU to itself?
#ifndef DM_SKINNY_LDS_PAD
#define DM_SKINNY_LDS_PAD 0
#endif
template <int BL, int NRB, int MODE, int EPI = 0>
__global__ void __launch_bounds__(SK_WAVES * 64) skinny_gemm_kernel(const SkinnyArgs g) {
  __shared__ float part[SK_WAVES * 64 * 16 + DM_SKINNY_LDS_PAD];
  __shared__ SkinnyShared sh;
  DM_CHAIN_PRIO();
  skinny_strip<BL, NRB, 1, MODE, EPI, MODE == 2 ? 2 : 4>(g, blockIdx.x, blockIdx.y * (16 * NRB), part, &sh, nullptr);   // grid.y = (16 NRB)-row chunks of M
}
// Posterior / prior head with the sampler in the epilogue: LayerNorm+ELU prologue, 32-wide strips (one categorical group
// per workgroup), M <= 64.
template <int NRB>
__global__ void __launch_bounds__(SK_WAVES * 64) skinny_gemm_sample_kernel(const SkinnyArgs g) {
  __shared__ float part[SK_WAVES * 64 * 32];
  __shared__ SkinnyShared sh;
  __shared__ float tile[64 * 33];
  DM_CHAIN_PRIO();
  skinny_strip<0, NRB, 2, 1, 1, 2>(g, blockIdx.x, blockIdx.y * (16 * NRB), part, &sh, tile);      // grid.y = (16 NRB)-row chunks of M
}
// Two independent products in ONE launch (the loops are bound by the number of launches, ~5 us of GPU time and ~6 us of
// host time each): the GRU's input and hidden gate products of a posterior step, the two backward-data products that
// feed step t-1's state gradient.  Workgroups [0, nb0) serve the first product, the rest the second.
// LNA0: the FIRST product's A operand goes through the LayerNorm+ELU prologue (gi = ELU(in_norm(x)) W_ih^T).
struct SkinnyPairArgs { SkinnyArgs g[2]; int nb0, nb; };      // nb = strips of both products; blocks beyond it serve the next (16 NRB)-row chunk
// MODE0 / MODE1: prologue of the first / second product (0 none, 1 LayerNorm+ELU forward, 2 LayerNorm+ELU backward)
// NCB1 = 2: the second product in 32-wide strips (softmax-backward epilogue: a strip is one categorical group)
template <int NRB, int MODE0, int MODE1, int NCB1 = 1>
__global__ void __launch_bounds__(SK_WAVES * 64) skinny_gemm_pair_kernel(const SkinnyPairArgs a) {
  __shared__ float part[SK_WAVES * 64 * 16 * NCB1 + (NCB1 == 1 ? DM_SKINNY_LDS_PAD : 0)];
  __shared__ SkinnyShared sh;
  DM_CHAIN_PRIO();
  const int b = blockIdx.x % a.nb, m0 = (blockIdx.x / a.nb) * (16 * NRB);
  if (b < a.nb0) skinny_strip<0, NRB, 1, MODE0, 0, MODE0 == 2 ? 2 : 4>(a.g[0], b, m0, part, &sh, nullptr);
  else skinny_strip<0, NRB, NCB1, MODE1, 0, (MODE1 == 2 || NCB1 == 2) ? 2 : 4>(a.g[1], b - a.nb0, m0, part, &sh, nullptr);
}

// max_m: 64 for the pair kernel (one chunk); the single-product kernel walks M in 64-row chunks (grid.y) up to
// g_skinny_max_m rows, where it still beats the tiled kernel's <= ~100 workgroups (imagination products of a
// data-parallel shard: M = T*B/8 = 350)
static const int g_skinny_max_m = getenv("DM_SKINNY_MAX_M") ? atoi(getenv("DM_SKINNY_MAX_M")) : 512;
static bool skinny_ok(const DmGemm& q, int max_m) {
  if (q.M > max_m || q.M < 1 || q.N < 1 || q.K < 16) return false;
  if (q.a_layout != 0 || q.a_maj || q.b_maj || q.mulref || q.c_tab || q.bias_mod) return false;
  if ((q.K & 3) || (q.lda & 3) || ((uintptr_t)q.A & 15)) return false;
  if (q.b_layout == 0 && ((q.ldb & 3) || ((uintptr_t)q.B & 15))) return false;
  if ((int64_t)q.N * q.K < (int64_t)64 * 1024) return false;             // tiny products: one tiled workgroup is fine
  return true;
}
static void skinny_fill(const DmGemm& q, SkinnyArgs& a) {
  a.A = q.A; a.B = q.B; a.C = q.C; a.bias = q.bias; a.add = q.add; a.row_zero = q.row_zero;
  a.M = q.M; a.N = q.N; a.K = q.K; a.lda = q.lda; a.ldb = q.ldb; a.ldc = q.ldc; a.ldadd = q.ldadd; a.flags = q.flags;
  a.ln_g = q.ln_g; a.ln_b = q.ln_b; a.ln_eps = q.ln_eps;
  a.lnb_x = q.lnb_x; a.lnb_ldx = q.lnb_ldx; a.lnb_stats = q.lnb_stats;
  a.eg_x = q.eg_x; a.eg_ldx = q.eg_ldx; a.eg_stats = q.eg_stats; a.eg_gamma = q.eg_gamma; a.eg_beta = q.eg_beta;
  a.eg_G = q.eg_G; a.eg_ldg = q.eg_ldg; a.eg_Gf = q.M <= 64 ? q.eg_Gf : nullptr; a.eg_ps = q.eg_ps;
  a.sm_logits = q.sm_logits; a.sm_ld = q.sm_ld; a.sm_dlogits = q.sm_dlogits; a.sm_ldd = q.sm_ldd;
  a.lnf_ps = q.lnf_ps; a.lnf_nps = q.lnf_nps; a.lnf_stats = q.lnf_stats; a.lnf_xw = q.lnf_xw; a.lnf_ldxw = q.lnf_ldxw; a.lnf_cs = q.lnf_cs;
  a.gb_gi = nullptr; a.gb_gh = nullptr; a.gb_hin = nullptr; a.gb_ldh = 0; a.gb_D = 0; a.gb_dgi = nullptr; a.gb_dgh = nullptr;
  a.gb_dprev = nullptr; a.gb_ldp = 0; a.gb_rz = nullptr; a.gb_dgif = nullptr; a.gb_dghf = nullptr;
  if (q.gates) {
    a.gb_gi = q.gates->gi; a.gb_gh = q.gates->gh; a.gb_hin = q.gates->h_in; a.gb_ldh = q.gates->ldh; a.gb_D = q.gates->D;
    a.gb_dgi = q.gates->dgi; a.gb_dgh = q.gates->dgh; a.gb_dprev = q.gates->dprev; a.gb_ldp = q.gates->ldp; a.gb_rz = q.gates->row_zero;
    if (q.M <= 64 && q.gates->dgi_frag && q.gates->dgh_frag) { a.gb_dgif = q.gates->dgi_frag; a.gb_dghf = q.gates->dgh_frag; }
  }
  a.u = nullptr; a.forced = nullptr; a.onehot = nullptr; a.ldo = 0; a.idx = nullptr; a.z_next = nullptr; a.next_reset = nullptr;
  const bool one_chunk = q.M <= 64;       // the fragment-major layout holds ONE 64-row chunk
  a.Af = one_chunk ? q.A_frag : nullptr; a.Cf = one_chunk ? q.C_frag : nullptr; a.znf = nullptr;
}
static const int g_skinny_disabled = getenv("DM_GEMM_NO_SKINNY") ? 1 : 0;      // A/B switch for scripts/gemm_bench.py
static const int g_skinny_nofuse = getenv("DM_SKINNY_NO_FUSE") ? 1 : 0;        // A/B switch: keep LayerNorm / sampler launches
static const int g_skinny_msplit = getenv("DM_SKINNY_MSPLIT") ? atoi(getenv("DM_SKINNY_MSPLIT")) : 2;      // A/B switch: 0 one workgroup per strip, 1 32-row halves, 2 also 16-row quarters

// Can a <= 64-row product with reduction length K take the LayerNorm+ELU prologue (and, for N % 32 == 0, the sampler
// epilogue)?  Shape-only test: rssm.hip picks the fused or the unfused schedule of the T loop with it.
bool dm_skinny_ln_ok(int M, int N, int K) {
  return !g_skinny_disabled && !g_skinny_nofuse && M >= 1 && M <= 64 && K >= 16 && K <= SK_LN_MAXK && (K & 3) == 0 &&
         (int64_t)N * K >= (int64_t)64 * 1024;
}

// Returns 1 if the launch was taken by the skinny kernel, 0 if the shape / alignment does not qualify, < 0 on error.
int dm_gemm_skinny_try(const DmGemm& q, hipStream_t stream) {
  if (q.sm_logits) return dm_fail(DM_E_SHAPE, "skinny gemm: the softmax-backward epilogue exists in the pair launch only");
  const bool folded = q.eg_x || q.lnf_ps;      // folded LayerNorm backward (producer / consumer side): skinny-only epilogues
  if (folded) {
    if (g_skinny_disabled || !skinny_ok(q, 64) || q.b_layout != 0 || q.ln_g || q.lnb_x)
      return dm_fail(DM_E_SHAPE, "skinny gemm: the folded LayerNorm backward is built for plain <= 64-row skinny products (M=%d N=%d K=%d)", q.M, q.N, q.K);
    if (q.eg_x && (!q.eg_stats || !q.eg_gamma || !q.eg_beta || !q.eg_ps || (!q.eg_G && !q.eg_Gf)))
      return dm_fail(DM_E_SHAPE, "skinny gemm: folded LayerNorm backward, producer side: statistics, gamma, beta, an output for g and the strip sums are required");
    if (q.lnf_ps && (!q.lnf_stats || !q.lnf_xw || !q.lnf_cs || q.lnf_nps < 1))
      return dm_fail(DM_E_SHAPE, "skinny gemm: folded LayerNorm backward, consumer side: statistics, x B^T and the weights' k-sums are required");
  }
  if (g_skinny_disabled || !skinny_ok(q, g_skinny_max_m)) return 0;
  if (q.C_frag && q.M > 64) return 0;      // the tiled path reports the misuse
  // beyond one chunk it pays only for short reductions over small weight matrices (measured at M = 350: 1000x1024
  // 31.9 -> 24.8 us, 400x400 15.1 -> 8.7 us; 1800x1000 equal; 400x1624 18.2 -> 19.9 us)
  if (q.M > 64 && (q.K > 1024 || (int64_t)q.N * q.K > (int64_t)1100 * 1024)) return 0;
  const bool lnb = q.lnb_x != nullptr, ln = q.ln_g != nullptr && !lnb;
  if ((ln || lnb) && (q.K > SK_LN_MAXK || q.b_layout != 0 || !q.ln_b || !q.ln_g || q.M > 64)) return 0;
  if (lnb && (!q.lnb_stats || (q.lnb_ldx & 3) || ((uintptr_t)q.lnb_x & 15))) return 0;
  if (q.gates && ((!lnb && !q.lnf_ps) || q.N != q.gates->D || q.b_layout != 0)) return 0;
  SkinnyArgs a;
  skinny_fill(q, a);
  // 33..64 rows: two 32-row workgroups per strip instead of one 64-row one (half the matrix-pipe and operand work per
  // workgroup on the chain's critical path; the strip's weights are read twice, from L2) while both fit the chip
  const int nst = dm_cdiv(q.N, 16);
  const bool quarters = g_skinny_msplit >= 2 && q.M > 16 && q.M <= 64 && dm_cdiv(q.M, 16) * nst <= 256;
  const bool halves = !quarters && g_skinny_msplit && q.M > 32 && q.M <= 64 && 2 * nst <= 256;
  const dim3 grid((unsigned)nst, (unsigned)(quarters ? dm_cdiv(q.M, 16) : halves ? 2 : dm_cdiv(q.M, 64)));
  const dim3 blk(SK_WAVES * 64);
#define SK_LAUNCH(BL_, MODE_, EPI_)                                                                                     \
  do {                                                                                                                  \
    if (q.M <= 16 || quarters) hipLaunchKernelGGL((skinny_gemm_kernel<BL_, 1, MODE_, EPI_>), grid, blk, 0, stream, a);  \
    else if (q.M <= 32 || halves) hipLaunchKernelGGL((skinny_gemm_kernel<BL_, 2, MODE_, EPI_>), grid, blk, 0, stream, a); \
    else hipLaunchKernelGGL((skinny_gemm_kernel<BL_, 4, MODE_, EPI_>), grid, blk, 0, stream, a);                        \
  } while (0)
  if (!lnb && q.gates) SK_LAUNCH(0, 0, 2);      // (consumer side of the folded LayerNorm backward)
  else if (lnb && q.gates) SK_LAUNCH(0, 2, 2);
  else if (lnb) SK_LAUNCH(0, 2, 0);
  else if (ln) SK_LAUNCH(0, 1, 0);
  else if (q.b_layout == 0) SK_LAUNCH(0, 0, 0);
  else SK_LAUNCH(1, 0, 0);
#undef SK_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dm_fail(DM_E_HIP, "skinny gemm: %s", hipGetErrorString(e));
  return 1;
}
// Both products in one launch if both qualify (k-contiguous B); otherwise two ordinary launches.
// q0 may carry a LayerNorm+ELU prologue (ln_g / ln_b): then the one-launch path is mandatory (callers check
// dm_skinny_ln_ok first).
int dm_gemm_pair_launch(const DmGemm& q0, const DmGemm& q1, void* ws, size_t ws_bytes, hipStream_t stream) {
  const bool ln0 = q0.ln_g != nullptr && !q0.lnb_x, lnb1 = q1.lnb_x != nullptr;
  DM_REQUIRE(!q0.eg_x && !q1.eg_x && !q0.lnf_ps && (!q1.lnf_ps || (q1.lnf_stats && q1.lnf_xw && q1.lnf_cs && q1.lnf_nps >= 1 && !q1.lnb_x && !q1.ln_g)),
             DM_E_SHAPE, "gemm pair: the folded LayerNorm backward is built for the consumer side on the second product");
  DM_REQUIRE(!q0.lnb_x && !(q1.ln_g && !q1.lnb_x) && !q0.gates && !q1.gates, DM_E_SHAPE,
             "gemm pair: built for a forward LayerNorm prologue on the first product or a backward one on the second");
  const bool fused = ln0 || lnb1;
  if (!g_skinny_disabled && skinny_ok(q0, 64) && skinny_ok(q1, 64) && q0.b_layout == 0 && q1.b_layout == 0 &&
      (!ln0 || (q0.K <= SK_LN_MAXK && q0.ln_b)) &&
      (!lnb1 || (q1.K <= SK_LN_MAXK && q1.ln_g && q1.ln_b && q1.lnb_stats && (q1.lnb_ldx & 3) == 0 && ((uintptr_t)q1.lnb_x & 15) == 0))) {
    SkinnyPairArgs a;
    skinny_fill(q0, a.g[0]);
    skinny_fill(q1, a.g[1]);
    const bool sm1 = q1.sm_logits != nullptr;      // 32-wide strips for the second product
    if (sm1 && (ln0 || lnb1 || !q1.sm_dlogits || (q1.N & 31)))
      return dm_fail(DM_E_SHAPE, "gemm pair: the softmax-backward epilogue is built for plain products with N %% 32 == 0");
    a.nb0 = dm_cdiv(q0.N, 16);
    a.nb = a.nb0 + dm_cdiv(q1.N, sm1 ? 32 : 16);
    const int mmax = q0.M > q1.M ? q0.M : q1.M;
    const bool quarters = g_skinny_msplit >= 2 && mmax > 16 && dm_cdiv(mmax, 16) * a.nb <= 256;
    const bool halves = !quarters && g_skinny_msplit && mmax > 32 && 2 * a.nb <= 256;
    const dim3 grid((unsigned)(quarters ? dm_cdiv(mmax, 16) * a.nb : halves ? 2 * a.nb : a.nb)), blk(SK_WAVES * 64);
#define SKP_LAUNCH(M0_, M1_)                                                                                            \
  do {                                                                                                                  \
    if (mmax <= 16 || quarters) hipLaunchKernelGGL((skinny_gemm_pair_kernel<1, M0_, M1_>), grid, blk, 0, stream, a);    \
    else if (mmax <= 32 || halves) hipLaunchKernelGGL((skinny_gemm_pair_kernel<2, M0_, M1_>), grid, blk, 0, stream, a); \
    else hipLaunchKernelGGL((skinny_gemm_pair_kernel<4, M0_, M1_>), grid, blk, 0, stream, a);                           \
  } while (0)
    if (ln0 && !lnb1) SKP_LAUNCH(1, 0);
    else if (lnb1 && !ln0) SKP_LAUNCH(0, 2);
    else if (sm1) {
      if (mmax <= 16 || quarters) hipLaunchKernelGGL((skinny_gemm_pair_kernel<1, 0, 0, 2>), grid, blk, 0, stream, a);
      else if (mmax <= 32 || halves) hipLaunchKernelGGL((skinny_gemm_pair_kernel<2, 0, 0, 2>), grid, blk, 0, stream, a);
      else hipLaunchKernelGGL((skinny_gemm_pair_kernel<4, 0, 0, 2>), grid, blk, 0, stream, a);
    } else if (!ln0 && !lnb1) SKP_LAUNCH(0, 0);
    else return dm_fail(DM_E_SHAPE, "gemm pair: forward and backward LayerNorm prologues in one launch are not built");
#undef SKP_LAUNCH
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  DM_REQUIRE(!fused && !q1.lnf_ps && !q1.sm_logits && !q0.sm_logits, DM_E_SHAPE, "gemm pair: LayerNorm prologue / folded backward requested but the one-launch skinny path does not apply");
  DM_TRY(dm_gemm_launch(q0, ws, ws_bytes, stream));
  return dm_gemm_launch(q1, ws, ws_bytes, stream);
}

// logits = ELU(LN(x)) W^T + b for M <= 64 rows, N = groups*32, then one straight-through categorical draw per 32-logit
// group in the epilogue (rssm.py:143-148 posterior head + sample; rssm.py:174-179 prior head + sample): ONE launch.
int dm_gemm_sample_launch(const DmGemm& q, const DmSample& sm, hipStream_t stream) {
  DM_REQUIRE(q.ln_g && q.ln_b && dm_skinny_ln_ok(q.M, q.N, q.K) && skinny_ok(q, 64) && q.b_layout == 0 && (q.N & 31) == 0,
             DM_E_SHAPE, "gemm_sample: shape M=%d N=%d K=%d does not qualify", q.M, q.N, q.K);
  DM_REQUIRE(sm.onehot && (sm.u || sm.forced) && (sm.ldo & 3) == 0 &&
                 (((uintptr_t)sm.onehot | (uintptr_t)sm.z_next) & 15) == 0,
             DM_E_SHAPE, "gemm_sample: sampler outputs must be 16-byte aligned");
  SkinnyArgs a;
  skinny_fill(q, a);
  a.u = sm.u; a.forced = sm.forced; a.onehot = sm.onehot; a.ldo = sm.ldo; a.idx = sm.idx; a.z_next = sm.z_next;
  a.next_reset = sm.next_reset;
  a.znf = sm.z_next ? sm.z_next_frag : nullptr;
  // row-split strips as in dm_gemm_skinny_try: every workgroup redoes the LayerNorm + ELU of ITS rows only, so quarters also
  // quarter that prologue (27 -> 1x us at 64 rows)
  const int nst = q.N / 32;
  const bool quarters = g_skinny_msplit >= 2 && q.M > 16 && dm_cdiv(q.M, 16) * nst <= 256;
  const bool halves = !quarters && g_skinny_msplit && q.M > 32 && 2 * nst <= 256;
  const dim3 grid((unsigned)nst, (unsigned)(quarters ? dm_cdiv(q.M, 16) : halves ? 2 : 1)), blk(SK_WAVES * 64);
  if (q.M <= 16 || quarters) hipLaunchKernelGGL((skinny_gemm_sample_kernel<1>), grid, blk, 0, stream, a);
  else if (q.M <= 32 || halves) hipLaunchKernelGGL((skinny_gemm_sample_kernel<2>), grid, blk, 0, stream, a);
  else hipLaunchKernelGGL((skinny_gemm_sample_kernel<4>), grid, blk, 0, stream, a);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
