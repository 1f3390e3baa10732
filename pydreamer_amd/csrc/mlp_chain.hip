// Whole-MLP forward in ONE launch for the 400-wide heads at small and mid row counts (common.py:37-65; a2c.py:37-39;
// decoders.py:257-319):   [Linear -> LayerNorm(eps) -> ELU] x L  ->  Linear(out_dim <= 32)
//
// Where it runs (rows >= dm_mlp_chain_min_rows, default 256 - round 5: 1024 before; at the 350 / 650 rows of the 8- / 4-way shards the
// one launch beats the 13 dependent per-layer launches by 0.1 ms per step - below the row-panel threshold): the actor inside the
// imagination rollout (M = T*B = 2500 rows per horizon step, dreamer.py:188-216) and the reward / terminal heads over the
// T*B posterior features.  There the per-layer form costs 3 launches per layer (GEMM + split-K reduce or LayerNorm) on
// N = 400-wide products that the tiled GEMM runs at 30-55 TF/s: 13 dependent launches for the actor, 15 times per step.
//
// Tiling: a workgroup owns 16 COMPLETE rows through all layers; the activation row block lives in LDS between layers and
// never visits HBM unless the caller wants it saved for backward.  Inside a layer the 4 waves split the 25 column blocks
// (7 + 6 + 6 + 6; v_mfma_f32_16x16x4_f32, <= 28 accumulator registers) and walk K together in PAIRS of 16-k groups, so
// every weight element is read exactly once per workgroup.  The weights come from a FRAGMENT-MAJOR copy
// (dm_mlp_chain_pack_launch, once per rollout): the 64 lanes of one load read one contiguous KiB in the MFMA operand
// order (lane l: weight row 16*blk + (l&15), k = 32*pair + 16*half + 4*(l>>4) .. +3; with bf16 operands a lane's 8 values
// are ONE 16-byte load, already rounded).  Gathering the same fragments from the row-major weights (16 rows x 64 B per
// instruction) runs at 16.6 B/clk/CU whatever the cache level - the kernel then streams its 4.5 MB at 40 GB/s per CU,
// 140 us in fp32 and no faster in bf16; a still earlier version with K split across the waves brought every line up from
// L2 twice (250 us).  Two pairs of loads are in flight under a pair's 56 MFMAs (buffer loads: scalar base + one vector
// offset).  LayerNorm needs whole rows: the waves exchange per-row partial sums through LDS (two-pass variance),
// normalise their own columns in registers, write the next layer's LDS block; the output layer is one more small MFMA
// product with all loads issued up front.
//
// (Round 5 measured a 13-wave x 2-block split of the same 16-row block - a quarter of the dependent MFMA chain per wave, 3-4 waves per
//  SIMD, double-buffered fragments within 128 registers, parity-green: 141.8 us per 2 500-row call against 141.9 for the 4 x 7 split,
//  bf16 step 20.25 vs 19.86 ms.  The call is bound by what one CU can pull of the weight stream (2.9 MB per 16-row block: 20 GB/s per
//  CU with ~40-50 KB in flight), not by the MFMA chain: removed.)
// Work split: ceil(rows/16) workgroups - 157 for the 2500-row rollout step; a row block costs 16 x 1.13 M MACs whatever
// the tiling (the 7-block wave: 102 + 3*25 groups x 28 MFMAs x 32 clk = 158 k cycles = 66 us in fp32).  Measured per
// 2500-row call: 130 us fp32, 65 us bf16 (per-layer launches: 167 / 136).
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CH_NBLK = 25;                    // hidden = 400
constexpr int CH_N = CH_NBLK * 16;
constexpr int CH_LD = CH_N + 4;                // LDS row stride (floats): rows stay 16-byte aligned
#ifndef CH_PRE
#define CH_PRE 0
#endif
constexpr int CH_WB = 7;                       // column blocks per wave (wave 0: 0..6; waves 1..3: 6 real blocks + a dummy)

struct ChainArgs {
  const float* x; int ldx, in_dim;
  int rows, layers, out_dim, ldout;
  const float* w[DM_MAX_MLP_LAYERS + 1];
  const float* b[DM_MAX_MLP_LAYERS + 1];
  const float* gamma[DM_MAX_MLP_LAYERS];
  const float* beta[DM_MAX_MLP_LAYERS];
  float eps;
  float* xpre[DM_MAX_MLP_LAYERS];              // all three optional (nullptr: nothing is saved for backward)
  float* stats[DM_MAX_MLP_LAYERS];
  float* y[DM_MAX_MLP_LAYERS];
  float* out;
  // fragment-major copies of the hidden-layer weights (dm_mlp_chain_pack_launch; null: gather from the row-major weights).
  // Layout per layer: [block nb][pair p][half h][lane group q][lane l&15][4 floats]: the 64 lanes of one weight-fragment load
  // read ONE contiguous KiB.  The row-major gather (16 rows x 64 B per instruction) is limited to 16.6 B/clk/CU by the
  // texture path whatever the cache level (scripts/microbench/l2_stream.hip), contiguous loads reach ~50.
  const float* wp[DM_MAX_MLP_LAYERS];
  // Layer 0 with a sparse tail (the one-hot latent of a feature row, rssm.py:83-84): the product runs over the first k0
  // input columns only (packed weights hold those k0 columns) and add0 (rows x 400, dense) - the sum of the weight rows the
  // tail's non-zeros name, made by dm_sparse_rows_launch / dm_z_embed_launch - joins the pre-activation before the
  // LayerNorm.  k0 = in_dim, add0 = null: the plain product.
  int k0;
  const float* add0;
  // fused one-hot sampler of the output row (DmChainSample, common.h); samp_u null: none
  const float* samp_u; float* samp_onehot; int samp_ld; int32_t* samp_idx;
};

__device__ __forceinline__ float chain_red16(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

// ---- the operand ring (round 6) ------------------------------------------------------------------------------------------
// What bounds the kernel was measured in round 6 (profiles/r06_mlp_chain_anatomy.txt: weight loads only 72 us, MFMAs only 93 us,
// neither 44 us, both 100 us per 16-row block): ~49 us of MFMA issue and ~44 us of layer epilogues, barriers and exposed
// latencies - the weight stream hides under the MFMAs.  So (a) the loop keeps its operands in a ring of small UNITS - fp32: one
// 16-k group (a lane's four k of A and of the wave's 7 weight blocks: 8 x 16 bytes), four units deep; a unit's loads go out
// right behind the MFMAs that consumed its slot and have three units of MFMAs (2 700 cycles) to land - in 128 registers instead
// of three whole pairs in 192: the kernel fits 256 registers and TWO workgroups share a CU, so one block's epilogue runs under
// the other's MFMAs (the 938-block calls of the late head window ran one block per CU, every epilogue exposed); (b) the next
// layer's first weight units are requested BEFORE the epilogue of the current layer; (c) the epilogue's barriers wait for LDS
// traffic only (__syncthreads also drains the global stores of the saved activations, ~2 us each) and there are three per
// layer instead of four.  Same k -> MFMA order as before (results equal the round-5 kernel's up to the compiler's fma
// contraction choices in the epilogue).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 chain_bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void chain_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One unit of operands.  UK = 16: fp32 product, one 16-k group (b[i] = the lane's 4 k of weight block i).  UK = 32: bf16
// product, one pair of 16-k groups = ONE v_mfma_f32_16x16x32_bf16 per block; PACKED 2: b[i] = the lane's 8 bf16 of block i in
// one 16-byte load, else b[i] / b2[i] = its two fp32 fragments (rounded in registers).
template <int UK, int PACKED>
struct ChainUnit {
  float4 a[UK / 16];
  float4 b[CH_WB];
  float4 b2[(UK == 32 && PACKED != 2) ? CH_WB : 1];
};
// Weight fragments are buffer loads: one descriptor per layer (wave-uniform), the block's byte offset in the scalar
// offset, and ONE 32-bit vector offset shared by the 7 loads (per-load 64-bit vector addresses made the register
// allocator recycle in-flight load destinations as address temporaries, which put an s_waitcnt vmcnt(0) at the top of
// the loop and serialised the pipeline).  PACKED: 0 = row-major weights (wblk = byte offset of the wave's first block),
// 1 = fragment-major fp32, 2 = fragment-major bf16 (wblk = first block of the wave; K-edge zeros are in the packed copy).
template <int UK, int PACKED>
__device__ __forceinline__ void chain_load_b(ChainUnit<UK, PACKED>& f, __amdgpu_buffer_rsrc_t W, unsigned wblk, int K, int u,
                                             int l15, int q, int last) {
  const int np = (K + 31) >> 5;
  if constexpr (PACKED == 0) {
#pragma unroll
    for (int h = 0; h < UK / 16; ++h) {
      int k = u * UK + h * 16 + 4 * q;
      k = k < K ? k : K - 4;
      const unsigned off = (unsigned)(l15 * K + k) * 4u;
#pragma unroll
      for (int i = 0; i < CH_WB; ++i) {
        const unsigned soff = wblk + (unsigned)(i == CH_WB - 1 ? last : i) * 16u * (unsigned)K * 4u;       // wave-uniform bytes
        if (h == 0) f.b[i] = chain_bload(W, off, soff);
        else if constexpr (UK == 32) f.b2[i] = chain_bload(W, off, soff);
      }
    }
  } else {
    const unsigned voff = (unsigned)(q * 16 + l15) * 16u;
#pragma unroll
    for (int i = 0; i < CH_WB; ++i) {
      const unsigned blk = wblk + (unsigned)(i == CH_WB - 1 ? last : i);
      if constexpr (PACKED == 2) f.b[i] = chain_bload(W, voff, (blk * (unsigned)np + (unsigned)u) * 1024u);
      else if constexpr (UK == 16) f.b[i] = chain_bload(W, voff, (blk * (unsigned)(2 * np) + (unsigned)u) * 1024u);
      else {
        f.b[i] = chain_bload(W, voff, ((blk * (unsigned)np + (unsigned)u) * 2u) * 1024u);
        f.b2[i] = chain_bload(W, voff, ((blk * (unsigned)np + (unsigned)u) * 2u + 1u) * 1024u);
      }
    }
  }
}
// k past K: the address is clamped into the row and the ACTIVATION fragment is zeroed when it is consumed (K is a multiple
// of 4, so a 16-byte fragment is valid or not as a whole).
template <bool FIRST, int UK, int PACKED>
__device__ __forceinline__ void chain_load_a(ChainUnit<UK, PACKED>& f, const float* A, const float* ybuf, int K, int u, int l15, int q) {
#pragma unroll
  for (int h = 0; h < UK / 16; ++h) {
    int k = u * UK + h * 16 + 4 * q;
    k = k < K ? k : K - 4;
    f.a[h] = FIRST ? *reinterpret_cast<const float4*>(A + k) : *reinterpret_cast<const float4*>(ybuf + l15 * CH_LD + k);
  }
}
// BF (dm_mlp_params.precision = 1, conf.amp): the unit's 32 k are ONE v_mfma_f32_16x16x32_bf16 per block - a lane's two
// fragments (k = 32u + 4q.. and 32u + 16 + 4q..) are its 8 operand values, rounded to bf16 (RNE) in registers; A and B use
// the same k order, which is all the product needs.
__device__ __forceinline__ bf16x8 chain_bf8(float4 lo, float4 hi) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 v = {dm_pack_bf16x2(lo.x, lo.y), dm_pack_bf16x2(lo.z, lo.w), dm_pack_bf16x2(hi.x, hi.y), dm_pack_bf16x2(hi.z, hi.w)};
  return __builtin_bit_cast(bf16x8, v);
}
template <int UK, int PACKED>
__device__ __forceinline__ void chain_mfma(f32x4 (&acc)[CH_WB], const ChainUnit<UK, PACKED>& f, int K, int u, int q) {
  if constexpr (UK == 32) {
    const bool v0 = u * 32 + 4 * q < K, v1 = u * 32 + 16 + 4 * q < K;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const bf16x8 a8 = chain_bf8(v0 ? f.a[0] : z, v1 ? f.a[1] : z);
#pragma unroll
    for (int i = 0; i < CH_WB; ++i) {
      bf16x8 b8;
      if constexpr (PACKED == 2) b8 = __builtin_bit_cast(bf16x8, f.b[i]);
      else b8 = chain_bf8(f.b[i], f.b2[i]);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
    }
    return;
  } else {
  const bool valid = u * 16 + 4 * q < K;
  const float aj[4] = {valid ? f.a[0].x : 0.f, valid ? f.a[0].y : 0.f, valid ? f.a[0].z : 0.f, valid ? f.a[0].w : 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < CH_WB; ++i) {
      const float4 bb = f.b[i];
      const float bv = j == 0 ? bb.x : j == 1 ? bb.y : j == 2 ? bb.z : bb.w;
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[j], bv, acc[i], 0, 0, 0);
    }
  }
}

// One layer's k-loop over the ring.  `pre`: the weight parts of units 0 .. DEPTH-1 are already in flight (requested in front of
// the previous layer's epilogue): only their activation parts are fetched here.
template <bool FIRST, int UK, int PACKED, int DEPTH>
__device__ __forceinline__ void chain_layer(f32x4 (&acc)[CH_WB], ChainUnit<UK, PACKED> (&f)[DEPTH], bool pre, const float* A,
                                            const float* ybuf, __amdgpu_buffer_rsrc_t W, unsigned wblk, int K, int l15, int q, int last) {
  const int nu = (K + UK - 1) / UK;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int u = d < nu ? d : 0;
    if (!pre) chain_load_b<UK, PACKED>(f[d], W, wblk, K, u, l15, q, last);
    chain_load_a<FIRST, UK, PACKED>(f[d], A, ybuf, K, u, l15, q);
  }
  for (int u0 = 0; u0 < nu; u0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int u = u0 + d;
      if (u >= nu) break;
      __builtin_amdgcn_sched_barrier(0);
      chain_mfma<UK, PACKED>(acc, f[d], K, u, q);
      __builtin_amdgcn_sched_barrier(0);
      if (u + DEPTH < nu) {                             // the slot just consumed takes the unit DEPTH ahead
        chain_load_b<UK, PACKED>(f[d], W, wblk, K, u + DEPTH, l15, q, last);
        chain_load_a<FIRST, UK, PACKED>(f[d], A, ybuf, K, u + DEPTH, l15, q);
      }
    }
  }
}

template <bool BF, int PACKED>
__device__ __forceinline__ void chain_body(const ChainArgs& g, float* ybuf, float (*red)[4][16], float (*outp)[16][32]) {
  constexpr int UK = BF ? 32 : 16;
  constexpr int DEPTH = BF ? (PACKED == 2 ? 3 : 2) : 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // tell the compiler it is wave-uniform
  const int l15 = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * 16;
  const int nb0 = wave == 0 ? 0 : 1 + 6 * wave;          // first block of this wave: 0, 7, 13, 19
  const int cnt = wave == 0 ? 7 : 6;                     // real blocks; block index cnt.. are dummies (clamped, ignored)
  const int last = cnt == CH_WB ? CH_WB - 1 : 0;
  // layer 0 reads its activation rows from global memory; rows past the end re-read the last row (never written back)
  const int arow = m0 + l15 < g.rows ? m0 + l15 : g.rows - 1;
  const float* A0 = g.x + (size_t)arow * g.ldx;
  // C/D map of the 16x16 MFMA: col = lane & 15 (+16*block), row = 4*(lane>>4) + r
  const int rbase = m0 + 4 * q;
  bool rok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) rok[r] = rbase + r < g.rows;
  const float inv_n = 1.0f / (float)CH_N;
  // 32-bit element offsets of this lane's four output rows (a wave-uniform base pointer + one VGPR offset per access: 64-bit
  // per-lane pointers were hoisted above the k-loop and spilled)
  int roff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) roff[r] = (rok[r] ? rbase + r : g.rows - 1) * CH_N + nb0 * 16 + l15;
  ChainUnit<UK, PACKED> f[DEPTH];
  bool pre = false;

  for (int l = 0; l < g.layers; ++l) {
    f32x4 acc[CH_WB];
#pragma unroll
    for (int i = 0; i < CH_WB; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int K = l == 0 ? g.k0 : CH_N;
    // all waves walk 7 blocks (wave 0 sets the pace anyway); the 7th of waves 1..3 is a dummy that re-reads the wave's
    // first block and is ignored below
    const int npr = (K + 31) >> 5;
    const __amdgpu_buffer_rsrc_t Wl = PACKED
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.wp[l]), 0, CH_NBLK * npr * (BF ? 1024 : 2048), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w[l]), 0, CH_N * K * 4, 0x00020000);
    const unsigned wblk = PACKED ? (unsigned)nb0 : (unsigned)(nb0 * 16) * (unsigned)K * 4u;
    // the epilogue's per-column parameters, fetched BEFORE the k-loop (dependent global round trips after it cost ~2 us
    // each on a workgroup that has nothing else to run)
    float pbi[CH_WB], pga[CH_WB], pbe[CH_WB];
    {
      const float* bias = g.b[l];
#pragma unroll
      for (int i = 0; i < CH_WB; ++i) pbi[i] = bias ? bias[(nb0 + (i < cnt ? i : 0)) * 16 + l15] : 0.f;
    }
    if (l == 0) chain_layer<true, UK, PACKED, DEPTH>(acc, f, pre, A0, nullptr, Wl, wblk, K, l15, q, last);
    else chain_layer<false, UK, PACKED, DEPTH>(acc, f, pre, nullptr, ybuf, Wl, wblk, K, l15, q, last);
    // the next layer's first weight units, requested in front of this layer's epilogue (fragment-major copies only: one
    // descriptor per layer, K = 400 for every layer but the first)
    pre = false;
    if (PACKED && CH_PRE && l + 1 < g.layers) {
      constexpr int KN = CH_N, NPN = (KN + 31) >> 5;
      const __amdgpu_buffer_rsrc_t Wn =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.wp[l + 1]), 0, CH_NBLK * NPN * (BF ? 1024 : 2048), 0x00020000);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) chain_load_b<UK, PACKED>(f[d], Wn, (unsigned)nb0, KN, d, l15, q, last);
      pre = true;
    }
    {      // the LayerNorm's gain / bias columns: requested here, consumed behind the two statistics barriers (round 6: held
           // across the k-loop they cost 14 registers of the 256 that let two workgroups share a CU)
      const float* gam = g.gamma[l];
      const float* bet = g.beta[l];
#pragma unroll
      for (int i = 0; i < CH_WB; ++i) {
        const int c = (nb0 + (i < cnt ? i : 0)) * 16 + l15;
        pga[i] = gam[c];
        pbe[i] = bet[c];
      }
    }
    if (l == 0 && g.add0) {      // the sparse tail's contribution (28 more values: loaded here, not held across the k-loop)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < CH_WB; ++i)
          if (i < cnt) acc[i][r] += g.add0[roff[r] + i * 16];
      }
    }
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CH_WB; ++i)
      if (i < cnt) {
        const float bi = pbi[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[i][r] += bi;
          s[r] += acc[i][r];
        }
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float t = chain_red16(s[r]);
      if (l15 == 0) red[0][wave][4 * q + r] = t;
    }
    chain_lds_barrier();                                 // also: every wave is done reading ybuf
    float mean[4], rstd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      mean[r] = ((red[0][0][4 * q + r] + red[0][1][4 * q + r]) + (red[0][2][4 * q + r] + red[0][3][4 * q + r])) * inv_n;
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CH_WB; ++i)
      if (i < cnt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = acc[i][r] - mean[r];
          ss[r] += d * d;
        }
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                        // (its own array: red[0] is still being read by slower waves)
      const float t = chain_red16(ss[r]);
      if (l15 == 0) red[1][wave][4 * q + r] = t;
    }
    chain_lds_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r)
      rstd[r] = 1.0f / sqrtf(((red[1][0][4 * q + r] + red[1][1][4 * q + r]) + (red[1][2][4 * q + r] + red[1][3][4 * q + r])) * inv_n + g.eps);
    if (g.stats[l] && wave == 0 && l15 == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (rok[r]) { g.stats[l][2 * (rbase + r)] = mean[r]; g.stats[l][2 * (rbase + r) + 1] = rstd[r]; }
    }
    if (g.xpre[l]) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (rok[r]) {
#pragma unroll
          for (int i = 0; i < CH_WB; ++i)
            if (i < cnt) g.xpre[l][roff[r] + i * 16] = acc[i][r];
        }
    }
#pragma unroll
    for (int i = 0; i < CH_WB; ++i)
      if (i < cnt) {
        const float ga = pga[i], be = pbe[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = dm_elu((acc[i][r] - mean[r]) * rstd[r] * ga + be);
      }
    if (g.y[l]) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (rok[r]) {
#pragma unroll
          for (int i = 0; i < CH_WB; ++i)
            if (i < cnt) g.y[l][roff[r] + i * 16] = acc[i][r];
        }
    }
    // the post-activation block goes to LDS: the next layer's (or the output layer's) A operand
#pragma unroll
    for (int i = 0; i < CH_WB; ++i)
      if (i < cnt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ybuf[(4 * q + r) * CH_LD + (nb0 + i) * 16 + l15] = acc[i][r];
      }
    chain_lds_barrier();
  }

  // The MLP's output layer out = y Wout^T + bout (out_dim <= 32) as one more product: two 16-column blocks (output rows past
  // out_dim re-read the last one and are dropped), the 25 k-groups split over the 4 waves (7 + 6 + 6 + 6), ALL loads issued up
  // front (one global round trip), partial blocks summed through LDS in a fixed order.
  {
    const float* wout = g.w[g.layers];
    const float* bout = g.b[g.layers];
    const int g0 = wave == 0 ? 0 : 1 + 6 * wave, gn = wave == 0 ? 7 : 6;
    const int o0 = l15 < g.out_dim ? l15 : g.out_dim - 1, o1 = 16 + l15 < g.out_dim ? 16 + l15 : g.out_dim - 1;
    float4 wa[7], w0[7], w1[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int k = (g0 + (i < gn ? i : 0)) * 16 + 4 * q;
      w0[i] = *reinterpret_cast<const float4*>(wout + (o0 * CH_N + k));
      w1[i] = *reinterpret_cast<const float4*>(wout + (o1 * CH_N + k));
      wa[i] = *reinterpret_cast<const float4*>(ybuf + l15 * CH_LD + k);
    }
    f32x4 c0 = (f32x4){0.f, 0.f, 0.f, 0.f}, c1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 7; ++i)
      if (i < gn) {
        const float aj[4] = {wa[i].x, wa[i].y, wa[i].z, wa[i].w};
        const float b0[4] = {w0[i].x, w0[i].y, w0[i].z, w0[i].w};
        const float b1[4] = {w1[i].x, w1[i].y, w1[i].z, w1[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[j], b0[j], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[j], b1[j], c1, 0, 0, 0);
        }
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      outp[wave][4 * q + r][l15] = c0[r];
      outp[wave][4 * q + r][16 + l15] = c1[r];
    }
    chain_lds_barrier();
    for (int e = tid; e < 16 * g.out_dim; e += 256) {
      const int r = e / g.out_dim, o = e % g.out_dim;
      if (m0 + r < g.rows)
        g.out[(m0 + r) * g.ldout + o] = ((outp[0][r][o] + outp[1][r][o]) + (outp[2][r][o] + outp[3][r][o])) +
                                                (bout ? bout[o] : 0.f);
    }
    // The categorical draw of the row (the rollout's action, dreamer.py:198-200), one lane per row: the logits are re-made from
    // the LDS partials by the expression above (the same float), then the shared rule in the operation order of
    // sample_onehot_kernel (elementwise.hip): max, sequential sum of exp, sequential cdf, idx = #{cdf_k <= u cdf_last}.
    if (g.samp_u && tid < 16 && m0 + tid < g.rows) {
      const int r = tid, C = g.out_dim, row = m0 + tid;
      auto logit = [&](int k) { return ((outp[0][r][k] + outp[1][r][k]) + (outp[2][r][k] + outp[3][r][k])) + (bout ? bout[k] : 0.f); };
      float mx = logit(0);
      for (int k = 1; k < C; ++k) mx = fmaxf(mx, logit(k));
      float sum = 0.f;
      for (int k = 0; k < C; ++k) sum += expf(logit(k) - mx);
      float total_p = 0.f;
      for (int k = 0; k < C; ++k) total_p += expf(logit(k) - mx) / sum;
      const float target = g.samp_u[row] * total_p;
      float cdf = 0.f;
      int idx = 0;
      for (int k = 0; k < C; ++k) {
        cdf += expf(logit(k) - mx) / sum;
        idx += (cdf <= target) ? 1 : 0;
      }
      if (idx > C - 1) idx = C - 1;
      float* o = g.samp_onehot + (size_t)row * g.samp_ld;
      for (int k = 0; k < C; ++k) o[k] = (k == idx) ? 1.f : 0.f;
      if (g.samp_idx) g.samp_idx[row] = idx;
    }
  }
}

// EXCL: the launch is ONE round (a block per CU at most) on a latency chain - the rollout's actor, 15 times per step - and must
// not share its SIMDs: measured (profiles/r06_mlp_chain_anatomy.txt), a 256-register build that lets other streams' waves
// co-reside runs 15 % faster alone and makes the step SLOWER (fp32 33.15 -> 33.6 ms, 7-column shard 9.22 -> 9.94 ms): the
// round-5 kernel's 465 registers had kept every other wave off its SIMDs.  The EXCL build claims the accumulator file up to
// a255 (one dead write), so a wave owns its SIMD again; the multi-round launches (the 938-block head windows) take the shared
// build, where a second block's MFMAs cover the first one's epilogue.
// (DM_CHAIN_EXCL_AGPR, compile-time: the highest accumulator register the exclusive build claims.  255 = the whole file, nobody fits
//  beside the wave; 127 / 191 leave room for one 128- / 96-register wave of another kernel - measured, profiles/r06_chain_exclusivity.txt)
#ifndef DM_CHAIN_EXCL_AGPR
#define DM_CHAIN_EXCL_AGPR 255
#endif
#define DM_STR2(x) #x
#define DM_STR(x) DM_STR2(x)
template <bool BF, int PACKED, bool EXCL>
__global__ void __launch_bounds__(256, (PACKED && !EXCL) ? 2 : 1) mlp_chain_fwd_kernel(const ChainArgs g) {      // (the row-major A/B form keeps one block per CU)
  __shared__ __attribute__((aligned(16))) float ybuf[16 * CH_LD];      // the next layer's input block
  __shared__ float red[2][4][16];                                      // per-wave row partials: [0] sums, [1] centred squares
  __shared__ float outp[4][16][32];                                    // per-wave output-layer partials
  if constexpr (EXCL) asm volatile("v_accvgpr_write_b32 a" DM_STR(DM_CHAIN_EXCL_AGPR) ", 0" ::: "a" DM_STR(DM_CHAIN_EXCL_AGPR));
  DM_CHAIN_PRIO();
  chain_body<BF, PACKED>(g, ybuf, red, outp);
}

// fragment-major copy of the hidden-layer weights, all layers in one launch: one thread per destination 16-byte group
struct ChainPackArgs {
  const float* w[DM_MAX_MLP_LAYERS];
  float* dst[DM_MAX_MLP_LAYERS];
  int K[DM_MAX_MLP_LAYERS];
  int ld[DM_MAX_MLP_LAYERS];                  // row stride of the source weights (> K for layer 0 of a sparse-tail pack)
  unsigned first[DM_MAX_MLP_LAYERS + 1];      // first group of each layer (prefix sums of 25 * npairs * 128)
  int layers;
};
// bf16 form: one 16-byte group = the 8 values (k = 32p + 4q .. +3 and 32p + 16 + 4q .. +3) a lane feeds to ONE
// v_mfma_f32_16x16x32_bf16; group index inside a layer = (nb * npairs + p) * 64 + q * 16 + l15
__global__ void __launch_bounds__(256) mlp_chain_pack_bf16_kernel(const ChainPackArgs a) {
  const unsigned total = a.first[a.layers];
  for (unsigned e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    int l = 0;
    while (l + 1 < a.layers && e >= a.first[l + 1]) ++l;
    const unsigned r = e - a.first[l];
    const int K = a.K[l], np = (K + 31) >> 5;
    const int l15 = r & 15, q = (r >> 4) & 3;
    const unsigned bp = r >> 6;
    const int p = bp % np, nb = bp / np;
    const float* src = a.w[l] + (size_t)(nb * 16 + l15) * a.ld[l];
    float v[8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = p * 32 + h * 16 + 4 * q + j;
        v[4 * h + j] = k < K ? src[k] : 0.f;
      }
    uint4 o;
    o.x = dm_pack_bf16x2(v[0], v[1]); o.y = dm_pack_bf16x2(v[2], v[3]);
    o.z = dm_pack_bf16x2(v[4], v[5]); o.w = dm_pack_bf16x2(v[6], v[7]);
    reinterpret_cast<uint4*>(a.dst[l])[r] = o;
  }
}
__global__ void __launch_bounds__(256) mlp_chain_pack_kernel(const ChainPackArgs a) {
  const unsigned total = a.first[a.layers];
  for (unsigned e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    int l = 0;
    while (l + 1 < a.layers && e >= a.first[l + 1]) ++l;
    const unsigned r = e - a.first[l];
    const int K = a.K[l], np = (K + 31) >> 5;
    const int l15 = r & 15, q = (r >> 4) & 3, h = (r >> 6) & 1;
    const unsigned bp = r >> 7;
    const int p = bp % np, nb = bp / np;
    const int k = p * 32 + h * 16 + 4 * q;
    const float* src = a.w[l] + (size_t)(nb * 16 + l15) * a.ld[l] + k;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k + 3 < K) v = make_float4(src[0], src[1], src[2], src[3]);        // K % 4 == 0 (host-checked)
    reinterpret_cast<float4*>(a.dst[l])[r] = v;
  }
}

// ---------------------------------------------------------------- host side ---------------------
static const int g_chain_off = getenv("DM_MLP_NO_CHAIN") ? 1 : 0;       // A/B switch: keep the per-layer launches
// A row block takes ~135 us however many there are (one CU walks all 1.13 M MACs per row: 66 us of MFMA issue plus the
// weight stream's latency).  Alone, the per-layer form's ~13 launches win below ~1000 rows (350 rows: 95 us of kernels,
// spread over all CUs by split-K); INSIDE the step their 12 dependent-launch gaps cost more than that: measured in round 5
// at the 350 / 650 rows of the 8- / 4-way shards, one launch beats the 13 by 0.1 ms per step; round 6 measured the 300 rows
// of a 6-column shard (ranks 2-7 of an 8-way split; `bench.py --emulate-world 8 --emulate-rank 7`, profiles/r06_chain_rows.txt).
// The default sits just below that smallest measured size.  DM_CHAIN_MIN_ROWS / dm_mlp_chain_min_rows override.
static int g_chain_min_rows = getenv("DM_CHAIN_MIN_ROWS") ? atoi(getenv("DM_CHAIN_MIN_ROWS")) : 256;
extern "C" int dm_mlp_chain_min_rows(int rows) {       // rows >= 1: set; returns the previous value
  const int prev = g_chain_min_rows;
  if (rows >= 1) g_chain_min_rows = rows;
  return prev;
}

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static const int g_chain_nopack = getenv("DM_CHAIN_NO_PACK") ? 1 : 0;      // A/B switch

bool dm_mlp_chain_ok(int rows, int in_dim, int hidden, int layers, int out_dim, const float* x, int ldx,
                     const dm_mlp_params* p) {
  if (g_chain_off || hidden != CH_N || rows < 1 || rows < g_chain_min_rows || layers < 1 || layers > DM_MAX_MLP_LAYERS || out_dim < 1 || out_dim > 32)
    return false;
  if ((in_dim & 3) != 0 || in_dim < 4 || (ldx & 3) != 0 || !al16(x)) return false;
  for (int l = 0; l < layers; ++l)
    if (!al16(p->w[l]) || !p->ln_g[l] || !p->ln_b[l]) return false;
  return true;
}

// fp32 calls only (with bf16 operands the product is cheaper than the gather, as measured for the row panels); the dense
// part keeps the 16-byte operand loads
bool dm_mlp_chain_sparse_ok(int in_dim, int sparse_cols) {
  static const int no_sparse = getenv("DM_MLP_NO_SPARSE") ? 1 : 0;         // A/B switch (shared with the row-panel path)
  const int dense = in_dim - sparse_cols;
  return !no_sparse && !g_chain_nopack && !dm_cur_precision() && sparse_cols > 0 && sparse_cols < in_dim && in_dim <= 4096 &&
         (dense & 7) == 0 && dense >= 32;
}
static size_t chain_pack_layer_floats(int K) { return (size_t)CH_NBLK * ((K + 31) / 32) * 512; }
size_t dm_mlp_chain_pack_floats(int in_dim, int layers) {
  size_t n = chain_pack_layer_floats(in_dim);
  for (int l = 1; l < layers; ++l) n += chain_pack_layer_floats(CH_N);
  return n;
}
// wpack: dm_mlp_chain_pack_floats(in_dim, layers) floats, 16-byte aligned.  A caller that runs the same weights several times
// (the H steps of a rollout) packs once.
int dm_mlp_chain_pack_launch(int in_dim, int layers, const dm_mlp_params* p, float* wpack, hipStream_t st, int k0) {
  DM_REQUIRE(wpack && al16(wpack) && (in_dim & 3) == 0 && layers >= 1 && layers <= DM_MAX_MLP_LAYERS, DM_E_SHAPE, "mlp_chain_pack");
  DM_REQUIRE(k0 >= 0 && k0 <= in_dim && (k0 & 3) == 0, DM_E_SHAPE, "mlp_chain_pack: k0=%d of in_dim=%d", k0, in_dim);
  if (k0 == 0) k0 = in_dim;
  ChainPackArgs a = {};
  a.layers = layers;
  const bool bf = dm_cur_precision() != 0;      // the copy's element type follows the precision of the call that will use it
  size_t off = 0;
  unsigned first = 0;
  for (int l = 0; l < layers; ++l) {
    const int K = l == 0 ? k0 : CH_N;
    a.w[l] = p->w[l]; a.dst[l] = wpack + off; a.K[l] = K; a.ld[l] = l == 0 ? in_dim : CH_N; a.first[l] = first;
    off += chain_pack_layer_floats(K);                                         // the layer offsets are the fp32 ones in both forms
    first += (unsigned)(chain_pack_layer_floats(K) / (bf ? 8 : 4));           // 16-byte groups: half as many in bf16
  }
  a.first[layers] = first;
  int blocks = dm_cdiv(first, 256);
  if (blocks > 2048) blocks = 2048;
  if (bf) hipLaunchKernelGGL(mlp_chain_pack_bf16_kernel, dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(mlp_chain_pack_kernel, dim3(blocks), dim3(256), 0, st, a);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// acts pointers per layer may be null (nothing saved); all row pointers already include the caller's row offset.
// wpack: fragment-major weights from dm_mlp_chain_pack_launch (null: the kernel gathers from the row-major weights).
int dm_mlp_chain_fwd_launch(int rows, int in_dim, int layers, int out_dim, const float* x, int ldx, const dm_mlp_params* p,
                            float* const* xpre, float* const* stats, float* const* y, float* out, int ldout, const float* wpack,
                            hipStream_t st, int k0, const float* add0, const DmChainSample* sample) {
  ChainArgs a = {};
  if (sample) {
    DM_REQUIRE(sample->u && sample->onehot && sample->ldo >= out_dim, DM_E_NULL, "mlp_chain: fused sampler arguments");
    a.samp_u = sample->u; a.samp_onehot = sample->onehot; a.samp_ld = sample->ldo; a.samp_idx = sample->idx;
  }
  if (k0 == 0) k0 = in_dim;
  DM_REQUIRE(k0 == in_dim || (add0 && wpack && !g_chain_nopack && k0 >= 4 && k0 < in_dim && (k0 & 3) == 0), DM_E_SHAPE,
             "mlp_chain: a sparse-tail layer 0 (k0=%d of %d) needs its addend and weights packed for k0", k0, in_dim);
  if (wpack && !g_chain_nopack) {
    size_t off = 0;
    for (int l = 0; l < layers; ++l) {
      a.wp[l] = wpack + off;
      off += chain_pack_layer_floats(l == 0 ? k0 : CH_N);
    }
  }
  a.k0 = k0; a.add0 = k0 < in_dim ? add0 : nullptr;
  a.x = x; a.ldx = ldx; a.in_dim = in_dim;
  a.rows = rows; a.layers = layers; a.out_dim = out_dim; a.ldout = ldout;
  for (int l = 0; l <= layers; ++l) { a.w[l] = p->w[l]; a.b[l] = p->b[l]; }
  for (int l = 0; l < layers; ++l) {
    a.gamma[l] = p->ln_g[l]; a.beta[l] = p->ln_b[l];
    a.xpre[l] = xpre ? xpre[l] : nullptr; a.stats[l] = stats ? stats[l] : nullptr; a.y[l] = y ? y[l] : nullptr;
  }
  a.eps = 1e-3f;
  a.out = out;
  double macs = (double)in_dim * CH_N + (double)(layers - 1) * CH_N * CH_N + (double)out_dim * CH_N;
  const int slot = dm_prof_slot_begin(22, 2.0 * rows * macs,
                                      4.0 * ((double)rows * in_dim + macs + (double)rows * out_dim +
                                             (xpre ? 2.0 * rows * CH_N * layers : 0.0)), st);
  // packed copies are bf16 when the kernel multiplies in bf16 (dm_mlp_chain_pack_launch); all layers or none are packed
  const dim3 grid((unsigned)dm_cdiv(rows, 16)), blk(256);
  const bool packed = a.wp[0] != nullptr;
  static const int excl_env = getenv("DM_CHAIN_EXCL") ? atoi(getenv("DM_CHAIN_EXCL")) : 1;      // A/B switch: 0 = the shared build everywhere
  const bool excl = excl_env && grid.x <= 256;
#define DM_CHAIN_LAUNCH(BF_, PK_)                                                                        \
  do {                                                                                                   \
    if (excl) hipLaunchKernelGGL((mlp_chain_fwd_kernel<BF_, PK_, true>), grid, blk, 0, st, a);          \
    else hipLaunchKernelGGL((mlp_chain_fwd_kernel<BF_, PK_, false>), grid, blk, 0, st, a);              \
  } while (0)
  if (dm_cur_precision()) {
    if (packed) DM_CHAIN_LAUNCH(true, 2);
    else DM_CHAIN_LAUNCH(true, 0);
  } else {
    if (packed) DM_CHAIN_LAUNCH(false, 1);
    else DM_CHAIN_LAUNCH(false, 0);
  }
#undef DM_CHAIN_LAUNCH
  dm_prof_slot_end(slot, st);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
