// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), LDS-tiled, 256-thread workgroups.
//
//   C[m,n] = epi( sum_k A(m,k) * B(n,k) )
//
// One kernel template serves every dense contraction of the DreamerV2 step:
//   Linear forward            y = x W^T      (A k-contiguous, B k-contiguous)   reference: common.py:37-65, rssm.py:138-146
//   Linear backward (data)    dx = dy W      (A k-contiguous, B n-contiguous)
//   Linear backward (weight)  dW = dy^T x    (A m-contiguous, B n-contiguous, split-K over the row dimension)
//   conv / conv-transpose as GEMM over im2col matrices (encoders.py:80-96, decoders.py:144-161)
//
// Three arithmetic modes, all with fp32 accumulation and fp32 results (template parameter BF):
//   0  fp32 operands on v_mfma_f32_32x32x2_f32 - an exact k-ordered fmaf chain (no TF32/xf32 on gfx950): the default
//   1  bf16 operands (conf.amp, BASELINE configs[2])
//
// Tiling: BMxBNx32 block tile, 4 waves in a 2x2 grid, each wave (BM/2)x(BN/2) as 32x32 MFMA blocks.
// LDS image per operand is chosen by the SOURCE layout so that both the global->LDS stores and the
// LDS->fragment reads stay conflict-free:
//   k-contiguous source: LDS [row][k] (row stride 36 floats), fragment = one ds_read_b128 per lane holding 4 k's
//   row-contiguous source: LDS [k][row], fragment = ds_read_b32 per k (lanes read consecutive floats)
// The k -> (mfma step, lane half) mapping is identical in both images: within a group of 8 k's, lanes 0-31
// feed k = 8g + j and lanes 32-63 feed k = 8g + 4 + j at step j (0..3); A and B therefore always agree.
// Global loads of tile t+1 are issued into registers before the MFMAs of tile t (register prefetch).
#include "common.h"
#include "dma_loop.h"
#include <stdlib.h>
#include <math.h>
#include <type_traits>
#include <mutex>
#include <atomic>


struct GemmKArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* add;
  const float* mulref;
  const uint8_t* row_zero;
  const int* a_maj; const int* a_min;   // optional separable-gather tables: element = A[a_maj[major] + a_min[minor]]
  const int* b_maj; const int* b_min;
  float* partial;
  const int2* c_tab;          // SC (scatter) epilogue: per output row {float offset of its class-(0,0) pixel, validity bits}
  int sc_cout, sc_wpitch;     // channels per parity class; floats between two output rows of the big image (wb * cout)
  int bias_mod;               // > 0: bias[col % bias_mod] (a 1x1 -> kxk transposed convolution: columns are (ky,kx,o))
  int M, N, K;
  int lda, ldb, ldc, ldadd, ldmul;
  int flags;
  int k_per_split;
  int nsplit;
  int tiles_m;
  int n_tiles, n_items;
  int tiles_n, n_fast;
  int n_groups;               // > 1: the fast tile dimension is cut into this many column groups (gemm_decode)
  int a_vec, b_vec;
  unsigned short* Ch;         // optional bf16 twin of C (same ldc), written with the final value
  int use_dma;                // host-side: this launch takes gemm_dma_kernel (not read by the kernels)
};

// Load a (ROWS x 32) operand tile into registers, zero-filled outside [0,nrows) x [k0,kend).  Branch-free: out-of-range
// groups read a clamped (always valid) address and are zeroed by a select, so edge tiles cost the same instruction
// stream as interior ones and the loads issue back to back ahead of the MFMAs.
// LAYOUT 0: the k index is contiguous in memory (major = row, minor = k); LAYOUT 1: the row index is (major = k).
// GATHER: an element lives at P[tmaj[major] + tmin[minor]] — the implicit-im2col operand (tmaj = start of a conv patch,
// tmin = offset of tap (ky,kx,c) inside it).
// VEC (chosen on the host, grid-uniform): the minor extent is a multiple of 4 and rows are 16-byte aligned, so every
// group of 4 minors is entirely inside or entirely outside and is one 16-byte load; otherwise 4 predicated scalar loads.
// KSEQ (bf16 path, row-contiguous sources with 256 % (ROWS/4) == 0): thread t owns NF4 CONSECUTIVE k of one 4-row group
// (k = (t / F4_PER_K) * NF4 + i) instead of k strided by 256 / F4_PER_K, so the transposing bf16 store can pack them
// into one 4/8-byte LDS write per row; global coalescing is unchanged (consecutive threads = consecutive row groups).
template <int ROWS, int LAYOUT, int NF4, bool GATHER, bool VEC, bool KSEQ = false>
__device__ __forceinline__ void gemm_load_tile(float4 (&r)[NF4], const float* __restrict__ P, int ld, int row0,
                                               int nrows, int k0, int kend, const int* __restrict__ tmaj,
                                               const int* __restrict__ tmin, int tid, unsigned& mask) {
  // `mask` gets one validity bit per loaded float; the zero-fill select is applied by gemm_store_tile AFTER the MFMAs
  // of the current tile — selecting here would consume the loads at once and drain vmcnt before the MFMAs start.
  mask = 0u;
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const int f = tid + i * 256;
    int major, minor, major_end, minor_end;
    if (LAYOUT == 0) {
      major = row0 + (f >> 3);
      minor = k0 + ((f & 7) << 2);
      major_end = nrows;
      minor_end = kend;
    } else {
      constexpr int F4_PER_K = ROWS / 4;
      if (KSEQ) {
        major = k0 + (tid / F4_PER_K) * NF4 + i;
        minor = row0 + ((tid % F4_PER_K) << 2);
      } else {
        major = k0 + f / F4_PER_K;
        minor = row0 + ((f % F4_PER_K) << 2);
      }
      major_end = kend;
      minor_end = nrows;
    }
    const bool okm = major < major_end;
    const int mj = okm ? major : 0;
    float4 v;
    if (VEC) {
      const bool ok = okm && minor < minor_end;
      const int mn = ok ? minor : 0;
      const size_t off = GATHER ? (size_t)(tmaj[mj] + tmin[mn]) : (size_t)mj * ld + mn;
      v = *reinterpret_cast<const float4*>(P + (ok ? off : 0));
      mask |= ok ? (0xFu << (4 * i)) : 0u;
    } else {
      float e[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = okm && minor + j < minor_end;
        const int mn = ok ? minor + j : 0;
        const size_t off = GATHER ? (size_t)(tmaj[mj] + tmin[mn]) : (size_t)mj * ld + mn;
        e[j] = P[ok ? off : 0];
        mask |= ok ? (1u << (4 * i + j)) : 0u;
      }
      v = make_float4(e[0], e[1], e[2], e[3]);
    }
    r[i] = v;
  }
}

template <int ROWS, int LAYOUT, int NF4>
__device__ __forceinline__ void gemm_store_tile(const float4 (&rr)[NF4], unsigned mask, float* S, int tid) {
  constexpr int LDK = 36;
  constexpr int LDM = ROWS + 4;
  float4 r[NF4];
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    r[i].x = (mask >> (4 * i + 0)) & 1u ? rr[i].x : 0.f;
    r[i].y = (mask >> (4 * i + 1)) & 1u ? rr[i].y : 0.f;
    r[i].z = (mask >> (4 * i + 2)) & 1u ? rr[i].z : 0.f;
    r[i].w = (mask >> (4 * i + 3)) & 1u ? rr[i].w : 0.f;
  }
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const int f = tid + i * 256;
    if (LAYOUT == 0) {
      const int row = f >> 3;
      const int kq = (f & 7) << 2;
      *reinterpret_cast<float4*>(&S[row * LDK + kq]) = r[i];
    } else {
      constexpr int F4_PER_K = ROWS / 4;
      const int kr = f / F4_PER_K;
      const int r4 = (f % F4_PER_K) << 2;
      *reinterpret_cast<float4*>(&S[kr * LDM + r4]) = r[i];
    }
  }
}

// ---- bf16 operand path (BASELINE configs[2]: mixed precision).  Operands are loaded as fp32 exactly as above and
// rounded to bf16 (round-to-nearest-even) on their way into LDS; products run on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation; results, activations and everything outside the GEMMs stay fp32.  The LDS image is [row][k] bf16
// (row stride 40 elements = 80 bytes, 16-byte aligned) for BOTH source layouts - a row-contiguous source is transposed
// by 2-byte stores - because the MFMA wants 8 consecutive k per lane: lane l feeds row l&31, k = 8*(l>>5) .. +7.
constexpr int LDKB = 40;
template <int ROWS, int LAYOUT, int NF4, bool KSEQ = false>
__device__ __forceinline__ void gemm_store_tile_bf16(const float4 (&rr)[NF4], unsigned mask, unsigned short* S, int tid) {
  if (LAYOUT == 1 && KSEQ) {      // rr[i] = rows r4..r4+3 at k = kq*NF4 + i: one packed write of NF4 bf16 per row
    constexpr int F4_PER_K = ROWS / 4;
    const int kq = (tid / F4_PER_K) * NF4;
    const int r4 = (tid % F4_PER_K) << 2;
    float h[4][NF4];
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      h[0][i] = (mask >> (4 * i + 0)) & 1u ? rr[i].x : 0.f;
      h[1][i] = (mask >> (4 * i + 1)) & 1u ? rr[i].y : 0.f;
      h[2][i] = (mask >> (4 * i + 2)) & 1u ? rr[i].z : 0.f;
      h[3][i] = (mask >> (4 * i + 3)) & 1u ? rr[i].w : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned short* d = &S[(r4 + j) * LDKB + kq];
      if (NF4 == 4) *reinterpret_cast<uint2*>(d) = make_uint2(dm_pack_bf16x2(h[j][0], h[j][1]), dm_pack_bf16x2(h[j][2], h[j][3]));
      else if (NF4 == 2) *reinterpret_cast<unsigned*>(d) = dm_pack_bf16x2(h[j][0], h[j][1]);
      else
#pragma unroll
        for (int i = 0; i < NF4; ++i) d[i] = (unsigned short)dm_f2bf(h[j][i]);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const float x = (mask >> (4 * i + 0)) & 1u ? rr[i].x : 0.f;
    const float y = (mask >> (4 * i + 1)) & 1u ? rr[i].y : 0.f;
    const float z = (mask >> (4 * i + 2)) & 1u ? rr[i].z : 0.f;
    const float w = (mask >> (4 * i + 3)) & 1u ? rr[i].w : 0.f;
    const unsigned p01 = dm_pack_bf16x2(x, y), p23 = dm_pack_bf16x2(z, w);
    const unsigned b0 = p01 & 0xFFFFu, b1 = p01 >> 16, b2 = p23 & 0xFFFFu, b3 = p23 >> 16;
    const int f = tid + i * 256;
    if (LAYOUT == 0) {
      const int row = f >> 3;
      const int kq = (f & 7) << 2;
      *reinterpret_cast<uint2*>(&S[row * LDKB + kq]) = make_uint2(p01, p23);
    } else {
      constexpr int F4_PER_K = ROWS / 4;
      const int kr = f / F4_PER_K;
      const int r4 = (f % F4_PER_K) << 2;
      S[(r4 + 0) * LDKB + kr] = (unsigned short)b0;
      S[(r4 + 1) * LDKB + kr] = (unsigned short)b1;
      S[(r4 + 2) * LDKB + kr] = (unsigned short)b2;
      S[(r4 + 3) * LDKB + kr] = (unsigned short)b3;
    }
  }
}

// Work item -> (tile, split).  Workgroup b is observed to run on XCD b % 8 (used for L2 affinity only, never for
// correctness): each XCD is given one CONTIGUOUS chunk of the item list, so the tiles an XCD's L2 sees share B panels
// (tile_m runs fastest inside a chunk).  Bijective for any item count (q = n/8, r = n%8).
__device__ __forceinline__ int gemm_item_of(int id, int n_items) {
  const int q = n_items >> 3, r = n_items & 7;
  const int xcd = id & 7, j = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

struct GemmItem {
  int m0, n0, kbeg, kend, nkt, split;
};
template <int BM, int BN>
__device__ __forceinline__ GemmItem gemm_decode(const GemmKArgs& g, int id) {
  const int item = gemm_item_of(id, g.n_items);
  const int tile = item % g.n_tiles;
  GemmItem it;
  it.split = item / g.n_tiles;
  // the index of the dimension with FEWER tiles runs fastest: the tiles that re-read one panel of the big operand
  // (activation rows for M >> N, patch rows for weight gradients) are then neighbours in the same XCD's chunk
  // Round 6: with n_groups > 1 the fast dimension is cut into that many groups, each walked completely (slow index inside a
  // group, fast index inside the group's width) before the next one.  An XCD's contiguous chunk of the item list then covers a
  // 2-D block of tiles - about (n_tiles / 8) / width rows of the slow operand x one group's width of the fast one - instead of a
  // few slow rows x ALL fast panels: eight private L2s fetch ~25 % fewer operand bytes on the 2 500 x 1 800 rollout products
  // (the eight L2s do not share lines, and a kernel boundary invalidates them: what one XCD's tiles need, that XCD fetches).
  // Same tiles, same arithmetic per tile: results are bit-identical for every grouping.
  const int F = g.n_fast ? g.tiles_n : g.tiles_m;
  int fast, slow;
  if (g.n_groups > 1) {
    const int S = g.n_fast ? g.tiles_m : g.tiles_n;
    const int w = F / g.n_groups, rem = F % g.n_groups;
    int base = 0, f0 = 0, wg = w + (rem > 0 ? 1 : 0);
    for (int gi = 0; gi < g.n_groups - 1; ++gi) {
      const int cnt = S * wg;
      if (tile < base + cnt) break;
      base += cnt; f0 += wg;
      wg = w + (gi + 1 < rem ? 1 : 0);
    }
    const int local = tile - base;
    fast = f0 + local % wg;
    slow = local / wg;
  } else {
    fast = tile % F;
    slow = tile / F;
  }
  if (g.n_fast) {
    it.n0 = fast * BN;
    it.m0 = slow * BM;
  } else {
    it.m0 = fast * BM;
    it.n0 = slow * BN;
  }
  it.kbeg = it.split * g.k_per_split;
  it.kend = min(g.K, it.kbeg + g.k_per_split);
  it.nkt = (it.kend > it.kbeg) ? (it.kend - it.kbeg + 31) / 32 : 0;
  return it;
}

// Epilogue shared by the tile kernels: `acc` holds the (BM/WGM) x (BN/WGN) block of this wave in the 32x32 MFMA C/D layout.
// `stage` (optional): 32 x 40 bf16 of LDS owned by this wave; the bf16 twin of a dense result then leaves as 16-byte
// stores of 8 columns (a 32 x 32 block = 2 store instructions instead of 16 two-byte ones - the two-byte form cost
// 170 us on a 2.25 M x 48 result).  The caller guarantees nobody else reads that LDS any more.
constexpr int EPI_STAGE_LD = 40;
template <int BM, int BN, int WGM, int WGN, bool SC, int MB, int NB>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[MB][NB], const GemmKArgs& g, const GemmItem& cur, int wm, int wn,
                                              int l31, int half, unsigned short* stage = nullptr) {
  const bool staged = !SC && stage && g.Ch && g.nsplit <= 1 && (g.ldc & 7) == 0 && (((uintptr_t)g.Ch) & 15) == 0;
  // C/D fragment map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int col = cur.n0 + wn * (BN / WGN) + nb * 32 + l31;
      int sc_off = 0, sc_need = 0, sc_o = 0;
      if (SC) {
        const int cls = col / g.sc_cout;
        sc_o = col - cls * g.sc_cout;
        sc_off = (cls >> 1) * g.sc_wpitch + (cls & 1) * g.sc_cout + sc_o;
        sc_need = cls;                                  // bit 1: odd output row needed, bit 0: odd output column needed
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = cur.m0 + wm * (BM / WGM) + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < g.M && col < g.N) {
          float v = acc[mb][nb][r];
          if (SC) {
            const int2 ct = g.c_tab[row];
            if ((sc_need & ~ct.y) == 0) {               // the odd row / column of this class pixel exists in the big image
              const size_t at = (size_t)ct.x + sc_off;
              if (g.bias) v += g.bias[sc_o];
              if (g.flags & DM_GEMM_ELU) v = dm_elu(v);
              if (g.mulref) v *= dm_elu_grad_from_y(g.mulref[at]);
              g.C[at] = v;
              if (g.Ch) g.Ch[at] = (unsigned short)dm_f2bf(v);
            }
          } else if (g.nsplit > 1) {
            g.partial[((size_t)cur.split * g.M + row) * g.N + col] = v;
          } else {
            if (g.row_zero && g.row_zero[row]) v = 0.f;
            if (g.bias) v += g.bias[g.bias_mod > 0 ? col % g.bias_mod : col];
            if (g.add) v += g.add[(size_t)row * g.ldadd + col];
            float* c = g.C + (size_t)row * g.ldc + col;
            if (g.flags & DM_GEMM_ACCUM) v += *c;
            if (g.flags & DM_GEMM_ELU) v = dm_elu(v);
            if (g.mulref) v *= dm_elu_grad_from_y(g.mulref[(size_t)row * g.ldmul + col]);
            *c = v;
            if (staged) stage[((r & 3) + 8 * (r >> 2) + 4 * half) * EPI_STAGE_LD + l31] = (unsigned short)dm_f2bf(v);
            else if (g.Ch) g.Ch[(size_t)row * g.ldc + col] = (unsigned short)dm_f2bf(v);
          }
        }
      }
      if (staged) {      // this wave's 32 x 32 block: lane -> (row lane >> 1, 16 columns), two 16-byte stores
        __builtin_amdgcn_wave_barrier();
        const int lane = half * 32 + l31;
        const int rl = lane >> 1, c0 = (lane & 1) * 16;
        const int row = cur.m0 + wm * (BM / WGM) + mb * 32 + rl;
        const int colb = cur.n0 + wn * (BN / WGN) + nb * 32 + c0;
        if (row < g.M) {
#pragma unroll
          for (int h8 = 0; h8 < 2; ++h8) {
            const int cc = colb + h8 * 8;
            const unsigned short* sp = &stage[rl * EPI_STAGE_LD + c0 + h8 * 8];
            unsigned short* dp = g.Ch + (size_t)row * g.ldc + cc;
            if (cc + 8 <= g.N) *reinterpret_cast<uint4*>(dp) = *reinterpret_cast<const uint4*>(sp);
            else
              for (int j = 0; j < 8 && cc + j < g.N; ++j) dp[j] = sp[j];
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

// One work item (tile x k-slice) per workgroup.  (A persistent tile walk with cross-tile prefetch was built and lost the
// A/B on MI355X — it costs the 3-8 workgroups/CU residency that hides the k-loop's load latency; numbers in
// profiles/r01_gemm_persist_ab.txt.)
// GA / GB: operand A / B is a separable gather (compile-time, so the plain instances keep their register budget).
// VEC: both operands take the 16-byte load path (host-checked alignment and extents).
// WGM x WGN: the 4 waves' grid over the block tile (2x2; 4x1 for the 128x96 tile, 1x4 for 96x128 - the conv stack is
// full of 96-wide operands (cnn_depth 48), which 64/128-wide tiles pad by 25-33 %).
// SC: stride-2 transposed-convolution output scatter.  The GEMM row is a class pixel (n, yy, xx), the column is
// (parity class (py,px), channel o): the result lands at big[n, 2yy+py, 2xx+px, o] (NHWC) - the four parity classes of a
// k = 4 / k = 6 transposed convolution share ONE gathered A operand (the input patch (yy-a, xx-b)), so the whole layer is
// one GEMM with N = 4*cout that writes the big image directly: no column matrix, no col2im pass (conv.hip).
template <int BM, int BN, int AL, int BL, bool VEC, bool GA = false, bool GB = false, int WGM = 2, int WGN = 2,
          int BF = 0, bool SC = false>
__global__ void __launch_bounds__(256) gemm_f32_kernel(const GemmKArgs g) {
  static_assert(WGM * WGN == 4 && BM % (32 * WGM) == 0 && BN % (32 * WGN) == 0, "wave grid must tile the block tile");
  constexpr int BK = 32;
  constexpr int LDK = 36;
  constexpr int LDMA = BM + 4, LDMB = BN + 4;
  constexpr int A_FLOATS = (AL == 0) ? BM * LDK : BK * LDMA;
  constexpr int B_FLOATS = (BL == 0) ? BN * LDK : BK * LDMB;
  constexpr int MB = BM / (32 * WGM), NB = BN / (32 * WGN);
  constexpr int A_F4 = BM * BK / 4 / 256, B_F4 = BN * BK / 4 / 256;
  constexpr bool A_KSEQ = BF != 0 && AL == 1 && 256 % (BM / 4) == 0, B_KSEQ = BF != 0 && BL == 1 && 256 % (BN / 4) == 0;
  // BF: 0 = fp32 operands on v_mfma_f32_32x32x2_f32, 1 = bf16 operands (conf.amp)
  constexpr int SMEM_FLOATS = A_FLOATS + B_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  float* As = smem;
  float* Bs = smem + A_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int wm = wave / WGN, wn = wave % WGN;

  const GemmItem cur = gemm_decode<BM, BN>(g, blockIdx.x);

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[A_F4], rb[B_F4];
  unsigned ma = 0u, mb_ = 0u;
  if (cur.nkt > 0) {
    gemm_load_tile<BM, AL, A_F4, GA, VEC, A_KSEQ>(ra, g.A, g.lda, cur.m0, g.M, cur.kbeg, cur.kend, g.a_maj, g.a_min, tid, ma);
    gemm_load_tile<BN, BL, B_F4, GB, VEC, B_KSEQ>(rb, g.B, g.ldb, cur.n0, g.N, cur.kbeg, cur.kend, g.b_maj, g.b_min, tid, mb_);
  }
  for (int kt = 0; kt < cur.nkt; ++kt) {
    unsigned short* Ah = reinterpret_cast<unsigned short*>(smem);
    unsigned short* Bh = Ah + BM * LDKB;
    if (BF == 1) {
      gemm_store_tile_bf16<BM, AL, A_F4, A_KSEQ>(ra, ma, Ah, tid);
      gemm_store_tile_bf16<BN, BL, B_F4, B_KSEQ>(rb, mb_, Bh, tid);
    } else {
      gemm_store_tile<BM, AL, A_F4>(ra, ma, As, tid);
      gemm_store_tile<BN, BL, B_F4>(rb, mb_, Bs, tid);
    }
    __syncthreads();
    if (kt + 1 < cur.nkt) {                                   // register prefetch under the MFMAs below
      const int k0 = cur.kbeg + (kt + 1) * BK;
      gemm_load_tile<BM, AL, A_F4, GA, VEC, A_KSEQ>(ra, g.A, g.lda, cur.m0, g.M, k0, cur.kend, g.a_maj, g.a_min, tid, ma);
      gemm_load_tile<BN, BL, B_F4, GB, VEC, B_KSEQ>(rb, g.B, g.ldb, cur.n0, g.N, k0, cur.kend, g.b_maj, g.b_min, tid, mb_);
    }
    __builtin_amdgcn_sched_barrier(0);                        // keep the loads AHEAD of the MFMAs (hipcc sinks them otherwise)
    if (BF == 1) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {                        // two 16-k MFMA steps per 32-k tile
        bf16x8 a8[MB], b8[NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          a8[mb] = *reinterpret_cast<const bf16x8*>(&Ah[(wm * (BM / WGM) + mb * 32 + l31) * LDKB + ks * 16 + half * 8]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          b8[nb] = *reinterpret_cast<const bf16x8*>(&Bh[(wn * (BN / WGN) + nb * 32 + l31) * LDKB + ks * 16 + half * 8]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[mb], b8[nb], acc[mb][nb], 0, 0, 0);
      }
    } else
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      float a[MB][4], b[NB][4];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int row = wm * (BM / WGM) + mb * 32 + l31;
        if (AL == 0) {
          const float4 t = *reinterpret_cast<const float4*>(&As[row * LDK + kg * 8 + half * 4]);
          a[mb][0] = t.x; a[mb][1] = t.y; a[mb][2] = t.z; a[mb][3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[mb][j] = As[(kg * 8 + half * 4 + j) * LDMA + row];
        }
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int row = wn * (BN / WGN) + nb * 32 + l31;
        if (BL == 0) {
          const float4 t = *reinterpret_cast<const float4*>(&Bs[row * LDK + kg * 8 + half * 4]);
          b[nb][0] = t.x; b[nb][1] = t.y; b[nb][2] = t.z; b[nb][3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[nb][j] = Bs[(kg * 8 + half * 4 + j) * LDMB + row];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb][j], b[nb][j], acc[mb][nb], 0, 0, 0);
    }
    __syncthreads();
  }

  gemm_epilogue<BM, BN, WGM, WGN, SC, MB, NB>(acc, g, cur, wm, wn, l31, half,
                                              reinterpret_cast<unsigned short*>(smem) + wave * 32 * EPI_STAGE_LD);
}

// ---- fp32 tile kernel with a direct-to-LDS operand pipeline (round 5) -----------------------------------------------------
// gemm_f32_kernel's k-step is: registers -> LDS (ds_write_b128), barrier, global loads of tile t+1 into registers, MFMAs,
// barrier.  Here the operand stream never touches a register:
//   * global_load_lds_dwordx4 (LDS-DMA) writes each 1-KiB piece of a tile straight into one of NS LDS stages, one barrier per
//     k-tile; tile t+NS-1 is issued right behind the barrier that retires tile t-1's reads, and a COUNTED s_waitcnt vmcnt leaves
//     the younger tiles in flight across it (raw s_barrier: __syncthreads() would drain them);
//   * an LDS-DMA lands lane-linear (wave-uniform base + 16 B x lane), so the conflict-free image of a k-contiguous operand -
//     [row][32 k] with the 16-byte chunk index XORed with (row >> 1) & 7 - is produced by permuting the SOURCE address of
//     each lane inside its row's 128-byte line (coalescing unchanged) and un-permuting in the fragment read; a row-contiguous
//     operand's [k][row] image is lane-linear as it is;
//   * the fragment reads are inline asm: hipcc treats an LDS-DMA in flight as an LDS store every ds_read may alias and puts
//     s_waitcnt vmcnt(0) in front of the first fragment read of a k-tile, which serialises the stream with the MFMAs
//     (scripts/microbench/gemm_lab.hip, profiles/r05_gemm_lab.txt: 117.9 -> 122.7 TF/s at 4096^3, production loop 110.8);
//     fragments are double-buffered by 8-k group, lgkmcnt counted by hand, sched_barrier(0) behind every wait.
// Same tiles, work-item order, k -> (MFMA step, lane half) map and epilogue as gemm_f32_kernel: results are BIT-IDENTICAL to
// it (tests/test_gpu_primitives.py::test_gemm_dma_equals_register_staged_loop).  16-byte-load shapes only (the host routes
// the rest to gemm_f32_kernel); edge rows read a clamped row, k-groups past the end of a ragged last tile a page of zeros.
template <int BM, int BN, int AL, int BL, bool GA, bool GB, int WGM, int WGN, bool SC, int NS>
__global__ void __launch_bounds__(256) gemm_dma_kernel(const GemmKArgs g) {
  static_assert(WGM * WGN == 4 && BM % (32 * WGM) == 0 && BN % (32 * WGN) == 0, "wave grid must tile the block tile");
  static_assert(NS >= 2 && NS <= 3, "two or three LDS stages");
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MB = WM / 32, NB = WN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int RD = MB * DmaFrag<AL>::READS + NB * DmaFrag<BL>::READS;       // LDS reads per 8-k group
  static_assert(RD <= 15, "lgkmcnt is a 4-bit counter");
  static_assert((AL == 0 || (MB - 1) * 32 + BM <= 255) && (BL == 0 || (NB - 1) * 32 + BN <= 255), "ds_read2_b32 offsets are 8 bits");
  extern __shared__ __attribute__((aligned(1024))) unsigned char dma_smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)dma_smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WGN, wn = wave % WGN;
  const GemmItem cur = gemm_decode<BM, BN>(g, blockIdx.x);

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (cur.nkt > 0) {
    DmaOperand<BM, AL, GA> sa;
    DmaOperand<BN, BL, GB> sb;
    sa.init(g.A, g.lda, cur.m0, g.M, g.a_maj, g.a_min, wave, lane);
    sb.init(g.B, g.ldb, cur.n0, g.N, g.b_maj, g.b_min, wave, lane);
    constexpr int NPT = DmaOperand<BM, AL, GA>::NP + DmaOperand<BN, BL, GB>::NP;       // LDS-DMA instructions per wave and tile

    // fragment addresses (bytes from the start of a stage).  Layout 0: row * 128 + ((2 kg + half) ^ swz) * 16 with
    // swz = (row >> 1) & 7 = (l31 >> 1) & 7 (block offsets are multiples of 32 rows); one VGPR per kg, blocks by immediate.
    // Layout 1: ((8 kg + 4 half + j) * ROWS + row) * 4; ds_read2_b32 takes (j, j+1) with dword offsets (32 mb, 32 mb + ROWS).
    const int swz = (l31 >> 1) & 7;
    unsigned fa[4], fb[4];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      fa[kg] = AL == 0 ? lds0 + (wm * WM + l31) * 128 + (((2 * kg + half) ^ swz) << 4)
                       : lds0 + (((8 * kg + 4 * half) * BM + wm * WM + l31) << 2);
      fb[kg] = BL == 0 ? lds0 + A_BYTES + (wn * WN + l31) * 128 + (((2 * kg + half) ^ swz) << 4)
                       : lds0 + A_BYTES + (((8 * kg + 4 * half) * BN + wn * WN + l31) << 2);
    }
    auto read_frags = [&](DmaFrag<AL> (&a)[MB], DmaFrag<BL> (&b)[NB], int kg, unsigned so) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        if constexpr (AL == 0) {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[mb].v) : "v"(fa[kg] + so), "n"(mb * 32 * 128));
        } else {
          asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(a[mb].lo) : "v"(fa[kg] + so), "n"(mb * 32), "n"(mb * 32 + BM));
          asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(a[mb].hi) : "v"(fa[kg] + so + 8 * BM), "n"(mb * 32), "n"(mb * 32 + BM));
        }
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if constexpr (BL == 0) {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[nb].v) : "v"(fb[kg] + so), "n"(nb * 32 * 128));
        } else {
          asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(b[nb].lo) : "v"(fb[kg] + so), "n"(nb * 32), "n"(nb * 32 + BN));
          asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(b[nb].hi) : "v"(fb[kg] + so + 8 * BN), "n"(nb * 32), "n"(nb * 32 + BN));
        }
      }
    };

    // prologue: tiles 0 .. NS-2 into stages 0 .. NS-2.
    // Gathered operands (NS == 2 only): the table entry of a lane's chunk, tk[k], is an ordinary load.  hipcc does not let an
    // LDS-DMA issue while a VGPR-destination load is pending (it drains vmcnt to 0 first - seen in the .s), so the fetch for
    // tile t+2 sits at the END of step t, behind the last MFMAs, where that drain coincides with the wait the two-stage ring
    // performs anyway at the top of step t+1.
    static_assert(!(GA || GB) || NS == 2, "gathered operands take the two-stage ring");
    sa.fetch_tab(cur.kbeg, cur.kend);
    sb.fetch_tab(cur.kbeg, cur.kend);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
      if (s < cur.nkt) {
        sa.issue(cur.kbeg + s * 32, cur.kend, dma_smem + s * STAGE, wave);
        sb.issue(cur.kbeg + s * 32, cur.kend, dma_smem + s * STAGE + A_BYTES, wave);
      }
    }
    sa.fetch_tab(cur.kbeg + 32, cur.kend);
    sb.fetch_tab(cur.kbeg + 32, cur.kend);

    int stage = 0;
    for (int kt = 0; kt < cur.nkt; ++kt) {
      // this wave's pieces of tile kt have landed once at most the (NS - 2) younger tiles are outstanding
      if (NS == 3 && kt + 1 < cur.nkt) dma_wait_vm<NPT>();
      else dma_wait_vm<0>();
      __builtin_amdgcn_s_barrier();           // ... and everybody else's; every wave is also done reading stage (kt - 1) % NS
      const unsigned so = (unsigned)stage * STAGE;
      DmaFrag<AL> a[2][MB];
      DmaFrag<BL> b[2][NB];
      read_frags(a[0], b[0], 0, so);
      const int nt = kt + NS - 1;
      int ns = stage + NS - 1;
      if (ns >= NS) ns -= NS;
      // the LDS-DMA of tile kt + NS - 1 goes out right here, in front of this tile's MFMAs (stage ns was last read during step
      // kt - 1: the barrier above).  Dealing it into the shadow of the first two groups' MFMAs instead - its address arithmetic
      // and M0 writes then cost no matrix-pipe time - was measured and is SLOWER (profiles/r05_dma_issue_placement.txt: 4096^3
      // 117.4 vs 118.8 TF/s, 2500 x 1800 x 1000 103.7 vs 99.8 us, step 34.03 vs 33.84 ms): with two stages a load has one
      // k-tile to land, and a quarter to a half of it was given away; the other resident workgroup covers the issue gap.
      if (nt < cur.nkt) {
        sa.issue(cur.kbeg + nt * 32, cur.kend, dma_smem + ns * STAGE, wave);
        sb.issue(cur.kbeg + nt * 32, cur.kend, dma_smem + ns * STAGE + A_BYTES, wave);
      }
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        const int c = kg & 1, n = c ^ 1;
        if (kg < 3) {
          read_frags(a[n], b[n], kg + 1, so);
          dma_wait_lgkm<RD>();
        } else dma_wait_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][mb].get(j), b[c][nb].get(j), acc[mb][nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if ((GA || GB) && nt + 1 < cur.nkt) {
        sa.fetch_tab(cur.kbeg + (nt + 1) * 32, cur.kend);
        sb.fetch_tab(cur.kbeg + (nt + 1) * 32, cur.kend);
      }
      if (++stage == NS) stage = 0;
    }
  }
  unsigned short* stg = nullptr;
  if (g.Ch) {      // (uniform) the twin leaves through LDS: every wave must be done reading its fragments first
    __syncthreads();
    stg = reinterpret_cast<unsigned short*>(dma_smem) + wave * 32 * EPI_STAGE_LD;
  }
  gemm_epilogue<BM, BN, WGM, WGN, SC, MB, NB>(acc, g, cur, wm, wn, l31, half, stg);
}

// ---- software-pipelined tile kernel for bf16 operands held as fp32 in memory ------------------------------------------
// gemm_f32_kernel's loop (store tile t, barrier, issue the loads of t+1, MFMAs of t, barrier) hides a load behind ONE tile's
// MFMAs.  That is right for the fp32 MFMA (4096 matrix-pipe cycles per 128x128x32 tile) and wrong for the bf16 pipe (256
// cycles): measured with that loop, bf16 operands reach 385 TF/s at 4096^3 (15 % of the pipe) - the k-step is as long as the
// load -> convert -> LDS chain, not the MFMAs.
// Here:
//   * two LDS slots and two register sets: while the MFMAs of tile t read slot t&1, the registers holding tile t+1 (loaded
//     during step t-1) are converted into slot (t+1)&1 and the loads of tile t+2 are in flight - every load has a whole
//     k-step to land, conversions and LDS writes sit in the MFMAs' issue shadow, ONE barrier per k-step;
//   * no validity masks anywhere: rows / columns past the edge read a clamped (valid) row and only feed output elements the
//     epilogue never stores; the k >= K groups of a ragged LAST k-tile are read from a page of zeros;
//   * per-thread element offsets are computed once; a k-step adds a uniform stride (or one table look-up for gathered
//     operands) instead of re-deriving addresses.
// Same LDS images, fragment reads, work-item order and epilogue as gemm_f32_kernel; 16-byte loads only (the host routes
// scalar-load shapes to gemm_f32_kernel).
template <int ROWS, int LAYOUT, int NF4, bool GATHER, bool KSEQ>
struct PipeOperand {
  const float* P;
  const int* tvar;          // gathered operands: the table indexed by k (layout 0: minor table, layout 1: major table)
  int ld;
  int off[NF4];             // this thread's fixed element offset per 16-byte load
  int kin[NF4];             // k index inside a tile of each load (layout 0: the same for all of them)
  __device__ __forceinline__ void init(const float* P_, int ld_, int row0, int nrows, const int* tmaj, const int* tmin, int tid) {
    P = P_; ld = ld_;
    tvar = LAYOUT == 0 ? tmin : tmaj;
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int f = tid + i * 256;
      if (LAYOUT == 0) {
        const int row = min(row0 + (f >> 3), nrows - 1);
        kin[i] = (f & 7) << 2;
        off[i] = GATHER ? tmaj[row] : row * ld + kin[i];
      } else {
        constexpr int F4_PER_K = ROWS / 4;
        kin[i] = KSEQ ? (tid / F4_PER_K) * NF4 + i : f / F4_PER_K;
        const int col = min(row0 + (((KSEQ ? tid : f) % F4_PER_K) << 2), nrows - 4);
        off[i] = GATHER ? tmin[col] : kin[i] * ld + col;
      }
    }
  }
  // tile starting at k0: a 16-byte group with k >= kend (ragged last tile) is read from a page of zeros instead, so nothing
  // downstream needs a mask (groups are wholly inside or outside: K % 4 == 0 on this path); branch-free
  __device__ __forceinline__ void load(float4 (&r)[NF4], int k0, int kend) const {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const bool ok = k0 + kin[i] < kend;
      size_t at;
      if (LAYOUT == 0) at = GATHER ? (size_t)off[i] + (size_t)tvar[ok ? k0 + kin[i] : 0] : (size_t)off[i] + (size_t)k0;
      else at = GATHER ? (size_t)off[i] + (size_t)tvar[ok ? k0 + kin[i] : 0] : (size_t)k0 * ld + (size_t)off[i];
      // (explicit global address space: a select between two generic pointers would make this a flat load)
      typedef float f32x4n __attribute__((ext_vector_type(4)));
      typedef const __attribute__((address_space(1))) f32x4n* gptr4;
      const uintptr_t src = ok ? (uintptr_t)(P + at) : (uintptr_t)dm_zero_page;
      const f32x4n v = *reinterpret_cast<gptr4>(src);
      r[i] = make_float4(v.x, v.y, v.z, v.w);
    }
  }
};

// Registers -> LDS in UNITS of one operand pair (2 fp32 -> one packed bf16 pair), so the kernel can deal
// them one at a time into the issue shadow of its MFMAs.  Same LDS image as gemm_store_tile_bf16.
// Units run in order u = 0, 1, ...; an even unit may leave its packed pieces in `hold` for the odd unit that completes the
// 8-byte LDS write.
template <int ROWS, int LAYOUT, int NF4, bool KSEQ>
struct PipeStash {
  static constexpr bool ROWWISE = LAYOUT == 1 && KSEQ;                 // one packed write per row: units walk the 4 rows
  static constexpr int UNITS = ROWWISE ? (NF4 == 4 ? 8 : 4) : 2 * NF4;
  static __device__ __forceinline__ float comp(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
  static __device__ __forceinline__ unsigned pack(float x, float y) { return dm_pack_bf16x2(x, y); }
  static __device__ __forceinline__ void unit(const float4 (&rr)[NF4], int u, unsigned short* S, int tid, unsigned& hold) {
    if (ROWWISE) {
      static_assert(!ROWWISE || NF4 == 4 || NF4 == 2, "row-wise packing covers 64- and 128-row operands");
      constexpr int F4_PER_K = ROWS / 4;
      const int kq = (tid / F4_PER_K) * NF4, r4 = (tid % F4_PER_K) << 2;
      if (NF4 == 4) {
        const int j = u >> 1;
        if ((u & 1) == 0) { hold = pack(comp(rr[0], j), comp(rr[1], j)); return; }
        const unsigned q = pack(comp(rr[NF4 > 2 ? 2 : 0], j), comp(rr[NF4 - 1], j));
        unsigned short* d = &S[(r4 + j) * LDKB + kq];
        *reinterpret_cast<uint2*>(d) = make_uint2(hold, q);
      } else {
        const unsigned q = pack(comp(rr[0], u), comp(rr[NF4 - 1], u));
        unsigned short* d = &S[(r4 + u) * LDKB + kq];
        *reinterpret_cast<unsigned*>(d) = q;
      }
      return;
    }
    const int i = u >> 1, f = tid + i * 256;
    if (LAYOUT == 0) {
      if ((u & 1) == 0) { hold = pack(rr[i].x, rr[i].y); return; }
      const unsigned q = pack(rr[i].z, rr[i].w);
      unsigned short* d = &S[(f >> 3) * LDKB + ((f & 7) << 2)];
      *reinterpret_cast<uint2*>(d) = make_uint2(hold, q);
    } else {      // 4 rows at one k: transposing 2-byte stores, two rows per unit
      constexpr int F4_PER_K = ROWS / 4;
      const int kr = f / F4_PER_K, r4 = ((f % F4_PER_K) << 2) + 2 * (u & 1);
      const unsigned q = (u & 1) ? pack(rr[i].z, rr[i].w) : pack(rr[i].x, rr[i].y);
      unsigned short* d = &S[r4 * LDKB + kr];
      d[0] = (unsigned short)(q & 0xFFFFu);
      d[LDKB] = (unsigned short)(q >> 16);
    }
  }
};

template <int BM, int BN, int AL, int BL, bool GA, bool GB, int WGM, int WGN, bool SC>
__global__ void __launch_bounds__(256) gemm_pipe_kernel(const GemmKArgs g) {
  static_assert(WGM * WGN == 4 && BM % (32 * WGM) == 0 && BN % (32 * WGN) == 0, "wave grid must tile the block tile");
  constexpr int BK = 32;
  constexpr int MB = BM / (32 * WGM), NB = BN / (32 * WGN);
  constexpr int A_F4 = BM * BK / 4 / 256, B_F4 = BN * BK / 4 / 256;
  constexpr bool A_KSEQ = AL == 1 && 256 % (BM / 4) == 0, B_KSEQ = BL == 1 && 256 % (BN / 4) == 0;
  constexpr int SLOT = (BM + BN) * LDKB;                       // bf16 elements per LDS slot
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * SLOT];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WGN, wn = wave % WGN;
  const GemmItem cur = gemm_decode<BM, BN>(g, blockIdx.x);

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (cur.nkt > 0) {
    PipeOperand<BM, AL, A_F4, GA, A_KSEQ> sa;
    PipeOperand<BN, BL, B_F4, GB, B_KSEQ> sb;
    sa.init(g.A, g.lda, cur.m0, g.M, g.a_maj, g.a_min, tid);
    sb.init(g.B, g.ldb, cur.n0, g.N, g.b_maj, g.b_min, tid);
    float4 ra[A_F4], rb[B_F4];                      // ONE register set: tile t+1 while step t runs (see `step`)
    typedef PipeStash<BM, AL, A_F4, A_KSEQ> SA;
    typedef PipeStash<BN, BL, B_F4, B_KSEQ> SB;
    constexpr int N_MFMA = 2 * MB * NB;
    // One k-step = the MFMAs of the tile in LDS slot `rs`, in a hand-made issue order (sched_barrier(0) pins it; left alone
    // the compiler emits all MFMAs first and the conversions behind them, and an in-order wave then leaves the matrix pipe
    // idle for the whole conversion - an MFMA occupies the pipe for 32 cycles = ~8 issue slots, which is where they go):
    //   MFMAs [N/8, N/2):  the conversion units of operand A of tile t+1 (registers `ra`) -> slot `ws`, then the loads of A of
    //                      tile t+2 into the same registers;
    //   MFMAs [N/2, 7N/8): the same for operand B.
    // A load is issued ~5/8 of a k-step before its first use; one register set serves all steps, so the loop body exists
    // once and the accumulators never move (a 2x-unrolled two-set loop made the allocator shuffle all 64 of them per step).
    auto step = [&](auto stash_tag, auto fetch_tag, int rs, int ws, int k_next2) {
      constexpr bool STASH = decltype(stash_tag)::value, FETCH = decltype(fetch_tag)::value;
      const unsigned short* Ah = smem + rs * SLOT;
      const unsigned short* Bh = Ah + BM * LDKB;
      unsigned short* Aw = smem + ws * SLOT;
      unsigned short* Bw = Aw + BM * LDKB;
      unsigned hold = 0u;
      constexpr int A_LO = N_MFMA / 8, A_HI = N_MFMA / 2, B_LO = N_MFMA / 2, B_HI = N_MFMA - N_MFMA / 8;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {                        // two 16-k MFMA steps per 32-k tile
        bf16x8 a8[MB], b8[NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          a8[mb] = *reinterpret_cast<const bf16x8*>(&Ah[(wm * (BM / WGM) + mb * 32 + l31) * LDKB + ks * 16 + half * 8]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          b8[nb] = *reinterpret_cast<const bf16x8*>(&Bh[(wn * (BN / WGN) + nb * 32 + l31) * LDKB + ks * 16 + half * 8]);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              const int j = (ks * MB + mb) * NB + nb;                         // MFMA index inside the k-step
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[mb], b8[nb], acc[mb][nb], 0, 0, 0);
              if (STASH && j >= A_LO && j < A_HI) {           // units [ceil((j-LO) U / span), ceil((j+1-LO) U / span)) of A
#pragma unroll
                for (int u = 0; u < SA::UNITS; ++u)
                  if (u >= ((j - A_LO) * SA::UNITS + (A_HI - A_LO) - 1) / (A_HI - A_LO) &&
                      u < ((j + 1 - A_LO) * SA::UNITS + (A_HI - A_LO) - 1) / (A_HI - A_LO))
                    SA::unit(ra, u, Aw, tid, hold);
                if (FETCH && j == A_HI - 1) sa.load(ra, k_next2, cur.kend);
              }
              if (STASH && j >= B_LO && j < B_HI) {
#pragma unroll
                for (int u = 0; u < SB::UNITS; ++u)
                  if (u >= ((j - B_LO) * SB::UNITS + (B_HI - B_LO) - 1) / (B_HI - B_LO) &&
                      u < ((j + 1 - B_LO) * SB::UNITS + (B_HI - B_LO) - 1) / (B_HI - B_LO))
                    SB::unit(rb, u, Bw, tid, hold);
                if (FETCH && j == B_HI - 1) sb.load(rb, k_next2, cur.kend);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
      }
    };
    const std::true_type yes;
    const std::false_type no;

    {      // prologue: tile 0 -> slot 0, tile 1 -> registers
      sa.load(ra, cur.kbeg, cur.kend);
      sb.load(rb, cur.kbeg, cur.kend);
      unsigned short* Aw = smem;
      unsigned short* Bw = Aw + BM * LDKB;
      unsigned hold = 0u;
#pragma unroll
      for (int u = 0; u < SA::UNITS; ++u) SA::unit(ra, u, Aw, tid, hold);
#pragma unroll
      for (int u = 0; u < SB::UNITS; ++u) SB::unit(rb, u, Bw, tid, hold);
      if (cur.nkt > 1) {
        sa.load(ra, cur.kbeg + BK, cur.kend);
        sb.load(rb, cur.kbeg + BK, cur.kend);
      }
      __syncthreads();
    }
    int kt = 0;
    for (; kt + 2 < cur.nkt; ++kt) {                       // tile kt in slot kt&1, tile kt+1 in registers, tile kt+2 to be loaded
      step(yes, yes, kt & 1, (kt & 1) ^ 1, cur.kbeg + (kt + 2) * BK);
      __syncthreads();
    }
    if (kt + 1 < cur.nkt) {                                // second-to-last tile: nothing left to load
      step(yes, no, kt & 1, (kt & 1) ^ 1, 0);
      __syncthreads();
      ++kt;
    }
    step(no, no, kt & 1, (kt & 1) ^ 1, 0);                 // last tile
  }
  unsigned short* stage = nullptr;
  if (g.Ch) {      // (uniform) the twin leaves through LDS: every wave must be done reading its fragments first
    __syncthreads();
    stage = smem + wave * 32 * EPI_STAGE_LD;
  }
  gemm_epilogue<BM, BN, WGM, WGN, SC, MB, NB>(acc, g, cur, wm, wn, l31, half, stage);
}

// ---- bf16-STORAGE tile kernel (conf.amp with operands that already live in HBM as bf16: weight twins kept by the optimizer,
// activation twins written by the producing epilogue).  The fp32-storage bf16 kernels above run at the rate their fp32 operands
// arrive (17-18 B/clk/CU = half that in elements); here a k-tile is 64 elements = the same 128 bytes per row and per load
// instruction pattern, so every byte that arrives carries twice the MACs and nothing is converted on the way:
//   * layout 0 (k contiguous): 16-byte chunks (8 bf16) go from global memory to the [row][k] LDS image as they are;
//   * layout 1 (row index contiguous): a chunk is 8 rows at one k; a thread owns NCH consecutive k of the same 8 rows and
//     transposes them in registers (v_perm) into one 8- / 4-byte LDS write per row;
//   * two LDS slots, one barrier per k-tile: the chunks of tile t+1 (in registers since the middle of step t-1) are written
//     between the two halves of step t's MFMAs and the loads of tile t+2 are issued right behind them;
//   * edge rows read a clamped row, k groups past the end read a page of zeros (K % 8 == 0): no masks.
// LDS row stride 72 bf16 (144 bytes): the ds_read_b128 fragment reads of 16 consecutive rows tile the 64 banks exactly.
// The 16-byte k-chunks of a row are XOR-swizzled with bits 4-6 of the row index (hswz): the transposing writes of a layout-1
// operand come from lanes 8 rows apart (8 * 144 B = 32 banks), which without the swizzle land 8-way on two bank groups.
constexpr int LDH = 72;
typedef unsigned int dm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int hswz(int row, int k) { return (k & 7) | ((((k >> 3) ^ (row >> 4)) & 7) << 3); }   // k in [0,64)
template <int ROWS, int LAYOUT, bool GATHER>
struct HOperand {
  static constexpr int NCH = ROWS * 8 / 256;                 // 16-byte chunks per thread per 64-k tile
  static constexpr int CPK = ROWS / 8;                       // layout 1: chunks per k
  static constexpr bool KSEQ = LAYOUT == 1 && 256 % CPK == 0 && (256 / CPK) * NCH == 64 && (NCH == 2 || NCH == 4);
  const unsigned short* P;
  const int* tvar;
  int ld;
  int off[NCH], kin[NCH];
  __device__ __forceinline__ void init(const unsigned short* P_, int ld_, int row0, int nrows, const int* tmaj, const int* tmin,
                                       int tid) {
    P = P_; ld = ld_;
    tvar = LAYOUT == 0 ? tmin : tmaj;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int f = tid + i * 256;
      if (LAYOUT == 0) {
        const int row = min(row0 + (f >> 3), nrows - 1);
        kin[i] = (f & 7) << 3;
        off[i] = GATHER ? tmaj[row] : row * ld + kin[i];
      } else {
        kin[i] = KSEQ ? (tid / CPK) * NCH + i : f / CPK;
        const int col = min(row0 + (((KSEQ ? tid : f) % CPK) << 3), nrows - 8);
        off[i] = GATHER ? tmin[col] : kin[i] * ld + col;
      }
    }
  }
  __device__ __forceinline__ void load(dm_u32x4 (&r)[NCH], int k0, int kend, bool nt = false) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const bool ok = k0 + kin[i] < kend;
      size_t at;
      if (LAYOUT == 0) at = GATHER ? (size_t)off[i] + (size_t)tvar[ok ? k0 + kin[i] : 0] : (size_t)off[i] + (size_t)k0;
      else at = GATHER ? (size_t)off[i] + (size_t)tvar[ok ? k0 + kin[i] : 0] : (size_t)k0 * ld + (size_t)off[i];
      typedef const __attribute__((address_space(1))) dm_u32x4* gptr4;
      const uintptr_t src = ok ? (uintptr_t)(P + at) : (uintptr_t)dm_zero_page;
      if (nt) r[i] = __builtin_nontemporal_load(reinterpret_cast<gptr4>(src));
      else r[i] = *reinterpret_cast<gptr4>(src);
    }
  }
  __device__ __forceinline__ void stash(const dm_u32x4 (&r)[NCH], unsigned short* S, int tid) const {
    if (LAYOUT == 0) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int f = tid + i * 256;
        *reinterpret_cast<dm_u32x4*>(&S[(f >> 3) * LDH + hswz(f >> 3, (f & 7) << 3)]) = r[i];
      }
    } else if (KSEQ) {       // r[i][p] = rows (r8 + 2p, r8 + 2p + 1) at k = kq + i
      const int kq = (tid / CPK) * NCH, r8 = (tid % CPK) << 3;
#pragma unroll
      for (int p2 = 0; p2 < 4; ++p2) {
        unsigned short* d0 = &S[(r8 + 2 * p2) * LDH + hswz(r8, kq)];      // rows r8 .. r8+7 share row >> 4
        const unsigned lo01 = __builtin_amdgcn_perm(r[NCH > 1 ? 1 : 0][p2], r[0][p2], 0x05040100u);
        const unsigned hi01 = __builtin_amdgcn_perm(r[NCH > 1 ? 1 : 0][p2], r[0][p2], 0x07060302u);
        if (NCH == 4) {
          const unsigned lo23 = __builtin_amdgcn_perm(r[NCH - 1][p2], r[NCH > 2 ? 2 : 0][p2], 0x05040100u);
          const unsigned hi23 = __builtin_amdgcn_perm(r[NCH - 1][p2], r[NCH > 2 ? 2 : 0][p2], 0x07060302u);
          *reinterpret_cast<uint2*>(d0) = make_uint2(lo01, lo23);
          *reinterpret_cast<uint2*>(d0 + LDH) = make_uint2(hi01, hi23);
        } else {
          *reinterpret_cast<unsigned*>(d0) = lo01;
          *reinterpret_cast<unsigned*>(d0 + LDH) = hi01;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int f = tid + i * 256;
        const int kk = f / CPK, r8 = (f % CPK) << 3;
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
          S[(r8 + 2 * p2) * LDH + hswz(r8, kk)] = (unsigned short)(r[i][p2] & 0xFFFFu);
          S[(r8 + 2 * p2 + 1) * LDH + hswz(r8, kk)] = (unsigned short)(r[i][p2] >> 16);
        }
      }
    }
  }
};

template <int BM, int BN, int AL, int BL, bool GA, bool GB, int WGM, int WGN, bool SC>
__global__ void __launch_bounds__(256, 2) gemm_h_kernel(const GemmKArgs g) {
  static_assert(WGM * WGN == 4 && BM % (32 * WGM) == 0 && BN % (32 * WGN) == 0, "wave grid must tile the block tile");
  constexpr int BK = 64;
  constexpr int MB = BM / (32 * WGM), NB = BN / (32 * WGN);
  constexpr int SLOT = (BM + BN) * LDH;
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * SLOT];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WGN, wn = wave % WGN;
  const GemmItem cur = gemm_decode<BM, BN>(g, blockIdx.x);
  const int nkt = cur.kend > cur.kbeg ? (cur.kend - cur.kbeg + BK - 1) / BK : 0;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nkt > 0) {
    typedef HOperand<BM, AL, GA> OA;
    typedef HOperand<BN, BL, GB> OB;
    OA sa;
    OB sb;
    sa.init(reinterpret_cast<const unsigned short*>(g.A), g.lda, cur.m0, g.M, g.a_maj, g.a_min, tid);
    sb.init(reinterpret_cast<const unsigned short*>(g.B), g.ldb, cur.n0, g.N, g.b_maj, g.b_min, tid);
    dm_u32x4 ra[OA::NCH], rb[OB::NCH];
    const bool nt = (g.flags & (1 << 20)) != 0;        // experiment switch DM_GEMM_H_NT: non-temporal operand loads
    sa.load(ra, cur.kbeg, cur.kend, nt);
    sb.load(rb, cur.kbeg, cur.kend, nt);
    sa.stash(ra, smem, tid);
    sb.stash(rb, smem + BM * LDH, tid);
    if (nkt > 1) {
      sa.load(ra, cur.kbeg + BK, cur.kend, nt);
      sb.load(rb, cur.kbeg + BK, cur.kend, nt);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const unsigned short* Ah = smem + (kt & 1) * SLOT;
      const unsigned short* Bh = Ah + BM * LDH;
      unsigned short* Aw = smem + ((kt & 1) ^ 1) * SLOT;
      unsigned short* Bw = Aw + BM * LDH;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 a8[MB], b8[NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          a8[mb] = *reinterpret_cast<const bf16x8*>(&Ah[(wm * (BM / WGM) + mb * 32 + l31) * LDH +
                                                        hswz(wm * (BM / WGM) + mb * 32 + l31, ks * 16 + half * 8)]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          b8[nb] = *reinterpret_cast<const bf16x8*>(&Bh[(wn * (BN / WGN) + nb * 32 + l31) * LDH +
                                                        hswz(wn * (BN / WGN) + nb * 32 + l31, ks * 16 + half * 8)]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[mb], b8[nb], acc[mb][nb], 0, 0, 0);
        if (ks == 1) {      // middle of the step: tile kt+1 registers -> the other slot, then the loads of tile kt+2
          __builtin_amdgcn_sched_barrier(0);
          if (kt + 1 < nkt) {
            sa.stash(ra, Aw, tid);
            sb.stash(rb, Bw, tid);
          }
          if (kt + 2 < nkt) {
            sa.load(ra, cur.kbeg + (kt + 2) * BK, cur.kend, nt);
            sb.load(rb, cur.kbeg + (kt + 2) * BK, cur.kend, nt);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    }
  }
  gemm_epilogue<BM, BN, WGM, WGN, SC, MB, NB>(acc, g, cur, wm, wn, l31, half, smem + wave * 32 * EPI_STAGE_LD);
}

__global__ void __launch_bounds__(256) gemm_splitk_reduce_kernel(const GemmKArgs g) {
  const size_t total = (size_t)g.M * g.N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / g.N);
    const int col = (int)(i % g.N);
    // the partials are added in split order (deterministic); the loads of 8 splits are issued together - one load in flight
    // per thread made this kernel latency-bound (19 splits of 400 x 1624: 30 us = 1.6 TB/s)
    const float* __restrict__ pp = g.partial + i;
    float v = 0.f;
    int s = 0;
    for (; s + 8 <= g.nsplit; s += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = pp[(size_t)(s + u) * total];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; s < g.nsplit; ++s) v += pp[(size_t)s * total];
    if (g.row_zero && g.row_zero[row]) v = 0.f;
    if (g.bias) v += g.bias[g.bias_mod > 0 ? col % g.bias_mod : col];
    if (g.add) v += g.add[(size_t)row * g.ldadd + col];
    float* c = g.C + (size_t)row * g.ldc + col;
    if (g.flags & DM_GEMM_ACCUM) v += *c;
    if (g.flags & DM_GEMM_ELU) v = dm_elu(v);
    if (g.mulref) v *= dm_elu_grad_from_y(g.mulref[(size_t)row * g.ldmul + col]);
    *c = v;
    if (g.Ch) g.Ch[(size_t)row * g.ldc + col] = (unsigned short)dm_f2bf(v);
  }
}

// ---- optional per-launch timing (HIP events on the launch stream), used by bench.py for the roofline line -------
#include <vector>
struct GemmProf {
  bool on = false;
  size_t cap = 0, n = 0;
  std::vector<hipEvent_t> ev;      // 2 per slot
  std::vector<long long> shape;    // 5 per slot: M, N, K, split count, gather/scatter flags (bench.py --shape-table)
  std::vector<double> flops, bytes;
  std::vector<int> kind;
};
static GemmProf g_prof;
static int prof_before(int kind, double flops, double bytes, hipStream_t st) {
  if (!g_prof.on || g_prof.n >= g_prof.cap) return -1;
  const int slot = (int)g_prof.n++;
  g_prof.flops[slot] = flops;
  g_prof.bytes[slot] = bytes;
  g_prof.kind[slot] = kind;
  (void)hipEventRecord(g_prof.ev[2 * slot], st);
  return slot;
}
static void prof_after(int slot, hipStream_t st) {
  if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], st);
}
// other translation units (panel.hip: kinds 20 = row-panel forward, 21 = row-panel backward; mlp_chain.hip: 22)
bool dm_prof_active() { return g_prof.on; }
int dm_prof_slot_begin(int kind, double flops, double bytes, hipStream_t st) { return prof_before(kind, flops, bytes, st); }
void dm_prof_slot_end(int slot, hipStream_t st) { prof_after(slot, st); }
extern "C" int dm_prof_begin(int max_launches) {
  DM_REQUIRE(max_launches > 0 && max_launches <= (1 << 20), DM_E_SHAPE, "prof_begin: max_launches %d", max_launches);
  if ((size_t)max_launches > g_prof.ev.size() / 2) {
    const size_t have = g_prof.ev.size();
    g_prof.ev.resize(2 * (size_t)max_launches);
    for (size_t i = have; i < g_prof.ev.size(); ++i)
      if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return dm_fail(DM_E_HIP, "prof_begin: hipEventCreate failed");
  }
  g_prof.flops.assign(max_launches, 0.0);
  g_prof.bytes.assign(max_launches, 0.0);
  g_prof.kind.assign(max_launches, 0);
  g_prof.shape.assign(5 * (size_t)max_launches, 0);
  g_prof.cap = max_launches;
  g_prof.n = 0;
  g_prof.on = true;
  return DM_OK;
}
// Per-launch rows of the profiled region, BEFORE dm_prof_end: rows[i*8 + {0..7}] = {kind, M, N, K, split count, flags
// (1 gathered A, 2 gathered B, 4 scatter epilogue, 8 bf16 operands), flops, milliseconds}; M = 0 for the non-GEMM kinds
// (row panels, whole-MLP kernel).  Returns the number of rows written (<= max_rows).  Synchronises on the events.
extern "C" int dm_prof_rows(double* rows, int max_rows) {
  DM_REQUIRE(rows && max_rows > 0, DM_E_NULL, "prof_rows: null output");
  int n = 0;
  for (size_t i = 0; i < g_prof.n && n < max_rows; ++i, ++n) {
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return dm_fail(DM_E_HIP, "prof_rows: event sync failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess)
      return dm_fail(DM_E_HIP, "prof_rows: hipEventElapsedTime failed");
    double* r = rows + 8 * (size_t)n;
    r[0] = g_prof.kind[i];
    for (int j = 0; j < 5; ++j) r[1 + j] = (double)g_prof.shape[5 * i + j];
    r[6] = g_prof.flops[i]; r[7] = ms;
  }
  return n;
}
// out[kind*4 + {0,1,2,3}] = {launches, flops, milliseconds, algorithmic bytes (4*(M*K + N*K + M*N))} for
// kind = tile*4 + a_layout*2 + b_layout (tile 0: 128x128, 1: 128x64, 2: 64x64, 3: 128x96, 4: 96x128), + 24 when the launch took
// gemm_dma_kernel; 20..22 panel / whole-MLP kernels; returns the number of recorded launches
// (negative on error).  Synchronises on the recorded events.
extern "C" int dm_prof_end(double* out, int nkinds) {
  DM_REQUIRE(out && nkinds >= 44, DM_E_SHAPE, "prof_end: need room for 44 kinds");
  g_prof.on = false;
  for (int i = 0; i < nkinds * 4; ++i) out[i] = 0.0;
  for (size_t i = 0; i < g_prof.n; ++i) {
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return dm_fail(DM_E_HIP, "prof_end: event sync failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess)
      return dm_fail(DM_E_HIP, "prof_end: hipEventElapsedTime failed");
    const int k = g_prof.kind[i];
    out[k * 4 + 0] += 1.0;
    out[k * 4 + 1] += g_prof.flops[i];
    out[k * 4 + 2] += ms;
    out[k * 4 + 3] += g_prof.bytes[i];
  }
  return (int)g_prof.n;
}

template <int BM, int BN, int WGM, int WGN>
static int gemm_pipe_dispatch(const GemmKArgs& a, int al, int bl, int gather, dim3 grid, hipStream_t stream) {
  if (a.c_tab) {
    if (gather != 1 || al != 0 || bl != 0) return dm_fail(DM_E_SHAPE, "gemm: the scatter epilogue is built for a gathered A, layout (0,0), 16-byte loads");
    hipLaunchKernelGGL((gemm_pipe_kernel<BM, BN, 0, 0, true, false, WGM, WGN, true>), grid, dim3(256), 0, stream, a);
  } else if (gather == 1) {
    if (al != 0 || bl != 0) return dm_fail(DM_E_SHAPE, "gemm: gathered A is built for layout (0,0) only");
    hipLaunchKernelGGL((gemm_pipe_kernel<BM, BN, 0, 0, true, false, WGM, WGN, false>), grid, dim3(256), 0, stream, a);
  } else if (gather == 2) {
    if (al != 1 || bl != 1) return dm_fail(DM_E_SHAPE, "gemm: gathered B is built for layout (1,1) only");
    hipLaunchKernelGGL((gemm_pipe_kernel<BM, BN, 1, 1, false, true, WGM, WGN, false>), grid, dim3(256), 0, stream, a);
  } else if (al == 0 && bl == 0) hipLaunchKernelGGL((gemm_pipe_kernel<BM, BN, 0, 0, false, false, WGM, WGN, false>), grid, dim3(256), 0, stream, a);
  else if (al == 0 && bl == 1) hipLaunchKernelGGL((gemm_pipe_kernel<BM, BN, 0, 1, false, false, WGM, WGN, false>), grid, dim3(256), 0, stream, a);
  else if (al == 1 && bl == 0) hipLaunchKernelGGL((gemm_pipe_kernel<BM, BN, 1, 0, false, false, WGM, WGN, false>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((gemm_pipe_kernel<BM, BN, 1, 1, false, false, WGM, WGN, false>), grid, dim3(256), 0, stream, a);
  return DM_OK;
}
static int gemm_pipe_tiles(int tc, const GemmKArgs& a, int al, int bl, int gather, dim3 grid, hipStream_t stream) {
  if (tc == 0) return gemm_pipe_dispatch<128, 128, 2, 2>(a, al, bl, gather, grid, stream);
  if (tc == 1) return gemm_pipe_dispatch<128, 64, 2, 2>(a, al, bl, gather, grid, stream);
  if (tc == 3) return gemm_pipe_dispatch<128, 96, 4, 1>(a, al, bl, gather, grid, stream);
  if (tc == 4) return gemm_pipe_dispatch<96, 128, 1, 4>(a, al, bl, gather, grid, stream);
  return gemm_pipe_dispatch<64, 64, 2, 2>(a, al, bl, gather, grid, stream);
}

template <int BM, int BN, int WGM, int WGN>
static int gemm_h_dispatch(const GemmKArgs& a, int al, int bl, int gather, dim3 grid, hipStream_t stream) {
  if (a.c_tab) {
    if (gather != 1 || al != 0 || bl != 0) return dm_fail(DM_E_SHAPE, "gemm: the scatter epilogue is built for a gathered A, layout (0,0)");
    hipLaunchKernelGGL((gemm_h_kernel<BM, BN, 0, 0, true, false, WGM, WGN, true>), grid, dim3(256), 0, stream, a);
  } else if (gather == 1) {
    if (al != 0 || bl != 0) return dm_fail(DM_E_SHAPE, "gemm: gathered A is built for layout (0,0) only");
    hipLaunchKernelGGL((gemm_h_kernel<BM, BN, 0, 0, true, false, WGM, WGN, false>), grid, dim3(256), 0, stream, a);
  } else if (gather == 2) {
    if (al != 1 || bl != 1) return dm_fail(DM_E_SHAPE, "gemm: gathered B is built for layout (1,1) only");
    hipLaunchKernelGGL((gemm_h_kernel<BM, BN, 1, 1, false, true, WGM, WGN, false>), grid, dim3(256), 0, stream, a);
  } else if (al == 0 && bl == 0) hipLaunchKernelGGL((gemm_h_kernel<BM, BN, 0, 0, false, false, WGM, WGN, false>), grid, dim3(256), 0, stream, a);
  else if (al == 0 && bl == 1) hipLaunchKernelGGL((gemm_h_kernel<BM, BN, 0, 1, false, false, WGM, WGN, false>), grid, dim3(256), 0, stream, a);
  else if (al == 1 && bl == 0) hipLaunchKernelGGL((gemm_h_kernel<BM, BN, 1, 0, false, false, WGM, WGN, false>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((gemm_h_kernel<BM, BN, 1, 1, false, false, WGM, WGN, false>), grid, dim3(256), 0, stream, a);
  return DM_OK;
}
static int gemm_h_tiles(int tc, const GemmKArgs& a, int al, int bl, int gather, dim3 grid, hipStream_t stream) {
  if (tc == 0) return gemm_h_dispatch<128, 128, 2, 2>(a, al, bl, gather, grid, stream);
  if (tc == 1) return gemm_h_dispatch<128, 64, 2, 2>(a, al, bl, gather, grid, stream);
  if (tc == 3) return gemm_h_dispatch<128, 96, 4, 1>(a, al, bl, gather, grid, stream);
  if (tc == 4) return gemm_h_dispatch<96, 128, 1, 4>(a, al, bl, gather, grid, stream);
  return gemm_h_dispatch<64, 64, 2, 2>(a, al, bl, gather, grid, stream);
}

// ---- gemm_dma_kernel launch: dynamic LDS (NS stages of (BM + BN) x 128 B), attribute set once per instantiation
static int g_dma_enabled = getenv("DM_GEMM_DMA") ? atoi(getenv("DM_GEMM_DMA")) : 1;
// 1 / 0: the direct-to-LDS main loop for fp32 16-byte-load products on / off (A/B and the bit-identity test), -1: query.
extern "C" int dm_gemm_dma_enable(int on) {
  if (on >= 0) g_dma_enabled = on > 2 ? 2 : on;      // 2: also for products of a few k-tiles (the bit-identity test)
  return g_dma_enabled;
}
template <void (*KERN)(const GemmKArgs)>
static int dma_launch(int lds_bytes, const GemmKArgs& a, dim3 grid, hipStream_t stream) {
  // the attribute belongs to the CURRENT DEVICE's function object: one flag per device (a process that touches a second GPU -
  // single-process multi-device tests - would otherwise launch the 48-72 KiB rings there without it)
  static std::atomic<int> done[DM_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DM_MAX_DEVICES) return dm_fail(DM_E_DEVICE, "gemm: hipGetDevice failed / device %d", dev);
  if (!done[dev].load(std::memory_order_acquire)) {
    hipError_t rc = hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (rc != hipSuccess) return dm_fail(DM_E_HIP, "gemm: hipFuncSetAttribute(%d B of LDS): %s", lds_bytes, hipGetErrorString(rc));
    done[dev].store(1, std::memory_order_release);
  }
  hipLaunchKernelGGL(KERN, grid, dim3(256), (size_t)lds_bytes, stream, a);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
template <int BM, int BN, int WGM, int WGN>
static int gemm_dma_dispatch(const GemmKArgs& a, int al, int bl, int gather, dim3 grid, hipStream_t stream) {
  // stages: three where two workgroups still fit a CU's 160 KiB with them (64 x 64: 48 KiB, 128 x 64: 72 KiB), else two
  constexpr int NS = (BM + BN) * 128 * 3 <= 72 * 1024 ? 3 : 2;
  constexpr int LDS = NS * (BM + BN) * 128;
  constexpr int LDS2 = 2 * (BM + BN) * 128;       // gathered operands: two stages (see the kernel's prologue comment)
  if (a.c_tab) {
    if (gather != 1 || al != 0 || bl != 0) return dm_fail(DM_E_SHAPE, "gemm: the scatter epilogue is built for a gathered A, layout (0,0), 16-byte loads");
    return dma_launch<gemm_dma_kernel<BM, BN, 0, 0, true, false, WGM, WGN, true, 2>>(LDS2, a, grid, stream);
  }
  if (gather == 1) {
    if (al != 0 || bl != 0) return dm_fail(DM_E_SHAPE, "gemm: gathered A is built for layout (0,0) only");
    return dma_launch<gemm_dma_kernel<BM, BN, 0, 0, true, false, WGM, WGN, false, 2>>(LDS2, a, grid, stream);
  }
  if (gather == 2) {
    if (al != 1 || bl != 1) return dm_fail(DM_E_SHAPE, "gemm: gathered B is built for layout (1,1) only");
    return dma_launch<gemm_dma_kernel<BM, BN, 1, 1, false, true, WGM, WGN, false, 2>>(LDS2, a, grid, stream);
  }
  if (al == 0 && bl == 0) return dma_launch<gemm_dma_kernel<BM, BN, 0, 0, false, false, WGM, WGN, false, NS>>(LDS, a, grid, stream);
  if (al == 0 && bl == 1) return dma_launch<gemm_dma_kernel<BM, BN, 0, 1, false, false, WGM, WGN, false, NS>>(LDS, a, grid, stream);
  if (al == 1 && bl == 0) return dma_launch<gemm_dma_kernel<BM, BN, 1, 0, false, false, WGM, WGN, false, NS>>(LDS, a, grid, stream);
  return dma_launch<gemm_dma_kernel<BM, BN, 1, 1, false, false, WGM, WGN, false, NS>>(LDS, a, grid, stream);
}

// gather: 0 none, 1 = A gathered (NT: conv forward / conv-transpose backward-data), 2 = B gathered (TN: conv weight grads)
template <int BM, int BN, bool V, int WGM = 2, int WGN = 2, int BF = 0>
static int gemm_dispatch(const GemmKArgs& a, int al, int bl, int gather, dim3 grid, hipStream_t stream) {
  if constexpr (V && BF == 0) {
    if (a.use_dma) return gemm_dma_dispatch<BM, BN, WGM, WGN>(a, al, bl, gather, grid, stream);
  }
  if (a.c_tab) {
    if (gather != 1 || al != 0 || bl != 0 || !V) return dm_fail(DM_E_SHAPE, "gemm: the scatter epilogue is built for a gathered A, layout (0,0), 16-byte loads");
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, 0, 0, true, true, false, WGM, WGN, BF, true>), grid, dim3(256), 0, stream, a);
  } else if (gather == 1) {
    if (al != 0 || bl != 0) return dm_fail(DM_E_SHAPE, "gemm: gathered A is built for layout (0,0) only");
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, 0, 0, V, true, false, WGM, WGN, BF>), grid, dim3(256), 0, stream, a);
  } else if (gather == 2) {
    if (al != 1 || bl != 1) return dm_fail(DM_E_SHAPE, "gemm: gathered B is built for layout (1,1) only");
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, 1, 1, V, false, true, WGM, WGN, BF>), grid, dim3(256), 0, stream, a);
  } else if (al == 0 && bl == 0) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, 0, 0, V, false, false, WGM, WGN, BF>), grid, dim3(256), 0, stream, a);
  else if (al == 0 && bl == 1) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, 0, 1, V, false, false, WGM, WGN, BF>), grid, dim3(256), 0, stream, a);
  else if (al == 1 && bl == 0) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, 1, 0, V, false, false, WGM, WGN, BF>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, 1, 1, V, false, false, WGM, WGN, BF>), grid, dim3(256), 0, stream, a);
  return DM_OK;
}

// operand precision of the call in progress on this host thread (common.h: DmPrecisionScope)
// ---- bf16 twin map of the composite call in progress (common.h DmTwinScope)
struct TwinRange { const float* base; size_t n; unsigned short* twin; bool valid; };
static thread_local TwinRange tl_twins[48];
static thread_local int tl_ntwins = 0;
static thread_local bool tl_twins_on = false;
static int g_twins_enabled = getenv("DM_BF16_NO_TWINS") ? 0 : 1;
// 1 / 0: switch the bf16-storage operand path of bf16-mode calls on / off (A/B and parity tests), -1: query.  Returns the state.
extern "C" int dm_bf16_twins_enable(int on) {
  if (on >= 0) g_twins_enabled = on ? 1 : 0;
  return g_twins_enabled;
}
DmTwinScope::DmTwinScope(bool bf16_mode) {
  opened = !tl_twins_on;               // composite calls do not nest; an inner scope leaves the outer one's map alone
  active = bf16_mode && g_twins_enabled;
  if (opened) { tl_twins_on = active; tl_ntwins = 0; }
}
DmTwinScope::~DmTwinScope() {
  if (opened) { tl_twins_on = false; tl_ntwins = 0; }
}
bool dm_twins_on() { return tl_twins_on; }
void dm_twin_add(const float* base, size_t n, unsigned short* twin, bool valid) {
  if (!tl_twins_on || !base || !twin || n == 0 || tl_ntwins >= 48) return;
  tl_twins[tl_ntwins++] = TwinRange{base, n, twin, valid};
}
static TwinRange* twin_find(const float* p) {
  if (!tl_twins_on || !p) return nullptr;
  for (int i = tl_ntwins - 1; i >= 0; --i)
    if (p >= tl_twins[i].base && p < tl_twins[i].base + tl_twins[i].n) return &tl_twins[i];
  return nullptr;
}
void dm_twin_mark(const float* p) {
  TwinRange* r = twin_find(p);
  if (r) r->valid = true;
}
unsigned short* dm_twin_of(const float* p, bool need_valid) {
  TwinRange* r = twin_find(p);
  if (!r || (need_valid && !r->valid)) return nullptr;
  return r->twin + (p - r->base);
}

#include <mutex>
#include <unordered_map>
static std::mutex g_arena_mu;
static std::unordered_map<const void*, bool> g_arena_twins;
void dm_twin_arena_note(const void* acts, bool written) {
  std::lock_guard<std::mutex> lk(g_arena_mu);
  if (g_arena_twins.size() > 4096) g_arena_twins.clear();       // stale keys of freed buffers: forgetting only disables the twin path once
  g_arena_twins[acts] = written;
}
bool dm_twin_arena_valid(const void* acts) {
  std::lock_guard<std::mutex> lk(g_arena_mu);
  auto it = g_arena_twins.find(acts);
  return it != g_arena_twins.end() && it->second;
}

static thread_local int tl_precision = 0;
int dm_cur_precision() { return tl_precision; }
DmPrecisionScope::DmPrecisionScope(int p) : prev(tl_precision) { tl_precision = p ? 1 : 0; }
DmPrecisionScope::~DmPrecisionScope() { tl_precision = prev; }


int dm_gemm_launch(const DmGemm& q, void* ws, size_t ws_bytes, hipStream_t stream) {
  DM_REQUIRE(q.M >= 0 && q.N >= 0 && q.K >= 1, DM_E_SHAPE, "gemm: bad dims M=%d N=%d K=%d (K must be >= 1)", q.M, q.N, q.K);
  if (q.M == 0 || q.N == 0) return DM_OK;
  DM_REQUIRE(((q.A && q.B) || (q.A_h && q.B_h)) && q.C, DM_E_NULL, "gemm: null operand");
  DM_REQUIRE((q.A_h == nullptr) == (q.B_h == nullptr), DM_E_NULL, "gemm: bf16-storage operands come in pairs");
  DM_REQUIRE((unsigned)q.a_layout < 2 && (unsigned)q.b_layout < 2, DM_E_SHAPE, "gemm: bad layout");
  DM_REQUIRE(q.a_maj || q.lda >= (q.a_layout == 0 ? q.K : q.M), DM_E_SHAPE, "gemm: lda %d too small", q.lda);
  DM_REQUIRE(q.b_maj || q.ldb >= (q.b_layout == 0 ? q.K : q.N), DM_E_SHAPE, "gemm: ldb %d too small", q.ldb);
  DM_REQUIRE((q.a_maj == nullptr) == (q.a_min == nullptr) && (q.b_maj == nullptr) == (q.b_min == nullptr), DM_E_NULL,
             "gemm: gather tables must come in (major, minor) pairs");
  DM_REQUIRE(q.c_tab || q.ldc >= q.N, DM_E_SHAPE, "gemm: ldc %d < N %d", q.ldc, q.N);
  DM_REQUIRE(!q.add || q.ldadd >= q.N, DM_E_SHAPE, "gemm: ldadd %d < N %d", q.ldadd, q.N);

  {   // <= 64-row products of the sequential RSSM chains: one-launch skinny kernel (gemm_skinny.hip)
    const int sk = (q.A_h || q.C_h) ? 0 : dm_gemm_skinny_try(q, stream);
    if (sk < 0) return sk;
    if (sk == 1) {
      // the skinny kernel keeps fp32 operands and writes no twin: a registered result gets its twin by a copy pass (dense
      // results only; otherwise the twin stays invalid and consumers read the fp32 values)
      unsigned short* t = (q.bf16 && dm_twins_on()) ? dm_twin_of(q.C, false) : nullptr;
      if (t && q.ldc == q.N) {
        const DmCvtSeg sg = {q.C, t, (size_t)q.M * q.N};
        DM_TRY(dm_to_bf16_multi_launch(&sg, 1, stream));
        dm_twin_mark(q.C);
      }
      return DM_OK;
    }
  }
  DM_REQUIRE(!q.ln_g && !q.lnb_x && !q.gates && !q.eg_x && !q.lnf_ps && !q.sm_logits, DM_E_SHAPE,
             "gemm: LayerNorm prologues / the gates epilogue are built for the <= 64-row skinny products only (M=%d K=%d)", q.M, q.K);
  // bf16-storage operands: given explicitly (both or neither), or found in the call's twin map (common.h DmTwinScope) when
  // BOTH operands have a valid twin and the shape meets the 16-byte chunk rules (8 elements along the minor axis); the
  // result's twin is written whenever the map has storage for it.
  const unsigned short* Ah = q.A_h;
  const unsigned short* Bh = q.B_h;
  unsigned short* Chh = q.C_h;
  if (q.bf16 && dm_twins_on()) {
    if (!Ah && !Bh) {
      const unsigned short* ta = dm_twin_of(q.A, true);
      const unsigned short* tb = dm_twin_of(q.B, true);
      const bool fits = ta && tb && ((((uintptr_t)ta | (uintptr_t)tb) & 15) == 0) &&
                        ((q.K & 7) == 0 || (q.a_layout == 1 && q.b_layout == 1)) &&
                        (q.a_maj ? q.a_tab_vec8 != 0 : (q.lda & 7) == 0) && (q.b_maj ? q.b_tab_vec8 != 0 : (q.ldb & 7) == 0) &&
                        (q.a_layout == 0 || ((q.M & 7) == 0 && q.M >= 8)) && (q.b_layout == 0 || ((q.N & 7) == 0 && q.N >= 8));
      // measured on the step's convolution products (profiles/r03_bf16_storage.txt): weight gradients (both operands
      // row-contiguous, K = pixels) gain 1.1-1.9x, k-contiguous products gain from K ~ 1000 up and lose below it (a 64-k tile
      // halves the k-steps that amortise a tile's prologue and epilogue)
      // ...; dense k-contiguous products (the 2 500-row imagination cell) gain from K ~ 400
      const int min_k = (q.a_maj || q.b_maj || q.c_tab) ? DM_HSTORE_MIN_K : DM_HSTORE_MIN_K_DENSE;
      if (fits && ((q.a_layout == 1 && q.b_layout == 1) || q.K >= min_k)) { Ah = ta; Bh = tb; }
    }
    if (!Chh && !q.no_twin) Chh = dm_twin_of(q.C, false);
    if (Chh) dm_twin_mark(q.C);
  }
  const bool hstore = Ah && Bh;
  DM_REQUIRE(!hstore || ((((uintptr_t)Ah | (uintptr_t)Bh) & 15) == 0 && ((q.K & 7) == 0 || (q.a_layout == 1 && q.b_layout == 1)) &&
                         (q.a_maj ? q.a_tab_vec8 != 0 : (q.lda & 7) == 0) && (q.b_maj ? q.b_tab_vec8 != 0 : (q.ldb & 7) == 0) &&
                         (q.a_layout == 0 || ((q.M & 7) == 0 && q.M >= 8)) && (q.b_layout == 0 || ((q.N & 7) == 0 && q.N >= 8))),
             DM_E_SHAPE, "gemm: bf16-storage operands need 16-byte aligned rows and extents in multiples of 8 (M=%d N=%d K=%d)",
             q.M, q.N, q.K);
  GemmKArgs a;
  a.Ch = Chh;
  a.A = q.A; a.B = q.B; a.C = q.C; a.bias = q.bias; a.add = q.add; a.mulref = q.mulref; a.row_zero = q.row_zero; a.partial = nullptr;
  a.M = q.M; a.N = q.N; a.K = q.K;
  a.lda = q.lda; a.ldb = q.ldb; a.ldc = q.ldc; a.ldadd = q.ldadd; a.ldmul = q.ldmul;
  a.flags = q.flags;
  a.a_maj = q.a_maj; a.a_min = q.a_min; a.b_maj = q.b_maj; a.b_min = q.b_min;
  a.c_tab = q.c_tab; a.sc_cout = q.sc_cout; a.sc_wpitch = q.sc_wpitch; a.bias_mod = q.bias_mod;
  DM_REQUIRE(!q.c_tab || (q.sc_cout > 0 && q.N % q.sc_cout == 0 && q.N / q.sc_cout == 4 && !q.add && !(q.flags & DM_GEMM_ACCUM)),
             DM_E_SHAPE, "gemm: scatter epilogue needs N = 4 * sc_cout, no addend, no accumulate");
  if (hstore) {
    a.A = reinterpret_cast<const float*>(Ah); a.B = reinterpret_cast<const float*>(Bh);
    static const int h_nt = getenv("DM_GEMM_H_NT") ? 1 : 0;
    if (h_nt) a.flags |= 1 << 20;
  }
  // 16-byte load path: aligned base, rows a multiple of 4 floats apart, and the vectorised (minor) extent a multiple
  // of 4 so that no group of 4 straddles the edge.  Minor extent: K for layout 0, M (resp. N) for layout 1.
  a.a_vec = hstore ? 1 : (((uintptr_t)q.A & 15) == 0 && (q.a_maj ? q.a_tab_vec != 0 : (q.lda & 3) == 0) &&
             (((q.a_layout == 0 ? q.K : q.M) & 3) == 0)) ? 1 : 0;
  a.b_vec = hstore ? 1 : (((uintptr_t)q.B & 15) == 0 && (q.b_maj ? q.b_tab_vec != 0 : (q.ldb & 3) == 0) &&
             (((q.b_layout == 0 ? q.K : q.N) & 3) == 0)) ? 1 : 0;

  // ---- tile / split-K selection (deterministic in the shape only) ------------------------------------------------
  // Cost model in MACs per CU, fitted offline (scripts/fit_gemm_model.py) to per-tile timings of 28 step shapes on
  // MI355X (profiles/r01_gemm_shapes_5tiles.txt; regret 140 us of 15.4 ms summed over all shapes vs the per-shape best
  // tile):
  //   a CU holds `avg` = workgroups/256 of this launch, at most R of them resident (R = 3 / 4 / 7 by registers);
  //   per k-tile it needs max(conc * BM*BN*32 / rate, L): MFMA time of the `conc` co-resident workgroups, or the
  //   ~1 MMAC-equivalent load latency when too few workgroups are resident to cover it (long reductions on few
  //   workgroups are latency-bound: weight gradients want 128x128 tiles AND >= 2 workgroups per CU);
  //   plus K_eq k-steps of prologue/epilogue per tile, plus split-K partial traffic and its reduce launch.
  // Split-K is considered only when the grid would not fill the chip (t < 256), up to the point where the partial
  // traffic (nsplit*M*N) reaches 1/2 of the operand traffic ((M+N)*K); full grids may take a light 2/4-way split to
  // even out a ragged last wave.
  const int ktiles = dm_cdiv(q.K, 32);
  const double out_elems = (double)q.M * q.N;
  int max_split = (int)(0.5 * ((double)q.M + q.N) * q.K / (out_elems > 0 ? out_elems : 1));
  if (max_split > ktiles / 2) max_split = ktiles / 2;
  if (max_split > 512) max_split = 512;
  {
    const size_t per = (size_t)q.M * q.N * sizeof(float);
    if (ws == nullptr || q.c_tab) max_split = 1;        // the scatter epilogue writes its result directly
    else if ((size_t)max_split * per > ws_bytes) max_split = (int)(ws_bytes / per);
  }
  if (max_split < 1) max_split = 1;
  static const int cand[5][2] = {{128, 128}, {128, 64}, {64, 64}, {128, 96}, {96, 128}};
  static const double keq[5] = {150.0, 60.0, 45.0, 45.0, 45.0};
  static const double rate[5] = {1.0, 0.9, 0.8, 0.75, 0.75};
  static const double resid[5] = {3.0, 4.0, 7.0, 4.0, 4.0};
  const double lat_macs = 1.0e6;
  // tuning overrides for scripts/gemm_bench.py only (unset in production): DM_GEMM_TILE=1|2|3 forces a candidate,
  // DM_GEMM_SPLIT=n forces the split count
  static const int force_tile = getenv("DM_GEMM_TILE") ? atoi(getenv("DM_GEMM_TILE")) : 0;
  static const int force_split = getenv("DM_GEMM_SPLIT") ? atoi(getenv("DM_GEMM_SPLIT")) : 0;
  int BM = 64, BN = 64, nsplit = 1;
  double best_cost = -1.0;
  const int kt1 = ktiles > 0 ? ktiles : 1;
  const bool dma_shape_pre = !hstore && !q.bf16 && a.a_vec && a.b_vec;
  for (int c = 0; c < 5; ++c) {
    static const int sc_tile = getenv("DM_SC_TILE") ? atoi(getenv("DM_SC_TILE")) : 0;      // tuning override: scatter-epilogue products only
    if (force_tile && c != force_tile - 1) continue;
    if (sc_tile && q.c_tab && c != sc_tile - 1) continue;
    static const int plain_tile = getenv("DM_PLAIN_TILE") ? atoi(getenv("DM_PLAIN_TILE")) : 0;   // tuning override: products without gather / scatter
    if (plain_tile && !q.c_tab && !q.a_maj && !q.b_maj && c != plain_tile - 1) continue;
    static const int ga_tile = getenv("DM_GA_TILE") ? atoi(getenv("DM_GA_TILE")) : 0;      // tuning override: gathered-operand products
    if (ga_tile && !q.c_tab && (q.a_maj || q.b_maj) && c != ga_tile - 1) continue;
    if (!(a.a_vec && a.b_vec) && c != 2) continue;        // the scalar-load variant exists for the 64x64 tile only
    const int bm = cand[c][0], bn = cand[c][1];
    const int64_t t = (int64_t)dm_cdiv(q.M, bm) * dm_cdiv(q.N, bn);
    int sp_fill = 1;
    if (t < 256) {
      sp_fill = dm_cdiv(512, t);
      if (sp_fill > max_split) sp_fill = max_split;
    }
    if (sp_fill > kt1) sp_fill = kt1;
    if (sp_fill < 1) sp_fill = 1;
    // grids of a few tiles also try 1.5x and 2x the fill split: a long reduction on few tiles is latency /
    // bandwidth bound per workgroup (per_kt = lat_macs below), so more, shorter workgroups per CU finish sooner
    // (decoder layer-4 weight gradient 48 x 144 x 2.25 M: 1096 -> 700 us); DM_GEMM_NO_WIDE_SPLIT=1 restores {1, fill}
    static const int no_wide = getenv("DM_GEMM_NO_WIDE_SPLIT") ? 1 : 0;
    const bool wide = t < 16 && !no_wide;          // (at 16+ tiles the extra partial traffic loses: 400 x 400 x 40 000 196 -> 225 us)
    const int sp15 = wide ? sp_fill + sp_fill / 2 : sp_fill, sp20 = wide ? 2 * sp_fill : sp_fill;
    // ... and the BALANCED splits floor(256 k / tiles), k = 2..5: just under k workgroups on EVERY CU.  When all workgroups
    // of a launch are resident at once it ends with its fullest CU, so the load model below takes ceil(workgroups per CU)
    // there: 52 tiles x 10 splits = 520 workgroups leave 8 CUs with 3 and run 1.5x longer than 52 x 19 = 988 (3.86 per CU).
    // Measured: 400 x 1624 x 40 000 820 -> 602 us, 96 x 1728 x 422 500 1889 -> 1301 us, 96 x 768 x 490 000 928 -> 702 us,
    // no shape slower, step -0.46 ms.  DM_GEMM_NO_BALANCED_SPLIT=1 restores the previous candidate set and fractional load.
    static const int balanced = getenv("DM_GEMM_NO_BALANCED_SPLIT") ? 0 : 1;
    int sps[10] = {1, sp_fill, t >= 256 ? 2 : sp_fill, t >= 256 ? 4 : sp_fill, sp15, sp20, sp_fill, sp_fill, sp_fill, sp_fill};
    if (balanced && t < 256)
      for (int k = 2; k <= 5; ++k) sps[4 + k] = (int)((256 * k) / t) > 0 ? (int)((256 * k) / t) : 1;
    for (int pass = 0; pass < 10; ++pass) {
      int sp = force_split > 0 ? force_split : sps[pass];
      if (force_split <= 0 && sp > max_split) sp = max_split;
      if (sp > kt1) sp = kt1;
      if (sp < 1) sp = 1;
      const double avg = (double)(t * sp) / 256.0;
      const double R = resid[c];
      const double waves = avg > R ? ceil(avg / R) : 1.0;
      const double conc = avg > R ? avg / waves : (avg > 1.0 ? (balanced ? ceil(avg - 1e-9) : avg) : 1.0);
      const double nkt = (double)dm_cdiv(kt1, sp);
      const double tm = (double)bm * bn * 32.0 / rate[c];
      const double per_kt = conc * tm > lat_macs ? conc * tm : lat_macs;
      double cost = waves * (nkt * per_kt + conc * bm * bn * keq[c] / rate[c]);
      if (sp > 1) cost += 0.4 * sp * out_elems + 1.2e6;
      // scatter-epilogue products (gather-form transposed convolution): the epilogue (table look-ups, 4 parity classes) is
      // dearer per tile than the model's, which was fitted to plain stores.  Measured on the step's three such products
      // (DM_SC_TILE sweep): 562500x192x384 1176 -> 1051 us and 122500x384x768 981 -> 837 us on 64x64, 562500x192x864
      // 2214 -> 2087 us on 128x96
      if (q.a_maj && !q.c_tab && q.N <= 64 && c == 2) cost *= 0.8;      // gathered patches x <= 64 channels (DM_GA_TILE sweep): 2250000x48x144 742 -> 597 us
      if (q.c_tab) {
        if (c == 2 && q.K <= 800) cost *= 0.8;
        if (c == 3 && q.N % 96 == 0 && q.K > 800) cost *= 0.8;
      }
      // with the LDS-DMA loop (scripts/gemm_tile_sweep.py, profiles/r05_tile_sweep.txt): a row-contiguous B on the 128 x 128 tile
      // is the one combination it does not speed up (2500 x 4800 x 1536: 425 us against 342 on 128 x 64); split weight gradients
      // of the 400-wide heads run 6-14 % faster on 64 x 64 tiles (three resident workgroups per CU with the 3-stage ring)
      if (!q.a_maj && !q.b_maj && !q.c_tab && dma_shape_pre && g_dma_enabled) {
        if (c == 0 && q.a_layout == 0 && q.b_layout == 1) cost *= 1.3;
        if (c == 2 && q.a_layout == 1 && q.b_layout == 1 && sp > 1 && t < 256 && dm_cdiv(kt1, sp) >= 14) cost *= 0.8;
      }
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; BM = bm; BN = bn; nsplit = sp; }
    }
  }
  // ---- which main loop (fp32 operands, 16-byte loads): gemm_dma_kernel when a work item has enough k-tiles to amortise the
  // ring's fill - below that the register-staged loop's higher residency (one LDS stage, 4-7 workgroups per CU) hides a
  // tile's prologue and epilogue better (profiles/r05_gemm_shapes_dma.txt vs _regstaged.txt with the switch at 2: K = 144 /
  // 192 / 384 products lose 8-15 %); both loops give the same bits, so the choice is free per call.  Layout 0 clamps edge
  // rows to the last row, layout 1 to the last full group of 4: >= 1 / >= 4 rows.
  static const int dma_min_kt = getenv("DM_GEMM_DMA_MIN_KT") ? atoi(getenv("DM_GEMM_DMA_MIN_KT")) : 14;
  const bool dma_shape = !hstore && !q.bf16 && a.a_vec && a.b_vec && (q.a_layout == 0 ? q.M >= 1 : q.M >= 4) && (q.b_layout == 0 ? q.N >= 1 : q.N >= 4);
  auto dma_for = [&](int sp) { return g_dma_enabled && dma_shape && (dm_cdiv(kt1, sp) >= dma_min_kt || g_dma_enabled >= 2); };
  // (A one-round 128 x 160 tile for the rollout's 2 500 x 1 800 gate products - 240 workgroups, one per CU, instead of 1 160
  // tiles of 64 x 64 = 4.53 per CU - was built and measured: 104.8 vs 106.3 us at K = 1000, 71.3 vs 68.4 at K = 600, step 33.94
  // vs 33.50 ms.  One 4-wave workgroup per CU leaves nobody to hide its barrier skew, prologue and epilogue: removed.)
  const int tiles_m = dm_cdiv(q.M, BM), tiles_n = dm_cdiv(q.N, BN);
  const int64_t tiles = (int64_t)tiles_m * tiles_n;
  int k_per_split = dm_cdiv(ktiles > 0 ? ktiles : 1, nsplit) * 32;
  nsplit = q.K > 0 ? dm_cdiv(q.K, k_per_split) : 1;
  a.k_per_split = k_per_split;
  a.nsplit = nsplit;
  a.tiles_m = tiles_m;
  a.partial = nsplit > 1 ? (float*)ws : nullptr;

  DM_REQUIRE(tiles * nsplit < (int64_t)1 << 30, DM_E_SHAPE, "gemm: too many work items");
  a.n_tiles = (int)tiles;
  a.tiles_n = tiles_n;
  a.n_fast = tiles_n <= tiles_m ? 1 : 0;
  a.n_items = (int)(tiles * nsplit);
  {      // column groups of the fast dimension (gemm_decode): the grouping that minimises the operand rows one XCD's chunk touches
    static const int grp_env = getenv("DM_GEMM_XCD_GROUPS") ? atoi(getenv("DM_GEMM_XCD_GROUPS")) : 0;      // 0 auto, 1 off, 2 / 4 forced
    const int F = a.n_fast ? tiles_n : tiles_m, S = a.n_fast ? tiles_m : tiles_n;
    const int bf = a.n_fast ? BN : BM, bs = a.n_fast ? BM : BN;
    int best_g = 1;
    if (nsplit == 1 && tiles >= 64 && !q.c_tab) {
      double best = -1.0;
      for (int G = 1; G <= 4; G *= 2) {
        if (grp_env > 1 && G != grp_env) continue;
        if (F / G < 2) break;
        const double width = (double)F / G;
        double srows = ((double)tiles / 8.0) / width;
        if (srows > S) srows = S;
        if (srows < 1.0) srows = 1.0;
        const double cost = srows * bs + width * bf;
        if (best < 0 || cost < best * 0.95) { best = cost; best_g = G; }      // (a new grouping has to win by 5 %)
      }
    }
    a.n_groups = grp_env == 1 ? 1 : best_g;
  }
  const int tc = (BM == 128 && BN == 128) ? 0 : (BM == 128 && BN == 64) ? 1 : (BM == 64 ? 2 : (BN == 96 ? 3 : 4));
  dim3 grid((unsigned)a.n_items);
  const int gather = q.a_maj ? 1 : (q.b_maj ? 2 : 0);
  DM_REQUIRE(!(q.a_maj && q.b_maj), DM_E_SHAPE, "gemm: only one gathered operand per call");
  a.use_dma = dma_for(nsplit) ? 1 : 0;
  // profiling kinds: tile * 4 + layouts for the register-staged loop (0..19), 20..22 the panel / whole-MLP kernels (panel.hip,
  // mlp_chain.hip), 24 + (tile * 4 + layouts) for gemm_dma_kernel
  const int kind = tc * 4 + q.a_layout * 2 + q.b_layout + (a.use_dma ? 24 : 0);
  const int slot = prof_before(kind, 2.0 * q.M * q.N * (double)q.K,
                               4.0 * ((double)q.M * q.K + (double)q.N * q.K + (double)q.M * q.N), stream);
  if (slot >= 0) {
    long long* sh = &g_prof.shape[5 * (size_t)slot];
    sh[0] = q.M; sh[1] = q.N; sh[2] = q.K; sh[3] = nsplit; sh[4] = (q.a_maj ? 1 : 0) | (q.b_maj ? 2 : 0) | (q.c_tab ? 4 : 0) | (q.bf16 ? 8 : 0) |
            (hstore ? 32 : 0);
  }
  int rc;
  const bool vec = a.a_vec && a.b_vec;
  static const int no_pipe = getenv("DM_GEMM_NO_PIPE") ? 1 : 0;      // A/B switch: the single-stage loop for the bf16-pipe modes
  // the pipelined kernel clamps edge rows instead of masking them: it needs >= 1 full row (layout 0) / >= 4 (layout 1 groups)
  const bool pipe_ok = vec && !no_pipe && (q.a_layout == 0 ? q.M >= 1 : q.M >= 4) && (q.b_layout == 0 ? q.N >= 1 : q.N >= 4);
  if (hstore) rc = gemm_h_tiles(tc, a, q.a_layout, q.b_layout, gather, grid, stream);
  else if (pipe_ok && q.bf16) rc = gemm_pipe_tiles(tc, a, q.a_layout, q.b_layout, gather, grid, stream);
  else if (vec && q.bf16) {
    if (tc == 0) rc = gemm_dispatch<128, 128, true, 2, 2, 1>(a, q.a_layout, q.b_layout, gather, grid, stream);
    else if (tc == 1) rc = gemm_dispatch<128, 64, true, 2, 2, 1>(a, q.a_layout, q.b_layout, gather, grid, stream);
    else if (tc == 3) rc = gemm_dispatch<128, 96, true, 4, 1, 1>(a, q.a_layout, q.b_layout, gather, grid, stream);
    else if (tc == 4) rc = gemm_dispatch<96, 128, true, 1, 4, 1>(a, q.a_layout, q.b_layout, gather, grid, stream);
    else rc = gemm_dispatch<64, 64, true, 2, 2, 1>(a, q.a_layout, q.b_layout, gather, grid, stream);
  } else if (vec) {
    if (tc == 0) rc = gemm_dispatch<128, 128, true>(a, q.a_layout, q.b_layout, gather, grid, stream);
    else if (tc == 1) rc = gemm_dispatch<128, 64, true>(a, q.a_layout, q.b_layout, gather, grid, stream);
    else if (tc == 3) rc = gemm_dispatch<128, 96, true, 4, 1>(a, q.a_layout, q.b_layout, gather, grid, stream);
    else if (tc == 4) rc = gemm_dispatch<96, 128, true, 1, 4>(a, q.a_layout, q.b_layout, gather, grid, stream);
    else rc = gemm_dispatch<64, 64, true>(a, q.a_layout, q.b_layout, gather, grid, stream);
  } else {      // rare shapes (K = action_dim = 18, single-row weight gradients): scalar loads, smallest tile only
    rc = gemm_dispatch<64, 64, false>(a, q.a_layout, q.b_layout, gather, grid, stream);
  }
  if (rc != DM_OK) return rc;
  prof_after(slot, stream);
  DM_LAUNCH_CHECK();
  if (nsplit > 1) {
    const size_t total = (size_t)q.M * q.N;
    int blocks = dm_cdiv(total, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a);
    DM_LAUNCH_CHECK();
  }
  if (q.C_frag) {      // the skinny kernel writes this copy in its epilogue; the tiled path owes it to the chain's next product
    DM_REQUIRE(q.M <= 64 && !q.c_tab, DM_E_SHAPE, "gemm: a fragment-major copy of C needs M <= 64 (M=%d)", q.M);
    DM_TRY(dm_frag_pack_launch(q.M, q.N, q.C, q.ldc, q.C_frag, stream));
  }
  return DM_OK;
}

extern "C" int dm_gemm_f32(int a_layout, int b_layout, int M, int N, int K, const float* A, int lda,
                           const float* B, int ldb, float* C, int ldc, const float* bias, const float* add,
                           int ldadd, int flags, void* ws, size_t ws_bytes, void* stream) {
  DmPrecisionScope prec(flags & DM_GEMM_BF16);
  DmGemm g;
  g.a_layout = a_layout; g.b_layout = b_layout;
  g.M = M; g.N = N; g.K = K;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.bias = bias; g.add = add; g.ldadd = ldadd; g.flags = flags & ~DM_GEMM_BF16;
  return dm_gemm_launch(g, ws, ws_bytes, (hipStream_t)stream);
}

// bf16-STORAGE product (operands already bf16 in HBM; fp32 accumulation and fp32 result, optional bf16 twin of the result):
// the primitive behind conf.amp's weight / activation twins.  Same layouts and leading-dimension meaning (in elements) as
// dm_gemm_f32; extents along the minor axis in multiples of 8, 16-byte aligned rows.
extern "C" int dm_gemm_bf16h(int a_layout, int b_layout, int M, int N, int K, const uint16_t* A, int lda, const uint16_t* B,
                             int ldb, float* C, int ldc, uint16_t* C_h, const float* bias, int flags, void* ws, size_t ws_bytes,
                             void* stream) {
  DmPrecisionScope prec(1);
  DmGemm g;
  g.a_layout = a_layout; g.b_layout = b_layout;
  g.M = M; g.N = N; g.K = K;
  g.A_h = A; g.lda = lda; g.B_h = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.C_h = C_h;
  g.bias = bias;
  g.flags = flags & ~DM_GEMM_BF16;
  return dm_gemm_launch(g, ws, ws_bytes, (hipStream_t)stream);
}
