// The posterior T loop (RSSMCore.forward, rssm.py:38-58, over RSSMCell.forward, rssm.py:125-153, with nn.GRUCell, rnn.py:40-67)
// as ONE persistent kernel that spans every compute unit and keeps the cell's weights STATIONARY IN LDS.
//
// Why: a posterior step is four dependent <= 64-row products whose weights (18 MB fp32 at deter 600 / hidden 1000 / 32x32
// latents) every launch of the launch-chain schedule re-streams from L2 with few workgroups - 74 us per step for 0.7 GFLOP.
// 18 MB is 70 KB per CU over 256 CUs: each workgroup (one per CU) owns a 4-COLUMN slice of every layer, loads it into its LDS
// once per call and keeps it for all T steps.  What crosses CUs per step are the small activation rows only.
//
// Roles (workgroup `me` has role X iff me < nX; a role's block b is columns [4b, 4b+4) of its layer):
//   A  x1 = z_mlp(z) + a_mlp(a): gather-sum over z_mlp^T rows at the sampled indices (one-hot z, rssm.py:138-139)   nA = Hd/4
//   B  gi = ELU(in_norm(x1)) W_ih^T, GRU gates for 4 hidden units (the r, z, n columns of a unit live together)     nB = D/4
//   C  x2 = h W_post_h^T + b + post_mlp_e(e); then gh of the NEXT step from the same h registers (off the chain)     nC = Hd/4
//   D  post logits = ELU(post_norm(x2)) W_post^T + b, 4 logits                                                      nD = S*C/4
//   L  (the first D workgroup of each latent group) z ~ OneHotCategoricalStraightThrough(post): the sampler contract of
//      elementwise.hip (sequential fp32 softmax / cumsum, idx = #{k: cdf_k <= u cdf_last})                            S leaders
//
// Exchange (measured first: scripts/microbench/allgather_xcd.hip, profiles/r04_allgather.txt).  A producer publishes its
// (rows x 4) block with ONE 16-byte write-through store per row (sc1); consumers read blocks with 16-byte sc1 loads (L1
// bypassed, no fence anywhere: buffer_inv costs ~1 us per workgroup on this part).  There are NO flags: every step has its own
// exchange buffers, the host fills the whole region with a poison pattern (0xFFFFFFFF, a NaN no arithmetic on finite data
// produces) before the launch, and a consumer simply re-loads a granule until none of its four words is poison.  Flag +
// payload costs 8 us per 250-block all-gather (4 us of flag propagation, 3.5 us of payload); the payload poll is one phase.
// Nothing is ever rewritten inside a launch, so there is no reuse hazard and no ordering requirement beyond per-word atomicity.
//
// Products: lane = batch row (rows <= 64), v_mfma_f32_4x4x1_16b_f32 with the weight slice as the A operand (lane & 3 = column)
// and the activation granule (4 consecutive k of the lane's row, exactly one 16-byte exchange granule) as the B operand: 4
// columns x 64 rows per instruction, no padding to 16 columns.  With <= 32 rows the idle lanes take other k blocks (KG k-groups
// of RL rows), summed by lane shuffles.  The 8 waves split K; partial sums meet in LDS in fixed order: run-to-run deterministic.
// LayerNorm statistics are lane-local sums (lane = row) over the granules a wave holds, two-pass like elementwise.hip.
//
// Every spin loop gives up after RL_SPIN_LIMIT polls and raises a sticky error word (also in host-visible memory); the next
// library call reports it.
#include "common.h"
#include <mutex>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned RL_POISON = 0xFFFFFFFFu;
constexpr int RL_THREADS = 512, RL_WAVES = 8;
constexpr unsigned RL_SPIN_LIMIT = 1u << 20;

struct RlArgs {
  int B, D, Hd, S, C, Z, ZP, F, t_begin, t_end;
  int nA, nB, nC, nD;
  const float *wzt, *zb, *wih, *bih, *whh, *bhh, *wph, *bph, *wpo, *bpo, *in_g, *in_b, *post_g, *post_b;
  const float *ea, *ee;
  const uint8_t* reset;
  const float* u;
  const int32_t* forced;
  float *x1, *gi, *gh, *hin, *zin, *feat, *x2, *post;
  int32_t* idx;
  char* xch;                    // exchange region, poisoned by the host: per step  xa | xh | xc | xd | xi
  unsigned step_bytes, off_h, off_c, off_d, off_i;
  unsigned* err;                // [0] sticky give-up flag of this launch (device), polled inside the spin loops
  unsigned* host_err;           // host-visible sticky flag (mapped pinned memory)
  unsigned long long* prof;     // optional: 16 phase tick sums of workgroup 0
  // LDS carve, in floats
  int l_wz, l_wg, l_wh, l_wc, l_wd, l_gb1, l_gb2, l_red, l_gh, l_hown, l_stage, l_flag;
  int wz_global;                // 1: the z_mlp^T slice does not fit beside the others (deter 1024): its 32 rows per batch row are gathered from L2
};

// ELU through the hardware exponential: x > 0 ? x : 2^(x log2 e) - 1.  Absolute error <= ~2 ulp(1) = 2.4e-7 - the size of one
// fp32 rounding of the O(1) activation itself - at 5 VALU operations instead of expm1f's ~30; every CU applies it to the
// WHOLE (rows x K) operand of its product, so with expm1f it was the longest phase of a step (8.5 of 21 us, measured).
__device__ __forceinline__ float rl_elu(float v) {
  // (max(v, 0) + min(e, 0) is no cheaper: gfx950 has packed mul / add / fma but no packed max / min or select)
  const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896341f) - 1.0f;
  return v > 0.f ? v : e;
}
__device__ __forceinline__ float rl_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }      // = elementwise.hip dm_sigmoid
__device__ __forceinline__ bool rl_valid(const u32x4 v) {
  return v.x != RL_POISON && v.y != RL_POISON && v.z != RL_POISON && v.w != RL_POISON;
}
__device__ __forceinline__ f32x4 rl_asf(const u32x4 v) {
  f32x4 r;
  r.x = __uint_as_float(v.x); r.y = __uint_as_float(v.y); r.z = __uint_as_float(v.z); r.w = __uint_as_float(v.w);
  return r;
}
__device__ __forceinline__ u32x4 rl_asu(const f32x4 v) {
  u32x4 r;
  r.x = __float_as_uint(v.x); r.y = __float_as_uint(v.y); r.z = __float_as_uint(v.z); r.w = __float_as_uint(v.w);
  return r;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rl_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
template <class ARGS>
__device__ __forceinline__ void rl_give_up(const ARGS& a) {
  __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(a.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <class ARGS>
__device__ __forceinline__ bool rl_dead(const ARGS& a) {
  return __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

// blocks of an operand rounded up to the granule grid of a workgroup (RL_WAVES waves x KG k-groups x MAXJ granules): LDS images
// of weights and LayerNorm gains are zero-padded to it so that the unrolled code needs no per-granule clamps or predicates
template <int RL>
__host__ __device__ constexpr int rl_pad_blocks(int P) { return (P + RL_WAVES * (64 / RL) - 1) / (RL_WAVES * (64 / RL)) * (RL_WAVES * (64 / RL)); }

// This wave's share of the P blocks of 4 k of an operand: granule j of lane (row, kg) is block p = (j * RL_WAVES + wave) * KG + kg,
// read as ONE 16-byte load at byte offset p * pstride + row * rstride of the resource.  The resource ends exactly behind block
// P - 1, so granules past P come back as zeros from the hardware's bounds check (no predicates in the unrolled code).
// poll: re-load a granule until none of its words is poison.  Returns false when the launch has given up.
template <int RL, class ARGS>
__device__ __forceinline__ bool rl_sweep(const ARGS& a, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned jstride, bool poll,
                                         f32x4 (&v)[32 / (64 / RL)]) {
  constexpr int MAXJ = 32 / (64 / RL);
  u32x4 raw[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) raw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (unsigned)j * jstride, 0, 16);
  bool alive = true;
  if (poll) {
    unsigned spins = 0;
    while (true) {
      bool ok = true;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j)
        if (!rl_valid(raw[j])) {
          ok = false;
          raw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (unsigned)j * jstride, 0, 16);
        }
      if (__all(ok)) break;
      if (++spins > RL_SPIN_LIMIT || ((spins & 63u) == 0 && rl_dead(a))) {
        rl_give_up(a);
        alive = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) v[j] = rl_asf(raw[j]);
  return alive;
}

// sum over the k-groups of a wave (lanes with equal lane % RL)
template <int RL>
__device__ __forceinline__ float rl_kg_sum(float x) {
#pragma unroll
  for (int off = RL; off < 64; off <<= 1) x += __shfl_xor(x, off, 64);
  return x;
}

// LayerNorm(eps) + ELU over the K = 4P values of every row, in place on the granules the workgroup's waves hold
// (common.py:48 eps 1e-3; two-pass variance like ln_elu_fwd_kernel).  gb: LDS image of (gamma | beta), K floats each;
// st: RL_WAVES * RL floats of LDS scratch.  Granules past P hold zeros and stay zero.
template <int RL>
__device__ __forceinline__ void rl_ln_elu(f32x4 (&v)[32 / (64 / RL)], int P, int row, int kg, int wave, const float* gb, float eps,
                                          float* st) {
  constexpr int KG = 64 / RL, MAXJ = 32 / KG;
  const float invK = 1.0f / (float)(4 * P);
  float npad = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) npad += ((j * RL_WAVES + wave) * KG + kg) >= P ? 4.f : 0.f;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  s = rl_kg_sum<RL>(s);
  __syncthreads();
  if (kg == 0) st[wave * RL + row] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < RL_WAVES; ++w) tot += st[w * RL + row];
  const float mean = tot * invK;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const float d0 = v[j].x - mean, d1 = v[j].y - mean, d2 = v[j].z - mean, d3 = v[j].w - mean;
    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  q -= npad * (mean * mean);      // this lane's zero granules past P each added 4 mean^2
  q = rl_kg_sum<RL>(q);
  __syncthreads();
  if (kg == 0) st[wave * RL + row] = q;
  __syncthreads();
  float tq = 0.f;
#pragma unroll
  for (int w = 0; w < RL_WAVES; ++w) tq += st[w * RL + row];
  const float rstd = 1.0f / sqrtf(tq * invK + eps);
  // gamma / beta images are padded with zeros to the granule grid: a granule past P becomes ELU(0 * .. + 0) = 0 again
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gb) + (wave * KG + kg);
  const f32x4* b4 = g4 + rl_pad_blocks<RL>(P);
  const int njp = rl_pad_blocks<RL>(P) / (RL_WAVES * KG);     // granules per lane the padded images cover (= MAXJ at production sizes)
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    if (njp < MAXJ && j >= njp) continue;         // small models: past the images (the granule holds zeros and stays zero)
    const f32x4 g = g4[j * RL_WAVES * KG], b = b4[j * RL_WAVES * KG];
    // x a + (b - mean a), a = rstd gamma: written without (x - mean) so that the compiler does not keep the variance pass's 128
    // differences alive for reuse here (it did: 444 spilled registers)
    const float a0 = rstd * g.x, a1 = rstd * g.y, a2 = rstd * g.z, a3 = rstd * g.w;
    v[j].x = rl_elu(fmaf(v[j].x, a0, fmaf(-mean, a0, b.x)));
    v[j].y = rl_elu(fmaf(v[j].y, a1, fmaf(-mean, a1, b.y)));
    v[j].z = rl_elu(fmaf(v[j].z, a2, fmaf(-mean, a2, b.z)));
    v[j].w = rl_elu(fmaf(v[j].w, a3, fmaf(-mean, a3, b.w)));
    if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
}

// acc[cs] += W_slice[cs] (4 columns) x granules: w is the LDS image [p][cs][col i][4 k] of the slice (NCS column sets).
// A dependent v_mfma_f32_4x4x1 chain issues every ~40 cycles, independent ones every 8 (scripts/microbench/allgather_xcd.hip):
// the even and odd k of a granule go to separate accumulators (2 NCS chains per wave, 2 waves per SIMD), summed at the end.
// FULL: run every granule of the grid (operands that went through rl_ln_elu: with a skip branch the compiler SINKS each
// granule's normalisation into its branch, but not the LDS loads of the gains, whose results then sit in - spilled - registers).
template <int RL, int NCS, bool FULL>
__device__ __forceinline__ void rl_dot(f32x4 (&acc)[NCS], const f32x4 (&v)[32 / (64 / RL)], int P, int kg, int wave, int lane,
                                       const float* w) {
  constexpr int KG = 64 / RL, MAXJ = 32 / KG;
  // (the image is zero-padded to the granule grid: no clamp; the address is base + a compile-time offset per (j, cs))
  const f32x4* w4 = reinterpret_cast<const f32x4*>(w) + ((wave * KG + kg) * NCS) * 4 + (lane & 3);
  const int nj = (P + RL_WAVES * KG - 1) / (RL_WAVES * KG);
  const bool full = FULL && nj == MAXJ;      // (small models: the padded LDS image ends before the granule grid does)
  f32x4 odd[NCS];
#pragma unroll
  for (int cs = 0; cs < NCS; ++cs) odd[cs] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (KG == 1 && NCS == 3) {
    // 64 rows, 12 columns: every 4-lane block multiplies the SAME k, so lane l < 12 holds column l (one ds_read_b128 per
    // granule instead of three) and the instruction's A-broadcast (cbsz 4: block `abid` feeds all 16 blocks) selects the
    // column set - verified by scripts/microbench/allgather_xcd.hip (layout modes 1 / 2)
    const f32x4* wl = reinterpret_cast<const f32x4*>(w) + wave * 12 + ((lane & 15) < 12 ? (lane & 15) : 11);
    f32x4 wnext = wl[0];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      if (full || j < nj) {
        const f32x4 ww = wnext;
        if (j + 1 < MAXJ) wnext = wl[(j + 1) * RL_WAVES * 12];      // the next granule's weights are in flight under this one's 12 MFMAs
        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww.x, v[j].x, acc[0], 4, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww.x, v[j].x, acc[1], 4, 1, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww.x, v[j].x, acc[2], 4, 2, 0);
        odd[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww.y, v[j].y, odd[0], 4, 0, 0);
        odd[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww.y, v[j].y, odd[1], 4, 1, 0);
        odd[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww.y, v[j].y, odd[2], 4, 2, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww.z, v[j].z, acc[0], 4, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww.z, v[j].z, acc[1], 4, 1, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww.z, v[j].z, acc[2], 4, 2, 0);
        odd[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww.w, v[j].w, odd[0], 4, 0, 0);
        odd[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww.w, v[j].w, odd[1], 4, 1, 0);
        odd[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww.w, v[j].w, odd[2], 4, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int cs = 0; cs < NCS; ++cs) {
      acc[cs].x += odd[cs].x; acc[cs].y += odd[cs].y; acc[cs].z += odd[cs].z; acc[cs].w += odd[cs].w;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    if (full || j < nj) {                         // (uniform) granules past the operand are zeros: skip their instructions
      f32x4 ww[NCS];
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs) ww[cs] = w4[(j * RL_WAVES * KG * NCS + cs) * 4];
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs) {
        acc[cs] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww[cs].x, v[j].x, acc[cs], 0, 0, 0);
        odd[cs] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww[cs].y, v[j].y, odd[cs], 0, 0, 0);
      }
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs) {
        acc[cs] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww[cs].z, v[j].z, acc[cs], 0, 0, 0);
        odd[cs] = __builtin_amdgcn_mfma_f32_4x4x1f32(ww[cs].w, v[j].w, odd[cs], 0, 0, 0);
      }
    }
    if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int cs = 0; cs < NCS; ++cs) {
    acc[cs].x += odd[cs].x; acc[cs].y += odd[cs].y; acc[cs].z += odd[cs].z; acc[cs].w += odd[cs].w;
  }
}

// Partial sums of the 8 waves -> red, after summing the k-groups.  4-column products: red[wave][row] (f32x4), 8 entries per
// row.  12-column products meet in two rounds so that red stays RL_RED_WAVES * RL * 12 floats: waves 4-7 hand theirs to waves
// 0-3, which add and publish; 4 entries per row.  rl_red_sum<NCS> knows which.
constexpr int RL_RED_WAVES = 4;
template <int RL, int NCS>
__device__ __forceinline__ void rl_reduce_store(f32x4 (&acc)[NCS], int row, int kg, int wave, float* red) {
#pragma unroll
  for (int cs = 0; cs < NCS; ++cs) {
    acc[cs].x = rl_kg_sum<RL>(acc[cs].x); acc[cs].y = rl_kg_sum<RL>(acc[cs].y);
    acc[cs].z = rl_kg_sum<RL>(acc[cs].z); acc[cs].w = rl_kg_sum<RL>(acc[cs].w);
  }
  f32x4* r4 = reinterpret_cast<f32x4*>(red);
  __syncthreads();      // earlier readers of red are done
  if (NCS == 1) {
    if (kg == 0) r4[wave * RL + row] = acc[0];
    __syncthreads();
  } else {
    if (wave >= RL_RED_WAVES && kg == 0) {
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs) r4[((wave - RL_RED_WAVES) * RL + row) * NCS + cs] = acc[cs];
    }
    __syncthreads();
    if (wave < RL_RED_WAVES) {
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs) {
        const f32x4 o = r4[(wave * RL + row) * NCS + cs];
        acc[cs].x += o.x; acc[cs].y += o.y; acc[cs].z += o.z; acc[cs].w += o.w;
      }
    }
    __syncthreads();
    if (wave < RL_RED_WAVES && kg == 0) {
#pragma unroll
      for (int cs = 0; cs < NCS; ++cs) r4[(wave * RL + row) * NCS + cs] = acc[cs];
    }
    __syncthreads();
  }
}
template <int RL, int NCS>
__device__ __forceinline__ float rl_red_sum(const float* red, int row, int c) {      // column c of (NCS * 4), waves in order
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < (NCS == 1 ? RL_WAVES : RL_RED_WAVES); ++w) s += red[((w * RL + row) * NCS) * 4 + c];
  return s;
}

// LDS image [p][cs][i][kk] <- W[(cs * colstride + c0 + i) * K + 4p + kk]   (W row-major (rows, K))
template <int NCS>
__device__ __forceinline__ void rl_fill(float* dst, const float* W, int K, int Ppad, int c0, int colstride, int tid) {
  const int P = K / 4;
  f32x4* d4 = reinterpret_cast<f32x4*>(dst);
  for (int i = tid; i < Ppad * NCS * 4; i += RL_THREADS) {
    const int p = i % Ppad, ci = i / Ppad, cs = ci / 4, col = ci % 4;
    d4[(p * NCS + cs) * 4 + col] = p < P ? *reinterpret_cast<const f32x4*>(W + (size_t)(cs * colstride + c0 + col) * K + 4 * p)
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// Sub-phase ticks (slots 8-14) sit between the LayerNorm and the product it feeds: they stop the compiler from interleaving
// the two, and at 64 rows per wave the split code spills 420 registers (5.2 vs 2.9 ms per T=50 call) - so they are compiled in
// only with -DRL_FINE_PROF; they are valid as measurements for <= 32 rows.
#ifdef RL_FINE_PROF
#define RL_FINE_TICK(k_) RL_TICK(k_)
#else
#define RL_FINE_TICK(k_) do { } while (0)
#endif
#define RL_TICK(k_)                                                  \
  do {                                                               \
    if (a.prof && me == 0 && tid == 0) {                             \
      const unsigned long long now_ = wall_clock64();                \
      a.prof[k_] += now_ - tick_;                                    \
      tick_ = now_;                                                  \
    }                                                                \
  } while (0)

template <int RL>
__global__ void __launch_bounds__(RL_THREADS) rssm_lds_fwd_kernel(const RlArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KG = 64 / RL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = lane % RL, kg = lane / RL;
  const int me = blockIdx.x;
  const int B = a.B, D = a.D, Hd = a.Hd, S = a.S, C = a.C, Z = a.Z, ZP = a.ZP, F = a.F;
  const int rowc = row < B ? row : B - 1;                 // rows past B repeat the last row (never stored)
  const bool roleA = me < a.nA, roleB = me < a.nB, roleC = me < a.nC, roleD = me < a.nD;
  const int cpg = C / 4;                                   // D workgroups per latent group
  const bool leader = roleD && (me % cpg) == 0;
  float* wz = lds + a.l_wz; float* wg = lds + a.l_wg; float* wh = lds + a.l_wh; float* wc = lds + a.l_wc; float* wd = lds + a.l_wd;
  float* gb1 = lds + a.l_gb1; float* gb2 = lds + a.l_gb2;      // (gamma | beta) of in_norm (role B) and post_norm (role D)
  float* red = lds + a.l_red; float* ghs = lds + a.l_gh; float* hown = lds + a.l_hown; float* stage = lds + a.l_stage;
  unsigned* lflag = reinterpret_cast<unsigned*>(lds + a.l_flag);
  unsigned long long tick_ = wall_clock64();

  // ---- weights into LDS, once
  if (roleA && !a.wz_global) {
    f32x4* d4 = reinterpret_cast<f32x4*>(wz);
    for (int e = tid; e < Z; e += RL_THREADS) d4[e] = *reinterpret_cast<const f32x4*>(a.wzt + (size_t)e * Hd + 4 * me);
  }
  const int PH = rl_pad_blocks<RL>(Hd / 4), PD = rl_pad_blocks<RL>(D / 4);      // operand blocks on the granule grid
  if (roleB) {
    rl_fill<3>(wg, a.wih, Hd, PH, 4 * me, D, tid);
    rl_fill<3>(wh, a.whh, D, PD, 4 * me, D, tid);
  }
  if (roleC) rl_fill<1>(wc, a.wph, D, PD, 4 * me, 0, tid);
  if (roleD) rl_fill<1>(wd, a.wpo, Hd, PH, 4 * me, 0, tid);
  for (int i = tid; i < 4 * PH; i += RL_THREADS) {
    if (roleB) { gb1[i] = i < Hd ? a.in_g[i] : 0.f; gb1[4 * PH + i] = i < Hd ? a.in_b[i] : 0.f; }
    if (roleD) { gb2[i] = i < Hd ? a.post_g[i] : 0.f; gb2[4 * PH + i] = i < Hd ? a.post_b[i] : 0.f; }
  }
  if (tid == 0) lflag[0] = 0u;
  __syncthreads();

  const unsigned blk = RL * 16u;                           // bytes of one exchange block
  constexpr int MAXJ = 32 / KG;
  f32x4 v[MAXJ];
  // a lane's first granule and the distance between its consecutive granules, in an exchange buffer of (RL x 4) blocks
  const unsigned xoff = (unsigned)(wave * KG + kg) * blk + (unsigned)rowc * 16u, xjs = (unsigned)(RL_WAVES * KG) * blk;
  // ---- prologue (B): the state the first step continues from was left by the launch schedule in row-major buffers
  if (roleB) {
    const size_t rp = (size_t)(a.t_begin - 1) * B, r0 = (size_t)a.t_begin * B;
    // gh of the first step from h of step t_begin - 1 (feature matrix, column 0), under the first step's reset mask
    // (row-major source: block p = 4 columns at byte 16 p of a row of F floats; the bounds check does not end the operand
    // here - the z columns follow h in a feature row - so granules past D / 4 are zeroed by hand)
    __amdgpu_buffer_rsrc_t rs = rl_rsrc(a.feat + rp * F, (unsigned)((size_t)B * F * 4));
    rl_sweep<RL>(a, rs, (unsigned)(wave * KG + kg) * 16u + (unsigned)rowc * (unsigned)F * 4u, (unsigned)(RL_WAVES * KG) * 16u, false, v);
    const float keep = a.reset[r0 + rowc] ? 0.f : 1.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const float kj = ((j * RL_WAVES + wave) * KG + kg) < D / 4 ? keep : 0.f;
      v[j].x *= kj; v[j].y *= kj; v[j].z *= kj; v[j].w *= kj;
    }
    f32x4 acc[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    rl_dot<RL, 3, false>(acc, v, D / 4, kg, wave, lane, wh);
    rl_reduce_store<RL, 3>(acc, row, kg, wave, red);
    if (tid < RL * 4) {
      const int r = tid % RL, un = tid / RL;
#pragma unroll
      for (int cs = 0; cs < 3; ++cs) ghs[r * 12 + cs * 4 + un] = rl_red_sum<RL, 3>(red, r, cs * 4 + un) + a.bhh[cs * D + 4 * me + un];
      hown[r * 4 + un] = a.hin[(r0 + (r < B ? r : B - 1)) * D + 4 * me + un];      // masked h_in of the first step
    }
    __syncthreads();
  }

  for (int t = a.t_begin; t < a.t_end; ++t) {
    const size_t r0 = (size_t)t * B;
    const bool more = t + 1 < a.t_end;
    char* xs = a.xch + (size_t)(t - a.t_begin) * a.step_bytes;
    unsigned xo = xoff;
    asm volatile("" : "+v"(xo));      // opaque per step: keeps the compiler from hoisting 32 granule offsets per sweep out of the loop
    // ---- A. x1 block = z_b + a_mlp(a) + sum_g z_mlp^T[g C + idx_g]   (reset rows: no latent term)      rssm.py:134-139
    if (roleA) {
      const int parts = RL_THREADS / RL, part = tid / RL, r = tid % RL;
      const int gpp = (S + parts - 1) / parts;
      // indices of step t-1: the exchange buffer of the previous step, or (first step) the launch schedule's idx array
      const int32_t* isrc = t == a.t_begin ? a.idx + (r0 - B) * S : reinterpret_cast<const int32_t*>(xs - a.step_bytes + a.off_i);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const bool live = r < B && !a.reset[r0 + r];
      bool alive = true;
      if (live) {
        const unsigned* ip = reinterpret_cast<const unsigned*>(isrc) + r * S + part * gpp;
        const int ng = S - part * gpp < gpp ? S - part * gpp : gpp;      // this thread's groups (<= 0: none)
        for (int q0 = 0; q0 < ng; q0 += 4) {
          unsigned ix[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            ix[q] = q0 + q < ng ? __hip_atomic_load(ip + q0 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
          unsigned spins = 0;
          while (alive && (ix[0] == RL_POISON || ix[1] == RL_POISON || ix[2] == RL_POISON || ix[3] == RL_POISON)) {
            if (++spins > RL_SPIN_LIMIT || ((spins & 63u) == 0 && rl_dead(a))) { rl_give_up(a); alive = false; break; }
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (ix[q] == RL_POISON) ix[q] = __hip_atomic_load(ip + q0 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (!alive) break;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q0 + q < ng) {
              const int e = (part * gpp + q0 + q) * C + (int)ix[q];
              const f32x4 w = a.wz_global ? *reinterpret_cast<const f32x4*>(a.wzt + (size_t)e * Hd + 4 * me) : reinterpret_cast<const f32x4*>(wz)[e];
              acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
            }
        }
      }
      if (!alive) lflag[0] = 1u;
      __syncthreads();
      reinterpret_cast<f32x4*>(red)[part * RL + r] = acc;
      __syncthreads();
      if (tid < RL && tid < B) {
        f32x4 tot = *reinterpret_cast<const f32x4*>(a.zb + 4 * me);
        const f32x4 e4 = *reinterpret_cast<const f32x4*>(a.ea + (r0 + tid) * Hd + 4 * me);
        tot.x += e4.x; tot.y += e4.y; tot.z += e4.z; tot.w += e4.w;
        for (int pq = 0; pq < parts; ++pq) {
          const f32x4 pa = reinterpret_cast<const f32x4*>(red)[pq * RL + tid];
          tot.x += pa.x; tot.y += pa.y; tot.z += pa.z; tot.w += pa.w;
        }
        __amdgpu_buffer_rsrc_t xr = rl_rsrc(xs, a.step_bytes);
        __builtin_amdgcn_raw_buffer_store_b128(rl_asu(tot), xr, (unsigned)me * blk + (unsigned)tid * 16u, 0, 16);
        *reinterpret_cast<f32x4*>(a.x1 + (r0 + tid) * Hd + 4 * me) = tot;
      }
    }
    RL_TICK(0);
    // ---- B. gi = ELU(in_norm(x1)) W_ih^T + b_ih ; GRU gates of 4 hidden units                        rssm.py:140-141, rnn.py:48-49
    if (roleB) {
      __amdgpu_buffer_rsrc_t rs = rl_rsrc(xs, a.step_bytes);
      __amdgpu_buffer_rsrc_t ra = rl_rsrc(xs, (unsigned)a.nA * blk);                 // the x1 blocks
      if (!rl_sweep<RL>(a, ra, xo, xjs, true, v)) lflag[0] = 1u;
      RL_TICK(1);
      rl_ln_elu<RL>(v, a.nA, row, kg, wave, gb1, 1e-3f, red);
      RL_FINE_TICK(8);
      f32x4 acc[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
      rl_dot<RL, 3, true>(acc, v, a.nA, kg, wave, lane, wg);
      RL_FINE_TICK(9);
      rl_reduce_store<RL, 3>(acc, row, kg, wave, red);
      RL_FINE_TICK(10);
      if (tid < RL * 4) {
        const int r = tid % RL, un = tid / RL, d = 4 * me + un;
        const float gir = rl_red_sum<RL, 3>(red, r, un) + a.bih[d];
        const float giz = rl_red_sum<RL, 3>(red, r, 4 + un) + a.bih[D + d];
        const float gin = rl_red_sum<RL, 3>(red, r, 8 + un) + a.bih[2 * D + d];
        const float ghr = ghs[r * 12 + un], ghz = ghs[r * 12 + 4 + un], ghn = ghs[r * 12 + 8 + un];
        const float rg = rl_sigmoid(gir + ghr);
        const float ug = rl_sigmoid(giz + ghz);
        const float ng = tanhf(gin + rg * ghn);
        const float h = hown[r * 4 + un];
        const float ho = (h - ng) * ug + ng;
        stage[r * 16 + un] = ho;
        stage[r * 16 + 4 + un] = gir; stage[r * 16 + 8 + un] = giz; stage[r * 16 + 12 + un] = gin;
      }
      __syncthreads();
      if (tid < RL && tid < B) {                    // publish h first (the chain), then the copies the backward pass reads
        const f32x4 h4 = *reinterpret_cast<const f32x4*>(stage + tid * 16);
        __builtin_amdgcn_raw_buffer_store_b128(rl_asu(h4), rs, a.off_h + (unsigned)me * blk + (unsigned)tid * 16u, 0, 16);
        *reinterpret_cast<f32x4*>(a.feat + (r0 + tid) * F + 4 * me) = h4;
        const bool rz = more && a.reset[r0 + B + tid];
        const f32x4 hn = rz ? f32x4{0.f, 0.f, 0.f, 0.f} : h4;
        if (more) *reinterpret_cast<f32x4*>(a.hin + (r0 + B + tid) * D + 4 * me) = hn;
        *reinterpret_cast<f32x4*>(hown + tid * 4) = hn;
      } else if (tid >= 64 && tid < 64 + RL && tid - 64 < B) {
        const int r = tid - 64;
#pragma unroll
        for (int cs = 0; cs < 3; ++cs) {
          *reinterpret_cast<f32x4*>(a.gi + (r0 + r) * 3 * D + cs * D + 4 * me) = *reinterpret_cast<const f32x4*>(stage + r * 16 + 4 + 4 * cs);
          *reinterpret_cast<f32x4*>(a.gh + (r0 + r) * 3 * D + cs * D + 4 * me) = *reinterpret_cast<const f32x4*>(ghs + r * 12 + 4 * cs);
        }
      }
      __syncthreads();
    }
    RL_TICK(2);
    // ---- C. x2 = h W_post_h^T + b + post_mlp_e(embed); then gh of step t+1 from the same h                rssm.py:143-144
    if (roleC || (roleB && more)) {      // (unit owners past the last hidden block - deter_dim > hidden_dim - still need h for their gh)
      __amdgpu_buffer_rsrc_t rs = rl_rsrc(xs, a.step_bytes);
      __amdgpu_buffer_rsrc_t rh = rl_rsrc(xs + a.off_h, (unsigned)a.nB * blk);      // the h blocks
      if (!rl_sweep<RL>(a, rh, xo, xjs, true, v)) lflag[0] = 1u;
      RL_TICK(3);
      f32x4 acc1[1] = {f32x4{0, 0, 0, 0}};
      if (roleC) {
        rl_dot<RL, 1, false>(acc1, v, a.nB, kg, wave, lane, wc);
        rl_reduce_store<RL, 1>(acc1, row, kg, wave, red);
      }
      if (roleC && tid < RL && tid < B) {
        f32x4 o = *reinterpret_cast<const f32x4*>(a.bph + 4 * me);
        const f32x4 e4 = *reinterpret_cast<const f32x4*>(a.ee + (r0 + tid) * Hd + 4 * me);
        o.x += e4.x + rl_red_sum<RL, 1>(red, tid, 0); o.y += e4.y + rl_red_sum<RL, 1>(red, tid, 1);
        o.z += e4.z + rl_red_sum<RL, 1>(red, tid, 2); o.w += e4.w + rl_red_sum<RL, 1>(red, tid, 3);
        __builtin_amdgcn_raw_buffer_store_b128(rl_asu(o), rs, a.off_c + (unsigned)me * blk + (unsigned)tid * 16u, 0, 16);
        *reinterpret_cast<f32x4*>(a.x2 + (r0 + tid) * Hd + 4 * me) = o;
      }
      RL_FINE_TICK(14);
      if (roleB && more) {
        const float keep = a.reset[r0 + B + rowc] ? 0.f : 1.f;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) { v[j].x *= keep; v[j].y *= keep; v[j].z *= keep; v[j].w *= keep; }
        f32x4 acc[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
        rl_dot<RL, 3, false>(acc, v, a.nB, kg, wave, lane, wh);
        rl_reduce_store<RL, 3>(acc, row, kg, wave, red);
        if (tid < RL * 4) {
          const int r = tid % RL, un = tid / RL;
#pragma unroll
          for (int cs = 0; cs < 3; ++cs)
            ghs[r * 12 + cs * 4 + un] = rl_red_sum<RL, 3>(red, r, cs * 4 + un) + a.bhh[cs * D + 4 * me + un];
        }
        __syncthreads();
      }
    }
    RL_TICK(4);
    // ---- D. 4 posterior logits = ELU(post_norm(x2)) W_post^T + b                                          rssm.py:145-146
    if (roleD) {
      __amdgpu_buffer_rsrc_t rs = rl_rsrc(xs, a.step_bytes);
      __amdgpu_buffer_rsrc_t rc = rl_rsrc(xs + a.off_c, (unsigned)a.nC * blk);      // the x2 blocks
      if (!rl_sweep<RL>(a, rc, xo, xjs, true, v)) lflag[0] = 1u;
      RL_TICK(5);
      rl_ln_elu<RL>(v, a.nC, row, kg, wave, gb2, 1e-3f, red);
      RL_FINE_TICK(11);
      f32x4 acc1[1] = {f32x4{0, 0, 0, 0}};
      rl_dot<RL, 1, true>(acc1, v, a.nC, kg, wave, lane, wd);
      RL_FINE_TICK(12);
      rl_reduce_store<RL, 1>(acc1, row, kg, wave, red);
      RL_FINE_TICK(13);
      if (tid < RL && tid < B) {
        f32x4 o = *reinterpret_cast<const f32x4*>(a.bpo + 4 * me);
        o.x += rl_red_sum<RL, 1>(red, tid, 0); o.y += rl_red_sum<RL, 1>(red, tid, 1);
        o.z += rl_red_sum<RL, 1>(red, tid, 2); o.w += rl_red_sum<RL, 1>(red, tid, 3);
        __builtin_amdgcn_raw_buffer_store_b128(rl_asu(o), rs, a.off_d + (unsigned)me * blk + (unsigned)tid * 16u, 0, 16);
        *reinterpret_cast<f32x4*>(a.post + (r0 + tid) * ZP + 4 * me) = o;
      }
      __syncthreads();
    }
    RL_TICK(6);
    // ---- L. z ~ OneHotCategoricalStraightThrough(post) for one latent group                             rssm.py:147-148,195-201
    // The rule's sums (softmax denominator, total, cdf) are sequential fp32 chains per row like sample_onehot_kernel
    // (bit-identical draws for identical logits); max, exp and the divisions are elementwise and spread over the workgroup.
    if (leader && cpg <= 8) {
      // C <= 32: 8 lanes per row (lane c holds the 4 logits of block c), 8 rows per wave, all 8 waves: exp and the divisions
      // run across lanes, the rule's three fp32 sums stay SEQUENTIAL in k - a carry handed from lane c to lane c + 1 by a
      // shuffle, (((0 + x0) + x1) + ...) exactly like sample_onehot_kernel - so the draw is bit-identical for identical
      // logits.  (One lane per row took 3.6 us of serial expf / divide code per step; the LDS version before it 7 us.)
      const int g = me / cpg;
      const int c = lane & 7, srow = wave * 8 + (lane >> 3), gl = lane & ~7;
      __amdgpu_buffer_rsrc_t rs = rl_rsrc(xs, a.step_bytes);
      const bool rowok = srow < B, mine = rowok && c < cpg;
      f32x4 x = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (mine) {
        const unsigned off = a.off_d + (unsigned)(me + c) * blk + (unsigned)srow * 16u;
        u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
        unsigned spins = 0;
        while (!rl_valid(raw)) {
          if (++spins > RL_SPIN_LIMIT || ((spins & 63u) == 0 && rl_dead(a))) { rl_give_up(a); lflag[0] = 1u; break; }
          __builtin_amdgcn_s_sleep(1);
          raw = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
        }
        x = rl_asf(raw);
      }
      float mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
      if (!rowok) mx = 0.f;
      f32x4 e = {0.f, 0.f, 0.f, 0.f};
      if (mine) { e.x = expf(x.x - mx); e.y = expf(x.y - mx); e.z = expf(x.z - mx); e.w = expf(x.w - mx); }
      float carry = 0.f, held = 0.f;
      for (int q = 0; q < cpg; ++q) {                        // sum = sequential over k
        if (c == q) { float t = carry; t += e.x; t += e.y; t += e.z; t += e.w; held = t; }
        carry = __shfl(held, gl + q, 64);
      }
      const float sum = carry;
      f32x4 pr = {0.f, 0.f, 0.f, 0.f};
      if (mine) { pr.x = e.x / sum; pr.y = e.y / sum; pr.z = e.z / sum; pr.w = e.w / sum; }
      carry = 0.f; held = 0.f;
      for (int q = 0; q < cpg; ++q) {                        // total = sequential over k
        if (c == q) { float t = carry; t += pr.x; t += pr.y; t += pr.z; t += pr.w; held = t; }
        carry = __shfl(held, gl + q, 64);
      }
      const float target = (rowok && !a.forced ? a.u[(r0 + srow) * S + g] : 0.f) * carry;
      carry = 0.f; held = 0.f;
      int cnt = 0;
      for (int q = 0; q < cpg; ++q) {                        // cdf = sequential over k; idx = #{k : cdf_k <= target}
        if (c == q) {
          float t = carry;
          t += pr.x; cnt += (t <= target) ? 1 : 0;
          t += pr.y; cnt += (t <= target) ? 1 : 0;
          t += pr.z; cnt += (t <= target) ? 1 : 0;
          t += pr.w; cnt += (t <= target) ? 1 : 0;
          held = t;
        }
        carry = __shfl(held, gl + q, 64);
      }
      if (!mine) cnt = 0;
      cnt += __shfl_xor(cnt, 1, 64); cnt += __shfl_xor(cnt, 2, 64); cnt += __shfl_xor(cnt, 4, 64);
      int ix = cnt > C - 1 ? C - 1 : cnt;
      if (rowok && a.forced) ix = a.forced[(r0 + srow) * S + g];
      if (mine) {
        if (c == 0) {
          __hip_atomic_store(reinterpret_cast<unsigned*>(xs + a.off_i) + srow * S + g, (unsigned)ix, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
          a.idx[(r0 + srow) * S + g] = ix;
        }
        const int k4 = 4 * c;
        const f32x4 o = {ix == k4 ? 1.f : 0.f, ix == k4 + 1 ? 1.f : 0.f, ix == k4 + 2 ? 1.f : 0.f, ix == k4 + 3 ? 1.f : 0.f};
        *reinterpret_cast<f32x4*>(a.feat + (r0 + srow) * F + D + g * C + k4) = o;
        if (more) {
          const bool rz = a.reset[r0 + B + srow];
          *reinterpret_cast<f32x4*>(a.zin + (r0 + B + srow) * Z + g * C + k4) = rz ? f32x4{0.f, 0.f, 0.f, 0.f} : o;
        }
      }
    } else if (leader) {
      const int g = me / cpg, ldl = C + 1;                   // row stride C + 1: conflict-free column walks
      float* lg = red;                                       // (RL, C + 1) logits -> exp -> probabilities
      float* rowv = red + RL * ldl;                          // (RL) row max, then row sum
      int* rowi = reinterpret_cast<int*>(rowv + RL);         // (RL) sampled index
      __amdgpu_buffer_rsrc_t rs = rl_rsrc(xs, a.step_bytes);
      if (tid < RL) {
        bool alive = true;
        const int rr = tid < B ? tid : B - 1;
        for (int q0 = 0; q0 < cpg && alive; q0 += 8) {      // 8 granules (32 logits) in flight per lane
          u32x4 raw[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int qq = q0 + q < cpg ? q0 + q : cpg - 1;
            raw[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, a.off_d + (unsigned)(me + qq) * blk + (unsigned)rr * 16u, 0, 16);
          }
          unsigned spins = 0;
          while (true) {
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (!rl_valid(raw[q])) {
                ok = false;
                const int qq = q0 + q < cpg ? q0 + q : cpg - 1;
                raw[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, a.off_d + (unsigned)(me + qq) * blk + (unsigned)rr * 16u, 0, 16);
              }
            if (ok) break;
            if (++spins > RL_SPIN_LIMIT || ((spins & 63u) == 0 && rl_dead(a))) { rl_give_up(a); alive = false; break; }
            __builtin_amdgcn_s_sleep(1);
          }
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (q0 + q < cpg) {
              const f32x4 x = rl_asf(raw[q]);
              float* o = lg + tid * ldl + 4 * (q0 + q);
              o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = x.w;
            }
        }
        if (!alive) lflag[0] = 1u;
        float mx = lg[tid * ldl];
        for (int k = 1; k < C; ++k) mx = fmaxf(mx, lg[tid * ldl + k]);
        rowv[tid] = mx;
      }
      __syncthreads();
      for (int i = tid; i < RL * C; i += RL_THREADS) {
        const int r = i / C, k = i - r * C;
        lg[r * ldl + k] = expf(lg[r * ldl + k] - rowv[r]);
      }
      __syncthreads();
      if (tid < RL) {
        float sum = 0.f;
        for (int k = 0; k < C; ++k) sum += lg[tid * ldl + k];
        rowv[tid] = sum;
      }
      __syncthreads();
      for (int i = tid; i < RL * C; i += RL_THREADS) {
        const int r = i / C, k = i - r * C;
        lg[r * ldl + k] = lg[r * ldl + k] / rowv[r];
      }
      __syncthreads();
      if (tid < RL) {
        int ix = 0;
        if (tid < B) {
          if (a.forced) {
            ix = a.forced[(r0 + tid) * S + g];
          } else {
            float total_p = 0.f;
            for (int k = 0; k < C; ++k) total_p += lg[tid * ldl + k];
            const float target = a.u[(r0 + tid) * S + g] * total_p;
            float cdf = 0.f;
            for (int k = 0; k < C; ++k) { cdf += lg[tid * ldl + k]; ix += (cdf <= target) ? 1 : 0; }
            if (ix > C - 1) ix = C - 1;
          }
          __hip_atomic_store(reinterpret_cast<unsigned*>(xs + a.off_i) + tid * S + g, (unsigned)ix, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
          a.idx[(r0 + tid) * S + g] = ix;
        }
        rowi[tid] = ix;
      }
      __syncthreads();
      for (int i = tid; i < B * C; i += RL_THREADS) {         // one-hot z into the feature matrix, next step's masked z
        const int r = i / C, k = i - r * C;
        const float o = rowi[r] == k ? 1.f : 0.f;
        a.feat[(r0 + r) * F + D + g * C + k] = o;
        if (more) a.zin[(r0 + B + r) * Z + g * C + k] = a.reset[r0 + B + r] ? 0.f : o;
      }
    }
    __syncthreads();
    RL_TICK(7);
    if (lflag[0]) return;      // this workgroup (or, through the sticky word, any other) gave up: the host reports it
  }
}

struct RlPlan {
  int rl, nA, nB, nC, nD, G;
  size_t lds_bytes;
  unsigned step_bytes, off_h, off_c, off_d, off_i;
  int l_wz, l_wg, l_wh, l_wc, l_wd, l_gb1, l_gb2, l_red, l_gh, l_hown, l_stage, l_flag;
  int wz_global;
};

int g_rssm_lds = getenv("DM_RSSM_LDS") ? atoi(getenv("DM_RSSM_LDS")) : 1;
int g_cus = -1, g_lds_max = -1;
unsigned* g_host_err = nullptr;      // mapped pinned word
unsigned* g_host_err_dev = nullptr;
std::mutex g_mu;

bool rl_plan_one(int B, int D, int Hd, int S, int C, int wz_global, RlPlan* pl) {
  if (B < 1 || B > 64 || C < 4 || (C & 3) || S < 1 || (D & 3) || (Hd & 3)) return false;
  RlPlan p;
  p.rl = B <= 8 ? 8 : (B <= 16 ? 16 : (B <= 32 ? 32 : 64));
  p.nA = Hd / 4; p.nB = D / 4; p.nC = Hd / 4; p.nD = S * C / 4;
  p.G = p.nA > p.nB ? p.nA : p.nB;
  if (p.nD > p.G) p.G = p.nD;
  if (p.G > 256) return false;
  const int Z = S * C;
  int off = 0;
  auto take = [&](int floats) { const int o = off; off += (floats + 3) & ~3; return o; };
  const int grid = RL_WAVES * (64 / p.rl);
  const int PH = (Hd / 4 + grid - 1) / grid * grid, PD = (D / 4 + grid - 1) / grid * grid;      // = rl_pad_blocks<rl>
  p.wz_global = wz_global;
  p.l_wz = take(wz_global ? 4 : Z * 4); p.l_wg = take(PH * 48); p.l_wh = take(PD * 48); p.l_wc = take(PD * 16); p.l_wd = take(PH * 16);
  p.l_gb1 = take(8 * PH); p.l_gb2 = take(8 * PH);
  int red = RL_RED_WAVES * p.rl * 12;                           // wave partials of a 12-column product (two rounds)
  if (red < RL_WAVES * p.rl * 4) red = RL_WAVES * p.rl * 4;     // ... of a 4-column product; LayerNorm statistics
  if (red < (RL_THREADS / p.rl) * p.rl * 4) red = (RL_THREADS / p.rl) * p.rl * 4;      // role A's part sums
  if (red < p.rl * (C + 3)) red = p.rl * (C + 3);               // the leader's logits (row stride C + 1), row values, indices
  p.l_red = take(red);
  p.l_gh = take(p.rl * 12); p.l_hown = take(p.rl * 4); p.l_stage = take(p.rl * 16); p.l_flag = take(4);
  p.lds_bytes = (size_t)off * 4;
  const unsigned blk = (unsigned)p.rl * 16u;
  p.off_h = (unsigned)p.nA * blk;
  p.off_c = p.off_h + (unsigned)p.nB * blk;
  p.off_d = p.off_c + (unsigned)p.nC * blk;
  p.off_i = p.off_d + (unsigned)p.nD * blk;
  p.step_bytes = p.off_i + (unsigned)dm_align_up((size_t)p.rl * S * 4, 256);
  *pl = p;
  return true;
}

bool rl_plan(int B, int D, int Hd, int S, int C, RlPlan* pl) {
  // everything in LDS if it fits 160 KB; else z_mlp^T stays in L2 (16 KB less: pydreamer's shipped Atari configuration, deter 1024, B <= 32)
  if (!rl_plan_one(B, D, Hd, S, C, 0, pl)) return false;
  if (pl->lds_bytes > 160 * 1024) return rl_plan_one(B, D, Hd, S, C, 1, pl);
  return true;
}

bool rl_device_ok(int G, size_t lds_bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_cus < 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) { (void)hipGetLastError(); g_cus = 0; return false; }
    g_cus = pr.multiProcessorCount;
    int v = 0;
    g_lds_max = hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess ? v : 65536;
    if (g_lds_max < 160 * 1024 && pr.major == 9) g_lds_max = 160 * 1024;      // gfx950: 160 KiB per workgroup (opt-in attribute)
    if (hipHostMalloc(reinterpret_cast<void**>(&g_host_err), 64, hipHostMallocMapped) == hipSuccess &&
        hipHostGetDevicePointer(reinterpret_cast<void**>(&g_host_err_dev), g_host_err, 0) == hipSuccess) {
      g_host_err[0] = 0u;
    } else {
      (void)hipGetLastError();
      g_host_err = nullptr; g_host_err_dev = nullptr;
    }
  }
  // one workgroup per CU, all resident at once: more than half of a CU's LDS each, and no more workgroups than CUs
  // All workgroups must be resident at once: never more workgroups than CUs.  Production sizes need more than half of a CU's LDS
  // each (so exactly one sits on every CU); smaller models are left to the launch chain, whose few small launches they do not
  // outweigh - except at switch level 2 (tests: the tiny reference goldens through these kernels).
  return g_host_err_dev && G <= g_cus && lds_bytes <= (size_t)g_lds_max && (lds_bytes > 80 * 1024 || g_rssm_lds >= 2);
}

// Two kernels that each need every CU at once must never be in flight together (each would hold some CUs and spin for the rest):
// within a process, a launch waits for the previous persistent launch of ANY stream and records itself behind it.  (Other
// processes' kernels are time-sliced against ours by the driver, not co-scheduled.)
hipEvent_t g_chip_lease = nullptr;
std::mutex g_lease_mu;
int rl_launch_exclusive(const void* fn, unsigned grid, void** args, size_t lds_bytes, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_lease_mu);
  if (!g_chip_lease && hipEventCreateWithFlags(&g_chip_lease, hipEventDisableTiming) != hipSuccess) {
    g_chip_lease = nullptr;
    return dm_fail(DM_E_HIP, "rssm_lds: cannot create the chip-lease event");
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
  if (!capturing && hipStreamWaitEvent(st, g_chip_lease, 0) != hipSuccess) (void)hipGetLastError();      // (never recorded yet: no-op)
  // Co-residency of the whole grid is CHECKED, not assumed: rl_raise_lds() asks the occupancy API for >= 1 workgroup of this
  // variant per CU at the LDS size it is launched with, and the grid never exceeds the CU count (rl_device_ok); the spin loops
  // are bounded.  A cooperative launch (hipLaunchCooperativeKernel: the runtime validates the same bound) was measured and is
  // NOT the default: it drains the other streams' queues around the kernel - the 25-column shard goes from 19.2 to 25.2 ms
  // per step, 13 / 7 columns +0.15 ms (profiles/r05_lds_coop.txt).  DM_RSSM_LDS_COOP=1 selects it.
  static const int coop = getenv("DM_RSSM_LDS_COOP") ? atoi(getenv("DM_RSSM_LDS_COOP")) : 0;
  bool launched = false;
  if (coop && !capturing) {
    if (hipLaunchCooperativeKernel(fn, dim3(grid), dim3(RL_THREADS), args, (unsigned)lds_bytes, st) == hipSuccess) launched = true;
    else (void)hipGetLastError();
  }
  if (!launched && hipLaunchKernel(fn, dim3(grid), dim3(RL_THREADS), args, lds_bytes, st) != hipSuccess)
    return dm_fail(DM_E_HIP, "rssm_lds: launch failed: %s", hipGetErrorString(hipGetLastError()));
  if (!capturing && hipEventRecord(g_chip_lease, st) != hipSuccess) (void)hipGetLastError();
  return DM_OK;
}

// The kernels need more dynamic LDS than the 64 KB default: raised once per variant; a driver that refuses leaves the launch chain in charge.
template <class FN>
bool rl_raise_lds(FN fn, int slot) {
  static int state[16] = {0};      // 0 untried, 1 ok, -1 refused
  std::lock_guard<std::mutex> lk(g_mu);
  if (state[slot] == 0) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, g_lds_max);
    if (e != hipSuccess) (void)hipGetLastError();
    state[slot] = e == hipSuccess ? 1 : -1;
    // ... and the occupancy API must grant at least one workgroup of this variant per CU at the LARGEST LDS size it is launched
    // with (registers, waves): the grid never exceeds the CU count (rl_device_ok), so one per CU makes it co-resident.
    if (state[slot] == 1) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(fn), RL_THREADS, (size_t)g_lds_max) != hipSuccess) {
        (void)hipGetLastError();
        nb = 0;
      }
      if (nb < 1) state[slot] = -1;
    }
  }
  return state[slot] == 1;
}
bool rl_fwd_ready(int rl) {
  switch (rl) {
    case 8: return rl_raise_lds(rssm_lds_fwd_kernel<8>, 0);
    case 16: return rl_raise_lds(rssm_lds_fwd_kernel<16>, 1);
    case 32: return rl_raise_lds(rssm_lds_fwd_kernel<32>, 2);
    default: return rl_raise_lds(rssm_lds_fwd_kernel<64>, 3);
  }
}
}  // namespace

// 1 / 0: run the posterior chain's steps as the LDS-weight-stationary persistent kernel when the shape qualifies / always
// as launches; -1: query.  Returns the state.
extern "C" int dm_rssm_lds_enable(int on) {
  if (on >= 0) g_rssm_lds = on > 2 ? 2 : on;      // 2: also for models whose slices need less than half a CU's LDS (tests)
  return g_rssm_lds;
}
// Sticky: non-zero once a persistent kernel of this process has given up inside a spin loop (its outputs are garbage).
extern "C" int dm_rssm_lds_status(void) { return g_host_err ? (int)*(volatile unsigned*)g_host_err : 0; }
// Acknowledge a give-up: returns the status word and clears it, so the error is REPORTED once (Dreamer.check_device_status), while
// the persistent kernel stays switched off for the life of the process (g_gave_up: every later call runs the launch chain).
static int g_gave_up = 0;
extern "C" int dm_rssm_lds_status_ack(void) {
  const int st = dm_rssm_lds_status();
  if (st != 0) {
    g_gave_up = st;
    *(volatile unsigned*)g_host_err = 0u;
  }
  return st;
}
extern "C" int dm_rssm_lds_gave_up(void) { return g_gave_up; }

// Above 32 rows the launch chain is the faster posterior loop since its LayerNorm stages are done once per row instead of once
// per consuming workgroup (rssm.hip ln_z, gemm_skinny.hip row-split strips): T = 50, B = 50 2.15 vs 2.8-2.9 ms alone, the
// step 34.26 vs 34.64 ms (bf16 19.71 vs 20.21); at 25 / 13 / 7 rows the persistent kernel wins inside the step (19.69 vs 20.30,
// 13.53 vs 14.41, 9.84 vs 10.84 ms; profiles/r04_bench_fwdchain.txt).  Switch level 2 (tests, microbenchmarks) lifts the cap.
static const int g_rssm_lds_max_b = getenv("DM_RSSM_LDS_MAX_B") ? atoi(getenv("DM_RSSM_LDS_MAX_B")) : 32;
bool dm_rssm_lds_ok(int B, int D, int Hd, int S, int C) {
  RlPlan p;
  // (once a persistent kernel has given up - dm_rssm_lds_status - every later call takes the launch chain instead of failing)
  return g_rssm_lds && dm_rssm_lds_status() == 0 && g_gave_up == 0 && (B <= g_rssm_lds_max_b || g_rssm_lds >= 2) && rl_plan(B, D, Hd, S, C, &p) &&
         rl_device_ok(p.G, p.lds_bytes) && rl_fwd_ready(p.rl);
}
size_t dm_rssm_lds_ws_floats(int B, int D, int Hd, int S, int C, int steps) {
  RlPlan p;
  if (!rl_plan(B, D, Hd, S, C, &p)) return 0;
  return ((size_t)p.step_bytes * (size_t)steps + 256) / 4 + 64;
}

static unsigned long long* g_rl_prof = nullptr;      // device, 16 words; allocated on first dm_rssm_lds_prof call
extern "C" int dm_rssm_lds_prof(unsigned long long* out16, int reset) {
  if (!g_rl_prof) {
    if (hipMalloc(reinterpret_cast<void**>(&g_rl_prof), 128) != hipSuccess) return -1;
    (void)hipMemset(g_rl_prof, 0, 128);
  }
  if (out16 && hipMemcpy(out16, g_rl_prof, 128, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (reset && hipMemset(g_rl_prof, 0, 128) != hipSuccess) return -1;
  return 0;
}

int dm_rssm_lds_launch(const DmRssmLds& q, hipStream_t st) {
  RlPlan p;
  if (!rl_plan(q.B, q.D, q.Hd, q.S, q.C, &p) || !rl_device_ok(p.G, p.lds_bytes)) return dm_fail(DM_E_SHAPE, "rssm_lds: shape does not qualify");
  DM_REQUIRE(dm_rssm_lds_status() == 0 && g_gave_up == 0, DM_E_HIP, "rssm_lds: an earlier persistent posterior kernel gave up in a spin loop");
  const int steps = q.t_end - q.t_begin;
  if (steps <= 0) return DM_OK;
  RlArgs a;
  a.B = q.B; a.D = q.D; a.Hd = q.Hd; a.S = q.S; a.C = q.C; a.Z = q.S * q.C; a.ZP = q.S * q.C; a.F = q.F;
  a.t_begin = q.t_begin; a.t_end = q.t_end;
  a.nA = p.nA; a.nB = p.nB; a.nC = p.nC; a.nD = p.nD;
  a.wzt = q.wzt; a.zb = q.zb; a.wih = q.wih; a.bih = q.bih; a.whh = q.whh; a.bhh = q.bhh; a.wph = q.wph; a.bph = q.bph;
  a.wpo = q.wpo; a.bpo = q.bpo; a.in_g = q.in_g; a.in_b = q.in_b; a.post_g = q.post_g; a.post_b = q.post_b;
  a.ea = q.ea; a.ee = q.ee; a.reset = q.reset; a.u = q.u; a.forced = q.forced;
  a.x1 = q.x1; a.gi = q.gi; a.gh = q.gh; a.hin = q.hin; a.zin = q.zin; a.feat = q.feat; a.x2 = q.x2; a.post = q.post; a.idx = q.idx;
  char* base = reinterpret_cast<char*>(q.ws);
  base = reinterpret_cast<char*>(dm_align_up(reinterpret_cast<size_t>(base), 256));
  a.err = reinterpret_cast<unsigned*>(base);
  a.xch = base + 256;
  a.step_bytes = p.step_bytes; a.off_h = p.off_h; a.off_c = p.off_c; a.off_d = p.off_d; a.off_i = p.off_i;
  a.host_err = g_host_err_dev;
  a.prof = g_rl_prof;
  a.wz_global = p.wz_global;
  a.l_wz = p.l_wz; a.l_wg = p.l_wg; a.l_wh = p.l_wh; a.l_wc = p.l_wc; a.l_wd = p.l_wd; a.l_gb1 = p.l_gb1; a.l_gb2 = p.l_gb2;
  a.l_red = p.l_red; a.l_gh = p.l_gh;
  a.l_hown = p.l_hown; a.l_stage = p.l_stage; a.l_flag = p.l_flag;
  const size_t xbytes = (size_t)p.step_bytes * steps;
  DM_REQUIRE((size_t)(a.xch - reinterpret_cast<char*>(q.ws)) + xbytes <= q.ws_floats * sizeof(float), DM_E_WORKSPACE,
             "rssm_lds: exchange region needs %zu bytes", xbytes + 512);
  if (hipMemsetAsync(a.err, 0, 256, st) != hipSuccess || hipMemsetAsync(a.xch, 0xFF, xbytes, st) != hipSuccess)
    return dm_fail(DM_E_HIP, "rssm_lds: memset failed");
  const void* fn = nullptr;
  switch (p.rl) {
    case 8: fn = reinterpret_cast<const void*>(rssm_lds_fwd_kernel<8>); break;
    case 16: fn = reinterpret_cast<const void*>(rssm_lds_fwd_kernel<16>); break;
    case 32: fn = reinterpret_cast<const void*>(rssm_lds_fwd_kernel<32>); break;
    default: fn = reinterpret_cast<const void*>(rssm_lds_fwd_kernel<64>); break;
  }
  if (!rl_fwd_ready(p.rl)) return dm_fail(DM_E_HIP, "rssm_lds: the driver refused %zu bytes of dynamic LDS", p.lds_bytes);
  void* args[] = {&a};
  DM_TRY(rl_launch_exclusive(fn, (unsigned)p.G, args, p.lds_bytes, st));
  DM_LAUNCH_CHECK();
  return DM_OK;
}

