// RSSM posterior sequence with BPTT, batched prior, and the imagination rollout.
// Reference: RSSMCore.forward (rssm.py:21-78), RSSMCell.forward / forward_prior / batch_prior (rssm.py:125-193),
// nn.GRUCell via GRUCellStack (rnn.py:40-67), Dreamer.dream (dreamer.py:188-216), ActorCritic.forward_actor (a2c.py:43-55).
//
// Design notes
//  * a_mlp(action) and post_mlp_e(embed) are loop-invariant and hoisted out of the T loop as two (T*B)-row GEMMs.
//  * h and z of every step are written straight into the (T*B, D+Z) feature matrix (h at column 0, z at column D),
//    so to_feature()'s concat (rssm.py:83-84) never materialises; consumers read sub-matrices through leading dims.
//  * Only the data-path GEMMs (d? @ W) are inside the sequential BPTT loop; every weight gradient is one
//    (T*B)-row split-K GEMM after the loop, every bias / LayerNorm-parameter gradient one column-sum.
//  * reset masks (rssm.py:41,134-135) are applied forward by a tiny row-scale kernel (the masked states are saved
//    for backward) and backward through the GEMM / GRU epilogues' row_zero option.
#include "common.h"
#include <stdlib.h>

struct RssmActs {
  float *ea, *ee, *hin, *zin, *x1, *st1, *za, *gi, *gh, *x2, *st2, *pin, *x3, *st3, *prin;
  float *gs, *gst;      // LayerNorm GRU cells: pre-LayerNorm gate sums (N,3D) and their statistics (N,6 per stack layer)
};
// A/B switch (DM_FWD_LN_Z=0): the posterior launch chain's gather kernel normalises its rows itself (see ln_z in dm_rssm_sequence_fwd_steps)
static const int g_fwd_ln_z = getenv("DM_FWD_LN_Z") ? atoi(getenv("DM_FWD_LN_Z")) : 1;
static inline int rssm_gru_kind(const dm_shape* s) { return (s->flags & DM_FLAG_GRU_MASK) >> DM_FLAG_GRU_SHIFT; }
static inline int rssm_gru_layers(const dm_shape* s) {
  return 1 + ((s->flags & DM_FLAG_GRU_LAYERS_MASK) >> DM_FLAG_GRU_LAYERS_SHIFT);
}
static size_t rssm_carve(const dm_shape* s, float* base, RssmActs* a) {
  const size_t N = (size_t)s->T * s->B, Hd = s->Hd, D = s->D, Z = (size_t)s->S * (s->C ? s->C : 1);   // width of z
  DmArena ar(base, (size_t)1 << 62);
  RssmActs t;
  t.ea = ar.take(N * Hd); t.ee = ar.take(N * Hd);
  t.hin = ar.take(N * D); t.zin = ar.take(N * Z);
  t.x1 = ar.take(N * Hd); t.st1 = ar.take(N * 2); t.za = ar.take(N * Hd);
  t.gi = ar.take(N * 3 * D); t.gh = ar.take(N * 3 * D);
  t.x2 = ar.take(N * Hd); t.st2 = ar.take(N * 2); t.pin = ar.take(N * Hd);
  t.x3 = ar.take(N * Hd); t.st3 = ar.take(N * 2); t.prin = ar.take(N * Hd);
  const bool lncell = rssm_gru_kind(s) != 0;
  t.gs = ar.take(lncell ? N * 3 * D : 0); t.gst = ar.take(lncell ? N * 6 * rssm_gru_layers(s) : 0);
  if (a) *a = t;
  return ar.off;
}
extern "C" size_t dm_rssm_acts_floats(const dm_shape* shp) {
  if (!shp) return 0;
  return rssm_carve(shp, nullptr, nullptr);
}

static int rssm_check(const dm_shape* s) {
  DM_REQUIRE(s->I == 1, DM_E_SHAPE, "rssm: iwae_samples=%d unsupported (only 1)", s->I);
  DM_REQUIRE(s->T >= 1 && s->B >= 1 && s->D >= 4 && s->Hd >= 4 && s->S >= 1 && (s->C >= 2 || s->C == 0) && s->A >= 1 && s->E >= 1,
             DM_E_SHAPE, "rssm: bad shape");
  DM_REQUIRE((s->D & 3) == 0, DM_E_SHAPE, "rssm: deter_dim must be a multiple of 4 (got %d)", s->D);
  DM_REQUIRE(rssm_gru_kind(s) <= 2, DM_E_SHAPE, "rssm: unknown recurrent cell kind %d", rssm_gru_kind(s));
  const int GL = rssm_gru_layers(s);
  DM_REQUIRE(s->D % (4 * GL) == 0, DM_E_SHAPE, "rssm: deter_dim=%d must be a multiple of 4*gru_layers (%d)", s->D, 4 * GL);
  return DM_OK;
}

// GRUCellStack (rnn.py:40-67): L cells (nn.GRUCell, or one of the two LayerNorm cells) of width ls = D/L.  Layer i reads x_i (x_0 = the cell input, x_i = the NEW
// state of layer i-1) and its own slice [i*ls, (i+1)*ls) of the incoming state, and writes the same slice of the new
// state.  The gate products of layer i live at columns [3*ls*i, 3*ls*(i+1)) of the (rows, 3D) gi / gh matrices (and of
// gs / dg for the LayerNorm cells, whose statistics of layer i are columns [6i, 6i+6) of the (rows, 6L) gst matrix).
struct GruStack {
  int L, ls, kind;
  const float *wih[4], *whh[4], *bih[4], *bhh[4];
  float *g_wih[4], *g_whh[4], *g_bih[4], *g_bhh[4];
  const float *lng[4][3], *lnb[4][3];      // LayerNorm cells: (reset, update, newval) or slot 0 = the one 3*ls-wide LayerNorm
  float *g_lng[4][3], *g_lnb[4][3];
};
static int gru_stack(const dm_shape* s, const float* const* p, float* const* g, GruStack* k) {
  k->L = rssm_gru_layers(s);
  k->ls = s->D / k->L;
  k->kind = rssm_gru_kind(s);
  for (int i = 0; i < k->L; ++i) {
    const int lb = i == 0 ? DM_RSSM_GRU_LN_G0 : DM_RSSM_GRU_L1_LN_G0 + 6 * (i - 1);
    for (int q = 0; q < 3; ++q) {
      k->lng[i][q] = p[lb + 2 * q]; k->lnb[i][q] = p[lb + 2 * q + 1];
      k->g_lng[i][q] = g ? g[lb + 2 * q] : nullptr; k->g_lnb[i][q] = g ? g[lb + 2 * q + 1] : nullptr;
      const bool need = k->kind == 1 || (k->kind == 2 && q == 0);
      DM_REQUIRE(!need || (k->lng[i][q] && k->lnb[i][q]), DM_E_NULL, "rssm: LayerNorm GRU layer %d without LayerNorm parameter %d",
                 i, q);
      DM_REQUIRE(!need || !g || (k->g_lng[i][q] && k->g_lnb[i][q]), DM_E_NULL,
                 "rssm: LayerNorm GRU layer %d without LayerNorm gradient slot %d", i, q);
    }
    const int b = i == 0 ? DM_RSSM_GRU_WIH : DM_RSSM_GRU_L1_WIH + 4 * (i - 1);
    k->wih[i] = p[b]; k->whh[i] = p[b + 1]; k->bih[i] = p[b + 2]; k->bhh[i] = p[b + 3];
    const bool biased = rssm_gru_kind(s) == 0;      // the LayerNorm cells have no gate biases (rnn.py:99-100)
    DM_REQUIRE(k->wih[i] && k->whh[i] && (!biased || (k->bih[i] && k->bhh[i])), DM_E_NULL,
               "rssm: GRU layer %d has a null parameter", i);
    if (g) {
      k->g_wih[i] = g[b]; k->g_whh[i] = g[b + 1]; k->g_bih[i] = g[b + 2]; k->g_bhh[i] = g[b + 3];
      DM_REQUIRE(k->g_wih[i] && k->g_whh[i] && (!biased || (k->g_bih[i] && k->g_bhh[i])), DM_E_NULL,
                 "rssm: GRU layer %d has a null gradient slot", i);
    }
  }
  return DM_OK;
}

// The cell's three norms (rssm.py:103-116) are nn.LayerNorm(eps 1e-3) or, with layer_norm=False, NoNorm (common.py:68-74:
// identity, no parameters): a null gain selects the activation alone, and no statistics / parameter gradients exist.
static int norm_elu_fwd(int rows, int n, const float* x, int ldx, const float* gamma, const float* beta, float eps, float* y,
                        int ldy, float* stats, hipStream_t st) {
  if (gamma) return dm_ln_elu_fwd_launch(rows, n, x, ldx, gamma, beta, eps, y, ldy, stats, st);
  return dm_elu_fwd_launch(rows, n, x, ldx, y, ldy, st);
}
static int norm_elu_bwd_dx(int rows, int n, const float* x, int ldx, const float* y, int ldy, const float* stats,
                           const float* gamma, const float* dy, int lddy, float* dx, int lddx, hipStream_t st) {
  if (gamma) return dm_ln_elu_bwd_dx_launch(rows, n, x, ldx, y, ldy, stats, gamma, dy, lddy, dx, lddx, st);
  return dm_elu_bwd_launch(rows, n, y, ldy, dy, lddy, dx, lddx, st);
}
static int norm_elu_bwd_params(int rows, int n, const float* x, int ldx, const float* y, int ldy, const float* stats,
                               const float* dy, int lddy, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                               hipStream_t st) {
  if (!dgamma && !dbeta) return DM_OK;
  return dm_ln_elu_bwd_params_launch(rows, n, x, ldx, y, ldy, stats, dy, lddy, dgamma, dbeta, ws, ws_bytes, st);
}

// y = x @ W^T (+ bias) (+ add)
static int linear(hipStream_t st, void* sk, size_t skb, int rows, int nout, int kin, const float* x, int ldx,
                  const float* W, const float* bias, const float* add, int ldadd, float* y, int ldy) {
  DmGemm q;
  q.M = rows; q.N = nout; q.K = kin;
  q.A = x; q.lda = ldx;
  q.B = W; q.ldb = kin;
  q.C = y; q.ldc = ldy;
  q.bias = bias; q.add = add; q.ldadd = ldadd;
  return dm_gemm_launch(q, sk, skb, st);
}
// dW[o][i] = sum_r dy[r][o] x[r][i]
static int wgrad(hipStream_t st, void* sk, size_t skb, int rows, int nout, int kin, const float* dy, int lddy,
                 const float* x, int ldx, float* dW, int accum = 0) {
  DmGemm q;
  q.a_layout = 1; q.b_layout = 1;
  q.M = nout; q.N = kin; q.K = rows;
  q.A = dy; q.lda = lddy;
  q.B = x; q.ldb = ldx;
  q.C = dW; q.ldc = kin;
  q.flags = accum ? DM_GEMM_ACCUM : 0;
  return dm_gemm_launch(q, sk, skb, st);
}
// dx[r][i] (+)= mask_r * sum_o dy[r][o] W[o][i]
static int dgrad(hipStream_t st, void* sk, size_t skb, int rows, int nout, int kin, const float* dy, int lddy,
                 const float* W, float* dx, int lddx, int accum, const uint8_t* row_zero) {
  DmGemm q;
  q.a_layout = 0; q.b_layout = 1;
  q.M = rows; q.N = kin; q.K = nout;
  q.A = dy; q.lda = lddy;
  q.B = W; q.ldb = kin;
  q.C = dx; q.ldc = lddx;
  q.flags = accum ? DM_GEMM_ACCUM : 0;
  q.row_zero = row_zero;
  return dm_gemm_launch(q, sk, skb, st);
}

// dx[r][i] (+)= mask_r * sum_o dy[r][o] Wt[i][o]   (Wt = W^T materialised once per backward pass: the B-row chain
// products of the BPTT loop then read their weights k-contiguous, the layout the skinny kernel streams fastest)
static int dgrad_t(hipStream_t st, void* sk, size_t skb, int rows, int nout, int kin, const float* dy, int lddy,
                   const float* Wt, float* dx, int lddx, int accum, const uint8_t* row_zero) {
  DmGemm q;
  q.M = rows; q.N = kin; q.K = nout;
  q.A = dy; q.lda = lddy;
  q.B = Wt; q.ldb = nout;
  q.C = dx; q.ldc = lddx;
  q.flags = accum ? DM_GEMM_ACCUM : 0;
  q.row_zero = row_zero;
  return dm_gemm_launch(q, sk, skb, st);
}
static int transpose(hipStream_t st, const float* W, float* Wt, int rows, int cols) {
  return dm_permute4_launch(W, Wt, 1, 1, rows, cols, 0, 1, 3, 2, st);
}

// One step of the stack, forward: 3 launches per layer.  `hout` may alias nothing of `hin`.
static int gru_stack_fwd(hipStream_t st, void* sk, size_t skb, const GruStack& k, int rows, int Hd, int D, const float* x0,
                         const float* hin, int ldh, float* gi, float* gh, float* hout, int ldo, float* h_next,
                         const uint8_t* next_reset, float* gs, float* gst) {
  const int ls = k.ls;
  for (int i = 0; i < k.L; ++i) {
    const float* x = i == 0 ? x0 : hout + (size_t)(i - 1) * ls;
    const int ldx = i == 0 ? Hd : ldo, kin = i == 0 ? Hd : ls;
    DM_TRY(linear(st, sk, skb, rows, 3 * ls, kin, x, ldx, k.wih[i], k.bih[i], nullptr, 0, gi + 3 * ls * i, 3 * D));
    DM_TRY(linear(st, sk, skb, rows, 3 * ls, ls, hin + i * ls, ldh, k.whh[i], k.bhh[i], nullptr, 0, gh + 3 * ls * i, 3 * D));
    if (k.kind == 0)
      DM_TRY(dm_gru_gates_fwd_launch(rows, ls, gi + 3 * ls * i, gh + 3 * ls * i, hin + i * ls, ldh, hout + i * ls, ldo,
                                     h_next ? h_next + i * ls : nullptr, next_reset, nullptr, nullptr, st, 3 * D, D));
    else
      DM_TRY(dm_gru_norm_fwd_launch(k.kind, rows, ls, gi + 3 * ls * i, gh + 3 * ls * i, hin + i * ls, ldh, k.lng[i], k.lnb[i],
                                    hout + i * ls, ldo, gs + 3 * ls * i, gst + 6 * i, h_next ? h_next + i * ls : nullptr,
                                    next_reset, st, 3 * D, 6 * k.L, D));
  }
  return DM_OK;
}

// Time steps [t0, t1) of the sequence; all buffers are the full (T*B)-row ones.  Step t0 > 0 continues from the state
// that step t0-1 left in `feat`, so consecutive ranges issued in order on one stream equal one full call; the encoder
// range that feeds them and the decoder range that consumes them can then run on other streams (see WorldModel._forward).
extern "C" int dm_rssm_sequence_fwd_steps(const dm_shape* s, int t0, int t1, const float* embed, const float* action,
                                          const uint8_t* reset, const float* h0, const float* z0, const float* u,
                                          const int32_t* forced_idx, const dm_rssm_params* P, float* acts, float* feat,
                                          float* post, float* prior, int32_t* idx, void* ws, size_t ws_bytes,
                                          void* stream) {
  DM_REQUIRE(s && embed && action && reset && h0 && z0 && P && acts && feat && post && prior && ws, DM_E_NULL,
             "rssm_sequence_fwd: null pointer");
  DM_REQUIRE(u || forced_idx, DM_E_NULL, "rssm_sequence_fwd: need uniforms or forced indices");
  DmPrecisionScope prec(s->flags & DM_FLAG_BF16);
  DM_TRY(rssm_check(s));
  DM_REQUIRE(t0 >= 0 && t0 <= t1 && t1 <= s->T, DM_E_SHAPE, "rssm_sequence_fwd: step range [%d,%d) outside 0..%d", t0, t1, s->T);
  if (t0 == t1) return DM_OK;
  DM_REQUIRE(ws_bytes >= DM_SPLITK_FLOATS * sizeof(float), DM_E_WORKSPACE, "rssm_sequence_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  // Z: width of z (S one-hot groups of C, or S Gaussian dimensions when C = 0); ZP: width of the posterior / prior
  // parameters (logits, or mean | raw std - rssm.py:112,117)
  const int T = s->T, B = s->B, D = s->D, Hd = s->Hd, S = s->S, C = s->C, Z = S * (C ? C : 1), ZP = S * (C ? C : 2);
  const int F = D + Z, E = s->E, A = s->A;
  const bool gauss = C == 0;
  (void)T;
  const int N = (t1 - t0) * B;                        // rows of this range
  const size_t q0 = (size_t)t0 * B;                   // its first row
  const size_t skb = DM_SPLITK_FLOATS * sizeof(float);
  RssmActs a;
  rssm_carve(s, acts, &a);
  const float* const* p = P->p;


  DM_TRY(linear(st, ws, skb, N, Hd, A, action + q0 * A, A, p[DM_RSSM_A_W], nullptr, nullptr, 0, a.ea + q0 * Hd, Hd));
  DM_TRY(linear(st, ws, skb, N, Hd, E, embed + q0 * E, E, p[DM_RSSM_POST_E_W], nullptr, nullptr, 0, a.ee + q0 * Hd, Hd));

  // Fused schedule (5 launches per step instead of 8) when the <= 64-row products qualify: the two LayerNorm+ELU stages
  // ride in the PROLOGUE of the product that consumes them (each workgroup recomputes the row statistics of its <= 64
  // rows from L2) and the straight-through sampler rides in the EPILOGUE of the posterior-logits product (one 32-logit
  // group per workgroup).  The post-LayerNorm activations `za` / `pin` that only the backward pass needs (weight
  // gradients, ELU') are then produced for ALL rows of the range by two batched launches after the loop.
  const int kind = rssm_gru_kind(s);
  const float* lng[3] = {p[DM_RSSM_GRU_LN_G0], p[DM_RSSM_GRU_LN_G1], p[DM_RSSM_GRU_LN_G2]};
  const float* lnb[3] = {p[DM_RSSM_GRU_LN_B0], p[DM_RSSM_GRU_LN_B1], p[DM_RSSM_GRU_LN_B2]};
  DM_REQUIRE(kind == 0 || (lng[0] && lnb[0] && (kind == 2 || (lng[1] && lnb[1] && lng[2] && lnb[2]))), DM_E_NULL,
             "rssm_sequence_fwd: LayerNorm GRU cell without its LayerNorm parameters");
  GruStack gk;
  DM_TRY(gru_stack(s, p, nullptr, &gk));
  const bool stacked = gk.L > 1;      // GRUCellStack with several layers: the unfused schedule, 3 launches per layer
  const bool normed = p[DM_RSSM_IN_G] != nullptr;      // layer_norm=False: all three norms are NoNorm (null parameters)
  DM_REQUIRE((p[DM_RSSM_POST_G] != nullptr) == normed && (p[DM_RSSM_PRIOR_G] != nullptr) == normed, DM_E_NULL,
             "rssm: the cell's three norms must be all LayerNorm or all NoNorm");
  const bool fuse_ln = normed && !stacked && !gauss && dm_skinny_ln_ok(B, 3 * D, Hd) && dm_skinny_ln_ok(B, ZP, Hd) &&
                       (ZP >= 64 * 1024 / Hd);
  static const int no_fuse_sample = getenv("DM_RSSM_NO_FUSE_SAMPLE") ? 1 : 0;      // A/B switch
  const bool fuse_sample = !no_fuse_sample && fuse_ln && C == 32 && (Z & 31) == 0 && (F & 3) == 0 && (D & 3) == 0 &&
                           (((uintptr_t)feat | (uintptr_t)a.zin) & 15) == 0;
  // Fragment-major copies of the chain's <= 64-row operands (common.h dm_frag_off), written by the kernel that produces
  // each operand next to its ordinary copy and read by the product that consumes it: z_in -> x1 -> (gi | gh from h_in)
  // -> h -> x2 -> z.  One buffer per operand is enough (producer and consumer alternate in stream order).
  static const int no_frag = getenv("DM_SKINNY_NO_FRAG") ? 1 : 0;          // A/B switch
  float *zinf = nullptr, *x1f = nullptr, *hinf = nullptr, *hf = nullptr, *x2f = nullptr;
  DmArena ar(ws, ws_bytes);
  ar.take(DM_SPLITK_FLOATS);
  // z_mlp of the sampled (one-hot) latent as a gather-sum over rows of z_mlp^T (dm_z_embed_launch): every step after the
  // first of a range takes its z from the sampler, whose indices are at hand; the first step's z comes from the caller as
  // a dense vector and keeps the product.
  static const int no_embed = getenv("DM_RSSM_NO_Z_EMBED") ? 1 : 0;        // A/B switch
  float* wzt = nullptr;
  if (!no_embed && !gauss && idx && t1 - t0 > 1 && dm_z_embed_ok(Hd)) {
    const size_t mark = ar.off;
    float* w = ar.take((size_t)Z * Hd);
    if (ar.ok) {
      wzt = w;
      DM_TRY(transpose(st, p[DM_RSSM_Z_W], wzt, Hd, Z));
    } else {      // optional: a small workspace keeps the product
      ar.off = mark; ar.ok = true;
    }
  }
  if (!no_frag && fuse_sample && kind == 0 && B <= 64) {
    float* f0 = ar.take(dm_frag_floats(Z)); float* f1 = ar.take(dm_frag_floats(Hd)); float* f2 = ar.take(dm_frag_floats(D));
    float* f3 = ar.take(dm_frag_floats(D)); float* f4 = ar.take(dm_frag_floats(Hd));
    if (ar.ok) { zinf = f0; x1f = f1; hinf = f2; hf = f3; x2f = f4; }
  }
  // The fused schedule's steps after the first of a range as ONE persistent kernel with the cell's weights stationary in
  // LDS (rssm_lds.hip): the first step runs as launches (its z is a dense vector from the caller) and leaves h, the masked
  // inputs and the indices the kernel's first step continues from.
  float* psync = nullptr;
  size_t psync_floats = 0;
  if (normed && !stacked && !gauss && kind == 0 && wzt && idx && (F & 3) == 0 && t1 - t0 >= 3 && dm_rssm_lds_ok(B, D, Hd, S, C)) {
    psync_floats = dm_rssm_lds_ws_floats(B, D, Hd, S, C, t1 - t0 - 1);
    float* sy = ar.take(psync_floats);
    if (ar.ok) psync = sy;
    else ar.ok = true;
  }
  const int t_launch_end = psync ? t0 + 1 : t1;
  // 8 launches per step otherwise: the reset masks of step t+1 are applied by the kernels that produce h_t and z_t (only
  // the first step of a range needs the stand-alone mask kernel), and the GRU's two gate products share one launch.
  for (int t = t0; t < t_launch_end; ++t) {
    const size_t r0 = (size_t)t * B;
    float* hin = a.hin + r0 * D;
    float* zin = a.zin + r0 * Z;
    if (t == t0) {
      const float* ph = t == 0 ? h0 : feat + (r0 - B) * F;
      const float* pz = t == 0 ? z0 : feat + (r0 - B) * F + D;
      const int ldp_h = t == 0 ? D : F, ldp_z = t == 0 ? Z : F;
      DM_TRY(dm_mask_rows2_launch(B, D, ph, ldp_h, hin, D, Z, pz, ldp_z, zin, Z, reset + r0, st));
      if (zinf) {
        DM_TRY(dm_frag_pack_launch(B, D, hin, D, hinf, st));
        DM_TRY(dm_frag_pack_launch(B, Z, zin, Z, zinf, st));
      }
    }
    const bool more = t + 1 < t1;
    float* hin_next = more ? a.hin + (r0 + B) * D : nullptr;
    float* zin_next = more ? a.zin + (r0 + B) * Z : nullptr;
    const uint8_t* reset_next = more ? reset + r0 + B : nullptr;
    // x = z_mlp(z) + a_mlp(a) ; za = ELU(in_norm(x))                                   rssm.py:138-140
    // (ln_z: the gather kernel owns complete rows, so it also normalises them and the gate product below runs plain - the
    //  prologue form makes each of that product's 226 workgroups redo the LayerNorm + ELU of the whole operand)
    const bool ln_z = fuse_ln && g_fwd_ln_z && wzt && t > t0 && x1f && !stacked && S <= 32;     // the row-per-workgroup form holds <= 32 groups (stoch_dim 64 / 96 take the branch below)
    if (ln_z) {
      DM_TRY(dm_z_embed_launch(B, Hd, S, C, idx + (r0 - B) * S, reset + r0, wzt, p[DM_RSSM_Z_B], a.ea + r0 * Hd, Hd,
                               nullptr, nullptr, a.x1 + r0 * Hd, Hd, nullptr, p[DM_RSSM_IN_G], p[DM_RSSM_IN_B], 1e-3f,
                               a.za + r0 * Hd, Hd, st, x1f, a.st1 + r0 * 2));
    } else if (wzt && t > t0) {
      DM_TRY(dm_z_embed_launch(B, Hd, S, C, idx + (r0 - B) * S, reset + r0, wzt, p[DM_RSSM_Z_B], a.ea + r0 * Hd, Hd,
                               nullptr, nullptr, a.x1 + r0 * Hd, Hd, x1f, nullptr, nullptr, 0.f, nullptr, 0, st));
    } else {
      DmGemm q;
      q.M = B; q.N = Hd; q.K = Z; q.A = zin; q.lda = Z; q.B = p[DM_RSSM_Z_W]; q.ldb = Z; q.C = a.x1 + r0 * Hd; q.ldc = Hd;
      q.bias = p[DM_RSSM_Z_B]; q.add = a.ea + r0 * Hd; q.ldadd = Hd;
      q.A_frag = zinf; q.C_frag = x1f;
      DM_TRY(dm_gemm_launch(q, ws, skb, st));
    }
    if (!fuse_ln)
      DM_TRY(norm_elu_fwd(B, Hd, a.x1 + r0 * Hd, Hd, p[DM_RSSM_IN_G], p[DM_RSSM_IN_B], 1e-3f, a.za + r0 * Hd, Hd,
                                  a.st1 + r0 * 2, st));
    // h = GRUCell(za, h_in)                                                             rssm.py:141
    if (stacked) {
      DM_TRY(gru_stack_fwd(st, ws, skb, gk, B, Hd, D, a.za + r0 * Hd, hin, D, a.gi + r0 * 3 * D, a.gh + r0 * 3 * D,
                           feat + r0 * F, F, hin_next, reset_next, kind ? a.gs + r0 * 3 * D : nullptr,
                           kind ? a.gst + r0 * 6 * gk.L : nullptr));
    } else {
      DmGemm gi_q, gh_q;
      gi_q.M = B; gi_q.N = 3 * D; gi_q.K = Hd; gi_q.A = a.za + r0 * Hd; gi_q.lda = Hd; gi_q.B = p[DM_RSSM_GRU_WIH]; gi_q.ldb = Hd;
      if (ln_z) {
        gi_q.A_frag = x1f;      // (holds ELU(in_norm(x1)) in this form)
      } else if (fuse_ln) {
        gi_q.A = a.x1 + r0 * Hd; gi_q.ln_g = p[DM_RSSM_IN_G]; gi_q.ln_b = p[DM_RSSM_IN_B]; gi_q.ln_eps = 1e-3f;
        gi_q.A_frag = x1f;
      }
      gh_q.A_frag = hinf;
      gi_q.C = a.gi + r0 * 3 * D; gi_q.ldc = 3 * D; gi_q.bias = p[DM_RSSM_GRU_BIH];
      gh_q.M = B; gh_q.N = 3 * D; gh_q.K = D; gh_q.A = hin; gh_q.lda = D; gh_q.B = p[DM_RSSM_GRU_WHH]; gh_q.ldb = D;
      gh_q.C = a.gh + r0 * 3 * D; gh_q.ldc = 3 * D; gh_q.bias = p[DM_RSSM_GRU_BHH];
      DM_TRY(dm_gemm_pair_launch(gi_q, gh_q, ws, skb, st));
    }
    if (stacked) {
    } else if (kind == 0)
      DM_TRY(dm_gru_gates_fwd_launch(B, D, a.gi + r0 * 3 * D, a.gh + r0 * 3 * D, hin, D, feat + r0 * F, F, hin_next,
                                     reset_next, hf, more ? hinf : nullptr, st));
    else
      DM_TRY(dm_gru_norm_fwd_launch(kind, B, D, a.gi + r0 * 3 * D, a.gh + r0 * 3 * D, hin, D, lng, lnb, feat + r0 * F, F,
                                    a.gs + r0 * 3 * D, a.gst + r0 * 6, hin_next, reset_next, st));
    // post = post_mlp(ELU(post_norm(post_mlp_h(h) + post_mlp_e(embed))))               rssm.py:143-146
    {
      DmGemm q;
      q.M = B; q.N = Hd; q.K = D; q.A = feat + r0 * F; q.lda = F; q.B = p[DM_RSSM_POST_H_W]; q.ldb = D;
      q.C = a.x2 + r0 * Hd; q.ldc = Hd; q.bias = p[DM_RSSM_POST_H_B]; q.add = a.ee + r0 * Hd; q.ldadd = Hd;
      q.A_frag = hf; q.C_frag = x2f;
      DM_TRY(dm_gemm_launch(q, ws, skb, st));
    }
    if (fuse_ln) {
      DmGemm pq;      // post = post_mlp(ELU(post_norm(x2))), LayerNorm in the prologue
      pq.M = B; pq.N = ZP; pq.K = Hd; pq.A = a.x2 + r0 * Hd; pq.lda = Hd; pq.B = p[DM_RSSM_POST_W]; pq.ldb = Hd;
      pq.C = post + r0 * ZP; pq.ldc = ZP; pq.bias = p[DM_RSSM_POST_OB];
      pq.ln_g = p[DM_RSSM_POST_G]; pq.ln_b = p[DM_RSSM_POST_B]; pq.ln_eps = 1e-3f;
      pq.A_frag = x2f;
      if (fuse_sample) {   // ... and z ~ OneHotCategoricalStraightThrough(post) in the epilogue        rssm.py:147-148
        DmSample sm;
        sm.u = u ? u + r0 * S : nullptr; sm.forced = forced_idx ? forced_idx + r0 * S : nullptr;
        sm.onehot = feat + r0 * F + D; sm.ldo = F; sm.idx = idx ? idx + r0 * S : nullptr;
        sm.z_next = zin_next; sm.next_reset = reset_next; sm.z_next_frag = zinf;
        DM_TRY(dm_gemm_sample_launch(pq, sm, st));
        continue;
      }
      DM_TRY(dm_gemm_launch(pq, ws, skb, st));
    } else {
      DM_TRY(norm_elu_fwd(B, Hd, a.x2 + r0 * Hd, Hd, p[DM_RSSM_POST_G], p[DM_RSSM_POST_B], 1e-3f, a.pin + r0 * Hd,
                                  Hd, a.st2 + r0 * 2, st));
      DM_TRY(linear(st, ws, skb, B, ZP, Hd, a.pin + r0 * Hd, Hd, p[DM_RSSM_POST_W], p[DM_RSSM_POST_OB], nullptr, 0,
                    post + r0 * ZP, ZP));
    }
    // z ~ OneHotCategoricalStraightThrough(post)                                       rssm.py:147-148
    DM_TRY(dm_sample_onehot_launch(B, S, C, post + r0 * ZP, ZP, u ? u + r0 * S : nullptr,
                                   forced_idx ? forced_idx + r0 * S : nullptr, feat + r0 * F + D, F,
                                   idx ? idx + r0 * S : nullptr, zin_next, reset_next, st));
  }
  if (psync) {
    DmRssmLds pq;
    pq.B = B; pq.D = D; pq.Hd = Hd; pq.S = S; pq.C = C; pq.F = F; pq.t_begin = t0 + 1; pq.t_end = t1;
    pq.wzt = wzt; pq.zb = p[DM_RSSM_Z_B];
    pq.wih = p[DM_RSSM_GRU_WIH]; pq.bih = p[DM_RSSM_GRU_BIH]; pq.whh = p[DM_RSSM_GRU_WHH]; pq.bhh = p[DM_RSSM_GRU_BHH];
    pq.wph = p[DM_RSSM_POST_H_W]; pq.bph = p[DM_RSSM_POST_H_B]; pq.wpo = p[DM_RSSM_POST_W]; pq.bpo = p[DM_RSSM_POST_OB];
    pq.in_g = p[DM_RSSM_IN_G]; pq.in_b = p[DM_RSSM_IN_B]; pq.post_g = p[DM_RSSM_POST_G]; pq.post_b = p[DM_RSSM_POST_B];
    pq.ea = a.ea; pq.ee = a.ee; pq.reset = reset; pq.u = u; pq.forced = forced_idx;
    pq.x1 = a.x1; pq.gi = a.gi; pq.gh = a.gh; pq.hin = a.hin; pq.zin = a.zin; pq.feat = feat; pq.x2 = a.x2; pq.post = post;
    pq.idx = idx; pq.ws = psync; pq.ws_floats = psync_floats;
    DM_TRY(dm_rssm_lds_launch(pq, st));
  }
  if (fuse_ln || psync) {     // what only the backward pass reads: post-LayerNorm activations + statistics of every row of the range
    DM_TRY(norm_elu_fwd(N, Hd, a.x1 + q0 * Hd, Hd, p[DM_RSSM_IN_G], p[DM_RSSM_IN_B], 1e-3f, a.za + q0 * Hd, Hd,
                                a.st1 + q0 * 2, st));
    DM_TRY(norm_elu_fwd(N, Hd, a.x2 + q0 * Hd, Hd, p[DM_RSSM_POST_G], p[DM_RSSM_POST_B], 1e-3f, a.pin + q0 * Hd, Hd,
                                a.st2 + q0 * 2, st));
  }
  // batch_prior over all (T*B) rows                                                    rssm.py:61,186-193
  DM_TRY(linear(st, ws, skb, N, Hd, D, feat + q0 * F, F, p[DM_RSSM_PRIOR_H_W], p[DM_RSSM_PRIOR_H_B], nullptr, 0,
                a.x3 + q0 * Hd, Hd));
  DM_TRY(norm_elu_fwd(N, Hd, a.x3 + q0 * Hd, Hd, p[DM_RSSM_PRIOR_G], p[DM_RSSM_PRIOR_B], 1e-3f, a.prin + q0 * Hd,
                              Hd, a.st3 + q0 * 2, st));
  DM_TRY(linear(st, ws, skb, N, ZP, Hd, a.prin + q0 * Hd, Hd, p[DM_RSSM_PRIOR_W], p[DM_RSSM_PRIOR_OB], nullptr, 0,
                prior + q0 * ZP, ZP));
  return DM_OK;
}
extern "C" int dm_rssm_sequence_fwd(const dm_shape* s, const float* embed, const float* action, const uint8_t* reset,
                                    const float* h0, const float* z0, const float* u, const int32_t* forced_idx,
                                    const dm_rssm_params* P, float* acts, float* feat, float* post, float* prior,
                                    int32_t* idx, void* ws, size_t ws_bytes, void* stream) {
  DM_REQUIRE(s, DM_E_NULL, "rssm_sequence_fwd: null shape");
  return dm_rssm_sequence_fwd_steps(s, 0, s->T, embed, action, reset, h0, z0, u, forced_idx, P, acts, feat, post, prior,
                                    idx, ws, ws_bytes, stream);
}

// A/B switch of the BPTT launch schedule's folded LayerNorm backward (include/dreamer_hip.h dm_bptt_fold_enable; DM_BPTT_FOLD=0 in the environment)
static int g_bptt_fold = getenv("DM_BPTT_FOLD") ? (atoi(getenv("DM_BPTT_FOLD")) ? 1 : 0) : 1;
extern "C" int dm_bptt_fold_enable(int on) {
  const int was = g_bptt_fold;
  if (on >= 0) g_bptt_fold = on ? 1 : 0;
  return was;
}
extern "C" int dm_rssm_sequence_bwd(const dm_shape* s, const float* embed, const float* action, const uint8_t* reset,
                                    const dm_rssm_params* P, const float* acts, const float* feat, const float* post,
                                    float* dfeat, float* dpost, float* dprior, const dm_rssm_grads* G, float* dembed,
                                    void* ws, size_t ws_bytes, void* stream) {
  DM_REQUIRE(s && embed && action && reset && P && acts && feat && post && dfeat && dpost && dprior && G && ws, DM_E_NULL,
             "rssm_sequence_bwd: null pointer");
  DmPrecisionScope prec(s->flags & DM_FLAG_BF16);
  DM_TRY(rssm_check(s));
  hipStream_t st = (hipStream_t)stream;
  const int T = s->T, B = s->B, D = s->D, Hd = s->Hd, S = s->S, C = s->C, Z = S * (C ? C : 1), ZP = S * (C ? C : 2);
  const int F = D + Z, E = s->E, A = s->A;        // Z: width of z, ZP: width of its distribution's parameters (see above)
  const bool gauss = C == 0;
  const int N = T * B;
  RssmActs a;
  rssm_carve(s, const_cast<float*>(acts), &a);
  const float* const* p = P->p;
  float* const* g = G->p;

  DmArena ar(ws, ws_bytes);
  float* sk = ar.take(DM_SPLITK_FLOATS);
  float* dprin = ar.take((size_t)N * Hd);
  float* dx3 = ar.take((size_t)N * Hd);
  float* dpin = ar.take((size_t)N * Hd);
  float* dx2 = ar.take((size_t)N * Hd);
  float* dgi = ar.take((size_t)N * 3 * D);
  float* dgh = ar.take((size_t)N * 3 * D);
  float* dza = ar.take((size_t)N * Hd);
  float* dx1 = ar.take((size_t)N * Hd);
  float* wt_post = ar.take((size_t)ZP * Hd);
  float* wt_post_h = ar.take((size_t)Hd * D);
  float* wt_ih = ar.take((size_t)3 * D * Hd);
  float* wt_hh = ar.take((size_t)3 * D * D);
  float* wt_z = ar.take((size_t)Hd * Z);
  const int kind = rssm_gru_kind(s);
  float* dgl = ar.take(kind ? (size_t)N * 3 * D : 0);       // LayerNorm GRU cells: gradients w.r.t. the LayerNorm outputs
  float* lnpg = ar.take(kind ? (size_t)3 * D : 0);
  float* lnpb = ar.take(kind ? (size_t)3 * D : 0);
  // the weight-gradient side stream's own split-K scratch and its own dx2 / dx1 (fused schedule: the LayerNorm backward of a
  // chunk of rows is redone there instead of being shared with the chain)
  float* sk_w = ar.take(DM_SPLITK_FLOATS);
  float* dx2_w = ar.take((size_t)N * Hd);
  float* dx1_w = ar.take((size_t)N * Hd);
  const float* lng[3] = {p[DM_RSSM_GRU_LN_G0], p[DM_RSSM_GRU_LN_G1], p[DM_RSSM_GRU_LN_G2]};
  const float* lnb[3] = {p[DM_RSSM_GRU_LN_B0], p[DM_RSSM_GRU_LN_B1], p[DM_RSSM_GRU_LN_B2]};
  DM_REQUIRE(ar.ok, DM_E_WORKSPACE, "rssm_sequence_bwd: workspace too small (need %zu floats)", ar.off);
  const size_t skb = DM_SPLITK_FLOATS * sizeof(float);

  // Parameter gradients are leaves of this pass: nothing reads them before the gradient clip.  They go to `sw` - the
  // library's weight-gradient side stream when the calling thread is armed (include/dreamer_hip.h dm_wgrad_side_arm), else
  // `st` itself - and the big ones are cut into up to four time chunks that are launched as soon as the BPTT loop has finished
  // their rows, so they run BESIDE the loop (a B-row latency chain that leaves most CUs idle) instead of behind it.  The chunk
  // boundaries, and with them every sum, are the same whether or not a side stream is used.
  hipStream_t const st_main = st;
  hipStream_t sw = st == (hipStream_t)stream ? dm_wgrad_side_stream(st) : st;      // (no side stream inside a chain-graph capture)

  // ---- prior branch, batched over all rows: the data gradient on st ...
  DM_TRY(dgrad(st, sk, skb, N, ZP, Hd, dprior, ZP, p[DM_RSSM_PRIOR_W], dprin, Hd, 0, nullptr));
  DM_TRY(norm_elu_bwd_dx(N, Hd, a.x3, Hd, a.prin, Hd, a.st3, p[DM_RSSM_PRIOR_G], dprin, Hd, dx3, Hd, st));
  DM_TRY(dgrad(st, sk, skb, N, Hd, D, dx3, Hd, p[DM_RSSM_PRIOR_H_W], dfeat, F, 1, nullptr));
  // ... its parameter gradients on sw
  DM_TRY(dm_wgrad_side_fork(st, sw));
  DM_TRY(wgrad(sw, sk_w, skb, N, ZP, Hd, dprior, ZP, a.prin, Hd, g[DM_RSSM_PRIOR_W]));
  DM_TRY(dm_colsum_launch(N, ZP, dprior, ZP, g[DM_RSSM_PRIOR_OB], sk_w, skb, sw));
  DM_TRY(norm_elu_bwd_params(N, Hd, a.x3, Hd, a.prin, Hd, a.st3, dprin, Hd, g[DM_RSSM_PRIOR_G],
                                     g[DM_RSSM_PRIOR_B], sk_w, skb, sw));
  DM_TRY(wgrad(sw, sk_w, skb, N, Hd, D, dx3, Hd, feat, F, g[DM_RSSM_PRIOR_H_W]));
  DM_TRY(dm_colsum_launch(N, Hd, dx3, Hd, g[DM_RSSM_PRIOR_H_B], sk_w, skb, sw));

  // (The BPTT loop as a second persistent LDS-weight-stationary kernel was built in round 4 - 1.6x faster than these launches
  // when it has the chip to itself, slower INSIDE the multi-stream step at every shard size measured, because it needs every CU
  // at once while the decoder backward wants them too: profiles/r04_ab_bptt.txt - and removed in round 5.)
  // ---- BPTT as launches.  The five backward-data products of a step multiply a B-row block by W (not W^T); transposing the
  // weights once here (22 MB, ~20 us) lets all 5*T of them stream k-contiguous rows.
  DM_TRY(transpose(st, p[DM_RSSM_POST_W], wt_post, ZP, Hd));
  DM_TRY(transpose(st, p[DM_RSSM_POST_H_W], wt_post_h, Hd, D));
  GruStack gk;
  DM_TRY(gru_stack(s, p, g, &gk));
  const bool stacked = gk.L > 1;
  if (!stacked) {
    DM_TRY(transpose(st, p[DM_RSSM_GRU_WIH], wt_ih, 3 * D, Hd));
    DM_TRY(transpose(st, p[DM_RSSM_GRU_WHH], wt_hh, 3 * D, D));
  }
  DM_TRY(transpose(st, p[DM_RSSM_Z_W], wt_z, Hd, Z));
  // Fused schedule (5 launches per step instead of 8), mirror of the forward T loop: both LayerNorm+ELU BACKWARD stages
  // ride in the prologue of the <= 64-row product that consumes their result, and the GRU gates backward rides in the
  // epilogue of the product that completes dh'.  dx1 / dx2 (needed by the batched weight gradients) are then produced for
  // all rows by two batched launches after the loop.
  const bool fuse_b = p[DM_RSSM_IN_G] != nullptr && !stacked && !gauss && kind == 0 && dm_skinny_ln_ok(B, D, Hd) &&
                      dm_skinny_ln_ok(B, Z, Hd) && (F & 3) == 0;
  // ... in FOLDED form (common.h DmGemm::eg_x): the product that makes dpin (dza) also turns it into g = dy ELU'(pre) gamma in
  // its epilogue - once, by the workgroup that owns the element, instead of once per consuming workgroup in a prologue - and
  // the consuming product is a plain one whose epilogue applies the two row-mean terms.  That needs x2 W_post_h and x1 W_z for
  // all rows (two batched products here) and the weights' column sums.
  bool fold = fuse_b && g_bptt_fold && B <= 64 && (int64_t)Hd * ZP >= (int64_t)64 * 1024 &&
              (int64_t)Hd * 3 * D >= (int64_t)64 * 1024 && (ZP & 3) == 0 && ((3 * D) & 3) == 0 && ZP >= 16;
  float *xw2 = nullptr, *xwz = nullptr, *cs2 = nullptr, *csz = nullptr, *eps2 = nullptr, *eps1 = nullptr;
  const int nstrip = (Hd + 15) / 16;
  if (fold) {
    const size_t mark = ar.off;
    xw2 = ar.take((size_t)N * D);
    xwz = ar.take((size_t)N * Z);
    cs2 = ar.take((size_t)D);
    csz = ar.take((size_t)Z);
    eps2 = ar.take((size_t)nstrip * 128);
    eps1 = ar.take((size_t)nstrip * 128);
    if (!ar.ok) { ar.off = mark; ar.ok = true; fold = false; }       // a small caller workspace keeps the prologue form
  }
  static const int no_fold_sm = getenv("DM_BPTT_NO_FOLD_SM") ? 1 : 0;      // A/B switch
  const bool fold_sm = fold && !gauss && C == 32 && Z == ZP && !no_fold_sm;
  if (fold) {
    DM_TRY(dgrad(st, sk, skb, N, Hd, D, a.x2, Hd, p[DM_RSSM_POST_H_W], xw2, D, 0, nullptr));      // x2 W_post_h
    DM_TRY(dgrad(st, sk, skb, N, Hd, Z, a.x1, Hd, p[DM_RSSM_Z_W], xwz, Z, 0, nullptr));           // x1 W_z
    DM_TRY(dm_colsum_launch(Hd, D, p[DM_RSSM_POST_H_W], D, cs2, sk, skb, st));
    DM_TRY(dm_colsum_launch(Hd, Z, p[DM_RSSM_Z_W], Z, csz, sk, skb, st));
  }
  // fragment-major copies (common.h dm_frag_off) of the two K = 3D operands of a step, dgi and dgh: written by the gates
  // backward epilogue, read by the two products that follow it
  static const int no_frag = getenv("DM_SKINNY_NO_FRAG") ? 1 : 0;
  float* dgif = (fuse_b && !no_frag && B <= 64) ? ar.take(dm_frag_floats(3 * D)) : nullptr;
  float* dghf = dgif ? ar.take(dm_frag_floats(3 * D)) : nullptr;
  float* dpinf = dgif ? ar.take(dm_frag_floats(Hd)) : nullptr;      // ... and of dpin, dza (written by the epilogue of the
  float* dzaf = dgif ? ar.take(dm_frag_floats(Hd)) : nullptr;       // product that makes them)
  if (!ar.ok) { dgif = nullptr; dghf = nullptr; dpinf = nullptr; dzaf = nullptr; }
  // time chunks of the batched weight gradients (single-layer cells): chunk c = steps [T*c/nchunk, T*(c+1)/nchunk); the loop
  // runs t downwards, so the LAST chunk completes first - it overwrites the gradient, the others accumulate
  const int nchunk = (stacked || B < 16) ? 1 : (T >= 16 ? 4 : T >= 8 ? 2 : 1);      // (a few-column shard: the chunks' extra launches cost more than they hide)
  int next_chunk = nchunk - 1;
  const float* dx2s = fuse_b ? dx2_w : dx2;
  const float* dx1s = fuse_b ? dx1_w : dx1;
  auto side_chunk = [&](int t) -> int {
    if (stacked || next_chunk < 0 || t != (int)((long long)T * next_chunk / nchunk)) return DM_OK;
    const int c = next_chunk--;
    const int t0 = (int)((long long)T * c / nchunk), t1 = (int)((long long)T * (c + 1) / nchunk);
    const size_t c0 = (size_t)t0 * B;
    const int rows = (t1 - t0) * B, acc = c != nchunk - 1;
    DM_TRY(dm_wgrad_side_fork(st, sw));            // rows [c0, c0 + rows) of dpost, dpin, dgi, dgh, dza (dx2, dx1) are final
    if (fuse_b) {
      DM_TRY(norm_elu_bwd_dx(rows, Hd, a.x2 + c0 * Hd, Hd, a.pin + c0 * Hd, Hd, a.st2 + c0 * 2, p[DM_RSSM_POST_G], dpin + c0 * Hd, Hd,
                             dx2_w + c0 * Hd, Hd, sw));
      DM_TRY(norm_elu_bwd_dx(rows, Hd, a.x1 + c0 * Hd, Hd, a.za + c0 * Hd, Hd, a.st1 + c0 * 2, p[DM_RSSM_IN_G], dza + c0 * Hd, Hd,
                             dx1_w + c0 * Hd, Hd, sw));
    }
    DM_TRY(wgrad(sw, sk_w, skb, rows, ZP, Hd, dpost + c0 * ZP, ZP, a.pin + c0 * Hd, Hd, g[DM_RSSM_POST_W], acc));
    DM_TRY(wgrad(sw, sk_w, skb, rows, Hd, D, dx2s + c0 * Hd, Hd, feat + c0 * F, F, g[DM_RSSM_POST_H_W], acc));
    DM_TRY(wgrad(sw, sk_w, skb, rows, Hd, E, dx2s + c0 * Hd, Hd, embed + c0 * E, E, g[DM_RSSM_POST_E_W], acc));
    DM_TRY(wgrad(sw, sk_w, skb, rows, 3 * D, Hd, dgi + c0 * 3 * D, 3 * D, a.za + c0 * Hd, Hd, g[DM_RSSM_GRU_WIH], acc));
    DM_TRY(wgrad(sw, sk_w, skb, rows, 3 * D, D, dgh + c0 * 3 * D, 3 * D, a.hin + c0 * D, D, g[DM_RSSM_GRU_WHH], acc));
    DM_TRY(wgrad(sw, sk_w, skb, rows, Hd, Z, dx1s + c0 * Hd, Hd, a.zin + c0 * Z, Z, g[DM_RSSM_Z_W], acc));
    DM_TRY(wgrad(sw, sk_w, skb, rows, Hd, A, dx1s + c0 * Hd, Hd, action + c0 * A, A, g[DM_RSSM_A_W], acc));
    return DM_OK;
  };
  for (int t = T - 1; t >= 0; --t) {
    const size_t r0 = (size_t)t * B;
    float* dft = dfeat + r0 * F;             // [dh' | dz'] of step t, complete at this point
    float* dpt = dpost + r0 * ZP;
    // straight-through sample: dpost += softmax'(post)^T dz'   (Gaussian: the reparameterised sample's (dmean, draw std))
    // (folded schedule, 32 classes: done in the epilogue of the product that completed dz' - the pair launch of step t+1)
    if (gauss) DM_TRY(dm_gauss_sample_bwd_launch(B, S, post + r0 * ZP, ZP, feat + r0 * F + D, F, dft + D, F, dpt, ZP, 1, st));
    else if (!(fold_sm && t < T - 1)) DM_TRY(dm_st_softmax_bwd_launch(B, S, C, post + r0 * ZP, ZP, dft + D, F, dpt, ZP, 1, st));
    // post_mlp, post_norm+ELU, post_mlp_h
    if (fuse_b) {
      DmGemm q3;   // dpin = dpost Wpost
      q3.M = B; q3.N = Hd; q3.K = ZP; q3.A = dpt; q3.lda = ZP; q3.B = wt_post; q3.ldb = ZP; q3.C = dpin + r0 * Hd; q3.ldc = Hd;
      q3.C_frag = dpinf;
      if (fold) {      // + g2 = dpin ELU'(pre2) gamma (row-major into the dx2 rows - re-made in batch after the loop - and fragment-major)
        q3.C_frag = nullptr;
        q3.eg_x = a.x2 + r0 * Hd; q3.eg_ldx = Hd; q3.eg_stats = a.st2 + r0 * 2; q3.eg_gamma = p[DM_RSSM_POST_G]; q3.eg_beta = p[DM_RSSM_POST_B];
        q3.eg_G = dx2 + r0 * Hd; q3.eg_ldg = Hd; q3.eg_Gf = dpinf; q3.eg_ps = eps2;
      }
      DM_TRY(dm_gemm_launch(q3, sk, skb, st));
    } else {
      DM_TRY(dgrad_t(st, sk, skb, B, ZP, Hd, dpt, ZP, wt_post, dpin + r0 * Hd, Hd, 0, nullptr));
    }
    if (fuse_b) {
      const uint8_t* rz = reset + r0;
      float* dprev = t > 0 ? dfeat + (r0 - B) * F : nullptr;
      DmGatesBwd gb;
      gb.gi = a.gi + r0 * 3 * D; gb.gh = a.gh + r0 * 3 * D; gb.h_in = a.hin + r0 * D; gb.ldh = D; gb.D = D;
      gb.dgi = dgi + r0 * 3 * D; gb.dgh = dgh + r0 * 3 * D; gb.dprev = dprev; gb.ldp = F; gb.row_zero = rz;
      gb.dgi_frag = dgif; gb.dgh_frag = dghf;
      DmGemm q4;     // dh' += LNbwd(dpin) Wph ; then the GRU gates backward on the completed dh'
      q4.M = B; q4.N = D; q4.K = Hd; q4.A = dpin + r0 * Hd; q4.lda = Hd; q4.B = wt_post_h; q4.ldb = Hd;
      q4.C = dft; q4.ldc = F; q4.flags = DM_GEMM_ACCUM;
      q4.ln_g = p[DM_RSSM_POST_G]; q4.ln_b = p[DM_RSSM_POST_B]; q4.lnb_x = a.x2 + r0 * Hd; q4.lnb_ldx = Hd;
      q4.lnb_stats = a.st2 + r0 * 2; q4.gates = &gb; q4.A_frag = dpinf;
      if (fold) {      // a plain product on g2; the row-mean terms enter in the epilogue, in front of the gates backward
        q4.A = dx2 + r0 * Hd; q4.ln_g = nullptr; q4.ln_b = nullptr; q4.lnb_x = nullptr; q4.lnb_stats = nullptr;
        q4.lnf_ps = eps2; q4.lnf_nps = nstrip; q4.lnf_stats = a.st2 + r0 * 2; q4.lnf_xw = xw2 + r0 * D; q4.lnf_ldxw = D; q4.lnf_cs = cs2;
      }
      DM_TRY(dm_gemm_launch(q4, sk, skb, st));
      {
        DmGemm q5;   // dza = dgi Wih
        q5.M = B; q5.N = Hd; q5.K = 3 * D; q5.A = dgi + r0 * 3 * D; q5.lda = 3 * D; q5.B = wt_ih; q5.ldb = 3 * D;
        q5.C = dza + r0 * Hd; q5.ldc = Hd; q5.A_frag = dgif; q5.C_frag = dzaf;
        if (fold) {
          q5.C_frag = nullptr;
          q5.eg_x = a.x1 + r0 * Hd; q5.eg_ldx = Hd; q5.eg_stats = a.st1 + r0 * 2; q5.eg_gamma = p[DM_RSSM_IN_G]; q5.eg_beta = p[DM_RSSM_IN_B];
          q5.eg_G = dx1 + r0 * Hd; q5.eg_ldg = Hd; q5.eg_Gf = dzaf; q5.eg_ps = eps1;
        }
        DM_TRY(dm_gemm_launch(q5, sk, skb, st));
      }
      if (t > 0) {   // both products into step t-1's [dh' | dz']; the second one consumes LNbwd(dza)
        DmGemm qh, qz;
        qh.M = B; qh.N = D; qh.K = 3 * D; qh.A = dgh + r0 * 3 * D; qh.lda = 3 * D; qh.B = wt_hh; qh.ldb = 3 * D;
        qh.A_frag = dghf;
        qh.C = dprev; qh.ldc = F; qh.flags = DM_GEMM_ACCUM; qh.row_zero = rz;
        qz.M = B; qz.N = Z; qz.K = Hd; qz.A = dza + r0 * Hd; qz.lda = Hd; qz.B = wt_z; qz.ldb = Hd;
        qz.C = dprev + D; qz.ldc = F; qz.flags = DM_GEMM_ACCUM; qz.row_zero = rz;
        qz.ln_g = p[DM_RSSM_IN_G]; qz.ln_b = p[DM_RSSM_IN_B]; qz.lnb_x = a.x1 + r0 * Hd; qz.lnb_ldx = Hd;
        qz.lnb_stats = a.st1 + r0 * 2; qz.A_frag = dzaf;
        if (fold) {
          qz.A = dx1 + r0 * Hd; qz.ln_g = nullptr; qz.ln_b = nullptr; qz.lnb_x = nullptr; qz.lnb_stats = nullptr;
          if (fold_sm) { qz.sm_logits = post + (r0 - B) * ZP; qz.sm_ld = ZP; qz.sm_dlogits = dpost + (r0 - B) * ZP; qz.sm_ldd = ZP; }
          qz.lnf_ps = eps1; qz.lnf_nps = nstrip; qz.lnf_stats = a.st1 + r0 * 2; qz.lnf_xw = xwz + r0 * Z; qz.lnf_ldxw = Z; qz.lnf_cs = csz;
        }
        DM_TRY(dm_gemm_pair_launch(qh, qz, sk, skb, st));
      }
      DM_TRY(side_chunk(t));
      continue;
    }
    DM_TRY(norm_elu_bwd_dx(B, Hd, a.x2 + r0 * Hd, Hd, a.pin + r0 * Hd, Hd, a.st2 + r0 * 2, p[DM_RSSM_POST_G],
                                   dpin + r0 * Hd, Hd, dx2 + r0 * Hd, Hd, st));
    DM_TRY(dgrad_t(st, sk, skb, B, Hd, D, dx2 + r0 * Hd, Hd, wt_post_h, dft, F, 1, nullptr));
    // GRU gates; the direct path dh'*u goes (masked) straight into step t-1's dh'
    const uint8_t* rz = reset + r0;
    float* dprev = t > 0 ? dfeat + (r0 - B) * F : nullptr;
    if (stacked) {
      // layers in reverse: layer i's input gradient lands in the new-state gradient of layer i-1 before that layer's own
      // gates run; its recurrent-input gradient goes (masked) into step t-1's dh' slice
      const int ls = gk.ls;
      for (int i = gk.L - 1; i >= 0; --i) {
        float* dgi_i = dgi + r0 * 3 * D + 3 * ls * i;
        float* dgh_i = dgh + r0 * 3 * D + 3 * ls * i;
        if (kind == 0)
          DM_TRY(dm_gru_gates_bwd_launch(B, ls, a.gi + r0 * 3 * D + 3 * ls * i, a.gh + r0 * 3 * D + 3 * ls * i,
                                         a.hin + r0 * D + i * ls, D, dft + i * ls, F, dgi_i, dgh_i,
                                         dprev ? dprev + i * ls : nullptr, F, 1, rz, st, 3 * D));
        else
          DM_TRY(dm_gru_norm_bwd_launch(kind, B, ls, a.gh + r0 * 3 * D + 3 * ls * i, a.hin + r0 * D + i * ls, D,
                                        a.gs + r0 * 3 * D + 3 * ls * i, a.gst + r0 * 6 * gk.L + 6 * i, gk.lng[i], gk.lnb[i],
                                        dft + i * ls, F, dgi_i, dgh_i, dgl + r0 * 3 * D + 3 * ls * i,
                                        dprev ? dprev + i * ls : nullptr, F, rz, st, 3 * D, 6 * gk.L));
        if (i > 0) DM_TRY(dgrad(st, sk, skb, B, 3 * ls, ls, dgi_i, 3 * D, gk.wih[i], dft + (i - 1) * ls, F, 1, nullptr));
        else DM_TRY(dgrad(st, sk, skb, B, 3 * ls, Hd, dgi_i, 3 * D, gk.wih[0], dza + r0 * Hd, Hd, 0, nullptr));
        if (dprev) DM_TRY(dgrad(st, sk, skb, B, 3 * ls, ls, dgh_i, 3 * D, gk.whh[i], dprev + i * ls, F, 1, rz));
      }
      DM_TRY(norm_elu_bwd_dx(B, Hd, a.x1 + r0 * Hd, Hd, a.za + r0 * Hd, Hd, a.st1 + r0 * 2, p[DM_RSSM_IN_G],
                                     dza + r0 * Hd, Hd, dx1 + r0 * Hd, Hd, st));
      if (t > 0) DM_TRY(dgrad_t(st, sk, skb, B, Hd, Z, dx1 + r0 * Hd, Hd, wt_z, dprev + D, F, 1, rz));
      continue;
    }
    if (kind == 0)
      DM_TRY(dm_gru_gates_bwd_launch(B, D, a.gi + r0 * 3 * D, a.gh + r0 * 3 * D, a.hin + r0 * D, D, dft, F,
                                     dgi + r0 * 3 * D, dgh + r0 * 3 * D, dprev, F, 1, rz, st));
    else
      DM_TRY(dm_gru_norm_bwd_launch(kind, B, D, a.gh + r0 * 3 * D, a.hin + r0 * D, D, a.gs + r0 * 3 * D, a.gst + r0 * 6, lng,
                                    lnb, dft, F, dgi + r0 * 3 * D, dgh + r0 * 3 * D, dgl + r0 * 3 * D, dprev, F, rz, st));
    DM_TRY(dgrad_t(st, sk, skb, B, 3 * D, Hd, dgi + r0 * 3 * D, 3 * D, wt_ih, dza + r0 * Hd, Hd, 0, nullptr));
    DM_TRY(norm_elu_bwd_dx(B, Hd, a.x1 + r0 * Hd, Hd, a.za + r0 * Hd, Hd, a.st1 + r0 * 2, p[DM_RSSM_IN_G],
                                   dza + r0 * Hd, Hd, dx1 + r0 * Hd, Hd, st));
    if (t > 0) {     // both products into step t-1's [dh' | dz'], one launch
      DmGemm qh, qz;
      qh.M = B; qh.N = D; qh.K = 3 * D; qh.A = dgh + r0 * 3 * D; qh.lda = 3 * D; qh.B = wt_hh; qh.ldb = 3 * D;
      qh.C = dprev; qh.ldc = F; qh.flags = DM_GEMM_ACCUM; qh.row_zero = rz;
      qz.M = B; qz.N = Z; qz.K = Hd; qz.A = dx1 + r0 * Hd; qz.lda = Hd; qz.B = wt_z; qz.ldb = Hd;
      qz.C = dprev + D; qz.ldc = F; qz.flags = DM_GEMM_ACCUM; qz.row_zero = rz;
      DM_TRY(dm_gemm_pair_launch(qh, qz, sk, skb, st));
    }
    DM_TRY(side_chunk(t));
  }

  // ---- the one data gradient left: dembed, for the encoder backward that follows on st
  if (fuse_b) DM_TRY(norm_elu_bwd_dx(N, Hd, a.x2, Hd, a.pin, Hd, a.st2, p[DM_RSSM_POST_G], dpin, Hd, dx2, Hd, st));
  if (dembed) DM_TRY(dgrad(st, sk, skb, N, Hd, E, dx2, Hd, p[DM_RSSM_POST_E_W], dembed, E, 0, nullptr));
  // ---- bias / LayerNorm gradients (column passes over all rows) and, for cell stacks, the weight gradients: on sw
  st = sw;
  sk = sk_w;
  if (stacked) DM_TRY(dm_wgrad_side_fork(st_main, sw));      // (single-layer cells forked at their last chunk)
  DM_TRY(dm_colsum_launch(N, ZP, dpost, ZP, g[DM_RSSM_POST_OB], sk, skb, st));
  DM_TRY(norm_elu_bwd_params(N, Hd, a.x2, Hd, a.pin, Hd, a.st2, dpin, Hd, g[DM_RSSM_POST_G], g[DM_RSSM_POST_B],
                                     sk, skb, st));
  DM_TRY(dm_colsum_launch(N, Hd, dx2s, Hd, g[DM_RSSM_POST_H_B], sk, skb, st));
  if (stacked) {
    DM_TRY(wgrad(st, sk, skb, N, ZP, Hd, dpost, ZP, a.pin, Hd, g[DM_RSSM_POST_W]));
    DM_TRY(wgrad(st, sk, skb, N, Hd, D, dx2, Hd, feat, F, g[DM_RSSM_POST_H_W]));
    DM_TRY(wgrad(st, sk, skb, N, Hd, E, dx2, Hd, embed, E, g[DM_RSSM_POST_E_W]));
  }
  if (stacked) {
    const int ls = gk.ls;
    for (int i = 0; i < gk.L; ++i) {
      const float* x = i == 0 ? a.za : feat + (size_t)(i - 1) * ls;
      const int ldx = i == 0 ? Hd : F, kin = i == 0 ? Hd : ls;
      DM_TRY(wgrad(st, sk, skb, N, 3 * ls, kin, dgi + 3 * ls * i, 3 * D, x, ldx, gk.g_wih[i]));
      DM_TRY(wgrad(st, sk, skb, N, 3 * ls, ls, dgh + 3 * ls * i, 3 * D, a.hin + i * ls, D, gk.g_whh[i]));
      if (kind == 0) {
        DM_TRY(dm_colsum_launch(N, 3 * ls, dgi + 3 * ls * i, 3 * D, gk.g_bih[i], sk, skb, st));
        DM_TRY(dm_colsum_launch(N, 3 * ls, dgh + 3 * ls * i, 3 * D, gk.g_bhh[i], sk, skb, st));
      } else {      // the layer's LayerNorm parameters: one column pass over its 3*ls gate columns of all rows
        DM_TRY(dm_gru_norm_param_grads_launch(kind, N, ls, a.gs + 3 * ls * i, a.gst + 6 * i, dgl + 3 * ls * i, lnpg, lnpb, st,
                                              3 * D, 6 * gk.L));
        const int parts = kind == 1 ? 3 : 1;
        const size_t len = (size_t)(kind == 1 ? ls : 3 * ls) * sizeof(float);
        for (int q = 0; q < parts; ++q)
          if (hipMemcpyAsync(gk.g_lng[i][q], lnpg + (size_t)q * ls, len, hipMemcpyDeviceToDevice, st) != hipSuccess ||
              hipMemcpyAsync(gk.g_lnb[i][q], lnpb + (size_t)q * ls, len, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return dm_fail(DM_E_HIP, "rssm_sequence_bwd: gradient copy failed");
      }
    }
  }
  if (stacked) {
  } else if (kind == 0) {
    DM_TRY(dm_colsum_launch(N, 3 * D, dgi, 3 * D, g[DM_RSSM_GRU_BIH], sk, skb, st));
    DM_TRY(dm_colsum_launch(N, 3 * D, dgh, 3 * D, g[DM_RSSM_GRU_BHH], sk, skb, st));
  } else {      // LayerNorm parameters of the cell: one batched column pass over all rows, then split into the thirds
    DM_TRY(dm_gru_norm_param_grads_launch(kind, N, D, a.gs, a.gst, dgl, lnpg, lnpb, st));
    const int parts = kind == 1 ? 3 : 1;
    const size_t len = (kind == 1 ? (size_t)D : (size_t)3 * D) * sizeof(float);
    float* const gdst[3] = {g[DM_RSSM_GRU_LN_G0], g[DM_RSSM_GRU_LN_G1], g[DM_RSSM_GRU_LN_G2]};
    float* const bdst[3] = {g[DM_RSSM_GRU_LN_B0], g[DM_RSSM_GRU_LN_B1], g[DM_RSSM_GRU_LN_B2]};
    for (int q = 0; q < parts; ++q) {
      DM_REQUIRE(gdst[q] && bdst[q], DM_E_NULL, "rssm_sequence_bwd: missing LayerNorm-GRU gradient slot %d", q);
      if (hipMemcpyAsync(gdst[q], lnpg + (size_t)q * D, len, hipMemcpyDeviceToDevice, st) != hipSuccess ||
          hipMemcpyAsync(bdst[q], lnpb + (size_t)q * D, len, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return dm_fail(DM_E_HIP, "rssm_sequence_bwd: gradient copy failed");
    }
  }
  DM_TRY(norm_elu_bwd_params(N, Hd, a.x1, Hd, a.za, Hd, a.st1, dza, Hd, g[DM_RSSM_IN_G], g[DM_RSSM_IN_B], sk, skb,
                                     st));
  DM_TRY(dm_colsum_launch(N, Hd, dx1s, Hd, g[DM_RSSM_Z_B], sk, skb, st));
  if (stacked) {
    DM_TRY(wgrad(st, sk, skb, N, Hd, Z, dx1, Hd, a.zin, Z, g[DM_RSSM_Z_W]));
    DM_TRY(wgrad(st, sk, skb, N, Hd, A, dx1, Hd, action, A, g[DM_RSSM_A_W]));
  }
  DM_TRY(dm_wgrad_side_mark(sw, st_main));
  return DM_OK;
}

// ---------------------------------------------------------------- imagination -------------------
// Progress marks of the NEXT dm_dream_rollout call of this thread (include/dreamer_hip.h): events[i] is recorded on the
// rollout's stream once horizon step steps[i] has been enqueued, i.e. when feature rows [0, (steps[i] + 2) * M) are final.
static thread_local int tl_marks_n = 0;
static thread_local int tl_mark_step[4];
static thread_local hipEvent_t tl_mark_ev[4];
extern "C" int dm_dream_rollout_marks(int n, const int* steps, void* const* events) {
  DM_REQUIRE(n >= 0 && n <= 4 && (n == 0 || (steps && events)), DM_E_SHAPE, "dream_rollout_marks: n=%d (0..4)", n);
  for (int i = 0; i < n; ++i) {
    DM_REQUIRE(events[i], DM_E_NULL, "dream_rollout_marks: null event %d", i);
    tl_mark_step[i] = steps[i];
    tl_mark_ev[i] = (hipEvent_t)events[i];
  }
  tl_marks_n = n;
  return DM_OK;
}
// every mark not recorded inside the loop (step out of range, or the chain ran as a captured / replayed graph) is recorded
// behind the whole call, so a waiter is never left with a stale event
struct DmRolloutMarks {
  int n;
  bool done[4] = {false, false, false, false};
  hipStream_t caller;
  explicit DmRolloutMarks(hipStream_t st) : n(tl_marks_n), caller(st) { tl_marks_n = 0; }
  void at_step(int i, hipStream_t st, bool eager) {
    if (!eager) return;
    for (int k = 0; k < n; ++k)
      if (!done[k] && tl_mark_step[k] == i) { (void)hipEventRecord(tl_mark_ev[k], st); done[k] = true; }
  }
  ~DmRolloutMarks() {
    for (int k = 0; k < n; ++k)
      if (!done[k]) (void)hipEventRecord(tl_mark_ev[k], caller);
  }
};

static int g_rollout_fuse_act = getenv("DM_ROLLOUT_NO_FUSE_ACT") ? 0 : 1;
// 1 / 0: the rollout's one-hot action draw in the output stage of the whole-MLP actor kernel / as its own launch; -1 queries.
extern "C" int dm_rollout_fuse_act_enable(int on) {
  if (on >= 0) g_rollout_fuse_act = on ? 1 : 0;
  return g_rollout_fuse_act;
}

extern "C" int dm_dream_rollout(const dm_shape* s, int M, const float* start, const dm_rssm_params* P,
                                const dm_mlp_params* actor, const float* u_act, const float* u_prior, float* feats,
                                float* actions, int32_t* act_idx, float* actor_acts, float* actor_logits, void* ws,
                                size_t ws_bytes, void* stream) {
  DM_REQUIRE(s && start && P && actor && u_act && u_prior && feats && actions && ws, DM_E_NULL,
             "dream_rollout: null pointer");
  DmPrecisionScope prec(s->flags & DM_FLAG_BF16);
  DM_TRY(rssm_check(s));
  DM_REQUIRE(M >= 1 && s->H >= 1, DM_E_SHAPE, "dream_rollout: M=%d H=%d", M, s->H);
  hipStream_t st = (hipStream_t)stream;
  const int H = s->H, D = s->D, Hd = s->Hd, S = s->S, C = s->C, Z = S * (C ? C : 1), ZP = S * (C ? C : 2);   // see fwd_steps
  const int F = D + Z, A = s->A;
  const int Hm = s->mlp_hidden, L = s->mlp_layers;
  const int adist = s->flags & 3;                 // 0 onehot, 1 tanh_normal, 2 normal_tanh
  DM_REQUIRE(adist <= 2, DM_E_SHAPE, "dream_rollout: unknown actor distribution %d", adist);
  const int AO = adist == 0 ? A : 2 * A;         // actor output width (a2c.py:35)
  const float* const* p = P->p;

  DmArena ar(ws, ws_bytes);
  float* sk = ar.take(DM_SPLITK_FLOATS);
  DM_REQUIRE((actor_acts == nullptr) == (actor_logits == nullptr), DM_E_NULL,
             "dream_rollout: actor_acts and actor_logits must be given together");
  float* macts = ar.take(actor_acts ? 0 : dm_mlp_acts_floats(M, Hm, L));
  float* logits_ws = ar.take(actor_acts ? 0 : (size_t)M * AO);
  float* ea = ar.take((size_t)M * Hd);
  float* x1 = ar.take((size_t)M * Hd);
  float* za = ar.take((size_t)M * Hd);
  float* stats = ar.take((size_t)M * 2);
  float* gi = ar.take((size_t)M * 3 * D);
  float* gh = ar.take((size_t)M * 3 * D);
  float* prior = ar.take((size_t)M * ZP);
  const int kind = rssm_gru_kind(s);
  float* gsw = ar.take(kind ? (size_t)M * 3 * D : 0);
  float* gstw = ar.take(kind ? (size_t)M * 6 * rssm_gru_layers(s) : 0);
  const float* lng[3] = {p[DM_RSSM_GRU_LN_G0], p[DM_RSSM_GRU_LN_G1], p[DM_RSSM_GRU_LN_G2]};
  const float* lnb[3] = {p[DM_RSSM_GRU_LN_B0], p[DM_RSSM_GRU_LN_B1], p[DM_RSSM_GRU_LN_B2]};
  // bf16 mode, plain single-layer GRU with LayerNorm: the cell's four 2 500-row products read bf16 twins (common.h DmTwinScope) -
  // per-call copies of their weights, za (written by the z_embed / LayerNorm kernels that produce it) and the h columns of
  // `feats` (written by the gates kernel; one range per step, so step 0's h, copied from `start`, stays on the fp32 path)
  DmTwinScope tw((s->flags & DM_FLAG_BF16) != 0);
  const bool tw_on = dm_twins_on() && kind == 0 && rssm_gru_layers(s) == 1 && p[DM_RSSM_IN_G] && p[DM_RSSM_PRIOR_G] && (F & 7) == 0;
  unsigned short* wih_h = (unsigned short*)ar.take(tw_on ? dm_half_floats((size_t)3 * D * Hd) : 0);
  unsigned short* whh_h = (unsigned short*)ar.take(tw_on ? dm_half_floats((size_t)3 * D * D) : 0);
  unsigned short* wph_h = (unsigned short*)ar.take(tw_on ? dm_half_floats((size_t)Hd * D) : 0);
  unsigned short* wp_h = (unsigned short*)ar.take(tw_on ? dm_half_floats((size_t)ZP * Hd) : 0);
  unsigned short* za_h = (unsigned short*)ar.take(tw_on ? dm_half_floats((size_t)M * Hd) : 0);
  unsigned short* feats_h = (unsigned short*)ar.take(tw_on ? dm_half_floats((size_t)(H + 1) * M * F) : 0);
  DM_REQUIRE(ar.ok, DM_E_WORKSPACE, "dream_rollout: workspace too small (need %zu floats)", ar.off);
  const size_t skb = DM_SPLITK_FLOATS * sizeof(float);
  DmRolloutMarks marks(st);
  const bool marks_eager = st == (hipStream_t)stream;
  // the actor's weights, fragment-major for the whole-MLP kernel: packed once for all H steps
  GruStack gk;
  DM_TRY(gru_stack(s, p, nullptr, &gk));
  if (tw_on) {
    const DmCvtSeg sg[4] = {{p[DM_RSSM_GRU_WIH], wih_h, (size_t)3 * D * Hd}, {p[DM_RSSM_GRU_WHH], whh_h, (size_t)3 * D * D},
                            {p[DM_RSSM_PRIOR_H_W], wph_h, (size_t)Hd * D}, {p[DM_RSSM_PRIOR_W], wp_h, (size_t)ZP * Hd}};
    DM_TRY(dm_to_bf16_multi_launch(sg, 4, st));
    for (int i = 0; i < 4; ++i) dm_twin_add(sg[i].src, sg[i].n, sg[i].dst, true);
    dm_twin_add(za, (size_t)M * Hd, za_h, false);
    for (int i = 1; i <= H && i < 40; ++i)
      dm_twin_add(feats + (size_t)i * M * F, (size_t)M * F, feats_h + (size_t)i * M * F, false);
  }
  // steps 1.. of the rollout read the z the prior sampler of the step before drew: z_mlp + in_norm + ELU become one
  // gather-sum launch over z_mlp^T (dm_z_embed_launch) instead of a (M x Hd x Z) product and a LayerNorm launch
  static const int no_embed = getenv("DM_RSSM_NO_Z_EMBED") ? 1 : 0;        // A/B switch
  float *wzt = nullptr, *wat = nullptr;
  int32_t *pidx = nullptr, *aidx = nullptr;
  if (!no_embed && C != 0 && H > 1 && dm_z_embed_ok(Hd)) {
    const size_t mark = ar.off;
    float* w = ar.take((size_t)Z * Hd);
    float* w2 = ar.take((size_t)A * Hd);
    int32_t* ix = reinterpret_cast<int32_t*>(ar.take((size_t)M * S));
    int32_t* ax = reinterpret_cast<int32_t*>(ar.take((size_t)M));
    if (ar.ok) {
      wzt = w; pidx = ix;
      DM_TRY(transpose(st, p[DM_RSSM_Z_W], wzt, Hd, Z));
      if (adist == 0) {       // one-hot actions: a_mlp(action) is a row of a_mlp^T too
        wat = w2; aidx = ax;
        DM_TRY(transpose(st, p[DM_RSSM_A_W], wat, Hd, A));
      }
    } else {      // optional buffers: the product path below needs none of them
      ar.off = mark; ar.ok = true;
    }
  }
  const float* actor_wpack = nullptr;
  // the actor's first layer sees [h | one-hot z]: its z columns are a gathered sum of W0^T rows (indices from the prior
  // sampler; step 0's z is a dense vector: its non-zeros are found by ballot), the MFMA product runs over h only
  float *actor_w0t = nullptr, *actor_add0 = nullptr;
  if (dm_mlp_chain_ok(M, F, Hm, L, AO, feats, F, actor) && !dm_panel_ok(M, Hm)) {
    float* wpk = ar.take(dm_mlp_chain_pack_floats(F, L));
    if (ar.ok) {
      int k0 = 0;
      if (C != 0 && pidx && dm_mlp_chain_sparse_ok(F, Z) && dm_z_embed_ok(Hm)) {
        const size_t mark = ar.off;
        float* wt = ar.take((size_t)F * Hm);
        float* ad = ar.take((size_t)M * Hm);
        if (ar.ok) {
          actor_w0t = wt; actor_add0 = ad; k0 = D;
          DM_TRY(transpose(st, actor->w[0], actor_w0t, Hm, F));
        } else {
          ar.off = mark; ar.ok = true;
        }
      }
      DM_TRY(dm_mlp_chain_pack_launch(F, L, actor, wpk, st, k0));
      actor_wpack = wpk;
    }
  }

  hipError_t e = hipMemcpyAsync(feats, start, (size_t)M * F * sizeof(float), hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return dm_fail(DM_E_HIP, "dream_rollout: %s", hipGetErrorString(e));
  for (int i = 0; i < H; ++i) {
    const float* cur = feats + (size_t)i * M * F;
    float* nxt = feats + (size_t)(i + 1) * M * F;
    float* act = actions + (size_t)i * M * A;
    // action ~ OneHotCategorical(actor(feature))                                        dreamer.py:195-200
    // with actor_acts the activations of all H steps are kept (rows i*M..) so ActorCritic's policy-gradient backward
    // reuses them instead of recomputing forward_actor(features[:-1]) (the reference's own TODO, a2c.py:119)
    float* logits = actor_acts ? actor_logits + (size_t)i * M * AO : logits_ws;
    if (actor_add0) {
      if (i == 0) DM_TRY(dm_sparse_rows_launch(M, Hm, Z, cur + D, F, actor_w0t + (size_t)D * Hm, actor_add0, Hm, st));
      else DM_TRY(dm_z_embed_launch(M, Hm, S, C, pidx, nullptr, actor_w0t + (size_t)D * Hm, nullptr, nullptr, 0, nullptr, nullptr,
                                    actor_add0, Hm, nullptr, nullptr, nullptr, 0.f, nullptr, 0, st));
    }
    const int asp = actor_add0 ? Z : 0;
    int32_t* ai = act_idx ? act_idx + (size_t)i * M : aidx;       // the sampled action's index (scratch if the caller wants none)
    // one-hot actors on the whole-MLP kernel: the action draw rides in that kernel's output stage (round 6; DM_ROLLOUT_NO_FUSE_ACT=1
    // keeps the stand-alone sampler launch - same rule, same operation order, bit-identical draws)
    const bool fuse_act = adist == 0 && actor_wpack && g_rollout_fuse_act;
    const DmChainSample samp = {u_act + (size_t)i * M, act, A, ai};
    if (actor_acts)
      DM_TRY(dm_mlp_fwd_launch(M, F, Hm, L, AO, cur, F, actor, actor_acts, H * M, i * M, logits, AO, sk, skb, st, actor_wpack, asp,
                               actor_add0, fuse_act ? &samp : nullptr));
    else DM_TRY(dm_mlp_fwd_launch(M, F, Hm, L, AO, cur, F, actor, macts, M, 0, logits, AO, sk, skb, st, actor_wpack, asp, actor_add0,
                                  fuse_act ? &samp : nullptr));
    if (fuse_act) {
    } else if (adist == 0)
      DM_TRY(dm_sample_onehot_launch(M, 1, A, logits, A, u_act + (size_t)i * M, nullptr, act, A, ai, nullptr, nullptr, st));
    else
      DM_TRY(dm_sample_continuous_launch(adist, M, A, logits, u_act + (size_t)i * M * A, act, st));
    // cell.forward_prior(action, None, (h, z))                                          rssm.py:155-184
    const bool embed = wzt && i > 0, embed_a = embed && wat && ai;
    if (!embed_a) DM_TRY(linear(st, sk, skb, M, Hd, A, act, A, p[DM_RSSM_A_W], nullptr, nullptr, 0, ea, Hd));
    if (embed) {
      DM_TRY(dm_z_embed_launch(M, Hd, S, C, pidx, nullptr, wzt, p[DM_RSSM_Z_B], embed_a ? nullptr : ea, Hd,
                               embed_a ? ai : nullptr, wat, nullptr, Hd, nullptr, p[DM_RSSM_IN_G], p[DM_RSSM_IN_B], 1e-3f,
                               za, Hd, st));
    } else {
      DM_TRY(linear(st, sk, skb, M, Hd, Z, cur + D, F, p[DM_RSSM_Z_W], p[DM_RSSM_Z_B], ea, Hd, x1, Hd));
      DM_TRY(norm_elu_fwd(M, Hd, x1, Hd, p[DM_RSSM_IN_G], p[DM_RSSM_IN_B], 1e-3f, za, Hd, stats, st));
    }
    if (gk.L > 1) {
      DM_TRY(gru_stack_fwd(st, sk, skb, gk, M, Hd, D, za, cur, F, gi, gh, nxt, F, nullptr, nullptr, gsw, gstw));
    } else {
      DM_TRY(linear(st, sk, skb, M, 3 * D, Hd, za, Hd, p[DM_RSSM_GRU_WIH], p[DM_RSSM_GRU_BIH], nullptr, 0, gi, 3 * D));
      DM_TRY(linear(st, sk, skb, M, 3 * D, D, cur, F, p[DM_RSSM_GRU_WHH], p[DM_RSSM_GRU_BHH], nullptr, 0, gh, 3 * D));
    }
    if (gk.L > 1) {
    } else if (kind == 0) DM_TRY(dm_gru_gates_fwd_launch(M, D, gi, gh, cur, F, nxt, F, nullptr, nullptr, nullptr, nullptr, st));
    else DM_TRY(dm_gru_norm_fwd_launch(kind, M, D, gi, gh, cur, F, lng, lnb, nxt, F, gsw, gstw, nullptr, nullptr, st));
    DM_TRY(linear(st, sk, skb, M, Hd, D, nxt, F, p[DM_RSSM_PRIOR_H_W], p[DM_RSSM_PRIOR_H_B], nullptr, 0, x1, Hd));
    DM_TRY(norm_elu_fwd(M, Hd, x1, Hd, p[DM_RSSM_PRIOR_G], p[DM_RSSM_PRIOR_B], 1e-3f, za, Hd, stats, st));
    DM_TRY(linear(st, sk, skb, M, ZP, Hd, za, Hd, p[DM_RSSM_PRIOR_W], p[DM_RSSM_PRIOR_OB], nullptr, 0, prior, ZP));
    DM_TRY(dm_sample_onehot_launch(M, S, C, prior, ZP, u_prior + (size_t)i * M * S, nullptr, nxt + D, F, pidx, nullptr,
                                   nullptr, st));
    marks.at_step(i, st, marks_eager);
  }
  return DM_OK;
}
