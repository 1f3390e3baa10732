// pydreamer MLP (common.py:37-65): [Linear -> LayerNorm(eps 1e-3) -> ELU] x L, Linear.
// Used by the reward / terminal decoders (decoders.py:257-319) and actor / critic / critic_target (a2c.py:37-39).
// Contractions run on gemm.hip (MFMA); LayerNorm+ELU and the bias / gamma / beta column sums are row kernels.
#include "common.h"
#include <stdlib.h>

struct MlpActs {
  float* xpre[DM_MAX_MLP_LAYERS];
  float* stats[DM_MAX_MLP_LAYERS];
  float* y[DM_MAX_MLP_LAYERS];
};
static size_t mlp_carve(int rows, int hidden, int layers, float* base, MlpActs* a) {
  DmArena ar(base, (size_t)1 << 62);
  for (int l = 0; l < layers; ++l) {
    float* xp = ar.take((size_t)rows * hidden);
    float* st = ar.take((size_t)rows * 2);
    float* yy = ar.take((size_t)rows * hidden);
    if (a) { a->xpre[l] = xp; a->stats[l] = st; a->y[l] = yy; }
  }
  return ar.off;
}
extern "C" size_t dm_mlp_acts_floats(int rows, int hidden, int layers) {
  if (rows < 0 || hidden < 0 || layers < 0 || layers > DM_MAX_MLP_LAYERS) return 0;
  return mlp_carve(rows, hidden, layers, nullptr, nullptr);
}

// Scratch (floats) any MLP call may need beyond the split-K region: activation ping-pong for calls without an `acts`
// buffer (3 x rows x hidden), backward ping-pong (2 x rows x hidden) and the panel kernels' column partials.
static size_t mlp_ws_floats(int rows, int hidden, int layers) {
  const size_t rh = dm_align_up((size_t)rows * hidden, 64);
  return DM_SPLITK_FLOATS + 3 * rh + dm_align_up((size_t)layers * dm_panel_count(rows) * 3 * hidden, 64) +
         dm_align_up((size_t)hidden * hidden, 64) + 256 +     // + the bf16 panel backward's transposed weights
         dm_align_up((size_t)hidden * 2048 + (size_t)layers * hidden * hidden / 2 + 64, 64) +     // + bf16 weight copies (in_dim <= 4096)
         rh + dm_align_up((size_t)hidden * 4096, 64);       // + sparse-column contribution and W0^T (sparse_cols, in_dim <= 4096)
}
extern "C" size_t dm_mlp_ws_floats(int rows, int hidden, int layers) {
  if (rows < 0 || hidden < 0 || layers < 0 || layers > DM_MAX_MLP_LAYERS) return 0;
  return mlp_ws_floats(rows, hidden, layers);
}

// `acts` is carved for acts_total_rows rows; this call fills rows [acts_row_off, acts_row_off + rows) of every
// per-layer array (the imagination rollout writes one horizon step at a time into one (H*M)-row activation set).
// acts == nullptr (heads nobody differentiates: critic_target, the dream's reward / terminal heads, inference): the
// activations ping-pong through `ws` and no pre-activation / statistics are written at all.
// Rows >= the panel threshold run each Linear -> LayerNorm -> ELU as ONE row-panel launch (panel.hip), with the output
// layer folded into the last one's epilogue; smaller batches keep GEMM + LayerNorm launches (the 64-row panels would
// leave most CUs idle there).
int dm_mlp_fwd_launch(int rows, int in_dim, int hidden, int layers, int out_dim, const float* x, int ldx,
                      const dm_mlp_params* p, float* acts, int acts_total_rows, int acts_row_off, float* out, int ldout,
                      void* ws, size_t ws_bytes, hipStream_t st, const float* chain_wpack, int sparse_cols,
                      const float* chain_add0, const DmChainSample* chain_sample) {
  DM_REQUIRE(layers >= 1 && layers <= DM_MAX_MLP_LAYERS, DM_E_SHAPE, "mlp: layers=%d", layers);
  DM_REQUIRE(sparse_cols >= 0 && sparse_cols < in_dim, DM_E_SHAPE, "mlp: sparse_cols=%d of in_dim=%d", sparse_cols, in_dim);
  DM_REQUIRE(ws_bytes >= DM_SPLITK_FLOATS * sizeof(float), DM_E_WORKSPACE, "mlp_fwd: workspace too small");
  MlpActs a;
  if (acts) {
    DM_REQUIRE(acts_row_off >= 0 && acts_row_off + rows <= acts_total_rows, DM_E_SHAPE, "mlp_fwd: acts row window");
    mlp_carve(acts_total_rows, hidden, layers, acts, &a);
    for (int l = 0; l < layers; ++l) {
      a.xpre[l] += (size_t)acts_row_off * hidden;
      a.stats[l] += (size_t)acts_row_off * 2;
      a.y[l] += (size_t)acts_row_off * hidden;
    }
  } else {
    DmArena ar(ws, ws_bytes);
    ar.take(DM_SPLITK_FLOATS);
    float* t0 = ar.take((size_t)rows * hidden);
    float* t1 = ar.take((size_t)rows * hidden);
    float* t2 = ar.take((size_t)rows * hidden);
    float* t3 = ar.take((size_t)rows * 2);
    DM_REQUIRE(ar.ok, DM_E_WORKSPACE, "mlp_fwd: workspace too small for the activation ping-pong (need %zu floats)", ar.off);
    for (int l = 0; l < layers; ++l) { a.xpre[l] = t2; a.stats[l] = t3; a.y[l] = (l & 1) ? t1 : t0; }
  }
  const size_t skb = DM_SPLITK_FLOATS * sizeof(float);
  const float* in = x;
  int ldin = ldx, kin = in_dim;
  // layer_norm=False (common.py:68-74 NoNorm): null LayerNorm parameters, Linear -> ELU per layer on the plain product path
  const bool normed = p->ln_g[0] != nullptr;
  for (int l = 0; l < layers; ++l)
    DM_REQUIRE((p->ln_g[l] != nullptr) == normed && (p->ln_b[l] != nullptr) == normed, DM_E_NULL,
               "mlp: layer %d mixes LayerNorm and NoNorm parameters", l);
  const bool panel = normed && dm_panel_ok(rows, hidden) && (in_dim & 3) == 0 && ((uintptr_t)p->w[0] & 15) == 0;
  if (!panel && dm_mlp_chain_ok(rows, in_dim, hidden, layers, out_dim, x, ldx, p)) {    // all layers + output in ONE launch
    const float* wpack = chain_wpack;
    // Sparse trailing columns (see the row-panel path below): layer 0 multiplies the dense columns only - for the DreamerV2
    // feature that is 58 % of the whole kernel's MFMA issue and 36 % of its weight stream - and the one-hot columns'
    // contribution is a gathered sum of W0^T rows, added before the LayerNorm.  A caller that packed the weights itself says
    // so by handing the addend (chain_add0) with them.
    int k0 = 0;
    const float* add0 = nullptr;
    if (wpack && chain_add0 && dm_mlp_chain_sparse_ok(in_dim, sparse_cols)) { k0 = in_dim - sparse_cols; add0 = chain_add0; }
    if (!wpack) {      // pack the weights fragment-major into the (otherwise unused) split-K region of the workspace
      const size_t need = dm_mlp_chain_pack_floats(in_dim, layers);
      if (need <= DM_SPLITK_FLOATS && ws_bytes >= need * sizeof(float) && ((uintptr_t)ws & 15) == 0) {
        if (dm_mlp_chain_sparse_ok(in_dim, sparse_cols) && (ldx & 3) == 0) {
          DmArena ar(ws, ws_bytes);       // scratch past the split-K region and the activation ping-pong (as the panel path)
          ar.take(DM_SPLITK_FLOATS);
          ar.take((size_t)rows * hidden); ar.take((size_t)rows * hidden); ar.take((size_t)rows * hidden); ar.take((size_t)rows * 2);
          float* g = ar.take((size_t)rows * hidden);
          float* w0t = ar.take((size_t)in_dim * hidden);
          if (ar.ok) {
            const int dense = in_dim - sparse_cols;
            DM_TRY(dm_permute4_launch(p->w[0], w0t, 1, 1, hidden, in_dim, 0, 1, 3, 2, st));       // W0 (hidden, in) -> W0^T (in, hidden)
            DM_TRY(dm_sparse_rows_launch(rows, hidden, sparse_cols, x + dense, ldx, w0t + (size_t)dense * hidden, g, hidden, st));
            k0 = dense; add0 = g;
          }
        }
        DM_TRY(dm_mlp_chain_pack_launch(in_dim, layers, p, (float*)ws, st, k0));
        wpack = (const float*)ws;
      }
    }
    return dm_mlp_chain_fwd_launch(rows, in_dim, layers, out_dim, x, ldx, p, acts ? a.xpre : nullptr,
                                   acts ? a.stats : nullptr, acts ? a.y : nullptr, out, ldout, wpack, st, k0, add0, chain_sample);
  }
  DM_REQUIRE(!chain_sample, DM_E_SHAPE, "mlp_fwd: the fused sampler rides in the whole-MLP kernel only (rows %d)", rows);
  if (panel) {
    const bool fuse_out = out_dim <= 32;
    // bf16 operands: one bf16 copy of the hidden-layer weights per call (the panels then stream half the bytes, unconverted)
    const unsigned short* wh[DM_MAX_MLP_LAYERS] = {};
    DmArena ar(ws, ws_bytes);       // optional scratch of this path, past the split-K region and the activation ping-pong
    ar.take(DM_SPLITK_FLOATS);
    ar.take((size_t)rows * hidden); ar.take((size_t)rows * hidden); ar.take((size_t)rows * hidden); ar.take((size_t)rows * 2);
    if (ar.ok && dm_cur_precision() && in_dim <= 4096 && (in_dim & 7) == 0) {
      const size_t mark = ar.off;
      const float* src[DM_MAX_MLP_LAYERS];
      unsigned short* dst[DM_MAX_MLP_LAYERS];
      int rws[DM_MAX_MLP_LAYERS], cls[DM_MAX_MLP_LAYERS];
      for (int l = 0; l < layers; ++l) {
        const int k = l == 0 ? in_dim : hidden;
        src[l] = p->w[l]; rws[l] = hidden; cls[l] = k;
        dst[l] = (unsigned short*)ar.take(((size_t)hidden * k + 1) / 2);
      }
      if (ar.ok) {
        DM_TRY(dm_panel_bf16_weights_launch(layers, src, dst, rws, cls, 0, st));
        for (int l = 0; l < layers; ++l) wh[l] = dst[l];
      } else {
        ar.off = mark; ar.ok = true;
      }
    }
    // Sparse trailing columns (the one-hot latent of a feature row): layer 0 multiplies only the dense columns, the
    // others contribute the sum of the weight rows their non-zeros name (x W^T = x_dense W_d^T + sum_e x_e W^T[e]);
    // for the DreamerV2 feature (600 dense + 32 of 1024) that is 63 % of the layer's flops replaced by a 32-row gather
    // (40 000 rows: 764 us -> 290 us + 135 us).  fp32 only: with bf16 operands the product is cheaper than the gather
    // (measured: step +0.4 ms), DM_MLP_SPARSE_BF16=1 forces it there for tests.
    static const int no_sparse = getenv("DM_MLP_NO_SPARSE") ? 1 : 0;         // A/B switch
    static const int sparse_bf16 = getenv("DM_MLP_SPARSE_BF16") ? 1 : 0;
    const float* addm = nullptr;
    int k0 = in_dim;
    const int dense = in_dim - sparse_cols;
    if (ar.ok && sparse_cols > 0 && !no_sparse && (!dm_cur_precision() || sparse_bf16) && in_dim <= 4096 && (dense & 7) == 0 && dense >= 32 && (ldx & 3) == 0) {
      float* g = ar.take((size_t)rows * hidden);
      float* w0t = ar.take((size_t)in_dim * hidden);
      if (ar.ok) {
        DM_TRY(dm_permute4_launch(p->w[0], w0t, 1, 1, hidden, in_dim, 0, 1, 3, 2, st));       // W0 (hidden, in) -> W0^T (in, hidden)
        DM_TRY(dm_sparse_rows_launch(rows, hidden, sparse_cols, x + dense, ldx, w0t + (size_t)dense * hidden, g, hidden, st));
        addm = g; k0 = dense;
      }
    }
    for (int l = 0; l < layers; ++l) {
      const bool last = l == layers - 1;
      const bool fo = last && fuse_out;
      DM_TRY(dm_panel_ln_fwd_launch(rows, hidden, l == 0 ? k0 : kin, in, ldin, p->w[l], p->b[l], p->ln_g[l], p->ln_b[l], 1e-3f,
                                    acts ? a.xpre[l] : nullptr, acts ? a.stats[l] : nullptr,
                                    (fo && !acts) ? nullptr : a.y[l], fo ? p->w[layers] : nullptr,
                                    fo ? p->b[layers] : nullptr, out, out_dim, ldout, st, wh[l], l == 0 ? in_dim : 0,
                                    l == 0 ? addm : nullptr));
      in = a.y[l]; ldin = hidden; kin = hidden;
    }
    if (fuse_out) return DM_OK;
  } else {
    for (int l = 0; l < layers; ++l) {
      DmGemm q;
      q.M = rows; q.N = hidden; q.K = kin;
      q.A = in; q.lda = ldin;
      q.B = p->w[l]; q.ldb = kin;
      q.C = a.xpre[l]; q.ldc = hidden;
      q.bias = p->b[l];
      DM_TRY(dm_gemm_launch(q, ws, skb, st));
      if (normed)
        DM_TRY(dm_ln_elu_fwd_launch(rows, hidden, a.xpre[l], hidden, p->ln_g[l], p->ln_b[l], 1e-3f, a.y[l], hidden,
                                    a.stats[l], st));
      else DM_TRY(dm_elu_fwd_launch(rows, hidden, a.xpre[l], hidden, a.y[l], hidden, st));
      in = a.y[l]; ldin = hidden; kin = hidden;
    }
  }
  DmGemm q;
  q.M = rows; q.N = out_dim; q.K = hidden;
  q.A = in; q.lda = hidden;
  q.B = p->w[layers]; q.ldb = hidden;
  q.C = out; q.ldc = ldout;
  q.bias = p->b[layers];
  return dm_gemm_launch(q, ws, skb, st);
}

extern "C" int dm_mlp_head_fwd(int rows, int in_dim, int hidden, int layers, int out_dim, const float* x, int ldx,
                               const dm_mlp_params* p, float* acts, float* out, void* ws, size_t ws_bytes, void* stream) {
  DM_REQUIRE(x && p && out && ws, DM_E_NULL, "mlp_head_fwd: null pointer");      // acts may be NULL (no backward)
  DmPrecisionScope prec(p->precision);
  return dm_mlp_fwd_launch(rows, in_dim, hidden, layers, out_dim, x, ldx, p, acts, rows, 0, out, out_dim, ws, ws_bytes,
                           (hipStream_t)stream);
}

extern "C" int dm_mlp_head_fwd_sparse(int rows, int in_dim, int sparse_cols, int hidden, int layers, int out_dim,
                                      const float* x, int ldx, const dm_mlp_params* p, float* acts, float* out, void* ws,
                                      size_t ws_bytes, void* stream) {
  DM_REQUIRE(x && p && out && ws, DM_E_NULL, "mlp_head_fwd_sparse: null pointer");
  DmPrecisionScope prec(p->precision);
  return dm_mlp_fwd_launch(rows, in_dim, hidden, layers, out_dim, x, ldx, p, acts, rows, 0, out, out_dim, ws, ws_bytes,
                           (hipStream_t)stream, nullptr, sparse_cols);
}

// Rows [row0, row0 + rows) of an MLP forward over rows_total rows: x, out and acts are the FULL arrays (acts carved for
// rows_total rows; may be NULL).  Row results do not depend on the window (no cross-row reduction in the forward).
extern "C" int dm_mlp_head_fwd_rows(int rows_total, int row0, int rows, int in_dim, int sparse_cols, int hidden, int layers,
                                    int out_dim, const float* x, int ldx, const dm_mlp_params* p, float* acts, float* out,
                                    void* ws, size_t ws_bytes, void* stream) {
  DM_REQUIRE(x && p && out && ws, DM_E_NULL, "mlp_head_fwd_rows: null pointer");
  DM_REQUIRE(row0 >= 0 && rows >= 1 && row0 + rows <= rows_total, DM_E_SHAPE, "mlp_head_fwd_rows: window [%d, %d) of %d rows", row0,
             row0 + rows, rows_total);
  DmPrecisionScope prec(p->precision);
  return dm_mlp_fwd_launch(rows, in_dim, hidden, layers, out_dim, x + (size_t)row0 * ldx, ldx, p, acts, rows_total, row0,
                           out + (size_t)row0 * out_dim, out_dim, ws, ws_bytes, (hipStream_t)stream, nullptr, sparse_cols);
}

extern "C" int dm_mlp_head_bwd(int rows, int in_dim, int hidden, int layers, int out_dim, const float* x, int ldx,
                               const dm_mlp_params* p, const float* acts, const float* dout, const dm_mlp_grads* g,
                               float* dx, int lddx, int dx_accum, void* ws, size_t ws_bytes, void* stream) {
  DM_REQUIRE(x && p && acts && dout && g && ws, DM_E_NULL, "mlp_head_bwd: null pointer");
  DmPrecisionScope prec(p->precision);
  DM_REQUIRE(layers >= 1 && layers <= DM_MAX_MLP_LAYERS, DM_E_SHAPE, "mlp: layers=%d", layers);
  hipStream_t st = (hipStream_t)stream;
  MlpActs a;
  mlp_carve(rows, hidden, layers, const_cast<float*>(acts), &a);
  DmArena ar(ws, ws_bytes);
  float* splitk = ar.take(DM_SPLITK_FLOATS);
  float* dy = ar.take((size_t)rows * hidden);
  float* dxp = ar.take((size_t)rows * hidden);
  const bool normed = p->ln_g[0] != nullptr;      // layer_norm=False: no LayerNorm parameters, no gradients for them
  const bool panel = normed && dm_panel_ok(rows, hidden) && ((uintptr_t)p->w[layers] & 15) == 0;
  const int npanels = dm_panel_count(rows);
  float* colpart = ar.take(panel ? (size_t)layers * npanels * 3 * hidden : 0);
  float* wt = (panel && dm_cur_precision()) ? ar.take((size_t)hidden * hidden) : nullptr;
  DM_REQUIRE(ar.ok, DM_E_WORKSPACE, "mlp_head_bwd: workspace too small (need %zu floats)", ar.off);
  // bf16 operands: bf16 copies of W_l^T (hidden x hidden, l >= 1) for the data-gradient panels, one launch for all layers
  const unsigned short* wth[DM_MAX_MLP_LAYERS] = {};
  if (panel && dm_cur_precision() && (hidden & 7) == 0 && layers > 1) {
    const float* src[DM_MAX_MLP_LAYERS];
    unsigned short* dst[DM_MAX_MLP_LAYERS];
    int rws[DM_MAX_MLP_LAYERS], cls[DM_MAX_MLP_LAYERS];
    for (int l = 1; l < layers; ++l) {
      src[l - 1] = p->w[l]; rws[l - 1] = hidden; cls[l - 1] = hidden;          // destination (in x out) = W_l^T, W_l is (out x in)
      dst[l - 1] = (unsigned short*)ar.take(((size_t)hidden * hidden + 1) / 2);
    }
    if (ar.ok) {
      DM_TRY(dm_panel_bf16_weights_launch(layers - 1, src, dst, rws, cls, 1, st));
      for (int l = 1; l < layers; ++l) wth[l] = dst[l - 1];
    }
  }
  const size_t skb = DM_SPLITK_FLOATS * sizeof(float);

  if (panel) {
    // Row-panel backward (panel.hip): the data-gradient product of layer l+1 and the LayerNorm/ELU backward of layer l
    // are ONE launch that also leaves per-panel column sums for dbias / dgamma / dbeta; one final launch adds them up.
    {
      DmGemm q;   // dW_L[o][h] = sum_r dout[r][o] y[r][h]
      q.a_layout = 1; q.b_layout = 1;
      q.M = out_dim; q.N = hidden; q.K = rows;
      q.A = dout; q.lda = out_dim;
      q.B = a.y[layers - 1]; q.ldb = hidden;
      q.C = g->w[layers]; q.ldc = hidden;
      DM_TRY(dm_gemm_launch(q, splitk, skb, st));
      DM_TRY(dm_colsum_launch(rows, out_dim, dout, out_dim, g->b[layers], splitk, skb, st));
    }
    float* cur = dxp;
    float* other = dy;
    DM_TRY(dm_panel_ln_bwd_launch(rows, hidden, out_dim, dout, out_dim, p->w[layers], a.xpre[layers - 1],
                                  a.stats[layers - 1], p->ln_g[layers - 1], p->ln_b[layers - 1], cur,
                                  colpart + (size_t)(layers - 1) * npanels * 3 * hidden, nullptr, st));
    for (int l = layers - 1; l >= 0; --l) {
      const float* in = l == 0 ? x : a.y[l - 1];
      const int ldin = l == 0 ? ldx : hidden;
      const int kin = l == 0 ? in_dim : hidden;
      DmGemm q;   // dW_l[h][i] = sum_r dxp_l[r][h] in[r][i]
      q.a_layout = 1; q.b_layout = 1;
      q.M = hidden; q.N = kin; q.K = rows;
      q.A = cur; q.lda = hidden;
      q.B = in; q.ldb = ldin;
      q.C = g->w[l]; q.ldc = kin;
      DM_TRY(dm_gemm_launch(q, splitk, skb, st));
      if (l > 0) {
        DM_TRY(dm_panel_ln_bwd_launch(rows, hidden, hidden, cur, hidden, p->w[l], a.xpre[l - 1], a.stats[l - 1],
                                      p->ln_g[l - 1], p->ln_b[l - 1], other, colpart + (size_t)(l - 1) * npanels * 3 * hidden,
                                      wt, st, wth[l]));
        float* t = cur; cur = other; other = t;
      } else if (dx) {
        DmGemm d;   // d(in)[r][i] = sum_h dxp_0[r][h] W_0[h][i]
        d.a_layout = 0; d.b_layout = 1;
        d.M = rows; d.N = kin; d.K = hidden;
        d.A = cur; d.lda = hidden;
        d.B = p->w[0]; d.ldb = kin;
        d.C = dx; d.ldc = lddx; d.flags = dx_accum ? DM_GEMM_ACCUM : 0;
        DM_TRY(dm_gemm_launch(d, splitk, skb, st));
      }
    }
    const float* parts[3 * DM_MAX_MLP_LAYERS];
    float* outs[3 * DM_MAX_MLP_LAYERS];
    for (int l = 0; l < layers; ++l) {
      const float* base = colpart + (size_t)l * npanels * 3 * hidden;
      parts[3 * l + 0] = base;              outs[3 * l + 0] = g->b[l];
      parts[3 * l + 1] = base + hidden;     outs[3 * l + 1] = g->ln_g[l];
      parts[3 * l + 2] = base + 2 * hidden; outs[3 * l + 2] = g->ln_b[l];
    }
    return dm_panel_colsum_final_launch(3 * layers, parts, outs, hidden, npanels, 3 * hidden, st);
  }

  // output layer
  {
    DmGemm q;   // dW_L[o][h] = sum_r dout[r][o] y[r][h]
    q.a_layout = 1; q.b_layout = 1;
    q.M = out_dim; q.N = hidden; q.K = rows;
    q.A = dout; q.lda = out_dim;
    q.B = a.y[layers - 1]; q.ldb = hidden;
    q.C = g->w[layers]; q.ldc = hidden;
    DM_TRY(dm_gemm_launch(q, splitk, skb, st));
    DM_TRY(dm_colsum_launch(rows, out_dim, dout, out_dim, g->b[layers], splitk, skb, st));
    DmGemm d;   // dy[r][h] = sum_o dout[r][o] W_L[o][h]
    d.a_layout = 0; d.b_layout = 1;
    d.M = rows; d.N = hidden; d.K = out_dim;
    d.A = dout; d.lda = out_dim;
    d.B = p->w[layers]; d.ldb = hidden;
    d.C = dy; d.ldc = hidden;
    DM_TRY(dm_gemm_launch(d, splitk, skb, st));
  }
  for (int l = layers - 1; l >= 0; --l) {
    const float* in = l == 0 ? x : a.y[l - 1];
    const int ldin = l == 0 ? ldx : hidden;
    const int kin = l == 0 ? in_dim : hidden;
    if (normed) {
      DM_TRY(dm_ln_elu_bwd_dx_launch(rows, hidden, a.xpre[l], hidden, a.y[l], hidden, a.stats[l], p->ln_g[l], dy, hidden,
                                     dxp, hidden, st));
      DM_TRY(dm_ln_elu_bwd_params_launch(rows, hidden, a.xpre[l], hidden, a.y[l], hidden, a.stats[l], dy, hidden,
                                         g->ln_g[l], g->ln_b[l], splitk, skb, st));
    } else {
      DM_TRY(dm_elu_bwd_launch(rows, hidden, a.y[l], hidden, dy, hidden, dxp, hidden, st));
    }
    DmGemm q;   // dW_l[h][i] = sum_r dxp[r][h] in[r][i]
    q.a_layout = 1; q.b_layout = 1;
    q.M = hidden; q.N = kin; q.K = rows;
    q.A = dxp; q.lda = hidden;
    q.B = in; q.ldb = ldin;
    q.C = g->w[l]; q.ldc = kin;
    DM_TRY(dm_gemm_launch(q, splitk, skb, st));
    DM_TRY(dm_colsum_launch(rows, hidden, dxp, hidden, g->b[l], splitk, skb, st));
    if (l > 0 || dx) {
      DmGemm d;   // d(in)[r][i] = sum_h dxp[r][h] W_l[h][i]
      d.a_layout = 0; d.b_layout = 1;
      d.M = rows; d.N = kin; d.K = hidden;
      d.A = dxp; d.lda = hidden;
      d.B = p->w[l]; d.ldb = kin;
      if (l > 0) { d.C = dy; d.ldc = hidden; }
      else { d.C = dx; d.ldc = lddx; d.flags = dx_accum ? DM_GEMM_ACCUM : 0; }
      DM_TRY(dm_gemm_launch(d, splitk, skb, st));
    }
  }
  return DM_OK;
}
