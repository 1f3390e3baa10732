// Shared host/device helpers for libdreamer_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "dreamer_hip.h"

int dm_fail(int code, const char* fmt, ...);

#define DM_MAX_DEVICES 16      // per-device once-flags (function attributes are per device)

// Wave priority of the latency-chain kernels (experiment, round 6): s_setprio raises a wave's instruction-issue priority on its SIMD
// over the co-resident waves of other streams' throughput kernels.  0 = leave the hardware default (compile-time: -DDM_CHAIN_SETPRIO=n).
#ifndef DM_CHAIN_SETPRIO
#define DM_CHAIN_SETPRIO 0
#endif
#define DM_CHAIN_PRIO() do { if (DM_CHAIN_SETPRIO > 0) __builtin_amdgcn_s_setprio(DM_CHAIN_SETPRIO); } while (0)
#define DM_LAUNCH_CHECK()                                                                      \
  do {                                                                                         \
    hipError_t e__ = hipGetLastError();                                                        \
    if (e__ != hipSuccess) return dm_fail(DM_E_HIP, "%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
  } while (0)

#define DM_TRY(expr)                 \
  do {                               \
    int rc__ = (expr);               \
    if (rc__ != DM_OK) return rc__;  \
  } while (0)

#define DM_REQUIRE(cond, code, ...)                     \
  do {                                                  \
    if (!(cond)) return dm_fail(code, __VA_ARGS__);     \
  } while (0)

static inline int dm_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline size_t dm_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over a caller-provided float buffer (acts / workspace carving).
struct DmArena {
  float* base;
  size_t cap;    // floats
  size_t off;    // floats
  bool ok;
  DmArena(void* p, size_t bytes) : base((float*)p), cap(bytes / sizeof(float)), off(0), ok(true) {}
  float* take(size_t nfloats) {
    size_t n = dm_align_up(nfloats, 64);   // 256-byte granules keep every carve 16B-aligned for float4 access
    if (base == nullptr || off + n > cap) { ok = false; off += n; return nullptr; }
    float* r = base + off;
    off += n;
    return r;
  }
};

// ---- weight-gradient side stream (lib.hip; include/dreamer_hip.h dm_wgrad_side_arm / _join) -------
// dm_wgrad_side_stream(st): the library's low-priority side stream if the calling thread is armed, else `st` itself.
// dm_wgrad_side_fork(st, sw): sw waits for everything enqueued on st so far (no-op when sw == st).
// dm_wgrad_side_mark(sw, st): records the "deferred work enqueued so far" event the join waits for (no-op when sw == st).
hipStream_t dm_wgrad_side_stream(hipStream_t st);
int dm_wgrad_side_fork(hipStream_t st, hipStream_t sw);
int dm_wgrad_side_mark(hipStream_t sw, hipStream_t st);

// ---- operand precision of the call in progress ---------------------------------------------------
// Every C-ABI entry point that runs contractions takes its precision from ITS OWN arguments (dm_shape.flags bit
// DM_FLAG_BF16, dm_mlp_params.precision, the DM_GEMM_BF16 flag of dm_gemm_f32) and holds it in a thread-local for the
// duration of the call, so the products it builds internally (DmGemm's default) inherit it; concurrent calls from
// different host threads / streams with different precisions do not interact.  0 = fp32, 1 = bf16 operands (RNE) with
// fp32 accumulation.
int dm_cur_precision();
struct DmPrecisionScope {
  int prev;
  explicit DmPrecisionScope(int p);
  ~DmPrecisionScope();
};

// ---- bf16 twins (conf.amp: operands STORED as bf16, gemm.hip gemm_h_kernel) ---------------------------------------------
// A composite entry point running in bf16 mode opens a DmTwinScope and registers, for each fp32 buffer it wants a bf16 copy
// of, the range and the twin's storage (scratch of its workspace, or a region of its activation arena when the backward
// pass needs the copy too).  dm_gemm_launch then (a) writes the twin of any result that lands in a registered range from its
// epilogue and marks it valid, (b) reads twins instead of fp32 operands when BOTH operands of a product have a valid one.
// Other producers (direct convolutions, pad / col2im / loss kernels) write their twin themselves and call dm_twin_mark.
// The map is thread-local and dies with the scope: nothing is remembered between calls (a backward entry point re-registers
// the arena twins its forward wrote as valid - acts must come from a forward call with the same dm_shape.flags).
// DM_BF16_NO_TWINS=1 (A/B switch) makes every scope inactive: the fp32-storage bf16 products of rounds 1-2.
struct DmTwinScope {
  bool active, opened;
  explicit DmTwinScope(bool bf16_mode);
  ~DmTwinScope();
};
bool dm_twins_on();                       // a scope is active on this thread
void dm_twin_add(const float* base, size_t n, unsigned short* twin, bool valid);
void dm_twin_mark(const float* p);        // the range holding p now has a complete twin
unsigned short* dm_twin_of(const float* p, bool need_valid);      // twin address of element p, or nullptr
// A forward entry point notes whether it wrote the arena twins of `acts`; the matching backward asks before trusting them
// (dm_bf16_twins_enable may have been flipped in between, or the forward ran without DM_FLAG_BF16): host-side table.
void dm_twin_arena_note(const void* acts, bool written);
bool dm_twin_arena_valid(const void* acts);
// dst[i] = bf16(src[i]) for up to 8 segments in one launch (weights of a call, small activations)
struct DmCvtSeg { const float* src; unsigned short* dst; size_t n; };
int dm_to_bf16_multi_launch(const DmCvtSeg* segs, int count, hipStream_t st);
static inline size_t dm_half_floats(size_t elems) { return (elems + 1) / 2; }      // floats that hold `elems` bf16
constexpr int DM_HSTORE_MIN_K = 1024;     // k-contiguous gathered / scattered products shorter than this stay on the fp32-storage kernels (gemm.hip)
constexpr int DM_HSTORE_MIN_K_DENSE = 400;

// ---- internal (C++) entry points shared between translation units -------------------------------
struct DmGatesBwd;
struct DmGemm {
  int bf16 = dm_cur_precision();                  // operand precision (see DmPrecisionScope)
  int a_layout = 0, b_layout = 0;
  int M = 0, N = 0, K = 0;
  const float* A = nullptr; int lda = 0;
  const float* B = nullptr; int ldb = 0;
  float* C = nullptr; int ldc = 0;
  const float* bias = nullptr;
  const float* add = nullptr; int ldadd = 0;
  const float* mulref = nullptr; int ldmul = 0;   // v *= ELU'(mulref[m,n]) (mulref holds ELU outputs), applied last
  const uint8_t* row_zero = nullptr;              // rows with row_zero[m] != 0 contribute 0 (reset masks), applied first
  // separable gather operands (implicit im2col): element = P[maj[major] + min[minor]] with major = the strided index
  // of the layout (row for layout 0, k for layout 1); *_tab_vec: 4 consecutive minors are 4 aligned consecutive floats
  const int* a_maj = nullptr; const int* a_min = nullptr; int a_tab_vec = 0;
  const int* b_maj = nullptr; const int* b_min = nullptr; int b_tab_vec = 0;
  int a_tab_vec8 = 0, b_tab_vec8 = 0;             // ... and 8 consecutive minors are 8 consecutive, 16-byte aligned bf16 of the twin
  int flags = 0;
  // scatter epilogue of the class-concatenated transposed convolution (see gemm.hip SC / conv.hip): c_tab[row] = {float
  // offset of the row's class-(0,0) output pixel, bit 1: odd output row exists, bit 0: odd output column exists}
  const int2* c_tab = nullptr; int sc_cout = 0, sc_wpitch = 0;
  int bias_mod = 0;                               // > 0: the bias is indexed by col % bias_mod
  // LayerNorm+ELU prologue on A (A holds pre-activations; the product uses ELU(LN(A))): <= 64-row skinny products only
  const float* ln_g = nullptr; const float* ln_b = nullptr; float ln_eps = 1e-3f;
  // LayerNorm+ELU BACKWARD prologue on A (A holds dy; the product uses dx): lnb_x = pre-activations, lnb_stats = (mean, rstd)
  const float* lnb_x = nullptr; int lnb_ldx = 0; const float* lnb_stats = nullptr;
  // The same backward in FOLDED form, split over the product that MAKES dy and the one that consumes dx (the BPTT launch schedule,
  // rssm.hip): dx B^T = rstd (g B^T - mean(g) cs - mean(g xhat) xhat B^T), g = dy ELU'(pre) gamma, xhat B^T = rstd (x B^T - mean cs).
  //   producer (C = dy): eg_x / eg_ldx pre-activations, eg_stats (mean, rstd), eg_gamma / eg_beta; its epilogue also writes
  //     g (eg_G row-major, leading dim eg_ldg, and / or eg_Gf fragment-major) and, per 16-column strip, the row sums of g and
  //     g xhat (eg_ps[strip][64][2]) - the transform is done ONCE, by the workgroup that owns the element;
  //   consumer (A = g, a plain product): lnf_ps / lnf_nps those partial sums, lnf_stats, lnf_xw = x B^T for these rows
  //     (leading dim lnf_ldxw), lnf_cs[n] = sum_k B(n,k); the correction enters in its epilogue.
  // <= 64-row skinny products only.
  const float* eg_x = nullptr; int eg_ldx = 0; const float* eg_stats = nullptr; const float* eg_gamma = nullptr; const float* eg_beta = nullptr;
  float* eg_G = nullptr; int eg_ldg = 0; float* eg_Gf = nullptr; float* eg_ps = nullptr;
  const float* lnf_ps = nullptr; int lnf_nps = 0; const float* lnf_stats = nullptr;
  const float* lnf_xw = nullptr; int lnf_ldxw = 0; const float* lnf_cs = nullptr;
  // straight-through softmax BACKWARD in the epilogue (skinny pair, second product, 32-class groups): the strip is one categorical
  // group of the completed C row (C = dz'), and  sm_dlogits += p (dz' - sum_group p dz'),  p = softmax(sm_logits)  (rssm.py:147-148)
  const float* sm_logits = nullptr; int sm_ld = 0; float* sm_dlogits = nullptr; int sm_ldd = 0;
  const struct DmGatesBwd* gates = nullptr;       // GRU gates backward in the epilogue (C = dh', N = D), skinny products only
  // Fragment-major copies of <= 64-row chain operands (dm_frag_off): A_frag mirrors A (the skinny kernel then loads its
  // MFMA fragments as contiguous KiB instead of 16 rows x 64 B per instruction); C_frag receives such a copy of C for
  // the NEXT product of the chain.  Both optional; a product that does not take the skinny path ignores A_frag and
  // fills C_frag with a pack launch, so callers may set them unconditionally.
  const float* A_frag = nullptr;
  float* C_frag = nullptr;
  // bf16-STORAGE operands (both or neither; same layouts, gather tables and leading dimensions, in elements): the product
  // reads these instead of A / B (gemm_h_kernel).  C_h: optional bf16 twin of the result (same ldc), any operand format.
  const unsigned short* A_h = nullptr;
  const unsigned short* B_h = nullptr;
  unsigned short* C_h = nullptr;
  bool no_twin = false;                           // do not write the result's twin even if the call's twin map has room for it
};
// Fragment-major layout of a <= 64-row block X[row][k]: the 16 B a lane of v_mfma_f32_16x16x4_f32 loads for a 16-k chunk
// (lane l: row 16*mb + (l&15), k = 16*c + 4*(l>>4) .. +3) sit at ((c*4 + mb)*4 + (l>>4))*16 + (l&15) in units of 16 B, so
// one load instruction of a wave reads ONE contiguous KiB.  Measured (scripts/microbench/l2_stream.hip): the row-major
// gather of the same fragments runs at 16.6 B/clk/CU whatever the cache level, contiguous loads at ~50.
__host__ __device__ __forceinline__ size_t dm_frag_off(int row, int k) {
  return ((((size_t)(k >> 4) * 4 + (row >> 4)) * 4 + ((k >> 2) & 3)) * 16 + (row & 15)) * 4 + (k & 3);
}
static inline size_t dm_frag_floats(int K) { return (size_t)dm_cdiv(K, 16) * 1024; }      // 64 rows x K rounded up to 16
int dm_frag_pack_launch(int rows, int K, const float* X, int ldx, float* Xf, hipStream_t st);
// x = bias + add + sum_s Wt[s*C + idx[r][s]] (+ Wt2[idx2[r]])  (z_mlp of a one-hot latent as a gather-sum; Wt = W^T, (S*C, n)
// row-major; idx2 / Wt2 optional: one more gathered row per output row, a_mlp of a one-hot action);
// y (optional) = ELU(LayerNorm(x)); x (optional when y is given); x_frag (optional, rows <= 64): fragment-major copy of x
// The posterior chain's steps [t_begin, t_end) as ONE persistent kernel whose workgroups (one per CU, all XCDs) keep their
// column slices of the cell's weights in LDS (rssm_lds.hip); all activation pointers are the bases of the full (T*B)-row
// buffers; step t_begin - 1 has been run by the launch schedule (its h in `feat`, its indices in `idx`, the masked inputs of
// step t_begin in `hin`).
struct DmRssmLds {
  int B, D, Hd, S, C, F, t_begin, t_end;
  const float *wzt, *zb, *wih, *bih, *whh, *bhh, *wph, *bph, *wpo, *bpo, *in_g, *in_b, *post_g, *post_b, *ea, *ee;
  const uint8_t* reset; const float* u; const int32_t* forced;
  float *x1, *gi, *gh, *hin, *zin, *feat, *x2, *post; int32_t* idx;
  float* ws; size_t ws_floats;      // dm_rssm_lds_ws_floats(...) floats: the per-step exchange buffers
};
bool dm_rssm_lds_ok(int B, int D, int Hd, int S, int C);
size_t dm_rssm_lds_ws_floats(int B, int D, int Hd, int S, int C, int steps);
int dm_rssm_lds_launch(const DmRssmLds& q, hipStream_t st);
extern "C" int dm_rssm_lds_enable(int on);
bool dm_z_embed_ok(int n);
// x[r][:] = sum over the non-zero e of z[r][e] * Wt[e][:]   (Wt: (Zc, n) row-major, n <= 1024, n % 4 == 0): exact for any z,
// cheap for rows of concatenated one-hot groups
int dm_sparse_rows_launch(int rows, int n, int Zc, const float* z, int ldz, const float* Wt, float* x, int ldx,
                          hipStream_t st);
int dm_z_embed_launch(int rows, int n, int S, int C, const int32_t* idx, const uint8_t* row_zero, const float* Wt,
                      const float* bias, const float* add, int ldadd, const int32_t* idx2, const float* Wt2, float* x, int ldx,
                      float* x_frag, const float* gamma, const float* beta, float eps, float* y, int ldy, hipStream_t st,
                      float* y_frag = nullptr, float* stats = nullptr);      // y_frag (rows <= 64): one workgroup per row, LayerNorm + ELU behind the sums, fragment-major copy of y
struct DmGatesBwd {
  const float* gi; const float* gh; const float* h_in; int ldh, D;
  float* dgi; float* dgh; float* dprev; int ldp; const uint8_t* row_zero;   // dprev (nullable) += mask * dh' * u
  float* dgi_frag = nullptr; float* dgh_frag = nullptr;                     // optional fragment-major copies (dm_frag_off, K = 3D)
};
// categorical sampler riding in the epilogue of a <= 64-row logits product (gemm_skinny.hip)
struct DmSample {
  const float* u = nullptr; const int32_t* forced = nullptr;   // uniforms (rows, groups) or forced indices
  float* onehot = nullptr; int ldo = 0;                        // one-hot sample rows (leading dim ldo)
  int32_t* idx = nullptr;                                      // optional (rows, groups)
  float* z_next = nullptr; const uint8_t* next_reset = nullptr;   // optional: next step's reset-masked sample input (dense rows)
  float* z_next_frag = nullptr;                                // optional: fragment-major copy of z_next (dm_frag_off)
};
bool dm_skinny_ln_ok(int M, int N, int K);
int dm_gemm_sample_launch(const DmGemm& q, const DmSample& sm, hipStream_t stream);
int dm_gemm_launch(const DmGemm& g, void* ws, size_t ws_bytes, hipStream_t stream);
int dm_gemm_skinny_try(const DmGemm& g, hipStream_t stream);   // gemm_skinny.hip: 1 = handled, 0 = not applicable, <0 error
// two independent products; one launch when both are <= 64-row skinny products, else two ordinary launches
int dm_gemm_pair_launch(const DmGemm& g0, const DmGemm& g1, void* ws, size_t ws_bytes, hipStream_t stream);

// element-wise / row-wise launchers (elementwise.hip)
int dm_colsum_launch(int rows, int n, const float* x, int ld, float* out, void* ws, size_t ws_bytes, hipStream_t st);
// layer_norm=False (common.py:68-74): y = ELU(x); dx = dy * ELU'(y)
int dm_elu_fwd_launch(int rows, int n, const float* x, int ldx, float* y, int ldy, hipStream_t st);
int dm_elu_bwd_launch(int rows, int n, const float* y, int ldy, const float* dy, int lddy, float* dx, int lddx,
                      hipStream_t st);
int dm_ln_elu_fwd_launch(int rows, int n, const float* x, int ldx, const float* gamma, const float* beta, float eps,
                         float* y, int ldy, float* stats, hipStream_t st);
int dm_ln_elu_bwd_dx_launch(int rows, int n, const float* x, int ldx, const float* y, int ldy, const float* stats,
                            const float* gamma, const float* dy, int lddy, float* dx, int lddx, hipStream_t st);
int dm_ln_elu_bwd_params_launch(int rows, int n, const float* x, int ldx, const float* y, int ldy, const float* stats,
                                const float* dy, int lddy, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                hipStream_t st);
// LayerNorm GRU cells (rnn.py:95-138): kind 1 = gru_layernorm, 2 = gru_layernorm_dv2; ln_g / ln_b: 3 pointers each
// (kind 1: LN_reset, LN_update, LN_newval of width D; kind 2: slot 0 = the one LayerNorm of width 3D).  One layer of a
// stack of such cells: D = the layer width, ldg / ldst / ldn = row strides of the gate matrices (gi, gh, gs, dgi, dgh, dg),
// of the statistics and of h_next (0: the packed single-cell strides 3D, 6, D).
int dm_gru_norm_fwd_launch(int kind, int rows, int D, const float* gi, const float* gh, const float* h_in, int ldh,
                           const float* const* ln_g, const float* const* ln_b, float* h_out, int ldo, float* gs, float* gst,
                           float* h_next, const uint8_t* next_reset, hipStream_t st, int ldg = 0, int ldst = 0, int ldn = 0);
int dm_gru_norm_bwd_launch(int kind, int rows, int D, const float* gh, const float* h_in, int ldh, const float* gs,
                           const float* gst, const float* const* ln_g, const float* const* ln_b, const float* dh_out, int lddh,
                           float* dgi, float* dgh, float* dg, float* dh_in, int lddi, const uint8_t* row_zero, hipStream_t st,
                           int ldg = 0, int ldst = 0);
int dm_gru_norm_param_grads_launch(int kind, int rows, int D, const float* gs, const float* gst, const float* dg, float* dgam,
                                   float* dbet, hipStream_t st, int ldg = 0, int ldst = 0);
// h_next (optional, rows x D): the NEXT step's masked state input, next_reset[r] ? 0 : h_out   (rssm.py:134)
// h_frag / h_next_frag (optional, rows <= 64): fragment-major copies (dm_frag_off) of h_out and of h_next
int dm_gru_gates_fwd_launch(int rows, int D, const float* gi, const float* gh, const float* h_in, int ldh, float* h_out,
                            int ldo, float* h_next, const uint8_t* next_reset, float* h_frag, float* h_next_frag,
                            hipStream_t st, int ldg = 0, int ldn = 0);      // ldg / ldn: row strides of gi, gh / h_next (0: 3*D / D)
// dh_in (+)= row_mask * dh_out*u  (accum: add into dh_in; row_zero: rows whose flag is set contribute 0)
int dm_gru_gates_bwd_launch(int rows, int D, const float* gi, const float* gh, const float* h_in, int ldh,
                            const float* dh_out, int lddh, float* dgi, float* dgh, float* dh_in, int lddi, int accum,
                            const uint8_t* row_zero, hipStream_t st, int ldg = 0);
// z_next (optional, rows x groups*C): the NEXT step's masked sample input, next_reset[r] ? 0 : onehot
int dm_sample_onehot_launch(int rows, int groups, int C, const float* logits, int ldl, const float* u,
                            const int32_t* forced, float* onehot, int ldo, int32_t* idx, float* z_next,
                            const uint8_t* next_reset, hipStream_t st);
// Gaussian latents (C = 0 in the sampler / KL launchers: parameter rows are (mean[S] | raw std[S]), std = 2 sigmoid + 0.1):
// backward of z = mean + std * eps into the parameter gradient (eps recovered from z)
int dm_gauss_sample_bwd_launch(int rows, int S, const float* par, int ldp, const float* z, int ldz, const float* dz,
                               int lddz, float* dpar, int lddp, int accum, hipStream_t st);
int dm_kl_fwd_launch(int rows, int S, int C, const float* post, const float* prior, float* kl, float* ep, float* eq,
                     hipStream_t st);
int dm_kl_bwd_launch(int rows, int S, int C, const float* post, const float* prior, float sp, float sq, float* dpost,
                     float* dprior, hipStream_t st);
int dm_st_softmax_bwd_launch(int rows, int groups, int C, const float* logits, int ldl, const float* dz, int lddz,
                             float* dlogits, int lddl, int accum, hipStream_t st);
int dm_mask_rows_launch(int rows, int n, const float* x, int ldx, const uint8_t* reset, float* y, int ldy, hipStream_t st);
int dm_mask_rows2_launch(int rows, int n1, const float* x1, int ldx1, float* y1, int ldy1, int n2, const float* x2,
                         int ldx2, float* y2, int ldy2, const uint8_t* reset, hipStream_t st);
int dm_sample_continuous_launch(int kind, int rows, int A, const float* params, const float* eps, float* action,
                                hipStream_t st);
int dm_mul_elu_grad_launch(size_t n, const float* dy, const float* yact, float* out, hipStream_t st);

// conv helpers (conv.hip)
int dm_im2col_s2_launch(int n, int hb, int wb, int c, int k, const float* big, int big_nchw, float* col, hipStream_t st);
int dm_col2im_s2_launch(int n, int hb, int wb, int c, int k, const float* col, const float* bias, int flags,
                        const float* elu_ref, float* big, hipStream_t st);
int dm_permute4_launch(const float* src, float* dst, int d0, int d1, int d2, int d3, int p0, int p1, int p2, int p3,
                       hipStream_t st);

// direct small-channel convolutions (conv_direct.hip): encoder layer 1 (3 -> d, k4 s2) and decoder layer 4 (d -> 3, k6 s2)
bool dm_enc_l1_direct_ok(int ch, int d, int img);
int dm_enc_l1_fwd_launch(int frames, int d, int u8, const void* image, const float* w, const float* bias, float* wt, float* y,
                         unsigned short* y_h, hipStream_t st);
size_t dm_enc_l1_wgrad_part_floats(int frames, int d);
int dm_enc_l1_wgrad_launch(int frames, int d, int u8, const void* image, const float* G, float* part, float* dW, void* ws,
                           size_t ws_bytes, hipStream_t st);
bool dm_dec_l4_direct_ok(int ch, int d, int hs, int k);
size_t dm_dec_l4_w4_floats(int d);
int dm_dec_l4_fwd_launch(int frames, int d, const float* x, const float* w, const float* bias, float* w4, float* out,
                         hipStream_t st);
bool dm_dec_l4_bwd_direct_ok(int ch, int d, int hs, int k);      // the image layer's backward as direct kernels (conv_direct.hip)
size_t dm_dec_l4_wp_floats(int d);
size_t dm_dec_l4_wgrad_part_floats(int frames, int d);
int dm_dec_l4_dgrad_launch(int frames, int d, const float* G4, const float* w, float* wp, const float* x3, float* dx,
                           unsigned short* dx_h, hipStream_t st);
int dm_dec_l4_wgrad_launch(int frames, int d, const float* G4, const float* x3, float* part, float* dW, void* ws,
                           size_t ws_bytes, hipStream_t st);

// fused MLP (mlp.hip)
// chain_wpack (optional): fragment-major weights already packed by the caller (dm_mlp_chain_pack_launch) for the whole-MLP
// kernel; null: packed here, per call, into the workspace
// chain_sample (with chain_wpack, out_dim <= 32): the one-hot categorical draw of the output row (dreamer.py:198-200: the
// rollout's action sampler) rides in the whole-MLP kernel's output stage - same rule and operation order as
// dm_sample_onehot_launch(rows, 1, out_dim, out, ...), so the drawn indices are bit-identical to the stand-alone sampler's
struct DmChainSample {
  const float* u;        // one uniform per row
  float* onehot;         // rows x out_dim (ldo floats per row)
  int ldo;
  int32_t* idx;          // optional
};
int dm_mlp_fwd_launch(int rows, int in_dim, int hidden, int layers, int out_dim, const float* x, int ldx,
                      const dm_mlp_params* p, float* acts, int acts_total_rows, int acts_row_off, float* out, int ldout,
                      void* ws, size_t ws_bytes, hipStream_t st, const float* chain_wpack = nullptr, int sparse_cols = 0,
                      const float* chain_add0 = nullptr, const DmChainSample* chain_sample = nullptr);
// chain_add0 (with chain_wpack packed for k0 = in_dim - sparse_cols): the caller already has the sparse columns' contribution
// (the imagination rollout knows the latent's indices and gathers it, rssm.hip)
// sparse_cols > 0: the LAST sparse_cols columns of x are mostly zero (one-hot latent groups); the row-panel path then
// multiplies only the dense columns and adds the sparse ones' contribution as a sum of weight rows (dm_sparse_rows_launch)

// row-panel Linear + LayerNorm/ELU kernels for the 400-wide MLP heads (panel.hip)
bool dm_panel_ok(int rows, int hidden);
int dm_panel_count(int rows);
int dm_panel_ln_fwd_launch(int rows, int hidden, int kin, const float* x, int ldx, const float* W, const float* b,
                           const float* gamma, const float* beta, float eps, float* xpre, float* stats, float* y,
                           const float* wout, const float* bout, float* out, int out_dim, int ldout, hipStream_t st,
                           const unsigned short* Wh = nullptr, int ldw = 0, const float* addm = nullptr);
// ldw: row stride of W / Wh when the product covers only the first kin of its columns (0: kin); addm (rows x hidden):
// added to the product before the LayerNorm (the other columns' contribution, computed elsewhere)
int dm_panel_bf16_weights_launch(int count, const float* const* w, unsigned short* const* dst, const int* rows, const int* cols,
                                 int transpose, hipStream_t st);
int dm_panel_ln_bwd_launch(int rows, int hidden, int kup, const float* dup, int lddup, const float* W, const float* xpre,
                           const float* stats, const float* gamma, const float* beta, float* dx, float* colpart, float* wt,
                           hipStream_t st, const unsigned short* Wth = nullptr);
int dm_panel_colsum_final_launch(int count, const float* const* part, float* const* out, int n, int npanels, int pstride,
                                 hipStream_t st);

// whole-MLP forward in one launch for the 400-wide heads below the panel threshold (mlp_chain.hip)
bool dm_mlp_chain_ok(int rows, int in_dim, int hidden, int layers, int out_dim, const float* x, int ldx,
                     const dm_mlp_params* p);
size_t dm_mlp_chain_pack_floats(int in_dim, int layers);
// k0 (0 = in_dim): layer 0 packed / multiplied over the first k0 input columns only; add0 (rows x 400): the remaining (sparse,
// one-hot) columns' contribution, added before the LayerNorm (dm_sparse_rows_launch / dm_z_embed_launch make it)
int dm_mlp_chain_pack_launch(int in_dim, int layers, const dm_mlp_params* p, float* wpack, hipStream_t st, int k0 = 0);
int dm_mlp_chain_fwd_launch(int rows, int in_dim, int layers, int out_dim, const float* x, int ldx, const dm_mlp_params* p,
                            float* const* xpre, float* const* stats, float* const* y, float* out, int ldout, const float* wpack,
                            hipStream_t st, int k0 = 0, const float* add0 = nullptr, const DmChainSample* sample = nullptr);
bool dm_mlp_chain_sparse_ok(int in_dim, int sparse_cols);      // the sparse-tail layer 0 applies (fp32 calls, aligned split)

int dm_prof_slot_begin(int kind, double flops, double bytes, hipStream_t st);      // gemm.hip: per-launch HIP-event timing
void dm_prof_slot_end(int slot, hipStream_t st);
extern "C" int dm_mlp_chain_min_rows(int rows);                                    // mlp_chain.hip: rows < 1 queries
bool dm_prof_active();                                                             // the per-launch profiler is recording

// split-K partial region carved at the front of every operator workspace
static const size_t DM_SPLITK_FLOATS = (size_t)16 * 1024 * 1024;

// bf16 operand path: round-to-nearest-even fp32 -> bf16 (bit pattern in the low 16 bits), 8-element MFMA fragments
// (v_cvt_pk_bf16_f32: two conversions per instruction - the integer form cost ~5 VALU operations per element and made
// the conversion, not the MFMAs or the loads, the bound of every bf16-operand main loop)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float dm_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned dm_pack_bf16x2(float lo, float hi) {       // lo in bits 0-15, hi in bits 16-31
  const dm_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ unsigned dm_f2bf(float x) { return dm_pack_bf16x2(x, 0.f) & 0xFFFFu; }
__device__ __forceinline__ uint2 dm_pack_bf16x4(float4 v) {
  return make_uint2(dm_pack_bf16x2(v.x, v.y), dm_pack_bf16x2(v.z, v.w));
}

__device__ __forceinline__ float dm_elu(float v) { return v > 0.f ? v : expm1f(v); }
// ELU'(x) expressed through y = ELU(x):  1 for x>0, exp(x) = y+1 otherwise.
__device__ __forceinline__ float dm_elu_grad_from_y(float y) { return y > 0.f ? 1.f : y + 1.f; }

__device__ __forceinline__ float dm_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float dm_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
