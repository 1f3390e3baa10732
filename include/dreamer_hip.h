/*
 * dreamer_hip.h — C-ABI of libdreamer_hip.so (MI355X / gfx950, HIP, fp32).
 *
 * This is the drop-in boundary of the DreamerV2 gradient-step hot path.  pydreamer itself has no FFI:
 * everything below is what a `torch.autograd.Function` in a pydreamer-style `Dreamer.training_step`
 * binds instead of the ATen ops the reference dispatches from its nn.Modules (file:line citations are
 * into /root/reference).  See INTEGRATION.md for the ctypes stub a maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous fp32 (unless typed otherwise), owned by the caller
 *    (torch tensors); the library never allocates or frees device memory and never synchronises;
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *  - every function returns 0 on success or a negative DM_E_* code; dm_last_error() gives the
 *    thread-local message; nothing throws across the ABI;
 *  - tensors are time-major: row index n = t*B + b for (T,B,...) tensors (reference layout,
 *    pydreamer/data.py:188, rssm.py:21-78); matrices are row-major with explicit leading dimension
 *    where a `ld*` argument is given;
 *  - workspaces: `ws` is scratch (size from dm_workspace_bytes), `acts` holds activations saved by a
 *    *_fwd call for the matching *_bwd call (size from dm_<op>_acts_floats); both are caller-allocated.
 */
#ifndef DREAMER_HIP_H
#define DREAMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM_OK 0
#define DM_E_SHAPE (-1)       /* bad / unsupported shape or config */
#define DM_E_WORKSPACE (-2)   /* workspace or acts buffer too small */
#define DM_E_HIP (-3)         /* HIP runtime error (captured with hipGetLastError) */
#define DM_E_DEVICE (-4)      /* not a gfx950 device */
#define DM_E_NULL (-5)        /* required pointer is null */

#define DM_MAX_MLP_LAYERS 8

/* model / batch geometry: the config/defaults.yaml keys the hot path reads (dreamer.py:23-58,237-277) */
typedef struct dm_shape {
  int32_t T, B, I;          /* batch_length, batch_size, iwae_samples (conv decoder: N = T*B*I frames; RSSM calls take B*I as B, I = 1) */
  int32_t H;                /* imag_horizon */
  int32_t D, Hd;            /* deter_dim, hidden_dim */
  int32_t S, C;             /* stoch_dim, stoch_discrete (Z = S*C) */
  int32_t E;                /* embed dim = 32*cnn_depth (encoders.py:76) */
  int32_t A;                /* action_dim */
  int32_t mlp_hidden;       /* 400 (a2c.py:16, decoders.py:259,289) */
  int32_t mlp_layers;       /* 4  (defaults.yaml:83,85; a2c.py:17) */
  int32_t cnn_depth;        /* 48 */
  int32_t img, img_ch;      /* 64, 3 */
  int32_t flags;            /* bits 0-1: actor distribution, 0 = onehot, 1 = tanh_normal, 2 = normal_tanh (a2c.py:43-55);
                               bit 4 (DM_FLAG_IMAGE_U8): the `image` / `target` pointers of the conv encoder / decoder are the
                               replay's native uint8 (N,H,W,C) frames; x/255-0.5 and HWC->CHW (preprocessing.py:21-29) happen
                               inside the first conv's patch loader and the MSE kernel */
} dm_shape;
#define DM_FLAG_IMAGE_U8 16
/* bits 5-6: recurrent cell (rnn.py:40-67 `gru_type`): 0 = gru (nn.GRUCell), 1 = gru_layernorm (NormGRUCell, rnn.py:95-114),
 * 2 = gru_layernorm_dv2 (NormGRUCellLateReset, rnn.py:117-138) */
#define DM_FLAG_GRU_SHIFT 5
/* bit 7: mixed precision (the reference's `amp` switch, train.py:166; BASELINE configs[2]/[4]) for THIS call: every
 * contraction the call runs takes its operands rounded to bf16 (RNE) into bf16 MFMA with fp32 accumulation; results,
 * LayerNorm, losses and storage stay fp32.  Precision is a per-call argument (here, dm_mlp_params.precision, DM_GEMM_BF16);
 * there is no process-wide switch. */
#define DM_FLAG_BF16 128
/* dm_shape.C = 0: Gaussian latents - z is S wide, the prior / posterior parameter rows 2*S wide (mean | raw std). */
/* bits 8-9: gru_layers - 1 (GRUCellStack depth, rnn.py:40-67; up to 4 layers, plain GRU cells only) */
#define DM_FLAG_GRU_LAYERS_SHIFT 8
#define DM_FLAG_GRU_LAYERS_MASK (3 << DM_FLAG_GRU_LAYERS_SHIFT)
#define DM_FLAG_GRU_MASK (3 << DM_FLAG_GRU_SHIFT)

/* ---------------------------------------------------------------- library ---------------------- */
int dm_version(void);                 /* ABI version, currently 13 (v2: LayerNorm-GRU slots; v3: per-call precision; v4: GRUCellStack layer slots;
                                         v5: dm_kl_sampled_gauss_*, dm_chain_graph_*, dm_fp32_mode - additions only;
                                         v6: LayerNorm slots of GRUCellStack layers 1..3, dm_rssm_params grows to 58;
                                         v7: dm_wgrad_side_arm / _join, dm_dream_rollout_marks, dm_mlp_head_fwd_rows - additions only;
                                         v8: dm_rssm_lds_* replace dm_rssm_persist_*;
                                         v9: dm_bptt_fold_enable added; the split-bf16 fp32 product mode and its dm_fp32_mode query removed;
                                         v10: dm_gemm_dma_enable added; the dm_chain_graph_ family (hipGraph replay of the launch chains: GPU-neutral in three rounds of
                                         measurement) and the persistent BPTT kernel with its switch dm_rssm_lds_bwd_enable - slower inside the step at every shard size - removed;
                                         dm_prof_end reports 44 kinds;
                                         v11: dm_dec_l4_bwd_direct_enable added;
                                         v12: dm_rssm_lds_status_ack / dm_rssm_lds_gave_up; the native exchange step dm_rccl_* / dm_allreduce_grads;
                                              dm_rollout_fuse_act_enable;
                                         v13: dm_wgrad_side_touch added */
const char* dm_last_error(void);      /* thread-local message of the last failing call */
int dm_device_check(void);            /* DM_OK iff the current HIP device is gfx950 */
size_t dm_workspace_bytes(const dm_shape* shp);   /* scratch needed by any call below for this shape */
/* A stream confined to the CUs whose bit is set in mask[0..words) (32 CUs per word): the host mirror reserves CUs for the
 * posterior loop's latency chain with it (no reference counterpart: torch exposes no CU masks). */
int dm_stream_create_cu_mask(const uint32_t* mask, int words, void** stream);
int dm_stream_destroy(void* stream);
/* Weight gradients off the critical chain.  In the world-model backward (train.py:189 loss_model.backward()) only the DATA
 * gradients are a chain - image decoder -> BPTT loop -> encoder; every weight / bias / LayerNorm gradient is a leaf that
 * nothing waits for until the gradient clip.  dm_wgrad_side_arm(1) arms the CALLING THREAD: while armed,
 * dm_conv_decoder_mse_bwd*() and dm_rssm_sequence_bwd() enqueue their parameter-gradient kernels on a library-owned
 * low-priority stream that waits (events) for the gradient rows they read, so they run beside the BPTT loop - a B-row
 * latency chain that leaves most CUs idle - instead of in front of / behind it (dm_rssm_sequence_bwd cuts its batched weight
 * gradients into up to four time chunks for this, armed or not: the sums are the same either way).
 * Contract while armed: the `ws` handed to such a call must not be reused by any other call before the join (the deferred
 * kernels read their gradient buffers, tables and split-K scratch there), and the parameter gradients are complete on
 * `stream` only after dm_wgrad_side_join(stream), which makes `stream` wait for the side stream and disarms the thread.
 * Unarmed (the default for every direct caller) everything is enqueued on the caller's stream.  Results are bit-identical
 * either way (same kernels, same arguments).  No reference counterpart (autograd orders these itself). */
int dm_wgrad_side_arm(int on);
int dm_wgrad_side_touch(void);          /* create the side stream now and give it one command: it then holds its hardware queue before later streams ask for one */
int dm_wgrad_side_join(void* stream);

/* ---------------------------------------------------------------- primitives ------------------- */
/* C[m,n] (ldc) = epi( sum_k A(m,k) * B(n,k) ), fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * a_layout 0: A(m,k) = A[m*lda+k]   1: A(m,k) = A[k*lda+m]
 * b_layout 0: B(n,k) = B[n*ldb+k]   1: B(n,k) = B[k*ldb+n]
 * epi(v) = act( v + bias[n] + add[m*ldadd+n] + (flags&DM_GEMM_ACCUM ? C[m,n] : 0) ), act = ELU if DM_GEMM_ELU.
 * Replaces torch.nn.functional.linear and its backward (common.py:37-65, rssm.py:138-146, rnn.py:48-49). */
#define DM_GEMM_ACCUM 1
#define DM_GEMM_ELU 2
#define DM_GEMM_BF16 256      /* this product only: operands rounded to bf16 (RNE) on their way into LDS, fp32 accumulation */
int dm_gemm_f32(int a_layout, int b_layout, int M, int N, int K,
                const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, const float* add, int ldadd, int flags,
                void* ws, size_t ws_bytes, void* stream);
/* The same product with operands STORED as bf16 (conf.amp's weight / activation twins: train.py:166 autocast casts them per
 * call, here they live in HBM in that format): fp32 accumulation, fp32 C and an optional bf16 twin C_h of it (same ldc).
 * Leading dimensions in elements; minor extents in multiples of 8, rows 16-byte aligned. */
int dm_gemm_bf16h(int a_layout, int b_layout, int M, int N, int K, const uint16_t* A, int lda, const uint16_t* B, int ldb,
                  float* C, int ldc, uint16_t* C_h, const float* bias, int flags, void* ws, size_t ws_bytes, void* stream);
/* Calls with DM_FLAG_BF16 keep bf16 twins of their activations / per-call weight copies and feed them to dm_gemm_bf16h's
 * kernel (activation arenas of such calls are larger: the *_acts_floats functions account for it; a backward entry point uses
 * the arena twins only if the forward call that filled that `acts` buffer wrote them - the library keeps a host-side note per
 * buffer).  1 / 0 switches that path on / off, -1 queries; returns the state.  Off = the fp32-storage products of dm_gemm_f32(DM_GEMM_BF16); results agree to fp32 summation order. */
int dm_bf16_twins_enable(int on);
/* fp32 products whose operands take the 16-byte load path run gemm_dma_kernel: global_load_lds (LDS-DMA) into a 2-3 stage LDS
 * ring, counted vmcnt across a raw s_barrier, inline-asm fragment reads (csrc/gemm.hip).  Same tiles, k order and epilogue as the
 * register-staged gemm_f32_kernel: bit-identical results, so the library picks per call: the LDS-DMA loop from 14 k-tiles per
 * work item up, the register-staged loop (higher residency) below.  1 / 0 switches it on / off, 2 = on for every k extent (the
 * bit-identity test), -1 queries; returns the state (default 1; DM_GEMM_DMA=0 / 2 in the environment). */
int dm_gemm_dma_enable(int on);
/* The image layer of the decoder (ConvTranspose2d(d -> 3, k6, s2), decoders.py:154-155) runs its backward - data gradient with
 * the ELU' of the layer below folded in, weight gradient - as two direct MFMA kernels that read the 3-channel output gradient
 * of a frame from LDS (csrc/conv_direct.hip) instead of as gather-form products through the generic tile (3 -> 4 channel pad, 48
 * columns in a 64-wide tile).  Same sums in another order (fp32 rounding only).  1 / 0 switches it on / off, -1 queries; returns
 * the state (default 1; DM_DEC_L4_BWD_GEMM=1 in the environment = off). */
int dm_dec_l4_bwd_direct_enable(int on);
/* The posterior T loop (rssm.py:38-58, cell rssm.py:125-153, nn.GRUCell rnn.py:40-67) runs, when the shape qualifies (plain
 * single-layer GRU, LayerNorm, categorical latents, B <= 64, the layer slices fit one CU's LDS), as ONE persistent kernel with
 * one workgroup per compute unit that keeps its 4-column slice of every layer's weights in LDS for all T steps and exchanges
 * the small activation rows through poison-filled per-step buffers (csrc/rssm_lds.hip) - instead of five
 * dependent launches per step that re-stream the weights.  Same arithmetic up to fp32 summation order, same sampler rule.
 * dm_rssm_lds_enable: 1 / 0 switches it on / off, 2 = on also for small models (slices under half a CU's LDS: tests), -1 queries;
 * returns the state (default 1; DM_RSSM_LDS=0 / 2 in the environment).
 * Co-residency is checked with the occupancy API before the first launch of a variant (>= 1 workgroup per CU at its LDS size; the grid
 * never exceeds the CU count) and the spin loops are bounded; DM_RSSM_LDS_COOP=1 launches cooperatively instead (measured slower: it drains
 * the other streams around the kernel).
 * dm_rssm_lds_status: non-zero once such a kernel has given up in a spin loop (that step's outputs are invalid; every later call
 * takes the launch chain).  dm_rssm_lds_status_ack: the same word, read AND cleared - the give-up is reported once
 * (Dreamer.check_device_status() / packed_metrics_host() raise on it) while the kernel stays switched off for the life of the
 * process (dm_rssm_lds_gave_up: the acknowledged status), so a trainer can restore a checkpoint and continue on the launch chain.
 * dm_rssm_lds_prof: 16 sums of clock ticks (100 MHz) of its workgroup 0, one per phase / sub-phase, since the last reset (diagnostic). */
int dm_rssm_lds_enable(int on);
int dm_bptt_fold_enable(int on);        /* launch schedule of the BPTT loop: the two LayerNorm+ELU backward stages of a step folded into the products
                                            that consume them, dx W = rstd (g W - mean(g) colsum(W) - mean(g xhat) xhat W), so those products start
                                            with their operand loads instead of a row reduction.  ON by default (DM_BPTT_FOLD=0); -1 queries. */
int dm_rssm_lds_status(void);
int dm_rssm_lds_status_ack(void);
int dm_rssm_lds_gave_up(void);
int dm_rssm_lds_prof(unsigned long long* out16, int reset);

/* y = ELU(LayerNorm(x; gamma, beta, eps)) row-wise; stats[r] = {mean, rstd}. (common.py:44-49, rssm.py:105-115) */
int dm_ln_elu_fwd(int rows, int n, const float* x, int ldx, const float* gamma, const float* beta, float eps,
                  float* y, int ldy, float* stats, void* stream);
/* dx from dy; dgamma/dbeta overwritten (column sums over rows). */
int dm_ln_elu_bwd(int rows, int n, const float* x, int ldx, const float* y, int ldy, const float* stats,
                  const float* gamma, const float* dy, int lddy, float* dx, int lddx,
                  float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);
/* out[c] = sum_r x[r*ld+c]  (bias gradients). */
int dm_colsum(int rows, int n, const float* x, int ld, float* out, void* ws, size_t ws_bytes, void* stream);

/* nn.GRUCell gate math (rnn.py:48-49; torch GRUCell: r,z,n row blocks).  gi,gh: (rows,3D) incl. biases. */
int dm_gru_gates_fwd(int rows, int D, const float* gi, const float* gh, const float* h_in, int ldh,
                     float* h_out, int ldo, void* stream);
/* dgi, dgh (rows,3D) overwritten; dh_in overwritten with the direct path dh_out*u (the gh path is added by the caller's GEMM). */
int dm_gru_gates_bwd(int rows, int D, const float* gi, const float* gh, const float* h_in, int ldh,
                     const float* dh_out, int lddh, float* dgi, float* dgh, float* dh_in, int lddi, void* stream);

/* OneHotCategorical(StraightThrough) sample (rssm.py:147-148,195-201; dreamer.py:198-200):
 * per group of C logits: p = softmax(logits); cdf = sequential fp32 cumsum(p);
 * idx = #{k : cdf_k <= u*cdf_{C-1}} clamped to C-1 (the inverse-CDF rule the oracle patches into torch.multinomial);
 * if forced_idx != NULL it is used instead of sampling.  onehot (rows, groups*C) with leading dim ldo.
 * C = 0 selects Gaussian latents (stoch_discrete = 0; rssm.py:202-203, functions.py:46-56 diag_normal): a `logits` row is
 * (mean[groups] | raw[groups]), std = 2 sigmoid(raw) + 0.1, `u` holds STANDARD-NORMAL draws and `onehot` receives
 * z = mean + std * u (Normal.rsample), groups wide; forced_idx must be NULL, idx (if given) is zero-filled. */
int dm_sample_onehot(int rows, int groups, int C, const float* logits, int ldl, const float* u,
                     const int32_t* forced_idx, float* onehot, int ldo, int32_t* idx, void* stream);

/* KL(post||prior), entropies (dreamer.py:326-343,369-379; torch/distributions/kl.py:248-252).  C = 0: rows of Gaussian
 * parameters (mean[S] | raw std[S]) and the Normal-Normal KL / entropies (torch/distributions/kl.py kl_normal_normal). */
int dm_kl_balance_fwd(int rows, int S, int C, const float* post, const float* prior,
                      float* kl, float* ent_post, float* ent_prior, void* stream);
/* dpost = scale_post * dKL/dpost, dprior = scale_prior * dKL/dprior (overwrite). */
int dm_kl_balance_bwd(int rows, int S, int C, const float* post, const float* prior,
                      float scale_post, float scale_prior, float* dpost, float* dprior, void* stream);
/* straight-through backward: dlogits (+)= softmax-jacobian(logits)^T dz per group (accumulate if accum). */
int dm_st_softmax_bwd(int rows, int groups, int C, const float* logits, int ldl, const float* dz, int lddz,
                      float* dlogits, int lddl, int accum, void* stream);

/* IWAE (iwae_samples = I > 1).  Row index n = (t*B + b)*I + i everywhere (rssm.py:35-41).
 * Sampled KL (dreamer.py:340-343): out[n] = log q(z_n) - log p(z_n), z given by its indices (rows,S). */
int dm_kl_sampled_fwd(int rows, int S, int C, const float* post, const float* prior, const int32_t* idx, float* out, void* stream);
/* dpost = scale*row_w[n]*(onehot - softmax(post)), dprior = -scale*row_w[n]*(onehot - softmax(prior)); row_w nullable. */
int dm_kl_sampled_bwd(int rows, int S, int C, const float* post, const float* prior, const int32_t* idx, float scale,
                      const float* row_w, float* dpost, float* dprior, void* stream);
/* The same for Gaussian latents (stoch_discrete = 0, rssm.py:202-203): post / prior rows are (mean | raw std), z (rows, S; row
 * stride ldz) is the reparameterised posterior sample, through which a gradient flows too: dz (row stride lddz) is
 * ACCUMULATED with scale*row_w[n]*d(log q - log p)/dz, dpost / dprior receive the explicit parameter gradients. */
int dm_kl_sampled_gauss_fwd(int rows, int S, const float* post, const float* prior, const float* z, int ldz, float* out,
                            void* stream);
int dm_kl_sampled_gauss_bwd(int rows, int S, const float* post, const float* prior, const float* z, int ldz, float scale,
                            const float* row_w, float* dpost, float* dprior, float* dz, int lddz, void* stream);
/* x (TB,I,W) -> out (TB,W): mode 0 mean over I, mode 2 sum, mode 1 (W = 1) -logavgexp(-x) (functions.py:97-102) with the
 * optional importance weights w_out (TB,I) = softmax_i(-x) = d out/d x_i. */
int dm_reduce_i(int TB, int I, int W, const float* x, int mode, float* out, float* w_out, void* stream);
/* out[r] = sum_j w[j]*x_j[r], j < count <= 8 (x: host array of device pointers, w: host floats): dreamer.py:362. */
int dm_combine_rows(int count, int64_t n, const float* const* x, const float* w, float* out, void* stream);
/* x[r, 0..n) *= w[r]*scale (ldx leading dim): per-sample importance weights on row gradients. */
int dm_scale_rows(int64_t rows, int n, float* x, int ldx, const float* w, float scale, void* stream);

/* rows of h and z multiplied by (1-reset[r]) (rssm.py:41,134-135). */
int dm_mask_rows(int rows, int n, const float* x, int ldx, const uint8_t* reset, float* y, int ldy, void* stream);

/* stride-2 "valid" conv patch gather / scatter used by ConvEncoder and ConvDecoder (encoders.py:80-96, decoders.py:144-161).
 * Small image (n, hs, ws) relates to big image (n, hb, wb) by y_big = 2*y_small + ky, ky in [0,k).
 * im2col: col[(n,ys,xs)][(ky,kx,c)] = big[n, 2ys+ky, 2xs+kx, c]   (big NHWC, or NCHW if big_nchw: col order (c,ky,kx))
 * col2im: big[n,y,x,c] = act( bias[c] + sum_{ky,kx valid} col[(n,ys,xs)][(ky,kx,c)] ) * (mul_elu_grad_of ? ELU'(ref) : 1) */
int dm_im2col_s2(int n, int hb, int wb, int c, int k, const float* big, int big_nchw, float* col, void* stream);
#define DM_C2I_ELU 1
int dm_col2im_s2(int n, int hb, int wb, int c, int k, const float* col, const float* bias, int flags,
                 const float* elu_ref, float* big, void* stream);

/* ---------------------------------------------------------------- fused operators -------------- */
/* MLP = [Linear, LayerNorm(eps 1e-3), ELU] x layers, Linear  (common.py:37-65). */
typedef struct dm_mlp_params {
  const float* w[DM_MAX_MLP_LAYERS + 1];   /* w[l]: (hidden,in_l) ; w[layers]: (out,hidden) */
  const float* b[DM_MAX_MLP_LAYERS + 1];
  const float* ln_g[DM_MAX_MLP_LAYERS];
  const float* ln_b[DM_MAX_MLP_LAYERS];
  int32_t precision;                       /* 0 = fp32; 1 = bf16 operands / fp32 accumulate for the calls given this struct */
  int32_t reserved_;
} dm_mlp_params;
typedef struct dm_mlp_grads {
  float* w[DM_MAX_MLP_LAYERS + 1];
  float* b[DM_MAX_MLP_LAYERS + 1];
  float* ln_g[DM_MAX_MLP_LAYERS];
  float* ln_b[DM_MAX_MLP_LAYERS];
} dm_mlp_grads;
size_t dm_mlp_acts_floats(int rows, int hidden, int layers);
/* scratch floats (incl. the split-K region) dm_mlp_head_fwd / dm_mlp_head_bwd need for `rows` rows */
size_t dm_mlp_ws_floats(int rows, int hidden, int layers);
/* acts may be NULL when no backward will follow (critic_target a2c.py:39,84; the dream's reward / terminal heads
 * dreamer.py:205-206; inference dreamer.py:92-111): activations then ping-pong through `ws` and nothing is saved.
 * For rows >= 16384 and hidden 400 each Linear -> LayerNorm -> ELU is ONE row-panel launch (csrc/panel.hip) and the output
 * layer (out_dim <= 32) rides in the last one's epilogue. */
int dm_mlp_head_fwd(int rows, int in_dim, int hidden, int layers, int out_dim,
                    const float* x, int ldx, const dm_mlp_params* p, float* acts, float* out,
                    void* ws, size_t ws_bytes, void* stream);
/* The same head when the LAST sparse_cols columns of x are known to be mostly zero - the one-hot latent part of a feature
 * row (rssm.py:83-84: feature = cat(h, z), z = 32 one-hot groups of 32).  Same result (x W^T = x_dense W_d^T + sum over the
 * non-zero e of x_e W^T[e], exact for ANY x); on the row-panel path layer 0 then multiplies only the dense columns and
 * adds the sparse ones as a sum of weight rows (63 % of that layer's flops for the 600 + 1024 feature).  acts / out as above,
 * dm_mlp_head_bwd unchanged. */
int dm_mlp_head_fwd_sparse(int rows, int in_dim, int sparse_cols, int hidden, int layers, int out_dim,
                           const float* x, int ldx, const dm_mlp_params* p, float* acts, float* out,
                           void* ws, size_t ws_bytes, void* stream);
/* Rows [row0, row0 + rows) of dm_mlp_head_fwd(_sparse) over rows_total rows; x, out and acts are the FULL arrays (acts
 * carved for rows_total rows, or NULL).  A row's result does not depend on the window, so a forward over 40 000 imagined
 * states can be issued in pieces as the rollout produces them (host mirror: the heads of the first horizon steps run on an
 * idle stream while the rollout finishes; dreamer.py:212-213, a2c.py:85,113 apply the heads to the stacked features). */
int dm_mlp_head_fwd_rows(int rows_total, int row0, int rows, int in_dim, int sparse_cols, int hidden, int layers, int out_dim,
                         const float* x, int ldx, const dm_mlp_params* p, float* acts, float* out,
                         void* ws, size_t ws_bytes, void* stream);

/* dout (rows,out_dim); grads overwritten; dx (rows,in_dim; ld lddx) written if non-null (accumulated if dx_accum). */
int dm_mlp_head_bwd(int rows, int in_dim, int hidden, int layers, int out_dim,
                    const float* x, int ldx, const dm_mlp_params* p, const float* acts, const float* dout,
                    const dm_mlp_grads* g, float* dx, int lddx, int dx_accum,
                    void* ws, size_t ws_bytes, void* stream);

/* loss epilogues of the dense heads: kind 0 = reward, 0.5*(mu-y)^2 + c (decoders.py:302-304); 1 = terminal,
 * BCE-with-logits (decoders.py:263-269).  loss[r]; dout[r] = scale * dloss/dout; mean_out[r] = mu or sigmoid(logit). */
int dm_head_loss(int kind, int rows, const float* out, const float* target, float scale, float loss_const,
                 float* loss, float* dout, float* mean_out, void* stream);

/* uint8 ingest (preprocessing.py:21-29 to_image): src (n, h*w, c) uint8 HWC -> dst (n, c, h*w) float32,
 * x/255 - 0.5.  Lets the trainer hand the replay's native uint8 frames to training_step(). */
int dm_preprocess_image_u8(int64_t n, int hw, int c, const uint8_t* src, float* dst, void* stream);

/* ConvEncoder (encoders.py:72-96): 4 x (Conv2d k4 s2 + ELU), Flatten.  w[i]: (Cout,Cin,4,4) torch layout. */
typedef struct dm_conv_params { const float* w[5]; const float* b[5]; } dm_conv_params;
typedef struct dm_conv_grads { float* w[5]; float* b[5]; } dm_conv_grads;
size_t dm_conv_encoder_acts_floats(const dm_shape* shp);
int dm_conv_encoder_fwd(const dm_shape* shp, const float* image /* (N,ch,64,64) */, const dm_conv_params* p,
                        float* acts, float* embed /* (N,E) torch (c,y,x) order */, void* ws, size_t ws_bytes, void* stream);
/* Frames [n0, n0+n) only (all buffers are the full-batch ones; every per-layer buffer is frame-major).  prepare != 0 also
 * builds what all ranges share (gather tables, repacked weights); n = 0 prepares only.  Lets the host pipeline time chunks
 * of encoder -> posterior loop -> decoder over three streams (the T-step loop is a latency chain that leaves most CUs idle). */
int dm_conv_encoder_fwd_rows(const dm_shape* shp, int n0, int n, int prepare, const float* image, const dm_conv_params* p,
                             float* acts, float* embed, void* ws, size_t ws_bytes, void* stream);
int dm_conv_encoder_bwd(const dm_shape* shp, const float* image, const dm_conv_params* p, const float* acts,
                        const float* dembed, const dm_conv_grads* g, void* ws, size_t ws_bytes, void* stream);

/* ConvDecoder + MSE (decoders.py:111-180): Linear F->32d, 4 x ConvTranspose2d (k 5,5,6,6; s2), ELU x3.
 * w[0],b[0] = Linear; w[1..4]: (Cin,Cout,k,k) torch layout.  loss_image[n] = 0.5*sum (pred-target)^2. */
size_t dm_conv_decoder_acts_floats(const dm_shape* shp);
/* float offset of the prediction (N, img, img, ch) NHWC inside `acts`: image_rec (decoders.py:177) is materialised by the
 * caller only when somebody reads it (pass image_rec = NULL to the forward) */
size_t dm_conv_decoder_pred_offset(const dm_shape* shp);
int dm_conv_decoder_mse_fwd(const dm_shape* shp, const float* feat, int ldf, const float* target,
                            const dm_conv_params* p, float* acts, float* loss_image, float* image_rec /* nullable, NCHW */,
                            void* ws, size_t ws_bytes, void* stream);
/* Frame-range form, see dm_conv_encoder_fwd_rows; the workspace need scales with n. */
int dm_conv_decoder_mse_fwd_rows(const dm_shape* shp, int n0, int n, int prepare, const float* feat, int ldf,
                                 const float* target, const dm_conv_params* p, float* acts, float* loss_image,
                                 float* image_rec, void* ws, size_t ws_bytes, void* stream);
/* dfeat (N,F) accumulated (+=) ; scale = image_weight / (T*B).  _rows: an optional per-frame factor on top (IWAE weights). */
int dm_conv_decoder_mse_bwd_rows(const dm_shape* shp, const float* feat, int ldf, const float* target,
                                 const dm_conv_params* p, const float* acts, float scale, const float* row_scale,
                                 const dm_conv_grads* g, float* dfeat, int lddf, void* ws, size_t ws_bytes, void* stream);
int dm_conv_decoder_mse_bwd(const dm_shape* shp, const float* feat, int ldf, const float* target,
                            const dm_conv_params* p, const float* acts, float scale,
                            const dm_conv_grads* g, float* dfeat, int lddf, void* ws, size_t ws_bytes, void* stream);

/* RSSM posterior sequence (rssm.py:21-78,125-153,186-193; rnn.py:40-67). Parameter order of dm_rssm_params.p[]: */
enum {
  DM_RSSM_Z_W = 0, DM_RSSM_Z_B, DM_RSSM_A_W, DM_RSSM_IN_G, DM_RSSM_IN_B,
  DM_RSSM_GRU_WIH, DM_RSSM_GRU_WHH, DM_RSSM_GRU_BIH, DM_RSSM_GRU_BHH,
  DM_RSSM_PRIOR_H_W, DM_RSSM_PRIOR_H_B, DM_RSSM_PRIOR_G, DM_RSSM_PRIOR_B, DM_RSSM_PRIOR_W, DM_RSSM_PRIOR_OB,
  DM_RSSM_POST_H_W, DM_RSSM_POST_H_B, DM_RSSM_POST_E_W, DM_RSSM_POST_G, DM_RSSM_POST_B, DM_RSSM_POST_W, DM_RSSM_POST_OB,
  /* LayerNorm GRU cells only (NULL for gru; then GRU_BIH / GRU_BHH are NULL instead - these cells have no biases):
   * gru_layernorm: (G0,B0) = ln_reset, (G1,B1) = ln_update, (G2,B2) = ln_newval, D floats each;
   * gru_layernorm_dv2: (G0,B0) = lnorm, 3D floats */
  DM_RSSM_GRU_LN_G0, DM_RSSM_GRU_LN_B0, DM_RSSM_GRU_LN_G1, DM_RSSM_GRU_LN_B1, DM_RSSM_GRU_LN_G2, DM_RSSM_GRU_LN_B2,
  /* GRUCellStack layers 1..3 (rnn.py:40-67, gru_layers > 1, gru_type = gru): weight_ih, weight_hh, bias_ih, bias_hh each;
   * layer 0 is the DM_RSSM_GRU_* group above.  Layer i maps (i == 0 ? hidden : D/L) -> D/L and owns state columns
   * [i*D/L, (i+1)*D/L). */
  DM_RSSM_GRU_L1_WIH, DM_RSSM_GRU_L1_WHH, DM_RSSM_GRU_L1_BIH, DM_RSSM_GRU_L1_BHH,
  DM_RSSM_GRU_L2_WIH, DM_RSSM_GRU_L2_WHH, DM_RSSM_GRU_L2_BIH, DM_RSSM_GRU_L2_BHH,
  DM_RSSM_GRU_L3_WIH, DM_RSSM_GRU_L3_WHH, DM_RSSM_GRU_L3_BIH, DM_RSSM_GRU_L3_BHH,
  /* ... and, for a stack of LayerNorm cells (gru_layers > 1 with gru_layernorm / gru_layernorm_dv2; ABI v6), the LayerNorm
   * parameters of layers 1..3 in the (G0,B0,G1,B1,G2,B2) order of DM_RSSM_GRU_LN_* above, D/L (dv2: 3D/L) floats each. */
  DM_RSSM_GRU_L1_LN_G0, DM_RSSM_GRU_L1_LN_B0, DM_RSSM_GRU_L1_LN_G1, DM_RSSM_GRU_L1_LN_B1, DM_RSSM_GRU_L1_LN_G2, DM_RSSM_GRU_L1_LN_B2,
  DM_RSSM_GRU_L2_LN_G0, DM_RSSM_GRU_L2_LN_B0, DM_RSSM_GRU_L2_LN_G1, DM_RSSM_GRU_L2_LN_B1, DM_RSSM_GRU_L2_LN_G2, DM_RSSM_GRU_L2_LN_B2,
  DM_RSSM_GRU_L3_LN_G0, DM_RSSM_GRU_L3_LN_B0, DM_RSSM_GRU_L3_LN_G1, DM_RSSM_GRU_L3_LN_B1, DM_RSSM_GRU_L3_LN_G2, DM_RSSM_GRU_L3_LN_B2,
  DM_RSSM_NPARAMS
};
typedef struct dm_rssm_params { const float* p[DM_RSSM_NPARAMS]; } dm_rssm_params;
typedef struct dm_rssm_grads { float* p[DM_RSSM_NPARAMS]; } dm_rssm_grads;
size_t dm_rssm_acts_floats(const dm_shape* shp);
/* feat (N,F): [h | z]; post, prior (N,Z) logits; idx (N,S).  u (N,S) uniforms; forced_idx nullable. */
int dm_rssm_sequence_fwd(const dm_shape* shp, const float* embed, const float* action, const uint8_t* reset,
                         const float* h0, const float* z0, const float* u, const int32_t* forced_idx,
                         const dm_rssm_params* p, float* acts, float* feat, float* post, float* prior, int32_t* idx,
                         void* ws, size_t ws_bytes, void* stream);
/* Time steps [t0, t1) only: step t0 > 0 continues from the state step t0-1 left in `feat`; consecutive ranges issued in
 * order on one stream equal one full call (rssm.py:38-58 is a plain loop over t). */
int dm_rssm_sequence_fwd_steps(const dm_shape* shp, int t0, int t1, const float* embed, const float* action,
                               const uint8_t* reset, const float* h0, const float* z0, const float* u,
                               const int32_t* forced_idx, const dm_rssm_params* p, float* acts, float* feat, float* post,
                               float* prior, int32_t* idx, void* ws, size_t ws_bytes, void* stream);
/* dfeat (N,F) from decoders/heads (consumed, overwritten as scratch), dpost/dprior (N,Z) from the KL term.
 * Produces parameter grads and dembed (N,E). */
int dm_rssm_sequence_bwd(const dm_shape* shp, const float* embed, const float* action, const uint8_t* reset,
                         const dm_rssm_params* p, const float* acts, const float* feat, const float* post,
                         float* dfeat, float* dpost, float* dprior,
                         const dm_rssm_grads* g, float* dembed, void* ws, size_t ws_bytes, void* stream);

/* Progress marks for the NEXT dm_dream_rollout call of the calling thread (n <= 4; cleared by that call): events[i] - a
 * hipEvent_t owned by the caller - is recorded on the rollout's stream when horizon step steps[i] (0-based) has been
 * enqueued, i.e. when feature rows [0, (steps[i] + 2) * M) are final; marks that cannot be placed inside the loop (step out
 * of range, chain-graph replay) are recorded behind the call.  Lets the caller start per-row work on the first horizon
 * steps on another stream while the rollout continues.  No reference counterpart. */
int dm_dream_rollout_marks(int n, const int* steps, void* const* events);

/* Imagination rollout (dreamer.py:188-216, rssm.py:155-184, a2c.py:43-55), no autograd graph (actor_grad=reinforce).
 * start (M,F) = [h|z] rows; feats (H+1,M,F); actions (H,M,A) one-hot (or continuous); act_idx (H,M) (onehot only);
 * u_act: (H,M) uniforms for the one-hot actor, or (H,M,A) standard-normal noise for continuous actors (shp->flags);
 * u_prior (H,M,S).
 * actor_acts (dm_mlp_acts_floats(H*M, hidden, layers) floats) + actor_logits (H*M, A): optional (both or neither);
 * when given, the actor activations of all H steps are kept for dm_mlp_head_bwd (rows = H*M). */
int dm_dream_rollout(const dm_shape* shp, int M, const float* start, const dm_rssm_params* cell,
                     const dm_mlp_params* actor, const float* u_act, const float* u_prior,
                     float* feats, float* actions, int32_t* act_idx, float* actor_acts, float* actor_logits,
                     void* ws, size_t ws_bytes, void* stream);

/* GAE + reality weight (a2c.py:81-108). All (J,M)/(H,M) row-major with J=H+1. */
int dm_gae_losses(int H, int M, float gamma, float lambda, const float* reward, const float* terminal,
                  const float* value_t, float* advantage, float* advantage_gae, float* value_target, float* weight,
                  void* stream);
/* actor (reinforce, onehot) loss rows (a2c.py:119-130): loss[r] = (-logpi(a)*adv - ent_w*H[pi])*w;
 * dlogits = scale * dloss/dlogits. */
int dm_actor_loss(int rows, int A, const float* logits, const int32_t* act_idx, const float* adv_gae,
                  const float* weight, float ent_w, float scale, float* loss, float* entropy, float* dlogits, void* stream);
/* Continuous actors (functions.py:59-78; a2c.py:43-55,119-130): kind 1 = tanh_normal, kind 2 = normal_tanh.
 * params (rows,2A) = [mean_raw | std_raw]; eps (rows,A) standard-normal noise (torch.normal restated);
 * action = tanh(mean + std*eps) (kind 1) or mean + std*eps (kind 2). */
int dm_sample_continuous(int kind, int rows, int A, const float* params, const float* eps, float* action, void* stream);
/* reinforce rows: loss[r] = (-log pi(a)*adv - ent_w*H)*w with TransformedDistribution(Normal, Tanh) log-prob (kind 1) and the
 * base Normal's entropy; dparams (rows,2A) = scale * dloss/dparams. */
int dm_actor_loss_continuous(int kind, int rows, int A, const float* params, const float* actions, const float* adv_gae,
                             const float* weight, float ent_w, float scale, float* loss, float* entropy, float* dparams,
                             void* stream);
/* critic loss rows (a2c.py:112-115): loss[r] = 0.5*(vt-v)^2*w ; dvalue = scale * -(vt-v)*w. */
int dm_critic_loss(int rows, const float* value, const float* value_target, const float* weight, float scale,
                   float* loss, float* dvalue, void* stream);

/* out[i] = scale[i] * sum(x_i[0..n_i)) for up to 32 arrays in one launch (losses / metrics, dreamer.py:362-379, a2c.py:133-147). */
typedef struct dm_reduce_item {
  const float* x; int64_t n; float scale;
  int32_t mode;            /* 0: sum x ; 1: sum (x - *center)^2 ; 2: sqrt(scale * sum (x - *center)^2)  (reward1.std(), a2c.py:140) */
  const float* center;     /* device scalar, mode 1 only */
} dm_reduce_item;
/* `out` is caller-provided: a step's losses and metrics are written side by side into ONE device buffer (Dreamer.metric_buffer),
 * so the trainer's ~20 per-metric .item() syncs (train.py:204-214) become one copy. */
int dm_multi_sum(int count, const dm_reduce_item* items /* host array */, float* out, void* stream);
/* out[0] = sum_i w[i] * x[i]  (x device, w host; count <= 16): loss_model from its weighted terms (dreamer.py:362-365). */
int dm_combine(int count, const float* x, const float* w, float* out, void* stream);

/* Optimizer (dreamer.py:60-87; torch.optim.AdamW defaults, clip_grad_norm_). */
/* norm_out[0] = ||g||_2 ; norm_out[1] = min(1, max_norm/(norm+1e-6)). */
int dm_multi_tensor_norm_clip(const float* grad, int64_t n, float max_norm, float* norm_out,
                              void* ws, size_t ws_bytes, void* stream);
/* x *= coef[0] (device scalar): the in-place gradient scaling of clip_grad_norm_. */
int dm_scale_inplace(float* x, int64_t n, const float* coef, void* stream);
/* AdamW step with the clip coefficient read from device memory (clip_coef may be NULL = 1). */
int dm_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  const float* clip_coef, void* stream);
int dm_copy_params(float* dst, const float* src, int64_t n, void* stream);   /* critic_target <- critic, a2c.py:151-152 */
/* Row threshold from which the 400-wide MLP heads run their whole forward as ONE launch (csrc/mlp_chain.hip; default 256,
 * below it the per-layer launches are faster).  rows >= 1 sets it; returns the previous value (rows < 1: query only). */
int dm_mlp_chain_min_rows(int rows);
/* The imagination rollout's one-hot action draw (dreamer.py:198-200) rides in the output stage of the whole-MLP actor kernel
 * (1, default) or runs as its own sampler launch (0); same rule and operation order: bit-identical draws.  -1 queries. */
int dm_rollout_fuse_act_enable(int on);
/* Optional per-launch timing of the GEMM kernel with HIP events on the launch stream (bench.py's roofline line).
 * dm_prof_begin arms up to max_launches slots; dm_prof_end synchronises on the events, fills
 * out[kind*4+{0,1,2,3}] = {launches, algorithmic flops (2MNK), milliseconds, algorithmic bytes 4(MK+NK+MN)} for
 * kind = tile*4 + a_layout*2 + b_layout with tile 0..4 = 128x128 / 128x64 / 64x64 / 128x96 / 96x128, kind 20 = row-panel
 * Linear+LayerNorm+ELU forward, 21 = row-panel backward, 22 = whole-MLP forward chain, 24 + kind = the same tile / layouts on the LDS-DMA
 * loop gemm_dma_kernel (44 kinds; nkinds >= 44; `out` holds 4*nkinds doubles) and returns the number of launches recorded.  (The <= 64-row skinny products are not in this set.) */
int dm_prof_begin(int max_launches);
int dm_prof_end(double* out, int nkinds);
/* Per-launch rows of the armed region (call before dm_prof_end): rows[i*8 + {0..7}] = {kind, M, N, K, split count, flags
 * (1 gathered A, 2 gathered B, 4 scatter epilogue, 8 bf16 operands, 32 bf16-storage operands), flops, milliseconds};
 * returns the row count. */
int dm_prof_rows(double* rows, int max_rows);
/* y = a*x + b*y */
int dm_axpby(int64_t n, float a, const float* x, float b, float* y, void* stream);

/* ---- the data-parallel exchange step, native (csrc/comm.hip; SURVEY 8(b) dm_allreduce_grads, 8(e)) ------------------------
 * The reference is single-process (no counterpart); pydreamer_amd/dist.py shards the batch axis and SUM-all-reduces each
 * optimizer group's flat fp32 gradient buffer before grad_clip (dreamer.py:73-87 then sees the global-batch gradient).
 * RCCL is bound with dlopen at first use (no link-time dependency; the instance torch already holds is preferred).
 *   dm_rccl_available()                1 when librccl could be bound, else 0 (never fails)
 *   dm_rccl_version()                  ncclGetVersion's code, 0 when unavailable
 *   dm_rccl_unique_id(id128)           rank 0: the 128-byte id of a new communicator (the host carries it to the other ranks)
 *   dm_rccl_comm_init(&comm, n, id, r) collective over the n ranks; the communicator is bound to the CURRENT device
 *   dm_allreduce_grads(buf, n, comm, stream)   buf[0..n) <- sum over ranks, in place, fp32, enqueued on `stream`; never
 *                                      synchronises.  One communicator per optimizer group: a group's collective is ordered by
 *                                      its stream alone (right behind the backward pass that filled the buffer).
 *   dm_rccl_comm_destroy(comm) */
int dm_rccl_available(void);
int dm_rccl_version(void);
int dm_rccl_unique_id(void* id128);
int dm_rccl_comm_init(void** comm, int nranks, const void* id128, int rank);
int dm_rccl_comm_destroy(void* comm);
int dm_allreduce_grads(void* buf, size_t n, void* comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DREAMER_HIP_H */
