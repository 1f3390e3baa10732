// Library-level entry points: version, error string, device check, workspace sizing.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <mutex>

static thread_local char g_err[512] = "";

int dm_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" int dm_version(void) { return 13; }
extern "C" const char* dm_last_error(void) { return g_err; }

extern "C" int dm_device_check(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return dm_fail(DM_E_DEVICE, "hipGetDevice failed (no HIP device visible)");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return dm_fail(DM_E_DEVICE, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return dm_fail(DM_E_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
  return DM_OK;
}

// A HIP stream restricted to a subset of the compute units (hipExtStreamCreateWithCUMask): `mask` has one bit per CU, 32
// per word.  Used by WorldModel's time-chunk pipeline to RESERVE a few CUs of every XCD for the posterior loop's latency
// chain, so its small kernels never queue behind the convolution GEMMs of the neighbouring streams.
extern "C" int dm_stream_create_cu_mask(const uint32_t* mask, int words, void** stream) {
  DM_REQUIRE(mask && stream && words >= 1, DM_E_NULL, "stream_create_cu_mask: null argument");
  hipStream_t s = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
  if (e != hipSuccess) return dm_fail(DM_E_HIP, "hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
  *stream = (void*)s;
  return DM_OK;
}
extern "C" int dm_stream_destroy(void* stream) {
  if (!stream) return DM_OK;
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) return dm_fail(DM_E_HIP, "hipStreamDestroy: %s", hipGetErrorString(e));
  return DM_OK;
}

// ---- weight-gradient side stream (see include/dreamer_hip.h) ---------------------------------------------------------
// One stream + two events per process (one process drives one GPU); the armed flag is per host thread, because the
// world-model backward is enqueued by one thread while others enqueue the rollout / actor-critic passes.
static std::mutex g_side_mu;
static hipStream_t g_side_stream = nullptr;
static hipEvent_t g_side_fork_ev[8] = {nullptr}, g_side_done_ev = nullptr;      // fork events rotate: one per fork point in flight
static unsigned g_side_fork_i = 0;
static int g_side_dev = -1;
static bool g_side_pending = false;
static thread_local bool tl_side_armed = false;

static int side_init_locked() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return dm_fail(DM_E_DEVICE, "wgrad_side: hipGetDevice failed");
  if (g_side_stream && dev == g_side_dev) return DM_OK;
  if (g_side_stream) return dm_fail(DM_E_DEVICE, "wgrad_side: created on device %d, called on device %d", g_side_dev, dev);
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  const char* pe = getenv("DM_WGRAD_SIDE_PRIO");
  const int prio = pe ? atoi(pe) : least;               // lowest priority: the chains' small kernels dispatch first
  // DM_WGRAD_SIDE_RESERVE_CUS=k (experiment): the side stream without the first k CUs of every 32 (hipExtStreamCreateWithCUMask),
  // so the deferred weight-gradient products never occupy the CUs the BPTT chain's small kernels are dispatched to
  const char* re = getenv("DM_WGRAD_SIDE_RESERVE_CUS");
  const int res = re ? atoi(re) : 0;
  hipError_t e;
  if (res > 0 && res < 32) {
    uint32_t mask[8];
    for (int i = 0; i < 8; ++i) mask[i] = 0xFFFFFFFFu ^ ((1u << res) - 1u);
    e = hipExtStreamCreateWithCUMask(&g_side_stream, 8, mask);
  } else {
    e = hipStreamCreateWithPriority(&g_side_stream, hipStreamNonBlocking, prio);
  }
  for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&g_side_fork_ev[i], hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&g_side_done_ev, hipEventDisableTiming);
  if (e != hipSuccess) return dm_fail(DM_E_HIP, "wgrad_side: %s", hipGetErrorString(e));
  g_side_dev = dev;
  return DM_OK;
}

extern "C" int dm_wgrad_side_arm(int on) {
  if (on) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    DM_TRY(side_init_locked());
  }
  tl_side_armed = on != 0;
  return DM_OK;
}

// Creates the side stream NOW and gives it one command, so that it holds its hardware queue before anything else (a communicator's
// internal streams, torch's communication stream) asks for one: round 6 measured that the order in which streams first get work
// decides which of them end up sharing a hardware queue (profiles/r06_force_dp.txt, runs C / G / H).
extern "C" int dm_wgrad_side_touch(void) {
  std::lock_guard<std::mutex> lk(g_side_mu);
  DM_TRY(side_init_locked());
  static float* pad = nullptr;
  if (!pad && hipMalloc(reinterpret_cast<void**>(&pad), 256) != hipSuccess) return dm_fail(DM_E_HIP, "wgrad_side_touch: hipMalloc failed");
  if (hipMemsetAsync(pad, 0, 256, g_side_stream) != hipSuccess) return dm_fail(DM_E_HIP, "wgrad_side_touch: memset failed");
  return DM_OK;
}

hipStream_t dm_wgrad_side_stream(hipStream_t st) { return tl_side_armed && g_side_stream ? g_side_stream : st; }

int dm_wgrad_side_fork(hipStream_t st, hipStream_t sw) {
  if (sw == st) return DM_OK;
  std::lock_guard<std::mutex> lk(g_side_mu);
  hipEvent_t ev = g_side_fork_ev[g_side_fork_i++ & 7];
  hipError_t e = hipEventRecord(ev, st);
  if (e == hipSuccess) e = hipStreamWaitEvent(sw, ev, 0);
  if (e != hipSuccess) return dm_fail(DM_E_HIP, "wgrad_side fork: %s", hipGetErrorString(e));
  return DM_OK;
}

int dm_wgrad_side_mark(hipStream_t sw, hipStream_t st) {
  if (sw == st) return DM_OK;
  std::lock_guard<std::mutex> lk(g_side_mu);
  hipError_t e = hipEventRecord(g_side_done_ev, sw);
  if (e != hipSuccess) return dm_fail(DM_E_HIP, "wgrad_side mark: %s", hipGetErrorString(e));
  g_side_pending = true;
  return DM_OK;
}

extern "C" int dm_wgrad_side_join(void* stream) {
  tl_side_armed = false;
  std::lock_guard<std::mutex> lk(g_side_mu);
  if (!g_side_pending) return DM_OK;
  hipError_t e = hipStreamWaitEvent((hipStream_t)stream, g_side_done_ev, 0);
  if (e != hipSuccess) return dm_fail(DM_E_HIP, "wgrad_side join: %s", hipGetErrorString(e));
  g_side_pending = false;
  return DM_OK;
}

// Scratch sizing: the largest transient of any fused operator at this shape, mirroring the arena carves in
// conv.hip / rssm.hip / mlp.hip (every carve is rounded up to 64 floats).  See DESIGN.md "HBM layout".
static size_t pad64(size_t n) { return (n + 63) / 64 * 64; }

extern "C" size_t dm_workspace_bytes(const dm_shape* s) {
  if (!s) return 0;
  const size_t N = (size_t)s->T * s->B * (s->I > 0 ? s->I : 1);
  const size_t d = (size_t)s->cnn_depth, ch = (size_t)s->img_ch;
  const size_t D = s->D, Hd = s->Hd, Z = (size_t)s->S * (s->C ? s->C : 2), A = s->A;    // C = 0 (Gaussian latents): parameters are 2S wide, z is S
  const size_t Hm = s->mlp_hidden, L = s->mlp_layers, H = s->H > 0 ? s->H : 1;
  const size_t SK = DM_SPLITK_FLOATS;
  // encoder backward: ga (N*31*31*d) + gb (N*14*14*2d) + dwr (8d*64d) + dxcol (max patch matrix, l>=1)
  size_t xc = N * 196 * 16 * d;
  if (N * 36 * 32 * d > xc) xc = N * 36 * 32 * d;
  if (N * 4 * 64 * d > xc) xc = N * 4 * 64 * d;
  // + gather-form data gradient: zero-padded copies of the 14x14x2d / 6x6x4d gradients, tables, class-concatenated weights
  size_t epad = N * 16 * 16 * 2 * d;
  if (N * 8 * 8 * 4 * d > epad) epad = N * 8 * 8 * 4 * d;
  // bf16 mode (DM_FLAG_BF16): + bf16 twins of gradient ping-pong buffers and per-call weight copies (common.h DmTwinScope)
  const size_t tw = (s->flags & DM_FLAG_BF16) ? 1 : 0;
  const size_t enc_bwd = SK + pad64(N * 961 * d) + pad64(N * 196 * 2 * d) + pad64(8 * d * 64 * d) + pad64(xc) + pad64(epad) +
                         3 * pad64(N * 225) + pad64(4 * 1024) + pad64(16 * 2 * d * 4 * d) + 1024 +
                         tw * (pad64(N * 961 * d / 2 + 1) + pad64(N * 196 * d + 1) + pad64(8 * 2 * d * 4 * d + 1));
  // decoder: column matrices rows_small * k*k*cout for layers 1..4
  size_t col = N * 25 * 4 * d;
  if (N * 25 * 25 * 2 * d > col) col = N * 25 * 25 * 2 * d;
  if (N * 169 * 36 * d > col) col = N * 169 * 36 * d;
  if (!dm_dec_l4_direct_ok((int)ch, (int)d, 30, 6) && N * 900 * 36 * ch > col) col = N * 900 * 36 * ch;   // layer 4 runs as a direct kernel otherwise
  size_t gmax = N * 25 * 4 * d;
  if (N * 169 * 2 * d > gmax) gmax = N * 169 * 2 * d;
  if (N * 900 * d > gmax) gmax = N * 900 * d;
  if (N * 4096 * ((ch + 3) / 4 * 4) > gmax) gmax = N * 4096 * ((ch + 3) / 4 * 4);   // image-layer gradient padded to 4 channels
  if (N * 32 * d > gmax) gmax = N * 32 * d;
  size_t wmax = 32 * d * 25 * 4 * d;
  if (4 * d * 25 * 2 * d > wmax) wmax = 4 * d * 25 * 2 * d;
  if (2 * d * 36 * d > wmax) wmax = 2 * d * 36 * d;
  if (d * 36 * ch > wmax) wmax = d * 36 * ch;
  // + gather-form transposed convolution (conv.hip): zero-padded input copy, tables, class-concatenated weights; bounded by the
  //   k = 6 layers: inputs 13x13x2d -> 17x17, 30x30xd -> 34x34, class pixels 15x15 / 32x32
  size_t gpad = N * 17 * 17 * 2 * d;
  if (N * 34 * 34 * d > gpad) gpad = N * 34 * 34 * d;
  const size_t gtab = N * 32 * 32;
  const size_t dec_fwd = SK + pad64(col) + pad64(gpad) + 3 * pad64(gtab) + pad64(9 * 1024) + pad64(4 * 2 * d * 9 * 2 * d) +
                         pad64(144 * d) + 1024 +      // + the direct layer-4 kernel's class-ordered weights
                         tw * pad64(2 * 2 * d * 9 * 2 * d + 1);
  // decoder backward: one gradient buffer per layer, one gather-table pair per layer, two split-K scratches (the weight
  // gradients may run on the side stream, conv.hip conv_decoder_mse_bwd_impl)
  const size_t r4 = (size_t)((ch + 3) / 4 * 4), q4 = 4;
  auto up4 = [&](size_t v) { return (v + q4 - 1) / q4 * q4; };
  const size_t g_l[5] = {N * 32 * d, N * 25 * up4(4 * d), N * 169 * up4(2 * d), N * 900 * up4(d), N * 4096 * r4};
  size_t g_sum = 0, g_tw = 0;
  for (int l = 0; l < 5; ++l) { g_sum += pad64(g_l[l]); g_tw += pad64(g_l[l] / 2 + 1); }
  const size_t dec_tabs = pad64(N) + pad64(N * 25) + pad64(N * 169) + pad64(N * 900) + pad64(25 * up4(4 * d)) + pad64(25 * up4(2 * d)) +
                          pad64(36 * up4(d)) + pad64(36 * r4);
  (void)gmax;
  const size_t l4_direct = dm_dec_l4_direct_ok((int)ch, (int)d, 30, 6)       // (sized whether or not the switch is on)
                               ? pad64(dm_dec_l4_wp_floats((int)d)) + pad64(dm_dec_l4_wgrad_part_floats((int)N, (int)d)) : 0;
  const size_t dec_bwd = 2 * SK + g_sum + 2 * pad64(wmax + 36 * 4 * d) + dec_tabs + 1024 + l4_direct +
                         tw * (g_tw + pad64((wmax + 36 * 4 * d) / 2 + 1));
  const size_t rssm_bwd = 2 * SK + 8 * pad64(N * Hd) + 3 * pad64(N * 3 * D) + 2 * pad64(Z * Hd) + pad64(Hd * D) +
                          pad64(3 * D * Hd) + pad64(3 * D * D) + 2 * pad64(3 * D) +   // + the transposed BPTT weights, LN-GRU dg
                          2 * pad64((3 * D + 15) / 16 * 1024) + 2 * pad64((Hd + 15) / 16 * 1024);   // + fragment-major dgi / dgh / dpin / dza
  // persistent posterior chain kernel (rssm_lds.hip): per-step exchange buffers (0 if the shape does not qualify)
  const size_t Zw = (size_t)s->S * (s->C ? s->C : 1);
  const size_t lds_fwd = SK + pad64(Zw * Hd) + 5 * pad64((Zw + 15) / 16 * 1024) +
                         (s->C ? pad64(dm_rssm_lds_ws_floats(s->B, s->D, s->Hd, s->S, s->C, s->T)) : 0) + 1024;
  const size_t rssm_bwd_lds = rssm_bwd + pad64(N * D) + pad64(N * Zw) + pad64(D) + pad64(Zw) + 2 * pad64((Hd + 15) / 16 * 128);      // (+ the folded launch schedule's x W products, column sums and strip sums)
  const size_t rows = (H + 1) * N;
  const size_t mlp_bwd = dm_mlp_ws_floats((int)rows, (int)Hm, (int)L);      // = SK + ping-pong + panel column partials
  const size_t dream = SK + L * (2 * pad64(N * Hm) + pad64(N * 2)) + pad64(N * 2 * A) + 3 * pad64(N * Hd) + pad64(N * 2) +
                       3 * pad64(N * 3 * D) + pad64(N * 6 * 4) + pad64(N * Z) +      // (LayerNorm-GRU statistics: 6 per stack layer)
                       pad64(25 * 512 * (((D + Z + 31) / 32) + (L > 0 ? L - 1 : 0) * ((Hm + 31) / 32))) +   // + fragment-major actor weights
                       pad64(Z * Hd) + pad64(A * Hd) + pad64(N * (size_t)s->S) + pad64(N) +   // + z_mlp^T, a_mlp^T and the sampled indices (z_embed)
                       pad64((D + (size_t)s->S * (s->C ? s->C : 1)) * Hm) + pad64(N * Hm) +   // + the actor's W0^T and its sparse-tail addend
                       tw * (pad64(3 * D * Hd / 2 + 1) + pad64(3 * D * D / 2 + 1) + pad64(Hd * D / 2 + 1) + pad64(Z * Hd / 2 + 1) +
                             pad64(N * Hd / 2 + 1) + pad64((H + 1) * N * (D + (size_t)s->S * (s->C ? s->C : 1)) / 2 + 1));   // + bf16 twins of the cell's weights, za and the h columns of feats
  size_t m = enc_bwd;
  if (dec_fwd > m) m = dec_fwd;
  if (dec_bwd > m) m = dec_bwd;
  if (rssm_bwd_lds > m) m = rssm_bwd_lds;
  if (lds_fwd > m) m = lds_fwd;
  if (mlp_bwd > m) m = mlp_bwd;
  if (dream > m) m = dream;
  return (m + 4096) * sizeof(float);
}
