// Direct small-channel convolutions: the two ends of the image stack, where one side of the layer has 3 channels.
//
//   encoder layer 1  (encoders.py:80-83)    Conv2d(3 -> d, k4, s2) + ELU on the 64x64 frame
//   decoder layer 4  (decoders.py:154-155)  ConvTranspose2d(d -> 3, k6, s2) onto the 64x64 frame
//
// As GEMMs these two have K = 48 resp. N = 108 and need an explicit patch / column matrix, because a 3-channel pixel is
// neither a 16-byte gather nor an MFMA-sized operand: N*31^2 x 48 floats (461 MB at Atari-literal) written by im2col and read
// twice, N*30^2 x 108 floats (972 MB) written by the product and read by col2im - 3.5 GB of HBM traffic and ~1.2 ms per step
// around ~25 GFLOP of work.  Here each is ONE kernel that reads the frame / the NHWC activation directly:
//
//   enc_l1_fwd_kernel    lane = output pixel, all d channels in registers; the 48 x d weights are wave-uniform -> scalar loads,
//                        the FMAs take them as SGPR operands (no LDS, no weight VGPRs); bias + ELU in the epilogue;
//                        float NCHW frames or the replay's uint8 HWC frames (x/255 - 0.5 in the loader)
//   enc_l1_wgrad_kernel  dW[o][tap] = sum_pixels G[pixel][o] patch[pixel][tap] on v_mfma_f32_16x16x4_f32 (K = pixels): the
//                        patch operand is gathered straight from the frame (tap block = input channel, lane&15 = (ky,kx));
//                        per-wave partials, summed in fixed order by dm_colsum_launch (deterministic)
//   dec_l4_fwd_kernel    lane = output pixel of ONE parity class (2yy+py, 2xx+px): an even-k stride-2 transposed convolution
//                        is, per class, a 3x3 convolution over the input; the class's 9 x d x 3 weights are wave-uniform
//                        (scalar loads), the input pixel vectors are 16-byte loads
// fp32 VALU arithmetic (v_fmac with an SGPR operand); bound: VALU issue (13 / 5.5 GFMA-lanes) and the output write.
#include "common.h"
#include <stdlib.h>

static inline int grid_for_px(size_t total, int per_block) { return (int)((total + per_block - 1) / per_block); }

bool dm_conv_direct_enabled() {
  static const int off = getenv("DM_CONV_NO_DIRECT") ? 1 : 0;      // A/B switch: the explicit patch / column matrices
  return !off;
}

// ---------------------------------------------------------------- encoder layer 1, forward --------
// wt: (48 taps, CO) with tap = c*16 + ky*4 + kx (the torch weight (CO,3,4,4) transposed once per call)
template <int CO, bool U8>
__global__ void __launch_bounds__(256) enc_l1_fwd_kernel(int npix, const void* __restrict__ image_,
                                                         const float* __restrict__ wt, const float* __restrict__ bias,
                                                         float* __restrict__ y, unsigned short* __restrict__ y_h) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const bool ok = p < npix;
  const int pc = ok ? p : npix - 1;
  const int n = pc / 961, r = pc - n * 961, py = r / 31, px = r - py * 31;
  float acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc[o] = bias[o];
#pragma unroll 1
  for (int cy = 0; cy < 12; ++cy) {               // (c, ky) pairs; the 4 kx taps of a pair are 4 consecutive pixels of one row
    const int c = cy >> 2, ky = cy & 3;
    float v[4];
    if (U8) {
      const uint8_t* img = (const uint8_t*)image_ + (size_t)n * 12288 + ((size_t)(2 * py + ky) * 64 + 2 * px) * 3 + c;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) v[kx] = (float)img[kx * 3] / 255.0f - 0.5f;      // preprocessing.py:21-29, as im2col_s2_u8hwc
    } else {
      const float* img = (const float*)image_ + (size_t)n * 12288 + (size_t)c * 4096 + (2 * py + ky) * 64 + 2 * px;
      const float2 a = *reinterpret_cast<const float2*>(img), b = *reinterpret_cast<const float2*>(img + 2);   // 2*px: 8-byte aligned
      v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    }
    const float* w = wt + (size_t)(c * 16 + ky * 4) * CO;        // wave-uniform: scalar loads
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
#pragma unroll
      for (int o = 0; o < CO; ++o) acc[o] = fmaf(v[kx], w[kx * CO + o], acc[o]);
  }
  if (ok) {
    float4* dst = reinterpret_cast<float4*>(y + (size_t)p * CO);
#pragma unroll
    for (int o = 0; o < CO; o += 4)
      dst[o >> 2] = make_float4(dm_elu(acc[o]), dm_elu(acc[o + 1]), dm_elu(acc[o + 2]), dm_elu(acc[o + 3]));
    if (y_h) {      // bf16 twin of the activations (conf.amp: the next layer's gathered operand, common.h DmTwinScope)
      uint2* dh = reinterpret_cast<uint2*>(y_h + (size_t)p * CO);
#pragma unroll
      for (int o = 0; o < CO; o += 4)
        dh[o >> 2] = dm_pack_bf16x4(make_float4(dm_elu(acc[o]), dm_elu(acc[o + 1]), dm_elu(acc[o + 2]), dm_elu(acc[o + 3])));
    }
  }
}

__global__ void __launch_bounds__(256) enc_l1_wt_kernel(int co, const float* __restrict__ w, float* __restrict__ wt) {
  const int e = blockIdx.x * 256 + threadIdx.x;                // wt[t][o] = w[o][t], t = c*16 + ky*4 + kx
  if (e < co * 48) wt[e] = w[(size_t)(e % co) * 48 + e / co];
}

bool dm_enc_l1_direct_ok(int ch, int d, int img) {
  return dm_conv_direct_enabled() && ch == 3 && img == 64 && (d == 8 || d == 16 || d == 32 || d == 48 || d == 64);
}
// y (frames*961, d) NHWC post-ELU; wt: scratch of 48*d floats (the transposed weights, written here)
int dm_enc_l1_fwd_launch(int frames, int d, int u8, const void* image, const float* w, const float* bias, float* wt,
                         float* y, unsigned short* y_h, hipStream_t st) {
  if (frames <= 0) return DM_OK;
  hipLaunchKernelGGL(enc_l1_wt_kernel, dim3(grid_for_px((size_t)d * 48, 256)), dim3(256), 0, st, d, w, wt);
  DM_LAUNCH_CHECK();
  const int npix = frames * 961;
  const dim3 grid(grid_for_px((size_t)npix, 256)), blk(256);
#define DM_ENC_L1(CO_)                                                                                              \
  if (d == CO_) {                                                                                                   \
    if (u8) hipLaunchKernelGGL((enc_l1_fwd_kernel<CO_, true>), grid, blk, 0, st, npix, image, wt, bias, y, y_h);      \
    else hipLaunchKernelGGL((enc_l1_fwd_kernel<CO_, false>), grid, blk, 0, st, npix, image, wt, bias, y, y_h);        \
  }
  DM_ENC_L1(8) DM_ENC_L1(16) DM_ENC_L1(32) DM_ENC_L1(48) DM_ENC_L1(64)
#undef DM_ENC_L1
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// ---------------------------------------------------------------- encoder layer 1, weight gradient
typedef float f32x4c __attribute__((ext_vector_type(4)));
constexpr int ENC_LD = 68;       // LDS row stride of the staged frame (floats): 16-byte aligned rows, the 16 taps of a pixel on 16 banks
// part[wave][o][48]: this wave's sum over its pixels of G[pixel][o] * patch[pixel][tap]   (o < CO).
// A workgroup stages one frame at a time in LDS (3 x 64 rows, converted to float once: coalesced 16-byte / 4-byte global
// reads instead of a 4-byte gather per MFMA operand - the first version, gathering from global memory, ran 542 us against
// 226 us for the product on the explicit patch matrix); its 4 waves split the frame's 961 output pixels in chunks of 64.
template <int OB, bool U8>
__global__ void __launch_bounds__(256) enc_l1_wgrad_kernel(int frames, int CO, const void* __restrict__ image_,
                                                           const float* __restrict__ G, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float img[3 * 64 * ENC_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  f32x4c acc[OB][3];
#pragma unroll
  for (int i = 0; i < OB; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4c){0.f, 0.f, 0.f, 0.f};
  const int tapoff = (l15 >> 2) * ENC_LD + (l15 & 3);           // this lane's tap (ky, kx) inside a channel plane
  for (int n = blockIdx.x; n < frames; n += gridDim.x) {
    __syncthreads();                                             // the previous frame's readers are done
    if (U8) {      // (64,64,3) bytes -> three float planes
      const uint8_t* src = (const uint8_t*)image_ + (size_t)n * 12288;
      for (int e = tid; e < 12288; e += 256) {
        const int c = e % 3, pix = e / 3;
        img[(c * 64 + (pix >> 6)) * ENC_LD + (pix & 63)] = (float)src[e] / 255.0f - 0.5f;
      }
    } else {
      const float4* src = reinterpret_cast<const float4*>((const float*)image_ + (size_t)n * 12288);
      for (int e = tid; e < 3072; e += 256) {                    // 3 x 64 rows x 16 float4
        const int row = e >> 4, c4 = e & 15;
        *reinterpret_cast<float4*>(&img[row * ENC_LD + 4 * c4]) = src[e];
      }
    }
    __syncthreads();
    const float* Gn = G + (size_t)n * 961 * CO;
    for (int p0 = wave * 64; p0 < 961; p0 += 256) {
      // all 16 k-steps' G operands first, unconditionally (clamped addresses, zeroed by a multiply): one latency per
      // 64-pixel chunk instead of one per k-step (a conditional load became a branch with the wait right behind it)
      float a[16][OB];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {                          // 4 pixels per MFMA k-step: pixel = p0 + 4*ks + (lane>>4)
        const int p = p0 + 4 * ks + q;
        const int pc = p < 961 ? p : 960;
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
          const int o = ob * 16 + l15;
          a[ks][ob] = Gn[(size_t)pc * CO + (o < CO ? o : CO - 1)] * ((p < 961 && o < CO) ? 1.f : 0.f);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int p = p0 + 4 * ks + q;
        const int pc = p < 961 ? p : 960;                        // a pixel past the frame multiplies a zeroed G row
        const int py = pc / 31, px = pc - py * 31;
        const float* base = &img[2 * py * ENC_LD + 2 * px + tapoff];
        float b[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) b[c] = base[c * 64 * ENC_LD];
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[ob][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][ob], b[c], acc[ob][c], 0, 0, 0);
      }
    }
  }
  // C/D map of the 16x16 MFMA: row (o) = 4*(lane>>4) + r, col (tap) = lane & 15
  float* dst = part + (size_t)(blockIdx.x * 4 + wave) * CO * 48;
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = ob * 16 + 4 * q + r;
        if (o < CO) dst[(size_t)o * 48 + c * 16 + l15] = acc[ob][c][r];
      }
}

static int enc_l1_wgrad_blocks(int frames) { return frames > 512 ? 512 : frames; }      // <= 2048 waves, whole frames per workgroup
size_t dm_enc_l1_wgrad_part_floats(int frames, int d) { return (size_t)enc_l1_wgrad_blocks(frames) * 4 * d * 48; }
// dW (d, 3, 4, 4) = sum over all pixels; G (frames*961, d) = dY0 * ELU'(Y0); part: dm_enc_l1_wgrad_part_floats(frames, d) of scratch
int dm_enc_l1_wgrad_launch(int frames, int d, int u8, const void* image, const float* G, float* part, float* dW, void* ws,
                           size_t ws_bytes, hipStream_t st) {
  if (frames <= 0) return DM_OK;
  const int blocks = enc_l1_wgrad_blocks(frames);
  const dim3 grid(blocks), blk(256);
  const int OB = (d + 15) / 16;
  DM_REQUIRE(OB >= 1 && OB <= 4, DM_E_SHAPE, "enc_l1_wgrad: cnn_depth %d", d);
#define DM_ENC_WG(OB_)                                                                                               \
  if (OB == OB_) {                                                                                                   \
    if (u8) hipLaunchKernelGGL((enc_l1_wgrad_kernel<OB_, true>), grid, blk, 0, st, frames, d, image, G, part);         \
    else hipLaunchKernelGGL((enc_l1_wgrad_kernel<OB_, false>), grid, blk, 0, st, frames, d, image, G, part);           \
  }
  DM_ENC_WG(1) DM_ENC_WG(2) DM_ENC_WG(3) DM_ENC_WG(4)
#undef DM_ENC_WG
  DM_LAUNCH_CHECK();
  return dm_colsum_launch(blocks * 4, d * 48, part, d * 48, dW, ws, ws_bytes, st);
}

// ---------------------------------------------------------------- decoder layer 4, forward --------
// w4[cls][a*3+b][c][4] = W[c][o][py+2a][px+2b] (o < 3, slot 3 = 0), cls = py*2 + px; W is the torch (d, 3, 6, 6) tensor
__global__ void __launch_bounds__(256) dec_l4_repack_kernel(int d, const float* __restrict__ w, float* __restrict__ w4) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 4 * 9 * d * 4) return;
  const int o = e & 3, c = (e >> 2) % d, ab = (e / (4 * d)) % 9, cls = e / (36 * d);
  const int ky = (cls >> 1) + 2 * (ab / 3), kx = (cls & 1) + 2 * (ab % 3);
  w4[e] = o < 3 ? w[(((size_t)c * 3 + o) * 6 + ky) * 6 + kx] : 0.f;
}
// out[n, 2yy+py, 2xx+px, o] = b[o] + sum_{a,b<3} sum_c x[n, yy-a, xx-b, c] * W[c][o][py+2a][px+2b]     (x: (n,30,30,d) NHWC)
// Workgroup = (frame, 8 class rows yy0 .. yy0+7); wave = parity class (py,px); lane = (column xx, half rh) and computes the
// FOUR pixels (yy0 + 4 rh + j, xx), j < 4.  The 10 input rows yy0-2 .. yy0+7 are staged in LDS with coalesced 16-byte reads
// (zero rows / columns outside the image, pixel stride d+4 floats: conflict-free ds_read_b128).  Measured history of this
// kernel at Atari-literal (the column-matrix path it replaces: 0.97 ms): pixel vectors read from global memory, 64 lanes x
// 16 B at a 192-byte stride per load: 4.1 ms (address-coalescing bound); LDS-staged input, one pixel per lane: 1.95 ms - the
// class's weights are wave-uniform, but behind a barrier the compiler does not scalarise global loads, so they arrive as
// vector loads, 4 per 12 FMAs; four pixels per lane share each weight load (4 per 48 FMAs).
constexpr int DEC_ROWS = 10, DEC_COLS = 34;
template <int D>
__global__ void __launch_bounds__(256) dec_l4_fwd_kernel(int frames, const float* __restrict__ x, const float* __restrict__ w4,
                                                         const float* __restrict__ bias, float* __restrict__ out) {
  constexpr int LDP = D + 4, D4 = D / 4;
  extern __shared__ __attribute__((aligned(16))) float rows[];      // [10 rows][34 columns (ix = -2 .. 31)][D + 4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int cls = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x >> 2, yy0 = (blockIdx.x & 3) * 8;
  for (int e = tid; e < DEC_ROWS * DEC_COLS * D4; e += 256) {     // D4 is a constant: the index split is multiply-shift
    const int c4 = e % D4, col = (e / D4) % DEC_COLS, r = e / (D4 * DEC_COLS);
    const int iy = yy0 - 2 + r, ix = col - 2;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < 30 && ix >= 0 && ix < 30) v = *reinterpret_cast<const float4*>(x + (((size_t)n * 30 + iy) * 30 + ix) * D + 4 * c4);
    *reinterpret_cast<float4*>(&rows[(r * DEC_COLS + col) * LDP + 4 * c4]) = v;
  }
  __syncthreads();
  const int py = cls >> 1, px = cls & 1;
  const int rh = lane >> 5, xx = lane & 31;
  // outputs as two float2 pairs per pixel, (o0, o1) and (o2, pad): every FMA is a v_pk_fma_f32 with the input value broadcast
  dm_f32x2 a01[4], a23[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { a01[j] = (dm_f32x2){bias[0], bias[1]}; a23[j] = (dm_f32x2){bias[2], 0.f}; }
#pragma unroll 1
  for (int ab = 0; ab < 9; ++ab) {
    // LDS row of pixel j: (yy0 + 4 rh + j) - a - (yy0 - 2) = 4 rh + j + 2 - a
    const float* src = &rows[((4 * rh + 2 - ab / 3) * DEC_COLS + (xx + 2 - ab % 3)) * LDP];
    const float4* wv = reinterpret_cast<const float4*>(w4 + ((size_t)(cls * 9 + ab) * D) * 4);
#pragma unroll 2
    for (int c4 = 0; c4 < D4; ++c4) {
      dm_f32x2 wlo[4], whi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 w = wv[c4 * 4 + i];
        wlo[i] = (dm_f32x2){w.x, w.y}; whi[i] = (dm_f32x2){w.z, w.w};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)j * DEC_COLS * LDP + 4 * c4);
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const dm_f32x2 b = {vv[i], vv[i]};
          a01[j] = __builtin_elementwise_fma(b, wlo[i], a01[j]);
          a23[j] = __builtin_elementwise_fma(b, whi[i], a23[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float* dst = out + (((size_t)n * 64 + 2 * (yy0 + 4 * rh + j) + py) * 64 + 2 * xx + px) * 3;
    dst[0] = a01[j].x; dst[1] = a01[j].y; dst[2] = a23[j].x;
  }
}

bool dm_dec_l4_direct_ok(int ch, int d, int hs, int k) {
  return dm_conv_direct_enabled() && ch == 3 && hs == 30 && k == 6 && (d == 8 || d == 16 || d == 32 || d == 48 || d == 64);
}
size_t dm_dec_l4_w4_floats(int d) { return (size_t)4 * 9 * d * 4; }
int dm_dec_l4_fwd_launch(int frames, int d, const float* x, const float* w, const float* bias, float* w4, float* out,
                         hipStream_t st) {
  if (frames <= 0) return DM_OK;
  hipLaunchKernelGGL(dec_l4_repack_kernel, dim3(grid_for_px(dm_dec_l4_w4_floats(d), 256)), dim3(256), 0, st, d, w, w4);
  DM_LAUNCH_CHECK();
  const size_t lds = (size_t)DEC_ROWS * DEC_COLS * (d + 4) * sizeof(float);       // 70.7 KB at d = 48: above the 64 KB default
#define DM_DEC_L4(D_)                                                                                                  \
  if (d == D_) {                                                                                                       \
    static bool attr_set[DM_MAX_DEVICES] = {false};      /* per device, like l4_raise_lds below */                  \
    int dev_ = 0;                                                                                                      \
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= DM_MAX_DEVICES) return dm_fail(DM_E_DEVICE, "dec_l4_fwd: hipGetDevice"); \
    if (!attr_set[dev_]) {                                                                                             \
      if (hipFuncSetAttribute((const void*)dec_l4_fwd_kernel<D_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
        return dm_fail(DM_E_HIP, "dec_l4_fwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");                \
      attr_set[dev_] = true;                                                                                           \
    }                                                                                                                  \
    hipLaunchKernelGGL((dec_l4_fwd_kernel<D_>), dim3(frames * 4), dim3(256), lds, st, frames, x, w4, bias, out);         \
  }
  DM_DEC_L4(8) DM_DEC_L4(16) DM_DEC_L4(32) DM_DEC_L4(48) DM_DEC_L4(64)
#undef DM_DEC_L4
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// ---------------------------------------------------------------- decoder layer 4, backward --------
// The image layer's backward as two direct MFMA kernels (round 5).  As gather-form products they ran through the generic
// 64 x 64 tile with the 3-channel gradient padded to 4 channels and d = 48 in a 64-wide tile: 2 250 000 x 48 x 144 and
// 48 x 144 x 2 250 000 at 48 TF/s of useful work (0.65 + 0.64 ms per step at Atari-literal) - 44 % of the issued MFMA work
// was padding.  Here a workgroup stages one frame of the output gradient dY (64 x 64 pixels, THREE channels, row stride RS
// floats) in LDS and both products read their dY operand from that image with computed addresses (the patch of input pixel
// (y, x) is the 6 x 18-float window at (2y, 6x)), on v_mfma_f32_16x16x4_f32 with K = 108 exactly:
//   dec_l4_dgrad_kernel   dX[pixel][c] = ELU'(x3[pixel][c]) * sum_k dYcol[pixel][k] W[c][k]      k = (ky*6 + kx)*3 + o
//                         M = channels (the weights: A operand, 27 x CB registers per lane for the whole kernel), N = 16
//                         pixels per MFMA, a wave walks the frame's 57 pixel blocks; each lane ends up with 4 consecutive
//                         channels of one pixel (16-byte loads of x3, 16-byte stores of dX, 8-byte stores of its bf16 twin)
//   dec_l4_wgrad_kernel   dW[c][k] = sum_pixels x3[pixel][c] dYcol[pixel][k]       K = pixels, 4 per MFMA; x3 straight from
//                         global memory (each 64-byte segment is read once), IB x 7 accumulator tiles per wave over ALL its
//                         frames; per-wave partials in the torch (d, 3, 6, 6) order, summed in fixed order by
//                         dm_colsum_launch (deterministic)
static int g_l4_bwd_direct = getenv("DM_DEC_L4_BWD_GEMM") ? 0 : 1;
extern "C" int dm_dec_l4_bwd_direct_enable(int on) {
  if (on >= 0) g_l4_bwd_direct = on ? 1 : 0;
  return g_l4_bwd_direct;
}
bool dm_dec_l4_bwd_direct_ok(int ch, int d, int hs, int k) { return g_l4_bwd_direct && dm_dec_l4_direct_ok(ch, d, hs, k); }

constexpr int L4_K = 108, L4_KS = 27;
// wp[(s*CB + cb)*64 + q*16 + l15] = W[c = 16 cb + l15][k = 4 s + q]  (0 for c >= d); W is the torch (d, 3, 6, 6) tensor
__global__ void __launch_bounds__(256) dec_l4_dgrad_repack_kernel(int d, int CB, const float* __restrict__ w,
                                                                  float* __restrict__ wp) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= L4_KS * CB * 64) return;
  const int l15 = e & 15, q = (e >> 4) & 3, cb = (e >> 6) % CB, s = e / (64 * CB);
  const int c = 16 * cb + l15, k = 4 * s + q;
  const int o = k % 3, kx = (k / 3) % 6, ky = k / 18;
  wp[e] = c < d ? w[(((size_t)c * 3 + o) * 6 + ky) * 6 + kx] : 0.f;
}

template <int RS>
__device__ __forceinline__ void l4_stage_frame(float* img, const float* __restrict__ G4, int n, int tid) {
  const float4* src = reinterpret_cast<const float4*>(G4) + (size_t)n * 4096;      // (64, 64, 4): channel 3 is the pad
#pragma unroll 4
  for (int e = tid; e < 4096; e += 256) {
    const float4 v = src[e];
    float* dst = &img[(e >> 6) * RS + (e & 63) * 3];
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z;
  }
}

template <int CB, int RS>
__global__ void __launch_bounds__(256) dec_l4_dgrad_kernel(int frames, int d, const float* __restrict__ G4,
                                                           const float* __restrict__ wp, const float* __restrict__ x3,
                                                           float* __restrict__ dx, unsigned short* __restrict__ dx_h) {
  extern __shared__ __attribute__((aligned(16))) float img[];      // [64 rows][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  float wreg[L4_KS][CB];
#pragma unroll
  for (int s = 0; s < L4_KS; ++s)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) wreg[s][cb] = wp[(s * CB + cb) * 64 + lane];
  int koff[L4_KS];
#pragma unroll
  for (int s = 0; s < L4_KS; ++s) {
    const int k = 4 * s + q, ky = k / 18;
    koff[s] = ky * RS + (k - 18 * ky);
  }
  for (int n = blockIdx.x; n < frames; n += gridDim.x) {
    __syncthreads();                                               // the previous frame's readers are done
    l4_stage_frame<RS>(img, G4, n, tid);
    __syncthreads();
    for (int blk = wave; blk < 57; blk += 4) {
      const int p = blk * 16 + l15;
      const int pc = p < 900 ? p : 899;
      const int y = pc / 30, x = pc - 30 * y;
      const float* base = &img[2 * y * RS + 6 * x];
      f32x4c acc[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) acc[cb] = (f32x4c){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < L4_KS; ++s) {
        const float b = base[koff[s]];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s][cb], b, acc[cb], 0, 0, 0);
      }
      // C/D map: row (channel) = 16 cb + 4 q + r, column (pixel) = l15
      if (p < 900) {
        const size_t row = ((size_t)n * 900 + p) * d;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          const int c0 = 16 * cb + 4 * q;
          if (c0 < d) {
            const float4 m = *reinterpret_cast<const float4*>(x3 + row + c0);
            const float4 v = make_float4(acc[cb][0] * dm_elu_grad_from_y(m.x), acc[cb][1] * dm_elu_grad_from_y(m.y),
                                         acc[cb][2] * dm_elu_grad_from_y(m.z), acc[cb][3] * dm_elu_grad_from_y(m.w));
            *reinterpret_cast<float4*>(dx + row + c0) = v;
            if (dx_h) *reinterpret_cast<uint2*>(dx_h + row + c0) = dm_pack_bf16x4(v);
          }
        }
      }
    }
  }
}

// part[(block*4 + wave)][c][o][ky][kx] (c < d): this wave's sum over its pixels of x3[pixel][c] * dYcol[pixel][k]
template <int IB, int RS>
__global__ void __launch_bounds__(256) dec_l4_wgrad_kernel(int frames, int d, const float* __restrict__ G4,
                                                           const float* __restrict__ x3, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float img[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  f32x4c acc[IB][7];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4c){0.f, 0.f, 0.f, 0.f};
  int koff[7];
#pragma unroll
  for (int kb = 0; kb < 7; ++kb) {
    const int k = 16 * kb + l15 < L4_K ? 16 * kb + l15 : L4_K - 1, ky = k / 18;      // columns >= 108 are never written out
    koff[kb] = ky * RS + (k - 18 * ky);
  }
  int cofs[IB];
  float cmask[IB];
#pragma unroll
  for (int ib = 0; ib < IB; ++ib) {
    const int c = 16 * ib + l15;
    cofs[ib] = c < d ? c : d - 1;
    cmask[ib] = c < d ? 1.f : 0.f;
  }
  for (int n = blockIdx.x; n < frames; n += gridDim.x) {
    __syncthreads();
    l4_stage_frame<RS>(img, G4, n, tid);
    __syncthreads();
    const float* xn = x3 + (size_t)n * 900 * d;
    // 57 chunks of 16 pixels (the last holds 4), dealt to the waves round-robin, the deal rotated per frame
    const int w0 = (wave + n) & 3;
    float a[4][IB], an[4][IB];
    auto load = [&](int chunk, float (&dst)[4][IB]) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int p = chunk * 16 + 4 * ks + q;
        const int pc = p < 900 ? p : 899;
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) dst[ks][ib] = xn[(size_t)pc * d + cofs[ib]] * (p < 900 ? cmask[ib] : 0.f);
      }
    };
    if (w0 < 57) load(w0, a);
    for (int chunk = w0; chunk < 57; chunk += 4) {
      if (chunk + 4 < 57) load(chunk + 4, an);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int p = chunk * 16 + 4 * ks + q;
        const int pc = p < 900 ? p : 899;                        // a pixel past the frame multiplies a zeroed x3 row
        const int y = pc / 30, x = pc - 30 * y;
        const float* base = &img[2 * y * RS + 6 * x];
        float b[7];
#pragma unroll
        for (int kb = 0; kb < 7; ++kb) b[kb] = base[koff[kb]];
#pragma unroll
        for (int ib = 0; ib < IB; ++ib)
#pragma unroll
          for (int kb = 0; kb < 7; ++kb) acc[ib][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][ib], b[kb], acc[ib][kb], 0, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) a[ks][ib] = an[ks][ib];
    }
  }
  // C/D map: row (channel) = 16 ib + 4 q + r, column (k) = 16 kb + l15
  float* dst = part + (size_t)(blockIdx.x * 4 + wave) * d * L4_K;
#pragma unroll
  for (int kb = 0; kb < 7; ++kb) {
    const int k = 16 * kb + l15;
    if (k < L4_K) {
      const int o = k % 3, kx = (k / 3) % 6, ky = k / 18;
      const int kt = o * 36 + ky * 6 + kx;
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * ib + 4 * q + r;
          if (c < d) dst[(size_t)c * L4_K + kt] = acc[ib][kb][r];
        }
    }
  }
}

constexpr int L4_RS_D = 196, L4_RS_W = 228;      // row strides of the staged frame: see the bank notes in DESIGN 4.1
size_t dm_dec_l4_wp_floats(int d) { return (size_t)L4_KS * ((d + 15) / 16) * 64; }
static int dec_l4_wgrad_blocks(int frames) { return frames > 512 ? 512 : frames; }
size_t dm_dec_l4_wgrad_part_floats(int frames, int d) { return (size_t)dec_l4_wgrad_blocks(frames) * 4 * d * L4_K; }

template <typename K>
static int l4_raise_lds(K kern, size_t lds, bool* done) {      // done: one flag per device (the attribute is per device)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DM_MAX_DEVICES) return dm_fail(DM_E_DEVICE, "dec_l4 backward: hipGetDevice failed / device %d", dev);
  if (!done[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return dm_fail(DM_E_HIP, "dec_l4 backward: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    done[dev] = true;
  }
  return DM_OK;
}

// dx (frames*900, d) = ELU'(x3) * (dY patches) W^T; G4 (frames, 64, 64, 4) = the padded output gradient; wp: scratch of
// dm_dec_l4_wp_floats(d) (written here); dx_h: optional bf16 twin of dx
int dm_dec_l4_dgrad_launch(int frames, int d, const float* G4, const float* w, float* wp, const float* x3, float* dx,
                           unsigned short* dx_h, hipStream_t st) {
  if (frames <= 0) return DM_OK;
  const int CB = (d + 15) / 16;
  DM_REQUIRE(CB >= 1 && CB <= 4 && (d & 3) == 0, DM_E_SHAPE, "dec_l4_dgrad: cnn_depth %d", d);
  hipLaunchKernelGGL(dec_l4_dgrad_repack_kernel, dim3(grid_for_px(dm_dec_l4_wp_floats(d), 256)), dim3(256), 0, st, d, CB, w, wp);
  DM_LAUNCH_CHECK();
  const size_t lds = (size_t)64 * L4_RS_D * sizeof(float);
#define DM_L4_DG(CB_)                                                                                                \
  if (CB == CB_) {                                                                                                   \
    static bool attr_set[DM_MAX_DEVICES] = {false};                                                                  \
    DM_TRY(l4_raise_lds(dec_l4_dgrad_kernel<CB_, L4_RS_D>, lds, attr_set));                                         \
    hipLaunchKernelGGL((dec_l4_dgrad_kernel<CB_, L4_RS_D>), dim3(frames), dim3(256), lds, st, frames, d, G4, wp, x3, dx, dx_h); \
  }
  DM_L4_DG(1) DM_L4_DG(2) DM_L4_DG(3) DM_L4_DG(4)
#undef DM_L4_DG
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// dW (d, 3, 6, 6) = sum over all pixels; part: dm_dec_l4_wgrad_part_floats(frames, d) of scratch
int dm_dec_l4_wgrad_launch(int frames, int d, const float* G4, const float* x3, float* part, float* dW, void* ws,
                           size_t ws_bytes, hipStream_t st) {
  if (frames <= 0) return DM_OK;
  const int IB = (d + 15) / 16;
  DM_REQUIRE(IB >= 1 && IB <= 4, DM_E_SHAPE, "dec_l4_wgrad: cnn_depth %d", d);
  const int blocks = dec_l4_wgrad_blocks(frames);
  const size_t lds = (size_t)64 * L4_RS_W * sizeof(float);
#define DM_L4_WG(IB_)                                                                                                \
  if (IB == IB_) {                                                                                                   \
    static bool attr_set[DM_MAX_DEVICES] = {false};                                                                  \
    DM_TRY(l4_raise_lds(dec_l4_wgrad_kernel<IB_, L4_RS_W>, lds, attr_set));                                         \
    hipLaunchKernelGGL((dec_l4_wgrad_kernel<IB_, L4_RS_W>), dim3(blocks), dim3(256), lds, st, frames, d, G4, x3, part); \
  }
  DM_L4_WG(1) DM_L4_WG(2) DM_L4_WG(3) DM_L4_WG(4)
#undef DM_L4_WG
  DM_LAUNCH_CHECK();
  return dm_colsum_launch(blocks * 4, d * L4_K, part, d * L4_K, dW, ws, ws_bytes, st);
}
