// ConvEncoder (encoders.py:72-96) and ConvDecoder + MSE (decoders.py:111-180) as GEMMs over stride-2 patch
// matrices.  Internal activations are NHWC so that a patch row (ky,kx,c) is contiguous in c and every
// gather/scatter kernel below moves 16-byte vectors; weights are re-laid once per call from the torch
// layouts (O,I,kh,kw) / (I,O,kh,kw) to (O,kh,kw,I) / (I,kh,kw,O) and gradients are permuted back.
//
//   conv   k4 s2 : Y[(n,ys,xs)][o]          = ELU( im2col(X)[(n,ys,xs)][(ky,kx,i)] . Wr[o][(ky,kx,i)] + b[o] )
//   convT  k  s2 : Ycol[(n,ys,xs)][(ky,kx,o)] = X[(n,ys,xs)][i] . Wr[i][(ky,kx,o)] ;  out = col2im(Ycol) + b (+ELU)
// The backward passes are the same two gathers with the roles swapped (convT backward-data is a conv).
// All kernels here are HBM-streaming; the contractions run in gemm.hip on the matrix cores.
#include "common.h"
#include <stdlib.h>

// ---------------------------------------------------------------- patch gather ------------------
// col[(i,ys,xs)][(ky,kx,cc)] = big[i, 2ys+ky, 2xs+kx, cc]      (NHWC, VEC floats of c per thread)
template <int VEC>
__global__ void __launch_bounds__(256) im2col_s2_nhwc_kernel(int n, int hb, int wb, int c, int k, int hs, int ws,
                                                             const float* __restrict__ big, float* __restrict__ col) {
  const int cv = c / VEC;
  const size_t total = (size_t)n * hs * ws * k * k * cv;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    size_t t = e;
    const int c4 = (int)(t % cv); t /= cv;
    const int kx = (int)(t % k); t /= k;
    const int ky = (int)(t % k); t /= k;
    const int xs = (int)(t % ws); t /= ws;
    const int ys = (int)(t % hs); t /= hs;
    const int i = (int)t;
    const size_t src = (((size_t)i * hb + (2 * ys + ky)) * wb + (2 * xs + kx)) * c + (size_t)c4 * VEC;
    if constexpr (VEC == 4) {
      reinterpret_cast<float4*>(col)[e] = *reinterpret_cast<const float4*>(big + src);
    } else {
      col[e] = big[src];
    }
  }
}

// layer-1 variant reading the (T,B,C,H,W) batch directly: col[(i,ys,xs)][(cc,ky,kx)] = big[i, cc, 2ys+ky, 2xs+kx]
__global__ void __launch_bounds__(256) im2col_s2_nchw_kernel(int n, int hb, int wb, int c, int k, int hs, int ws,
                                                             const float* __restrict__ big, float* __restrict__ col) {
  const size_t total = (size_t)n * hs * ws * c * k * k;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    size_t t = e;
    const int kx = (int)(t % k); t /= k;
    const int ky = (int)(t % k); t /= k;
    const int cc = (int)(t % c); t /= c;
    const int xs = (int)(t % ws); t /= ws;
    const int ys = (int)(t % hs); t /= hs;
    const int i = (int)t;
    col[e] = big[(((size_t)i * c + cc) * hb + (2 * ys + ky)) * wb + (2 * xs + kx)];
  }
}

// the same patch matrix straight from the replay's native uint8 (T,B,H,W,C) frames (preprocessing.py:21-29 `to_image`:
// x/255 - 0.5 and HWC -> CHW happen HERE, in the first conv's patch loader - SURVEY 8(f) N1): the float image is never
// materialised, the frames cross HBM as 1 byte per element.  col[(i,ys,xs)][(cc,ky,kx)] = u8[i, 2ys+ky, 2xs+kx, cc]/255 - 0.5
__global__ void __launch_bounds__(256) im2col_s2_u8hwc_kernel(int n, int hb, int wb, int c, int k, int hs, int ws,
                                                              const uint8_t* __restrict__ big, float* __restrict__ col) {
  const size_t total = (size_t)n * hs * ws * c * k * k;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    size_t t = e;
    const int kx = (int)(t % k); t /= k;
    const int ky = (int)(t % k); t /= k;
    const int cc = (int)(t % c); t /= c;
    const int xs = (int)(t % ws); t /= ws;
    const int ys = (int)(t % hs); t /= hs;
    const int i = (int)t;
    col[e] = (float)big[(((size_t)i * hb + (2 * ys + ky)) * wb + (2 * xs + kx)) * c + cc] / 255.0f - 0.5f;
  }
}

// big[i,y,x,cc] = epi( bias[cc] + sum_{ky,kx : (y-ky),(x-kx) even, in range} col[(i,(y-ky)/2,(x-kx)/2)][(ky,kx,cc)] )
template <int VEC>
__global__ void __launch_bounds__(256) col2im_s2_kernel(int n, int hb, int wb, int c, int k, int hs, int ws,
                                                        const float* __restrict__ col, const float* __restrict__ bias,
                                                        int flags, const float* __restrict__ elu_ref,
                                                        float* __restrict__ big, unsigned short* __restrict__ big_h) {
  const int cv = c / VEC;
  const size_t total = (size_t)n * hb * wb * cv;
  const size_t rowlen = (size_t)k * k * c;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    size_t t = e;
    const int c4 = (int)(t % cv); t /= cv;
    const int x = (int)(t % wb); t /= wb;
    const int y = (int)(t % hb); t /= hb;
    const int i = (int)t;
    float acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = bias ? bias[c4 * VEC + q] : 0.f;
    for (int ky = y & 1; ky < k && ky <= y; ky += 2) {
      const int ys = (y - ky) >> 1;
      if (ys >= hs) continue;
      for (int kx = x & 1; kx < k && kx <= x; kx += 2) {
        const int xs = (x - kx) >> 1;
        if (xs >= ws) continue;
        const float* p = col + (((size_t)i * hs + ys) * ws + xs) * rowlen + (size_t)(ky * k + kx) * c + (size_t)c4 * VEC;
        if constexpr (VEC == 4) {
          const float4 v = *reinterpret_cast<const float4*>(p);
          acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        } else {
          acc[0] += p[0];
        }
      }
    }
    const size_t dst = e * VEC;
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      float v = acc[q];
      if (flags & DM_C2I_ELU) v = dm_elu(v);
      if (elu_ref) v *= dm_elu_grad_from_y(elu_ref[dst + q]);
      big[dst + q] = v;
      if (big_h) big_h[dst + q] = (unsigned short)dm_f2bf(v);
    }
  }
}

static inline int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

int dm_im2col_s2_launch(int n, int hb, int wb, int c, int k, const float* big, int big_nchw, float* col, hipStream_t st) {
  // "valid" stride-2 geometry: a trailing row/column that no window reaches is simply never read (31 -> 14 at k=4)
  DM_REQUIRE(hb >= k && wb >= k && k >= 1, DM_E_SHAPE, "im2col_s2: big %dx%d smaller than k=%d", hb, wb, k);
  const int hs = (hb - k) / 2 + 1, ws = (wb - k) / 2 + 1;
  const size_t total = (size_t)n * hs * ws * k * k * c;
  if (total == 0) return DM_OK;
  if (big_nchw) {
    hipLaunchKernelGGL(im2col_s2_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, st, n, hb, wb, c, k, hs, ws, big, col);
  } else if ((c & 3) == 0 && (((uintptr_t)big | (uintptr_t)col) & 15) == 0) {
    hipLaunchKernelGGL((im2col_s2_nhwc_kernel<4>), dim3(grid_for(total / 4)), dim3(256), 0, st, n, hb, wb, c, k, hs, ws,
                       big, col);
  } else {
    hipLaunchKernelGGL((im2col_s2_nhwc_kernel<1>), dim3(grid_for(total)), dim3(256), 0, st, n, hb, wb, c, k, hs, ws, big,
                       col);
  }
  DM_LAUNCH_CHECK();
  return DM_OK;
}

int dm_col2im_s2_launch(int n, int hb, int wb, int c, int k, const float* col, const float* bias, int flags,
                        const float* elu_ref, float* big, hipStream_t st) {
  // rows/columns of `big` that no window reaches receive only the bias (zero gradient in the conv backward use)
  DM_REQUIRE(hb >= k && wb >= k && k >= 1, DM_E_SHAPE, "col2im_s2: big %dx%d smaller than k=%d", hb, wb, k);
  const int hs = (hb - k) / 2 + 1, ws = (wb - k) / 2 + 1;
  const size_t total = (size_t)n * hb * wb * c;
  if (total == 0) return DM_OK;
  unsigned short* big_h = dm_twin_of(big, false);      // bf16 mode: the result's twin (common.h DmTwinScope), written here
  if (big_h) dm_twin_mark(big);
  if ((c & 3) == 0 && (((uintptr_t)big | (uintptr_t)col) & 15) == 0) {
    hipLaunchKernelGGL((col2im_s2_kernel<4>), dim3(grid_for(total / 4)), dim3(256), 0, st, n, hb, wb, c, k, hs, ws, col,
                       bias, flags, elu_ref, big, big_h);
  } else {
    hipLaunchKernelGGL((col2im_s2_kernel<1>), dim3(grid_for(total)), dim3(256), 0, st, n, hb, wb, c, k, hs, ws, col, bias,
                       flags, elu_ref, big, big_h);
  }
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_im2col_s2(int n, int hb, int wb, int c, int k, const float* big, int big_nchw, float* col, void* stream) {
  DM_REQUIRE(big && col, DM_E_NULL, "im2col_s2: null pointer");
  return dm_im2col_s2_launch(n, hb, wb, c, k, big, big_nchw, col, (hipStream_t)stream);
}
extern "C" int dm_col2im_s2(int n, int hb, int wb, int c, int k, const float* col, const float* bias, int flags,
                            const float* elu_ref, float* big, void* stream) {
  DM_REQUIRE(big && col, DM_E_NULL, "col2im_s2: null pointer");
  return dm_col2im_s2_launch(n, hb, wb, c, k, col, bias, flags, elu_ref, big, (hipStream_t)stream);
}

// dst[j0,j1,j2,j3] = src[i0,i1,i2,i3] with j_a = i_{p_a}  (dst dims = (d[p0],d[p1],d[p2],d[p3]))
__global__ void __launch_bounds__(256) permute4_kernel(const float* __restrict__ src, float* __restrict__ dst, int d0,
                                                       int d1, int d2, int d3, int p0, int p1, int p2, int p3) {
  const int d[4] = {d0, d1, d2, d3};
  const int p[4] = {p0, p1, p2, p3};
  const size_t total = (size_t)d0 * d1 * d2 * d3;
  const size_t sstride[4] = {(size_t)d1 * d2 * d3, (size_t)d2 * d3, (size_t)d3, 1};
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    size_t t = e;
    size_t s = 0;
#pragma unroll
    for (int a = 3; a >= 0; --a) {
      const int dim = d[p[a]];
      const int j = (int)(t % dim);
      t /= dim;
      s += (size_t)j * sstride[p[a]];
    }
    dst[e] = src[s];
  }
}
int dm_permute4_launch(const float* src, float* dst, int d0, int d1, int d2, int d3, int p0, int p1, int p2, int p3,
                       hipStream_t st) {
  const size_t total = (size_t)d0 * d1 * d2 * d3;
  if (total == 0) return DM_OK;
  hipLaunchKernelGGL(permute4_kernel, dim3(grid_for(total)), dim3(256), 0, st, src, dst, d0, d1, d2, d3, p0, p1, p2, p3);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// ---------------------------------------------------------------- implicit im2col tables ----------
// The stride-2 patch matrix col[(i,ys,xs)][(ky,kx,c)] = big[i, 2ys+ky, 2xs+kx, c] (NHWC) is a SEPARABLE gather:
//   address = rowoff[(i,ys,xs)] + koff[(ky,kx,c)],  rowoff = ((i*hb + 2ys)*wb + 2xs)*c,  koff = ky*wb*c + kx*c + cc
// so the GEMM loaders (gemm.hip, gather operands) read patches straight from the activation tensor and the patch
// matrix is never materialised.  Both tables are tiny (rows + k*k*c ints) and L2-resident.
__global__ void __launch_bounds__(256) conv_tables_kernel(int n, int hb, int wb, int c, int k, int hs, int ws,
                                                          int* __restrict__ rowoff, int* __restrict__ koff) {
  const int rows = n * hs * ws;
  const int kc = k * c;
  const int kdim = k * kc;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < rows + kdim; e += gridDim.x * 256) {
    if (e < rows) {
      const int xs = e % ws, ys = (e / ws) % hs, i = e / (ws * hs);
      rowoff[e] = ((i * hb + 2 * ys) * wb + 2 * xs) * c;
    } else {
      const int q = e - rows;
      koff[q] = (q / kc) * (wb * c) + q % kc;
    }
  }
}
static int conv_tables_launch(int n, int hb, int wb, int c, int k, int* rowoff, int* koff, hipStream_t st) {
  const int hs = (hb - k) / 2 + 1, ws = (wb - k) / 2 + 1;
  DM_REQUIRE((int64_t)n * hb * wb * c < ((int64_t)1 << 31), DM_E_SHAPE, "conv tables: tensor exceeds 2^31 elements");
  const int total = n * hs * ws + k * k * c;
  hipLaunchKernelGGL(conv_tables_kernel, dim3(grid_for(total)), dim3(256), 0, st, n, hb, wb, c, k, hs, ws, rowoff, koff);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// ---------------------------------------------------------------- gather-form transposed convolution ----------
// ConvTranspose2d(k even, stride 2) WITHOUT a column matrix (decoders.py:144-161; the same identity gives the data
// gradient of a stride-2 Conv2d).  out[n, 2yy+py, 2xx+px, o] = b[o] + sum_{a,b,i} X[n, yy-a, xx-b, i] W[i][o][py+2a][px+2b]
// with a, b in [0, k/2): every parity class (py,px) reads the SAME input patch, so the layer is ONE GEMM
//   M = n*Hc*Wc class pixels (Hc = hs + k/2 - 1),  K = (k/2)^2 * cin (gathered from a zero-padded copy of X),
//   N = 4*cout (the four classes side by side), result scattered straight into the NHWC big image (gemm.hip SC epilogue).
// vs. the column-matrix form: no Ycol write + col2im read (2 x 2.9 GB for decoder layer 3 at Atari-literal), at the price
// of (Hc/hs)^2 more MACs (the zero border is multiplied too: 1.33x for 13 -> 30).
__global__ void __launch_bounds__(256) convt_pad_kernel(int n, int hs, int ws, int c4, int P, const float4* __restrict__ x,
                                                        float4* __restrict__ xp) {
  const int hp = hs + 2 * P, wp = ws + 2 * P;
  const size_t total = (size_t)n * hp * wp * c4;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    size_t t = e;
    const int cc = (int)(t % c4); t /= c4;
    const int xx = (int)(t % wp) - P; t /= wp;
    const int yy = (int)(t % hp) - P; t /= hp;
    const int i = (int)t;
    const bool in = yy >= 0 && yy < hs && xx >= 0 && xx < ws;
    xp[e] = in ? x[(((size_t)i * hs + yy) * ws + xx) * c4 + cc] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
// The same zero-padded copy written as bf16 (the gathered operand of a bf16-storage product, common.h DmTwinScope)
__global__ void __launch_bounds__(256) convt_pad_h_kernel(int n, int hs, int ws, int c4, int P, const float4* __restrict__ x,
                                                          uint2* __restrict__ xp) {
  const int hp = hs + 2 * P, wp = ws + 2 * P;
  const size_t total = (size_t)n * hp * wp * c4;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    size_t t = e;
    const int cc = (int)(t % c4); t /= c4;
    const int xx = (int)(t % wp) - P; t /= wp;
    const int yy = (int)(t % hp) - P; t /= hp;
    const int i = (int)t;
    const bool in = yy >= 0 && yy < hs && xx >= 0 && xx < ws;
    xp[e] = in ? dm_pack_bf16x4(x[(((size_t)i * hs + yy) * ws + xx) * c4 + cc]) : make_uint2(0u, 0u);
  }
}
// rowoff[(n,yy,xx)] = offset of padded pixel (yy+P, xx+P);  koff[(a,b,i)] = -(a*wp + b)*cin + i;
// ctab[(n,yy,xx)] = {offset of big[n, 2yy, 2xx, 0], bit 1: 2yy+1 < hb, bit 0: 2xx+1 < wb}
__global__ void __launch_bounds__(256) convt_tables_kernel(int n, int hs, int ws, int cin, int ta, int hb, int wb, int cout,
                                                           int* __restrict__ rowoff, int* __restrict__ koff,
                                                           int2* __restrict__ ctab) {
  const int P = ta - 1, hp = hs + 2 * P, wp = ws + 2 * P;
  const int Hc = hs + ta - 1, Wc = ws + ta - 1;
  const int rows = n * Hc * Wc, kdim = ta * ta * cin;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < rows + kdim; e += gridDim.x * 256) {
    if (e < rows) {
      const int xx = e % Wc, yy = (e / Wc) % Hc, i = e / (Wc * Hc);
      rowoff[e] = ((i * hp + yy + P) * wp + xx + P) * cin;
      ctab[e] = make_int2(((i * hb + 2 * yy) * wb + 2 * xx) * cout, ((2 * yy + 1 < hb) ? 2 : 0) | ((2 * xx + 1 < wb) ? 1 : 0));
    } else {
      const int q = e - rows;
      const int ci = q % cin, b = (q / cin) % ta, a = q / (cin * ta);
      koff[q] = -(a * wp + b) * cin + ci;
    }
  }
}
// Wcat[(py,px,o)][(a,b,i)] = W[i][o][py+2a][px+2b]      (W: torch ConvTranspose2d layout (cin, cout, k, k))
__global__ void __launch_bounds__(256) convt_repack_kernel(int cin, int cout, int k, const float* __restrict__ w,
                                                           float* __restrict__ wcat) {
  const int ta = k / 2, kdim = ta * ta * cin;
  const int total = 4 * cout * kdim;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int kk = e % kdim, nn = e / kdim;
    const int ci = kk % cin, b = (kk / cin) % ta, a = kk / (cin * ta);
    const int o = nn % cout, cls = nn / cout, py = cls >> 1, px = cls & 1;
    wcat[e] = w[(((size_t)ci * cout + o) * k + (py + 2 * a)) * k + (px + 2 * b)];
  }
}
static const int g_convt_gather_off = getenv("DM_CONVT_COLUMN") ? 1 : 0;       // A/B switch: keep the column-matrix form
// layers this form is used for: even kernel, 16-byte gathers, and enough output channels that N = 4*cout fills MFMA tiles
// (and an input large enough that the multiplied zero border stays under 40 % extra MACs: (Hc/hs)^2 <= 1.4, i.e. hs >= 6 for
// k = 4 and hs >= 11 for k = 6)
static bool convt_gather_ok(int k, int cin, int cout, int hs, size_t n) {
  const int Hc = hs + k / 2 - 1;
  return !g_convt_gather_off && (k & 1) == 0 && (cin & 3) == 0 && cout >= 16 && 10 * Hc * Hc <= 14 * hs * hs &&
         n * (size_t)(2 * (hs - 1) + k + 1) * (2 * (hs - 1) + k + 1) * cout < ((size_t)1 << 31);
}

// ---------------------------------------------------------------- MSE (decoders.py:163-167) -----
// pred NHWC (n,hw,c), target NCHW (n,c,hw).  loss[n] = 0.5*sum (pred-target)^2 ; dpred = scale*(pred-target) (NHWC);
// rec = pred in NCHW.  One block per frame, fixed reduction order.
// U8: the target is the replay's uint8 HWC frame, converted on the fly (x/255 - 0.5); it shares the prediction's NHWC
// index, so both streams are coalesced.  tdiv: iwae_samples (decoders.py:163-167 expands the target over I: prediction
// frame i compares with target frame i / I);  row_scale (optional): per-frame factor on dpred (IWAE importance weights).
template <bool U8>
__global__ void __launch_bounds__(256) mse_image_kernel(int hw, int c, const float* __restrict__ pred,
                                                        const void* __restrict__ target_, int tdiv, float scale,
                                                        const float* __restrict__ row_scale,
                                                        float* __restrict__ loss, float* __restrict__ dpred, int dpad,
                                                        float* __restrict__ rec, unsigned short* __restrict__ dpred_h) {
  __shared__ float red[4];
  const int i = blockIdx.x;
  const int per = hw * c;
  const float* pr = pred + (size_t)i * per;
  const float* tg = (const float*)target_ + (size_t)(i / tdiv) * per;
  const uint8_t* tg8 = (const uint8_t*)target_ + (size_t)(i / tdiv) * per;
  if (row_scale) scale *= row_scale[i];
  float s = 0.f;
  for (int e = threadIdx.x; e < per; e += 256) {
    const int pix = e / c, cc = e % c;
    const float pv = pr[e];
    const float tv = U8 ? (float)tg8[e] / 255.0f - 0.5f : tg[(size_t)cc * hw + pix];
    const float d = pv - tv;
    s += d * d;
    if (dpred) {        // dpad >= c channels per pixel; the pad channels are written as zeros (see the 4-channel trick in backward)
      float* dp = dpred + ((size_t)i * hw + pix) * dpad;
      dp[cc] = scale * d;
      if (cc == c - 1)
        for (int q = c; q < dpad; ++q) dp[q] = 0.f;
      if (dpred_h) {      // bf16 twin of the gradient (common.h DmTwinScope)
        unsigned short* dh = dpred_h + ((size_t)i * hw + pix) * dpad;
        dh[cc] = (unsigned short)dm_f2bf(scale * d);
        if (cc == c - 1)
          for (int q = c; q < dpad; ++q) dh[q] = 0;
      }
    }
    if (rec) rec[(size_t)i * per + (size_t)cc * hw + pix] = pv;
  }
  s = dm_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && loss) loss[i] = 0.5f * (red[0] + red[1] + red[2] + red[3]);
}

// ---------------------------------------------------------------- uint8 ingest ---
// preprocessing.py:21-29 `to_image`: uint8 (n, h*w, c) HWC frames -> float32 (n, c, h*w) CHW in [-0.5, 0.5], x/255 - 0.5 in
// fp32 (correctly rounded division, same values as numpy's).  One pass: 1 byte read + 4 bytes written per element, so
// a replay batch crosses PCIe and HBM as 31 MB instead of 123 MB and the host-side astype/transposes disappear.
__global__ void __launch_bounds__(256) preprocess_u8_kernel(size_t n, int hw, int c, const uint8_t* __restrict__ src,
                                                            float* __restrict__ dst) {
  const size_t total = n * hw * c;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t i = e / ((size_t)hw * c);
    const int rem = (int)(e % ((size_t)hw * c));
    const int cc = rem / hw, pix = rem % hw;          // e indexes the CHW (output) side: coalesced writes
    dst[e] = (float)src[i * hw * c + (size_t)pix * c + cc] / 255.0f - 0.5f;
  }
}
extern "C" int dm_preprocess_image_u8(int64_t n, int hw, int c, const uint8_t* src, float* dst, void* stream) {
  DM_REQUIRE(src && dst, DM_E_NULL, "preprocess_image_u8: null pointer");
  DM_REQUIRE(n >= 0 && hw >= 1 && c >= 1, DM_E_SHAPE, "preprocess_image_u8: bad shape n=%lld hw=%d c=%d", (long long)n, hw, c);
  if (n == 0) return DM_OK;
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid_for((size_t)n * hw * c)), dim3(256), 0, (hipStream_t)stream, (size_t)n,
                     hw, c, src, dst);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// dm_shape.flags bit 4: `image` / `target` pointers are uint8 (N, H, W, C) frames (the replay's native format)
static inline bool shape_u8(const dm_shape* s) { return (s->flags & DM_FLAG_IMAGE_U8) != 0; }
static int mse_launch(bool u8, int n, int hw, int c, const float* pred, const void* target, int tdiv, float scale,
                      const float* row_scale, float* loss, float* dpred, int dpad, float* rec, hipStream_t st) {
  unsigned short* dh = dpred ? dm_twin_of(dpred, false) : nullptr;      // bf16 mode: the gradient's twin, written here
  if (dh) dm_twin_mark(dpred);
  if (u8) hipLaunchKernelGGL((mse_image_kernel<true>), dim3(n), dim3(256), 0, st, hw, c, pred, target, tdiv, scale, row_scale, loss, dpred, dpad, rec, dh);
  else hipLaunchKernelGGL((mse_image_kernel<false>), dim3(n), dim3(256), 0, st, hw, c, pred, target, tdiv, scale, row_scale, loss, dpred, dpad, rec, dh);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// ---------------------------------------------------------------- geometry ----------------------
struct EncGeom {
  int N, ch, d;
  int hb[4], hs[4], cin[4], cout[4];
  size_t rows[4], kdim[4];
  bool direct0;      // layer 1 runs as a direct convolution (conv_direct.hip): no explicit patch matrix
  bool twins;        // bf16 mode: the arena also holds bf16 twins of y[0..2] and of the repacked weights (common.h DmTwinScope)
  explicit EncGeom(const dm_shape* s) {
    twins = (s->flags & DM_FLAG_BF16) != 0;
    N = s->T * s->B * (s->I > 0 ? s->I : 1);
    ch = s->img_ch; d = s->cnn_depth;
    direct0 = dm_enc_l1_direct_ok(ch, d, s->img);
    int h = s->img;
    const int ci[4] = {ch, d, 2 * d, 4 * d};
    const int co[4] = {d, 2 * d, 4 * d, 8 * d};
    for (int l = 0; l < 4; ++l) {
      hb[l] = h;
      hs[l] = (h - 4) / 2 + 1;
      cin[l] = ci[l]; cout[l] = co[l];
      rows[l] = (size_t)N * hs[l] * hs[l];
      kdim[l] = (size_t)16 * ci[l];
      h = hs[l];
    }
  }
  bool valid(const dm_shape* s) const {
    return s->img == 64 && hs[3] == 2 && s->E == 8 * d * 4 && ch >= 1 && d >= 1;
  }
};

struct DecGeom {
  int N, ch, d, F;
  int k[5], hsm[5], hbg[5], cin[5], cout[5];   // index 1..4
  size_t rows_s[5], rows_b[5];
  bool twins;       // bf16 mode: the arena also holds bf16 twins of x[0..2] and wr[1..4] (common.h DmTwinScope)
  explicit DecGeom(const dm_shape* s) {
    twins = (s->flags & DM_FLAG_BF16) != 0;
    N = s->T * s->B * (s->I > 0 ? s->I : 1);
    ch = s->img_ch; d = s->cnn_depth;
    F = s->D + s->S * (s->C ? s->C : 1);      // Gaussian latents (C = 0) are S wide
    const int kk[5] = {0, 5, 5, 6, 6};
    const int ci[5] = {0, 32 * d, 4 * d, 2 * d, d};
    const int co[5] = {0, 4 * d, 2 * d, d, ch};
    int h = 1;
    for (int l = 1; l <= 4; ++l) {
      k[l] = kk[l]; cin[l] = ci[l]; cout[l] = co[l];
      hsm[l] = h;
      hbg[l] = 2 * (h - 1) + kk[l];
      rows_s[l] = (size_t)N * h * h;
      rows_b[l] = (size_t)N * hbg[l] * hbg[l];
      h = hbg[l];
    }
  }
  bool valid(const dm_shape* s) const { return hbg[4] == s->img && s->img == 64; }
};

// ---------------------------------------------------------------- encoder -----------------------
struct EncActs {
  float* wr[4];    // repacked weights (l>=1)
  float* xcol[4];  // explicit patch matrix: layer 0 only (reads the NCHW batch, 3 channels)
  int* rowoff[4];  // implicit-im2col tables (l>=1), see conv_tables_kernel
  int* koff[4];
  float* y[4];     // post-ELU NHWC outputs
  unsigned short* yh[4];    // bf16 mode: twins of y[0..2] and wr[1..3], behind everything else (fp32 offsets do not move)
  unsigned short* wrh[4];
};
static size_t enc_carve(const EncGeom& g, float* base, size_t cap_floats, EncActs* a) {
  DmArena ar(base, cap_floats * sizeof(float));
  for (int l = 0; l < 4; ++l) {
    float* w = ar.take(l == 0 ? 0 : (size_t)g.cout[l] * g.kdim[l]);
    float* xc = ar.take(l == 0 && !g.direct0 ? g.rows[l] * g.kdim[l] : 0);
    float* ro = ar.take(l == 0 ? 0 : g.rows[l]);
    float* ko = ar.take(l == 0 ? 0 : g.kdim[l]);
    float* yy = ar.take(g.rows[l] * g.cout[l]);
    if (a) { a->wr[l] = w; a->xcol[l] = xc; a->rowoff[l] = (int*)ro; a->koff[l] = (int*)ko; a->y[l] = yy; }
  }
  for (int l = 0; l < 4; ++l) {
    float* yh = ar.take(g.twins && l < 3 ? dm_half_floats(g.rows[l] * g.cout[l]) : 0);
    float* wh = ar.take(g.twins && l > 0 ? dm_half_floats((size_t)g.cout[l] * g.kdim[l]) : 0);
    if (a) { a->yh[l] = (unsigned short*)yh; a->wrh[l] = (unsigned short*)wh; }
  }
  return ar.off;
}
static void enc_register_twins(const EncGeom& g, const EncActs& a, bool acts_valid, bool weights_valid) {
  if (!dm_twins_on()) return;
  for (int l = 0; l < 3; ++l) dm_twin_add(a.y[l], g.rows[l] * g.cout[l], a.yh[l], acts_valid);
  for (int l = 1; l < 4; ++l) dm_twin_add(a.wr[l], (size_t)g.cout[l] * g.kdim[l], a.wrh[l], weights_valid);
}
extern "C" size_t dm_conv_encoder_acts_floats(const dm_shape* shp) {
  if (!shp) return 0;
  EncGeom g(shp);
  return enc_carve(g, nullptr, 0, nullptr);
}

// NHWC (n, hw, c) <-> torch flatten order (n, c*hw + pix)
__global__ void __launch_bounds__(256) nhwc_to_chw_flat_kernel(size_t n, int hw, int c, const float* __restrict__ src,
                                                               float* __restrict__ dst, int to_chw) {
  const size_t total = n * hw * c;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t i = e / ((size_t)hw * c);
    const int rem = (int)(e % ((size_t)hw * c));
    const int pix = rem / c, cc = rem % c;        // e indexes the NHWC side
    const size_t o = i * hw * c + (size_t)cc * hw + pix;
    if (to_chw) dst[o] = src[e];
    else dst[e] = src[o];
  }
}

// Frames [n0, n0+n) of the batch; `acts` / `embed` / `image` are the FULL-batch buffers (every per-layer buffer is
// frame-major, so a frame range is a contiguous slice of each).  prepare != 0 additionally builds what all ranges share
// (gather tables, repacked weights) - callers that pipeline ranges over several streams prepare once, with n = 0, on
// the stream every range stream waits on.
extern "C" int dm_conv_encoder_fwd_rows(const dm_shape* shp, int n0, int n, int prepare, const float* image,
                                        const dm_conv_params* p, float* acts, float* embed, void* ws, size_t ws_bytes,
                                        void* stream) {
  DM_REQUIRE(shp && image && p && acts && embed && ws, DM_E_NULL, "conv_encoder_fwd: null pointer");
  DmPrecisionScope prec(shp->flags & DM_FLAG_BF16);
  EncGeom g(shp);
  DM_REQUIRE(g.valid(shp), DM_E_SHAPE, "conv_encoder: unsupported geometry (img=%d, E=%d, depth=%d)", shp->img, shp->E,
             shp->cnn_depth);
  DM_REQUIRE(n0 >= 0 && n >= 0 && n0 + n <= g.N, DM_E_SHAPE, "conv_encoder_fwd: frame range [%d,%d) outside 0..%d", n0,
             n0 + n, g.N);
  hipStream_t st = (hipStream_t)stream;
  EncActs a;
  enc_carve(g, acts, (size_t)1 << 60, &a);
  DM_REQUIRE(ws_bytes >= DM_SPLITK_FLOATS * sizeof(float), DM_E_WORKSPACE, "conv_encoder_fwd: workspace too small");
  DmTwinScope tw((shp->flags & DM_FLAG_BF16) != 0);
  enc_register_twins(g, a, false, !prepare && dm_twin_arena_valid(acts));
  dm_twin_arena_note(acts, dm_twins_on());
  if (prepare) {
    for (int l = 1; l < 4; ++l) {
      DM_TRY(conv_tables_launch(g.N, g.hb[l], g.hb[l], g.cin[l], 4, a.rowoff[l], a.koff[l], st));
      DM_TRY(dm_permute4_launch(p->w[l], a.wr[l], g.cout[l], g.cin[l], 4, 4, 0, 2, 3, 1, st));
    }
    if (dm_twins_on()) {
      DmCvtSeg sg[3];
      for (int l = 1; l < 4; ++l) { sg[l - 1] = DmCvtSeg{a.wr[l], a.wrh[l], (size_t)g.cout[l] * g.kdim[l]}; dm_twin_mark(a.wr[l]); }
      DM_TRY(dm_to_bf16_multi_launch(sg, 3, st));
    }
  }
  if (n == 0) return DM_OK;
  for (int l = 0; l < 4; ++l) {
    const size_t r0 = (size_t)n0 * g.hs[l] * g.hs[l];          // first patch row of the range in layer l
    if (l == 0 && g.direct0) {      // 3 -> d channels: one direct kernel on the frame, no patch matrix (conv_direct.hip)
      DmArena ar(ws, ws_bytes);
      ar.take(DM_SPLITK_FLOATS);
      float* wt = ar.take((size_t)48 * g.d);
      DM_REQUIRE(ar.ok, DM_E_WORKSPACE, "conv_encoder_fwd: workspace too small for the layer-1 weights");
      const size_t frame = (size_t)g.ch * g.hb[0] * g.hb[0];
      const void* img = shape_u8(shp) ? (const void*)((const uint8_t*)image + (size_t)n0 * frame)
                                      : (const void*)(image + (size_t)n0 * frame);
      unsigned short* y0h = dm_twin_of(a.y[0] + r0 * g.cout[0], false);
      DM_TRY(dm_enc_l1_fwd_launch(n, g.d, shape_u8(shp) ? 1 : 0, img, p->w[0], p->b[0], wt, a.y[0] + r0 * g.cout[0], y0h, st));
      if (y0h) dm_twin_mark(a.y[0]);
      continue;
    }
    if (l == 0) {
      const size_t frame = (size_t)g.ch * g.hb[0] * g.hb[0];
      if (shape_u8(shp)) {
        const size_t total = (size_t)n * g.hs[0] * g.hs[0] * g.kdim[0];
        hipLaunchKernelGGL(im2col_s2_u8hwc_kernel, dim3(grid_for(total)), dim3(256), 0, st, n, g.hb[0], g.hb[0], g.cin[0], 4,
                           g.hs[0], g.hs[0], (const uint8_t*)image + (size_t)n0 * frame, a.xcol[0] + r0 * g.kdim[0]);
        DM_LAUNCH_CHECK();
      } else {
        DM_TRY(dm_im2col_s2_launch(n, g.hb[0], g.hb[0], g.cin[0], 4, image + (size_t)n0 * frame, 1,
                                   a.xcol[0] + r0 * g.kdim[0], st));
      }
    }
    DmGemm q;
    q.a_layout = 0; q.b_layout = 0;
    q.M = n * g.hs[l] * g.hs[l]; q.N = g.cout[l]; q.K = (int)g.kdim[l];
    if (l == 0) { q.A = a.xcol[0] + r0 * g.kdim[0]; q.lda = q.K; }
    else {
      q.A = a.y[l - 1]; q.a_maj = a.rowoff[l] + r0; q.a_min = a.koff[l];
      q.a_tab_vec = (g.cin[l] & 3) == 0; q.a_tab_vec8 = (g.cin[l] & 7) == 0;
    }
    q.B = l == 0 ? p->w[0] : a.wr[l]; q.ldb = q.K;
    q.C = a.y[l] + r0 * g.cout[l]; q.ldc = q.N;
    q.bias = p->b[l];
    q.flags = DM_GEMM_ELU;
    DM_TRY(dm_gemm_launch(q, ws, DM_SPLITK_FLOATS * sizeof(float), st));
  }
  const size_t tot = (size_t)n * 4 * g.cout[3];
  hipLaunchKernelGGL(nhwc_to_chw_flat_kernel, dim3(grid_for(tot)), dim3(256), 0, st, (size_t)n, 4, g.cout[3],
                     a.y[3] + (size_t)n0 * 4 * g.cout[3], embed + (size_t)n0 * 4 * g.cout[3], 1);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
extern "C" int dm_conv_encoder_fwd(const dm_shape* shp, const float* image, const dm_conv_params* p, float* acts,
                                   float* embed, void* ws, size_t ws_bytes, void* stream) {
  DM_REQUIRE(shp, DM_E_NULL, "conv_encoder_fwd: null shape");
  return dm_conv_encoder_fwd_rows(shp, 0, shp->T * shp->B * (shp->I > 0 ? shp->I : 1), 1, image, p, acts, embed, ws,
                                  ws_bytes, stream);
}

extern "C" int dm_conv_encoder_bwd(const dm_shape* shp, const float* image, const dm_conv_params* p, const float* acts,
                                   const float* dembed, const dm_conv_grads* gr, void* ws, size_t ws_bytes, void* stream) {
  DM_REQUIRE(shp && p && acts && dembed && gr && ws, DM_E_NULL, "conv_encoder_bwd: null pointer");
  DmPrecisionScope prec(shp->flags & DM_FLAG_BF16);
  EncGeom g(shp);
  DM_REQUIRE(g.valid(shp), DM_E_SHAPE, "conv_encoder: unsupported geometry");
  hipStream_t st = (hipStream_t)stream;
  EncActs a;
  enc_carve(g, const_cast<float*>(acts), (size_t)1 << 60, &a);
  DmArena ar(ws, ws_bytes);
  float* splitk = ar.take(DM_SPLITK_FLOATS);
  const size_t gmax = g.rows[0] * g.cout[0];
  float* ga = ar.take(gmax);
  float* gb = ar.take(g.rows[1] * g.cout[1]);
  float* dwr = ar.take((size_t)g.cout[3] * g.kdim[3]);
  size_t xcmax = 0;
  for (int l = 1; l < 4; ++l) if (g.rows[l] * g.kdim[l] > xcmax) xcmax = g.rows[l] * g.kdim[l];
  float* dxcol = ar.take(xcmax);
  // gather-form data gradient (see convt_* above: the data gradient of a stride-2 convolution IS a transposed convolution)
  size_t padmax = 0, tabmax = 0, wcmax = 0;
  for (int l = 1; l < 4; ++l)
    if (convt_gather_ok(4, g.cout[l], g.cin[l], g.hs[l], (size_t)g.N)) {
      const size_t hp = g.hs[l] + 2, Hc = g.hs[l] + 1;
      if ((size_t)g.N * hp * hp * g.cout[l] > padmax) padmax = (size_t)g.N * hp * hp * g.cout[l];
      if ((size_t)g.N * Hc * Hc > tabmax) tabmax = (size_t)g.N * Hc * Hc;
      if ((size_t)16 * g.cin[l] * g.cout[l] > wcmax) wcmax = (size_t)16 * g.cin[l] * g.cout[l];
    }
  float* gpad = ar.take(padmax);
  int* t_rowoff = (int*)ar.take(tabmax);
  int2* t_ctab = (int2*)ar.take(2 * tabmax);
  int* t_koff = (int*)ar.take(wcmax ? 4 * 1024 : 0);
  float* wcat = ar.take(wcmax);
  // bf16 mode: twins of the two gradient ping-pong buffers and of the class-concatenated weights; the zero-padded gradient
  // copy is written as bf16 only (it lives in the fp32 copy's storage)
  DmTwinScope tw((shp->flags & DM_FLAG_BF16) != 0);
  const bool tw_on = dm_twins_on();
  unsigned short* ga_h = (unsigned short*)ar.take(tw_on ? dm_half_floats(gmax) : 0);
  unsigned short* gb_h = (unsigned short*)ar.take(tw_on ? dm_half_floats(g.rows[1] * g.cout[1]) : 0);
  unsigned short* wcat_h = (unsigned short*)ar.take(tw_on ? dm_half_floats(wcmax) : 0);
  DM_REQUIRE(ar.ok, DM_E_WORKSPACE, "conv_encoder_bwd: workspace too small (need %zu floats)", ar.off);
  const size_t skb = DM_SPLITK_FLOATS * sizeof(float);
  const bool arena_tw = dm_twin_arena_valid(acts);      // the forward that filled `acts` wrote its twins
  enc_register_twins(g, a, arena_tw, arena_tw);
  dm_twin_add(ga, gmax, ga_h, false);
  dm_twin_add(gb, g.rows[1] * g.cout[1], gb_h, false);

  // dY3 (NHWC) = permute(dembed) ; G = dY3 * ELU'(Y3)
  float* G = gb;      // layer-3 grads are small; ping-pong between ga / gb going down
  const size_t tot3 = (size_t)g.N * 4 * g.cout[3];
  hipLaunchKernelGGL(nhwc_to_chw_flat_kernel, dim3(grid_for(tot3)), dim3(256), 0, st, (size_t)g.N, 4, g.cout[3], dembed, G,
                     0);
  DM_LAUNCH_CHECK();
  DM_TRY(dm_mul_elu_grad_launch(tot3, G, a.y[3], G, st));
  if (tw_on) {
    const DmCvtSeg sg = {G, dm_twin_of(G, false), tot3};
    DM_TRY(dm_to_bf16_multi_launch(&sg, 1, st));
    dm_twin_mark(G);
  }
  for (int l = 3; l >= 0; --l) {
    const int rows = (int)g.rows[l], co = g.cout[l], kd = (int)g.kdim[l];
    DM_TRY(dm_colsum_launch(rows, co, G, co, gr->b[l], splitk, skb, st));
    if (l == 0 && g.direct0) {      // patches re-gathered from the frame inside the weight-gradient kernel (conv_direct.hip)
      DM_REQUIRE(image, DM_E_NULL, "conv_encoder_bwd: the direct layer-1 weight gradient reads the image");
      DM_REQUIRE(dm_enc_l1_wgrad_part_floats(g.N, g.d) <= xcmax, DM_E_WORKSPACE, "conv_encoder_bwd: partial buffer too small");
      DM_TRY(dm_enc_l1_wgrad_launch(g.N, g.d, shape_u8(shp) ? 1 : 0, image, G, dxcol, gr->w[0], splitk, skb, st));
      continue;
    }
    DmGemm q;   // dWr[o][kidx] = sum_rows G[row][o] * Xcol[row][kidx]
    q.a_layout = 1; q.b_layout = 1;
    q.M = co; q.N = kd; q.K = rows;
    q.A = G; q.lda = co;
    if (l == 0) { q.B = a.xcol[0]; q.ldb = kd; }
    else {
      q.B = a.y[l - 1]; q.b_maj = a.rowoff[l]; q.b_min = a.koff[l];
      q.b_tab_vec = (g.cin[l] & 3) == 0; q.b_tab_vec8 = (g.cin[l] & 7) == 0;
    }
    q.C = (l == 0) ? gr->w[0] : dwr; q.ldc = kd;
    DM_TRY(dm_gemm_launch(q, splitk, skb, st));
    if (l > 0 && convt_gather_ok(4, co, g.cin[l], g.hs[l], (size_t)g.N)) {
      // dX[n, 2yy+py, 2xx+px, i] = ELU'(Y_{l-1}) * sum_{a,b,o} G[n, yy-a, xx-b, o] W[o][i][py+2a][px+2b]: one implicit GEMM
      // over a zero-padded copy of G, scattered straight into the NHWC gradient of layer l-1 (no dXcol, no col2im)
      DM_TRY(dm_permute4_launch(dwr, gr->w[l], co, 4, 4, g.cin[l], 0, 3, 1, 2, st));
      const int hs = g.hs[l], hb = g.hb[l], Hc = hs + 1, kdim4 = 4 * co, ci = g.cin[l];
      DM_REQUIRE(kdim4 <= 4 * 1024, DM_E_SHAPE, "conv_encoder_bwd: gather-form K %d exceeds the offset table", kdim4);
      float* Gn = (G == ga) ? gb : ga;
      if (l == 1) Gn = ga;   // layer-0 output grads are the largest buffer
      const size_t padn = (size_t)g.N * (hs + 2) * (hs + 2) * (co / 4);
      const bool hpath = tw_on && (co & 7) == 0 && kdim4 >= DM_HSTORE_MIN_K;      // bf16-storage product: padded gradient + weights as bf16
      if (hpath) hipLaunchKernelGGL(convt_pad_h_kernel, dim3(grid_for(padn)), dim3(256), 0, st, g.N, hs, hs, co / 4, 1, (const float4*)G,
                                    (uint2*)gpad);
      else hipLaunchKernelGGL(convt_pad_kernel, dim3(grid_for(padn)), dim3(256), 0, st, g.N, hs, hs, co / 4, 1, (const float4*)G,
                              (float4*)gpad);
      DM_LAUNCH_CHECK();
      hipLaunchKernelGGL(convt_tables_kernel, dim3(grid_for((size_t)g.N * Hc * Hc + kdim4)), dim3(256), 0, st, g.N, hs, hs, co, 2,
                         hb, hb, ci, t_rowoff, t_koff, t_ctab);
      DM_LAUNCH_CHECK();
      hipLaunchKernelGGL(convt_repack_kernel, dim3(grid_for((size_t)4 * ci * kdim4)), dim3(256), 0, st, co, ci, 4, p->w[l], wcat);
      DM_LAUNCH_CHECK();
      if (hpath) {
        const DmCvtSeg sg = {wcat, wcat_h, (size_t)4 * ci * kdim4};
        DM_TRY(dm_to_bf16_multi_launch(&sg, 1, st));
      }
      if (2 * Hc != hb) {      // odd input extent (31): the last row / column is reached by no window - its gradient is zero
        hipError_t e = hipMemsetAsync(Gn, 0, (size_t)g.N * hb * hb * ci * sizeof(float), st);
        if (e == hipSuccess && dm_twin_of(Gn, false))
          e = hipMemsetAsync(dm_twin_of(Gn, false), 0, (size_t)g.N * hb * hb * ci * sizeof(unsigned short), st);
        if (e != hipSuccess) return dm_fail(DM_E_HIP, "conv_encoder_bwd: %s", hipGetErrorString(e));
      }
      DmGemm d;
      d.M = g.N * Hc * Hc; d.N = 4 * ci; d.K = kdim4;
      d.A = gpad; d.a_maj = t_rowoff; d.a_min = t_koff; d.a_tab_vec = 1; d.a_tab_vec8 = 1;
      if (hpath) { d.A = nullptr; d.A_h = (const unsigned short*)gpad; d.B_h = wcat_h; }
      d.B = wcat; d.ldb = kdim4;
      d.C = Gn;
      d.c_tab = t_ctab; d.sc_cout = ci; d.sc_wpitch = hb * ci;
      d.mulref = a.y[l - 1];
      d.no_twin = l == 1;      // the layer-0 gradient is read by the direct weight-gradient kernel and the bias sum only (fp32)
      DM_TRY(dm_gemm_launch(d, splitk, skb, st));
      G = Gn;
    } else if (l > 0) {
      DM_TRY(dm_permute4_launch(dwr, gr->w[l], co, 4, 4, g.cin[l], 0, 3, 1, 2, st));
      DmGemm d;   // dXcol[row][kidx] = sum_o G[row][o] * Wr[o][kidx]
      d.a_layout = 0; d.b_layout = 1;
      d.M = rows; d.N = kd; d.K = co;
      d.A = G; d.lda = co;
      d.B = a.wr[l]; d.ldb = kd;
      d.C = dxcol; d.ldc = kd;
      DM_TRY(dm_gemm_launch(d, splitk, skb, st));
      float* Gn = (G == ga) ? gb : ga;
      if (l == 1) Gn = ga;   // layer-0 output grads are the largest buffer
      DM_TRY(dm_col2im_s2_launch(g.N, g.hb[l], g.hb[l], g.cin[l], 4, dxcol, nullptr, 0, a.y[l - 1], Gn, st));
      G = Gn;
    }
  }
  return DM_OK;
}

// 4-channel trick for the image layer's backward (cout = 3): the output gradient is written with 4 channels per pixel (the
// 4th = 0) and the weights are padded to match, so its patches are 16-byte gathers like every other layer's and the
// explicit patch matrix of the 3-channel gradient (972 MB written + read at Atari-literal) is never built.
// dst (I, k, k, O4) <- src (I, O, k, k)  (torch ConvTranspose2d layout), zero for o >= O
__global__ void __launch_bounds__(256) convt_pad_cout_kernel(int I, int O, int O4, int k, const float* __restrict__ src,
                                                             float* __restrict__ dst) {
  const int total = I * k * k * O4;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int o = e % O4, kx = (e / O4) % k, ky = (e / (O4 * k)) % k, i = e / (O4 * k * k);
    dst[e] = o < O ? src[(((size_t)i * O + o) * k + ky) * k + kx] : 0.f;
  }
}
// dst (I, O, k, k) <- src (I, k, k, O4), dropping the pad channels
__global__ void __launch_bounds__(256) convt_unpad_cout_kernel(int I, int O, int O4, int k, const float* __restrict__ src,
                                                               float* __restrict__ dst) {
  const int total = I * O * k * k;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int kx = e % k, ky = (e / k) % k, o = (e / (k * k)) % O, i = e / (k * k * O);
    dst[e] = src[(((size_t)i * k + ky) * k + kx) * O4 + o];
  }
}

// ---------------------------------------------------------------- decoder -----------------------
struct DecActs {
  float* wr[5];
  float* x[5];     // x[0] = fc output (N,32d); x[l] = NHWC activations after layer l (x[4] = prediction)
  unsigned short* xh[5];     // bf16 mode: twins of x[0..2] and wr[1..4], behind everything else (dm_conv_decoder_pred_offset holds)
  unsigned short* wrh[5];
};
static inline size_t dec_x_elems(const DecGeom& g, int l) { return l == 0 ? (size_t)g.N * g.cin[1] : g.rows_b[l] * g.cout[l]; }
static inline size_t dec_w_elems(const DecGeom& g, int l) { return (size_t)g.cin[l] * g.k[l] * g.k[l] * g.cout[l]; }
static size_t dec_carve(const DecGeom& g, float* base, size_t cap_floats, DecActs* a) {
  DmArena ar(base, cap_floats * sizeof(float));
  float* x0 = ar.take((size_t)g.N * g.cin[1]);
  if (a) a->x[0] = x0;
  for (int l = 1; l <= 4; ++l) {
    float* w = ar.take((size_t)g.cin[l] * g.k[l] * g.k[l] * g.cout[l]);
    float* xx = ar.take(g.rows_b[l] * g.cout[l]);
    if (a) { a->wr[l] = w; a->x[l] = xx; }
  }
  for (int l = 0; l <= 4; ++l) {
    float* xh = ar.take(g.twins && l < 3 ? dm_half_floats(dec_x_elems(g, l)) : 0);
    float* wh = ar.take(g.twins && l > 0 ? dm_half_floats(dec_w_elems(g, l)) : 0);
    if (a) { a->xh[l] = (unsigned short*)xh; a->wrh[l] = (unsigned short*)wh; }
  }
  return ar.off;
}
static void dec_register_twins(const DecGeom& g, const DecActs& a, bool acts_valid, bool weights_valid) {
  if (!dm_twins_on()) return;
  // (x[3], the 30x30xd input of the image layer, gets none: writing it costs more than its one reader, that layer's weight gradient, gains)
  for (int l = 0; l < 3; ++l) dm_twin_add(a.x[l], dec_x_elems(g, l), a.xh[l], acts_valid);
  for (int l = 1; l <= 4; ++l) dm_twin_add(a.wr[l], dec_w_elems(g, l), a.wrh[l], weights_valid);
}
extern "C" size_t dm_conv_decoder_acts_floats(const dm_shape* shp) {
  if (!shp) return 0;
  DecGeom g(shp);
  return dec_carve(g, nullptr, 0, nullptr);
}

// Float offset, inside the `acts` buffer of dm_conv_decoder_mse_fwd, of the decoder's prediction (N, img, img, ch) NHWC:
// lets the caller materialise `image_rec` (decoders.py:177) only when somebody reads it.
extern "C" size_t dm_conv_decoder_pred_offset(const dm_shape* shp) {
  if (!shp) return 0;
  DecGeom g(shp);
  size_t off = dm_align_up((size_t)g.N * g.cin[1], 64);                  // mirrors dec_carve
  for (int l = 1; l <= 4; ++l) {
    off += dm_align_up((size_t)g.cin[l] * g.k[l] * g.k[l] * g.cout[l], 64);
    if (l == 4) break;
    off += dm_align_up(g.rows_b[l] * g.cout[l], 64);
  }
  return off;
}

// Frames [n0, n0+n); buffers are the full-batch ones (see dm_conv_encoder_fwd_rows).  The patch-matrix workspace
// scales with n, so a range stream needs only 1/chunks of the full-batch workspace.
extern "C" int dm_conv_decoder_mse_fwd_rows(const dm_shape* shp, int n0, int n, int prepare, const float* feat, int ldf,
                                            const float* target, const dm_conv_params* p, float* acts, float* loss_image,
                                            float* image_rec, void* ws, size_t ws_bytes, void* stream) {
  DM_REQUIRE(shp && feat && target && p && acts && ws, DM_E_NULL, "conv_decoder_fwd: null pointer");
  DmPrecisionScope prec(shp->flags & DM_FLAG_BF16);
  DecGeom g(shp);
  DM_REQUIRE(g.valid(shp), DM_E_SHAPE, "conv_decoder: unsupported geometry (img=%d)", shp->img);
  DM_REQUIRE(n0 >= 0 && n >= 0 && n0 + n <= g.N, DM_E_SHAPE, "conv_decoder_fwd: frame range [%d,%d) outside 0..%d", n0,
             n0 + n, g.N);
  hipStream_t st = (hipStream_t)stream;
  DecActs a;
  dec_carve(g, acts, (size_t)1 << 60, &a);
  DmTwinScope tw((shp->flags & DM_FLAG_BF16) != 0);
  const bool tw_on = dm_twins_on();
  dec_register_twins(g, a, false, !prepare && dm_twin_arena_valid(acts));
  dm_twin_arena_note(acts, tw_on);
  if (prepare) {
    for (int l = 1; l <= 4; ++l)
      DM_TRY(dm_permute4_launch(p->w[l], a.wr[l], g.cin[l], g.cout[l], g.k[l], g.k[l], 0, 2, 3, 1, st));
    if (tw_on) {
      DmCvtSeg sg[4];
      for (int l = 1; l <= 4; ++l) { sg[l - 1] = DmCvtSeg{a.wr[l], a.wrh[l], dec_w_elems(g, l)}; dm_twin_mark(a.wr[l]); }
      DM_TRY(dm_to_bf16_multi_launch(sg, 4, st));
    }
  }
  if (n == 0) return DM_OK;
  DmArena ar(ws, ws_bytes);
  float* splitk = ar.take(DM_SPLITK_FLOATS);
  const bool direct4 = dm_dec_l4_direct_ok(g.ch, g.d, g.hsm[4], g.k[4]);      // d -> 3 channels: conv_direct.hip, no column matrix
  size_t colmax = 0;
  for (int l = 1; l <= (direct4 ? 3 : 4); ++l) {
    const size_t c = (size_t)n * g.hsm[l] * g.hsm[l] * g.k[l] * g.k[l] * g.cout[l];
    if (c > colmax) colmax = c;
  }
  float* ycol = ar.take(colmax);
  float* w4 = ar.take(direct4 ? dm_dec_l4_w4_floats(g.d) : 0);
  // gather-form layers: zero-padded input copy, gather / scatter tables, class-concatenated weights
  size_t padmax = 0, tabmax = 0, wcmax = 0;
  for (int l = 1; l <= 4; ++l)
    if (convt_gather_ok(g.k[l], g.cin[l], g.cout[l], g.hsm[l], (size_t)n)) {
      const int ta = g.k[l] / 2, hp = g.hsm[l] + 2 * (ta - 1), Hc = g.hsm[l] + ta - 1;
      const size_t pad = (size_t)n * hp * hp * g.cin[l], tab = (size_t)n * Hc * Hc, wc = (size_t)4 * g.cout[l] * ta * ta * g.cin[l];
      if (pad > padmax) padmax = pad;
      if (tab > tabmax) tabmax = tab;
      if (wc > wcmax) wcmax = wc;
    }
  float* xpad = ar.take(padmax);
  int* t_rowoff = (int*)ar.take(tabmax);
  int2* t_ctab = (int2*)ar.take(2 * tabmax);
  int* t_koff = (int*)ar.take(wcmax ? 9 * 1024 : 0);
  float* wcat = ar.take(wcmax);
  unsigned short* wcat_h = (unsigned short*)ar.take(tw_on ? dm_half_floats(wcmax) : 0);
  DM_REQUIRE(ar.ok, DM_E_WORKSPACE, "conv_decoder_fwd: workspace too small (need %zu floats)", ar.off);
  const size_t skb = DM_SPLITK_FLOATS * sizeof(float);
  {
    DmGemm q;   // x0 = feat W^T + b
    q.M = n; q.N = g.cin[1]; q.K = g.F;
    q.A = feat + (size_t)n0 * ldf; q.lda = ldf;
    q.B = p->w[0]; q.ldb = g.F;
    q.C = a.x[0] + (size_t)n0 * g.cin[1]; q.ldc = q.N;
    q.bias = p->b[0];
    DM_TRY(dm_gemm_launch(q, splitk, skb, st));
  }
  for (int l = 1; l <= 4; ++l) {
    const int kk = g.k[l] * g.k[l];
    const float* xin = l == 1 ? a.x[0] + (size_t)n0 * g.cin[1]
                              : a.x[l - 1] + (size_t)n0 * g.hbg[l - 1] * g.hbg[l - 1] * g.cout[l - 1];
    if (l == 4 && direct4) {
      DM_TRY(dm_dec_l4_fwd_launch(n, g.d, xin, p->w[4], p->b[4], w4, a.x[4] + (size_t)n0 * g.hbg[4] * g.hbg[4] * g.cout[4], st));
      continue;
    }
    if (convt_gather_ok(g.k[l], g.cin[l], g.cout[l], g.hsm[l], (size_t)n)) {
      const int ta = g.k[l] / 2, hs = g.hsm[l], hb = g.hbg[l], Hc = hs + ta - 1, kdim = ta * ta * g.cin[l];
      DM_REQUIRE(kdim <= 9 * 1024, DM_E_SHAPE, "conv_decoder_fwd: gather-form K %d exceeds the offset table", kdim);
      const size_t padn = (size_t)n * (hs + 2 * (ta - 1)) * (hs + 2 * (ta - 1)) * (g.cin[l] / 4);
      const bool hpath = tw_on && (g.cin[l] & 7) == 0 && kdim >= DM_HSTORE_MIN_K;       // bf16-storage product: padded input + weights as bf16
      if (hpath) hipLaunchKernelGGL(convt_pad_h_kernel, dim3(grid_for(padn)), dim3(256), 0, st, n, hs, hs, g.cin[l] / 4, ta - 1,
                                    (const float4*)xin, (uint2*)xpad);
      else hipLaunchKernelGGL(convt_pad_kernel, dim3(grid_for(padn)), dim3(256), 0, st, n, hs, hs, g.cin[l] / 4, ta - 1,
                              (const float4*)xin, (float4*)xpad);
      DM_LAUNCH_CHECK();
      hipLaunchKernelGGL(convt_tables_kernel, dim3(grid_for((size_t)n * Hc * Hc + kdim)), dim3(256), 0, st, n, hs, hs, g.cin[l],
                         ta, hb, hb, g.cout[l], t_rowoff, t_koff, t_ctab);
      DM_LAUNCH_CHECK();
      hipLaunchKernelGGL(convt_repack_kernel, dim3(grid_for((size_t)4 * g.cout[l] * kdim)), dim3(256), 0, st, g.cin[l], g.cout[l],
                         g.k[l], p->w[l], wcat);
      DM_LAUNCH_CHECK();
      if (hpath) {
        const DmCvtSeg sg = {wcat, wcat_h, (size_t)4 * g.cout[l] * kdim};
        DM_TRY(dm_to_bf16_multi_launch(&sg, 1, st));
      }
      DmGemm q;   // big[n, 2yy+py, 2xx+px, o] = act( b[o] + patch(n,yy,xx) . Wcat[(py,px,o)] )
      q.M = n * Hc * Hc; q.N = 4 * g.cout[l]; q.K = kdim;
      q.A = xpad; q.a_maj = t_rowoff; q.a_min = t_koff; q.a_tab_vec = 1; q.a_tab_vec8 = 1;
      if (hpath) { q.A = nullptr; q.A_h = (const unsigned short*)xpad; q.B_h = wcat_h; }
      q.B = wcat; q.ldb = kdim;
      q.C = a.x[l] + (size_t)n0 * hb * hb * g.cout[l];
      q.c_tab = t_ctab; q.sc_cout = g.cout[l]; q.sc_wpitch = hb * g.cout[l];
      q.bias = p->b[l];
      q.flags = l < 4 ? DM_GEMM_ELU : 0;
      DM_TRY(dm_gemm_launch(q, splitk, skb, st));
      continue;
    }
    DmGemm q;   // Ycol[(n,ys,xs)][(ky,kx,o)] = X[(n,ys,xs)][i] * Wr[i][(ky,kx,o)]
    q.a_layout = 0; q.b_layout = 1;
    q.M = n * g.hsm[l] * g.hsm[l]; q.N = kk * g.cout[l]; q.K = g.cin[l];
    q.A = xin; q.lda = q.K;
    q.B = a.wr[l]; q.ldb = q.N;
    q.C = ycol; q.ldc = q.N;
    if (g.hsm[l] == 1) {     // a 1x1 input: no windows overlap, the column matrix IS the NHWC output - bias + ELU in the epilogue
      q.C = a.x[l] + (size_t)n0 * g.hbg[l] * g.hbg[l] * g.cout[l];
      q.bias = p->b[l]; q.bias_mod = g.cout[l];
      q.flags = l < 4 ? DM_GEMM_ELU : 0;
      DM_TRY(dm_gemm_launch(q, splitk, skb, st));
      continue;
    }
    DM_TRY(dm_gemm_launch(q, splitk, skb, st));
    DM_TRY(dm_col2im_s2_launch(n, g.hbg[l], g.hbg[l], g.cout[l], g.k[l], ycol, p->b[l], l < 4 ? DM_C2I_ELU : 0, nullptr,
                               a.x[l] + (size_t)n0 * g.hbg[l] * g.hbg[l] * g.cout[l], st));
  }
  if (loss_image || image_rec) {
    // targets are indexed by frame / I (I = iwae_samples, decoders.py:163-167), so the kernel gets the full target base
    const size_t per = (size_t)g.hbg[4] * g.hbg[4] * g.ch;
    const int I = shp->I > 0 ? shp->I : 1;
    DM_REQUIRE(n0 % I == 0, DM_E_SHAPE, "conv_decoder_fwd: frame range must start on a multiple of iwae_samples");
    const void* tbase = shape_u8(shp) ? (const void*)((const uint8_t*)target + (size_t)(n0 / I) * per)
                                      : (const void*)(target + (size_t)(n0 / I) * per);
    DM_TRY(mse_launch(shape_u8(shp), n, g.hbg[4] * g.hbg[4], g.ch, a.x[4] + n0 * per, tbase, I, 0.f, nullptr,
                      loss_image ? loss_image + n0 : nullptr, nullptr, g.ch, image_rec ? image_rec + n0 * per : nullptr, st));
  }
  return DM_OK;
}
extern "C" int dm_conv_decoder_mse_fwd(const dm_shape* shp, const float* feat, int ldf, const float* target,
                                       const dm_conv_params* p, float* acts, float* loss_image, float* image_rec,
                                       void* ws, size_t ws_bytes, void* stream) {
  DM_REQUIRE(shp, DM_E_NULL, "conv_decoder_fwd: null shape");
  return dm_conv_decoder_mse_fwd_rows(shp, 0, shp->T * shp->B * (shp->I > 0 ? shp->I : 1), 1, feat, ldf, target, p, acts,
                                      loss_image, image_rec, ws, ws_bytes, stream);
}

static int conv_decoder_mse_bwd_impl(const dm_shape* shp, const float* feat, int ldf, const float* target,
                                     const dm_conv_params* p, const float* acts, float scale, const float* row_scale,
                                     const dm_conv_grads* gr, float* dfeat, int lddf, void* ws, size_t ws_bytes, void* stream);
extern "C" int dm_conv_decoder_mse_bwd(const dm_shape* shp, const float* feat, int ldf, const float* target,
                                       const dm_conv_params* p, const float* acts, float scale, const dm_conv_grads* gr,
                                       float* dfeat, int lddf, void* ws, size_t ws_bytes, void* stream) {
  return conv_decoder_mse_bwd_impl(shp, feat, ldf, target, p, acts, scale, nullptr, gr, dfeat, lddf, ws, ws_bytes, stream);
}
// row_scale (N floats, nullable): per-frame factor on the image-loss gradient - the IWAE importance weights of
// loss_model = -logavgexp(-loss_tbi) (dreamer.py:362-365, functions.py:97-102)
extern "C" int dm_conv_decoder_mse_bwd_rows(const dm_shape* shp, const float* feat, int ldf, const float* target,
                                            const dm_conv_params* p, const float* acts, float scale, const float* row_scale,
                                            const dm_conv_grads* gr, float* dfeat, int lddf, void* ws, size_t ws_bytes,
                                            void* stream) {
  return conv_decoder_mse_bwd_impl(shp, feat, ldf, target, p, acts, scale, row_scale, gr, dfeat, lddf, ws, ws_bytes, stream);
}
static int conv_decoder_mse_bwd_impl(const dm_shape* shp, const float* feat, int ldf, const float* target,
                                     const dm_conv_params* p, const float* acts, float scale, const float* row_scale,
                                     const dm_conv_grads* gr, float* dfeat, int lddf, void* ws, size_t ws_bytes, void* stream) {
  DM_REQUIRE(shp && feat && target && p && acts && gr && ws, DM_E_NULL, "conv_decoder_bwd: null pointer");
  DmPrecisionScope prec(shp->flags & DM_FLAG_BF16);
  DecGeom g(shp);
  DM_REQUIRE(g.valid(shp), DM_E_SHAPE, "conv_decoder: unsupported geometry");
  hipStream_t st = (hipStream_t)stream;
  DecActs a;
  dec_carve(g, const_cast<float*>(acts), (size_t)1 << 60, &a);
  DmArena ar(ws, ws_bytes);
  float* splitk = ar.take(DM_SPLITK_FLOATS);
  // every layer's output gradient is gathered implicitly: channel counts that are not a multiple of 4 (the 3-channel image
  // layer) are padded to 4 (co4) in the gradient buffer and in a padded copy of the weights
  int co4[5] = {0, 0, 0, 0, 0};
  size_t gmax = 0, wmax = 0, wpadmax = 0;
  for (int l = 1; l <= 4; ++l) {
    co4[l] = (g.cout[l] + 3) & ~3;
    if (g.rows_b[l] * co4[l] > gmax) gmax = g.rows_b[l] * co4[l];
    const size_t w = (size_t)g.cin[l] * g.k[l] * g.k[l] * co4[l];
    if (w > wmax) wmax = w;
    if (co4[l] != g.cout[l] && w > wpadmax) wpadmax = w;
  }
  if ((size_t)g.N * g.cin[1] > gmax) gmax = (size_t)g.N * g.cin[1];
  // One gradient buffer per layer (together smaller than a ping-pong pair of the largest) and one gather table pair per layer:
  // a layer's weight / bias gradient only reads G[l], its tables and the saved activations, so those launches can sit on
  // the weight-gradient side stream (common.h, dm_wgrad_side_*) and run beside whatever follows the DATA-gradient chain on
  // `st` - the BPTT loop - instead of in front of it.  Not armed: sw == st and the order below is simply "as enqueued".
  float* Gl[5];
  unsigned short* Gl_h[5];
  size_t gsz[5];
  gsz[0] = (size_t)g.N * g.cin[1];
  for (int l = 1; l <= 4; ++l) gsz[l] = g.rows_b[l] * co4[l];
  for (int l = 0; l <= 4; ++l) Gl[l] = ar.take(gsz[l]);
  (void)gmax;
  float* dwr = ar.take(wmax);
  float* wpad = ar.take(wpadmax);
  float* bsum = ar.take(64);
  int* rowoff_l[5];
  int* koff_l[5];
  for (int l = 1; l <= 4; ++l) {
    rowoff_l[l] = (int*)ar.take(g.rows_s[l]);
    koff_l[l] = (int*)ar.take((size_t)g.k[l] * g.k[l] * co4[l]);
  }
  float* splitk_w = ar.take(DM_SPLITK_FLOATS);          // the side stream's own split-K scratch
  // image layer (d -> 3 channels): both backward products as direct kernels (conv_direct.hip) reading the padded gradient
  const bool direct4 = dm_dec_l4_bwd_direct_ok(g.ch, g.d, g.hsm[4], g.k[4]);
  float* l4_wp = ar.take(direct4 ? dm_dec_l4_wp_floats(g.d) : 0);
  float* l4_part = ar.take(direct4 ? dm_dec_l4_wgrad_part_floats(g.N, g.d) : 0);
  // bf16 mode: twins of the gradient buffers and of the padded image-layer weights
  DmTwinScope tw((shp->flags & DM_FLAG_BF16) != 0);
  const bool tw_on = dm_twins_on();
  for (int l = 0; l <= 4; ++l) Gl_h[l] = (unsigned short*)ar.take(tw_on ? dm_half_floats(gsz[l]) : 0);
  unsigned short* wpad_h = (unsigned short*)ar.take(tw_on ? dm_half_floats(wpadmax) : 0);
  DM_REQUIRE(ar.ok, DM_E_WORKSPACE, "conv_decoder_bwd: workspace too small (need %zu floats)", ar.off);
  const size_t skb = DM_SPLITK_FLOATS * sizeof(float);
  const bool arena_tw = dm_twin_arena_valid(acts);      // the forward that filled `acts` wrote its twins
  dec_register_twins(g, a, arena_tw, arena_tw);
  for (int l = 0; l <= 4; ++l)
    if (!(direct4 && l == 4)) dm_twin_add(Gl[l], gsz[l], Gl_h[l], false);      // (the direct kernels read the fp32 gradient)
  dm_twin_add(wpad, wpadmax, wpad_h, false);
  hipStream_t sw = dm_wgrad_side_stream(st);

  // G4 = scale * (pred - target), NHWC with co4[4] channels per pixel
  float* G = Gl[4];
  DM_TRY(mse_launch(shape_u8(shp), g.N, g.hbg[4] * g.hbg[4], g.ch, a.x[4], (const void*)target, shp->I > 0 ? shp->I : 1, scale,
                    row_scale, nullptr, G, co4[4], nullptr, st));
  for (int l = 4; l >= 1; --l) {
    const int kk = g.k[l] * g.k[l];
    const int co = co4[l];
    const bool padded = co != g.cout[l];
    const int ncol = kk * co;
    const int rows_s = (int)g.rows_s[l];
    int* rowoff = rowoff_l[l];
    int* koff = koff_l[l];
    float* Gn = Gl[l - 1];
    if (direct4 && l == 4) {
      DM_TRY(dm_wgrad_side_fork(st, sw));
      unsigned short* gnh = dm_twin_of(Gn, false);
      DM_TRY(dm_dec_l4_dgrad_launch(g.N, g.d, G, p->w[4], l4_wp, a.x[3], Gn, gnh, st));
      if (gnh) dm_twin_mark(Gn);
      DM_TRY(dm_colsum_launch((int)g.rows_b[l], co, G, co, bsum, splitk_w, skb, sw));
      hipError_t e = hipMemcpyAsync(gr->b[l], bsum, (size_t)g.cout[l] * sizeof(float), hipMemcpyDeviceToDevice, sw);
      if (e != hipSuccess) return dm_fail(DM_E_HIP, "conv_decoder_bwd: %s", hipGetErrorString(e));
      DM_TRY(dm_dec_l4_wgrad_launch(g.N, g.d, G, a.x[3], l4_part, gr->w[4], splitk_w, skb, sw));
      G = Gn;
      continue;
    }
    // patches of the output gradient, gathered implicitly (16-byte gathers: co is a multiple of 4)
    DM_TRY(conv_tables_launch(g.N, g.hbg[l], g.hbg[l], co, g.k[l], rowoff, koff, st));
    DM_TRY(dm_wgrad_side_fork(st, sw));                  // G[l] and its tables are enqueued: the side stream may read them
    // ---- data gradient (the chain the BPTT loop waits for), on st
    if (padded) {
      hipLaunchKernelGGL(convt_pad_cout_kernel, dim3(grid_for((size_t)g.cin[l] * ncol)), dim3(256), 0, st, g.cin[l], g.cout[l], co,
                         g.k[l], p->w[l], wpad);
      DM_LAUNCH_CHECK();
      if (tw_on) {
        const DmCvtSeg sg = {wpad, wpad_h, (size_t)g.cin[l] * ncol};
        DM_TRY(dm_to_bf16_multi_launch(&sg, 1, st));
        dm_twin_mark(wpad);
      }
    }
    {
      DmGemm d;   // dX[row][i] = sum_col dYcol[row][col] * Wr[i][col]   (* ELU'(X_{l-1}) for l-1 >= 1)
      d.a_layout = 0; d.b_layout = 0;
      d.M = rows_s; d.N = g.cin[l]; d.K = ncol;
      d.A = G; d.a_maj = rowoff; d.a_min = koff; d.a_tab_vec = 1;
      d.a_tab_vec8 = (co & 7) == 0 || (co == 4 && (g.k[l] & 1) == 0);
      d.B = padded ? wpad : a.wr[l]; d.ldb = ncol;
      d.C = Gn; d.ldc = g.cin[l];
      if (l - 1 >= 1) { d.mulref = a.x[l - 1]; d.ldmul = g.cin[l]; }
      DM_TRY(dm_gemm_launch(d, splitk, skb, st));
    }
    // ---- bias and weight gradient, on sw
    if (padded) {
      DM_TRY(dm_colsum_launch((int)g.rows_b[l], co, G, co, bsum, splitk_w, skb, sw));
      hipError_t e = hipMemcpyAsync(gr->b[l], bsum, (size_t)g.cout[l] * sizeof(float), hipMemcpyDeviceToDevice, sw);
      if (e != hipSuccess) return dm_fail(DM_E_HIP, "conv_decoder_bwd: %s", hipGetErrorString(e));
    } else {
      DM_TRY(dm_colsum_launch((int)g.rows_b[l], co, G, co, gr->b[l], splitk_w, skb, sw));
    }
    DmGemm q;   // dWr[i][(ky,kx,o)] = sum_rows X[row][i] * dYcol[row][(ky,kx,o)]
    q.a_layout = 1; q.b_layout = 1;
    q.M = g.cin[l]; q.N = ncol; q.K = rows_s;
    q.A = a.x[l - 1]; q.lda = g.cin[l];
    q.B = G; q.b_maj = rowoff; q.b_min = koff; q.b_tab_vec = 1;
    q.b_tab_vec8 = (co & 7) == 0 || (co == 4 && (g.k[l] & 1) == 0);      // 8 minors = 8 channels, or 2 pixels x 4 channels of an even-width window
    q.C = dwr; q.ldc = ncol;
    DM_TRY(dm_gemm_launch(q, splitk_w, skb, sw));
    if (padded) {
      hipLaunchKernelGGL(convt_unpad_cout_kernel, dim3(grid_for((size_t)g.cin[l] * kk * g.cout[l])), dim3(256), 0, sw, g.cin[l],
                         g.cout[l], co, g.k[l], dwr, gr->w[l]);
      DM_LAUNCH_CHECK();
    } else {
      DM_TRY(dm_permute4_launch(dwr, gr->w[l], g.cin[l], g.k[l], g.k[l], g.cout[l], 0, 3, 1, 2, sw));
    }
    G = Gn;
  }
  // fc: x0 = feat W^T + b
  const int O = g.cin[1];
  DM_TRY(dm_wgrad_side_fork(st, sw));
  if (dfeat) {
    DmGemm d;   // dfeat[n][f] += sum_o G[n][o] W[o][f]
    d.a_layout = 0; d.b_layout = 1;
    d.M = g.N; d.N = g.F; d.K = O;
    d.A = G; d.lda = O;
    d.B = p->w[0]; d.ldb = g.F;
    d.C = dfeat; d.ldc = lddf;
    d.flags = DM_GEMM_ACCUM;
    DM_TRY(dm_gemm_launch(d, splitk, skb, st));
  }
  DM_TRY(dm_colsum_launch(g.N, O, G, O, gr->b[0], splitk_w, skb, sw));
  {
    DmGemm q;   // dW[o][f] = sum_n G[n][o] feat[n][f]
    q.a_layout = 1; q.b_layout = 1;
    q.M = O; q.N = g.F; q.K = g.N;
    q.A = G; q.lda = O;
    q.B = feat; q.ldb = ldf;
    q.C = gr->w[0]; q.ldc = g.F;
    DM_TRY(dm_gemm_launch(q, splitk_w, skb, sw));
  }
  DM_TRY(dm_wgrad_side_mark(sw, st));
  return DM_OK;
}
