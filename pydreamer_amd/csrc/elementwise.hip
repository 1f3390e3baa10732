// Row-wise / element-wise kernels of the DreamerV2 step (HBM- or latency-bound; no MFMA here):
// LayerNorm+ELU, column sums, GRU gate math, categorical sampler, KL, straight-through backward, masks,
// head losses, GAE, actor / critic losses, multi-array reductions.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// LayerNorm(eps) + ELU, one wave per row (common.py:44-49; rssm.py:105-115; torch LayerNorm: biased variance)
// ------------------------------------------------------------------------------------------------
// CACHE = true keeps the row in registers (n <= 1024: 16 values per lane), so x / y / dy are read from memory once.
template <bool CACHE>
__global__ void __launch_bounds__(256) ln_elu_fwd_kernel(int rows, int n, const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, float* __restrict__ y, int ldy, float* __restrict__ stats,
                                                         unsigned short* __restrict__ y_h) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  unsigned short* yh = y_h ? y_h + (size_t)row * ldy : nullptr;      // bf16 twin of y, same leading dimension (common.h DmTwinScope)
  float xc[16];
  float s = 0.f;
  if (CACHE) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = lane + 64 * j;
      xc[j] = c < n ? xr[c] : 0.f;
      s += xc[j];
    }
  } else {
    for (int c = lane; c < n; c += 64) s += xr[c];
  }
  const float mean = dm_wave_sum(s) / (float)n;
  float v = 0.f;
  if (CACHE) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float d = (lane + 64 * j < n) ? xc[j] - mean : 0.f;
      v += d * d;
    }
  } else {
    for (int c = lane; c < n; c += 64) {
      const float d = xr[c] - mean;
      v += d * d;
    }
  }
  const float var = dm_wave_sum(v) / (float)n;
  const float rstd = 1.0f / sqrtf(var + eps);
  float* yr = y + (size_t)row * ldy;
  if (CACHE) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = lane + 64 * j;
      if (c < n) {
        const float o = dm_elu((xc[j] - mean) * rstd * gamma[c] + beta[c]);
        yr[c] = o;
        if (yh) yh[c] = (unsigned short)dm_f2bf(o);
      }
    }
  } else {
    for (int c = lane; c < n; c += 64) {
      const float o = dm_elu((xr[c] - mean) * rstd * gamma[c] + beta[c]);
      yr[c] = o;
      if (yh) yh[c] = (unsigned short)dm_f2bf(o);
    }
  }
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * ELU'(y) * gamma
template <bool CACHE>
__global__ void __launch_bounds__(256) ln_elu_bwd_dx_kernel(int rows, int n, const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ y, int ldy,
                                                            const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy, int lddy,
                                                            float* __restrict__ dx, int lddx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float mean = stats[2 * row], rstd = stats[2 * row + 1];
  const float* xr = x + (size_t)row * ldx;
  const float* yr = y + (size_t)row * ldy;
  const float* dyr = dy + (size_t)row * lddy;
  float gc[16], hc[16];
  float sg = 0.f, sgx = 0.f;
  if (CACHE) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = lane + 64 * j;
      gc[j] = 0.f; hc[j] = 0.f;
      if (c < n) {
        gc[j] = dyr[c] * dm_elu_grad_from_y(yr[c]) * gamma[c];
        hc[j] = (xr[c] - mean) * rstd;
      }
      sg += gc[j];
      sgx += gc[j] * hc[j];
    }
  } else {
    for (int c = lane; c < n; c += 64) {
      const float g = dyr[c] * dm_elu_grad_from_y(yr[c]) * gamma[c];
      const float xh = (xr[c] - mean) * rstd;
      sg += g;
      sgx += g * xh;
    }
  }
  sg = dm_wave_sum(sg) / (float)n;
  sgx = dm_wave_sum(sgx) / (float)n;
  float* dxr = dx + (size_t)row * lddx;
  if (CACHE) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = lane + 64 * j;
      if (c < n) dxr[c] = rstd * (gc[j] - sg - hc[j] * sgx);
    }
  } else {
    for (int c = lane; c < n; c += 64) {
      const float g = dyr[c] * dm_elu_grad_from_y(yr[c]) * gamma[c];
      const float xh = (xr[c] - mean) * rstd;
      dxr[c] = rstd * (g - sg - xh * sgx);
    }
  }
}

// Column partial sums over row chunks.  MODE 0: sum x.  MODE 1: LayerNorm dgamma/dbeta.
// Block = 64 columns x 4 row lanes, 4 independent accumulators per thread; partial[(chunk*nq + q)*n + col].
template <int MODE>
__global__ void __launch_bounds__(256) colsum_partial_kernel(int rows, int n, int rows_per_chunk,
                                                             const float* __restrict__ x, int ldx,
                                                             const float* __restrict__ y, int ldy,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ dy, int lddy,
                                                             float* __restrict__ partial) {
  __shared__ float red[2][4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const int chunk = blockIdx.y;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(rows, r0 + rows_per_chunk);
  float a0 = 0.f, a1 = 0.f;
  if (col < n) {
    if (MODE == 0) {
      float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
      int r = r0 + ry;
      for (; r + 12 < r1; r += 16) {
        b0 += x[(size_t)r * ldx + col];
        b1 += x[(size_t)(r + 4) * ldx + col];
        b2 += x[(size_t)(r + 8) * ldx + col];
        b3 += x[(size_t)(r + 12) * ldx + col];
      }
      for (; r < r1; r += 4) b0 += x[(size_t)r * ldx + col];
      a0 = (b0 + b1) + (b2 + b3);
    } else {
      for (int r = r0 + ry; r < r1; r += 4) {
        const float dl = dy[(size_t)r * lddy + col] * dm_elu_grad_from_y(y[(size_t)r * ldy + col]);
        const float xh = (x[(size_t)r * ldx + col] - stats[2 * r]) * stats[2 * r + 1];
        a0 += dl * xh;   // dgamma
        a1 += dl;        // dbeta
      }
    }
  }
  red[0][ry][cx] = a0;
  red[1][ry][cx] = a1;
  __syncthreads();
  if (ry == 0 && col < n) {
    const int nq = (MODE == 0) ? 1 : 2;
    float s0 = red[0][0][cx] + red[0][1][cx] + red[0][2][cx] + red[0][3][cx];
    partial[((size_t)chunk * nq + 0) * n + col] = s0;
    if (MODE == 1) {
      float s1 = red[1][0][cx] + red[1][1][cx] + red[1][2][cx] + red[1][3][cx];
      partial[((size_t)chunk * nq + 1) * n + col] = s1;
    }
  }
}

// Narrow dense matrices (n <= 64, ld == n; conv bias gradients over millions of pixels): the matrix is walked as a
// flat array in slabs of W = 256 - 256 % n consecutive floats (a multiple of n), one float per thread, so every
// thread stays on ONE column (t % n) and every wave reads contiguous memory.  partial[chunk*n + col]; fixed order.
__global__ void __launch_bounds__(256) colsum_flat_kernel(long long total, int n, int W, long long chunk_elems,
                                                          const float* __restrict__ x, float* __restrict__ partial) {
  __shared__ float sh[256];
  const long long base = (long long)blockIdx.x * chunk_elems;      // multiple of W, hence of n
  const long long end = min(total, base + chunk_elems);
  float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
  if ((int)threadIdx.x < W) {
    long long i = base + threadIdx.x;
    for (; i + 3LL * W < end; i += 4LL * W) {
      b0 += x[i];
      b1 += x[i + W];
      b2 += x[i + 2LL * W];
      b3 += x[i + 3LL * W];
    }
    for (; i < end; i += W) b0 += x[i];
  }
  sh[threadIdx.x] = (b0 + b1) + (b2 + b3);
  __syncthreads();
  if ((int)threadIdx.x < n) {
    float s = 0.f;
    for (int t = threadIdx.x; t < W; t += n) s += sh[t];      // threads t == col (mod n) hold column `col`
    partial[(size_t)blockIdx.x * n + threadIdx.x] = s;
  }
}

// out_q[col] = sum_chunks partial[(c*nq+q)*n + col]; block = 32 columns x 8 chunk lanes
__global__ void __launch_bounds__(256) colsum_final_kernel(int n, int chunks, int nq, const float* __restrict__ partial,
                                                           float* __restrict__ out0, float* __restrict__ out1) {
  __shared__ float red[8][32];
  const int cx = threadIdx.x & 31, cy = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cx;
  for (int q = 0; q < nq; ++q) {
    float s = 0.f;
    if (col < n) {
      // four loads in flight per thread, added in the SAME order as the plain loop (625 chunks of a 40 000-row head were 78 dependent
      // load + add rounds: 11 us for a 400-column sum)
      const float* p = partial + (size_t)q * n + col;
      const size_t st = (size_t)nq * n;
      int c = cy;
      for (; c + 24 < chunks; c += 32) {
        const float v0 = p[(size_t)c * st], v1 = p[(size_t)(c + 8) * st], v2 = p[(size_t)(c + 16) * st], v3 = p[(size_t)(c + 24) * st];
        s += v0; s += v1; s += v2; s += v3;
      }
      for (; c < chunks; c += 8) s += p[(size_t)c * st];
    }
    red[cy][cx] = s;
    __syncthreads();
    if (cy == 0 && col < n) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red[k][cx];
      (q == 0 ? out0 : out1)[col] = t;
    }
    __syncthreads();
  }
}

static int colsum_plan(int rows, size_t budget_floats, int n, int nq, int* rows_per_chunk) {
  int chunks = dm_cdiv(rows, 64);
  if (chunks > 1024) chunks = 1024;
  const size_t per = (size_t)n * nq;
  if (per > 0 && (size_t)chunks * per > budget_floats) chunks = (int)(budget_floats / per);
  if (chunks < 1) chunks = 1;
  *rows_per_chunk = dm_cdiv(rows, chunks);
  return dm_cdiv(rows, *rows_per_chunk);
}

int dm_colsum_launch(int rows, int n, const float* x, int ld, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
  if (n <= 0) return DM_OK;
  if (rows <= 0) {
    (void)hipMemsetAsync(out, 0, (size_t)n * sizeof(float), st);
    return DM_OK;
  }
  DM_REQUIRE(ws && ws_bytes >= (size_t)n * sizeof(float), DM_E_WORKSPACE, "colsum: workspace too small");
  const size_t budget = ws_bytes / sizeof(float);
  int chunks;
  if (n <= 64 && ld == n && (long long)rows * n >= (1 << 16)) {
    const long long total = (long long)rows * n;
    const int W = 256 - 256 % n;                          // slab width: the largest multiple of n that fits a block
    long long want = total / ((long long)W * 16);         // ~16 slabs per block
    if (want > 2048) want = 2048;
    if (want < 1) want = 1;
    if ((size_t)want * n > budget) want = (long long)(budget / n);
    long long chunk_elems = (total + want - 1) / want;
    chunk_elems = (chunk_elems + W - 1) / W * W;          // multiple of W (hence of n)
    chunks = (int)((total + chunk_elems - 1) / chunk_elems);
    hipLaunchKernelGGL(colsum_flat_kernel, dim3(chunks), dim3(256), 0, st, total, n, W, chunk_elems, x, (float*)ws);
  } else {
    int rpc;
    chunks = colsum_plan(rows, budget, n, 1, &rpc);
    hipLaunchKernelGGL((colsum_partial_kernel<0>), dim3(dm_cdiv(n, 64), chunks), dim3(256), 0, st, rows, n, rpc, x, ld,
                       nullptr, 0, nullptr, nullptr, 0, (float*)ws);
  }
  DM_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3(dm_cdiv(n, 32)), dim3(256), 0, st, n, chunks, 1, (const float*)ws, out,
                     nullptr);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

int dm_ln_elu_fwd_launch(int rows, int n, const float* x, int ldx, const float* gamma, const float* beta, float eps,
                         float* y, int ldy, float* stats, hipStream_t st) {
  if (rows <= 0) return DM_OK;
  unsigned short* y_h = dm_twin_of(y, false);        // bf16 mode: y's twin, written here
  if (y_h) dm_twin_mark(y);
  if (n <= 1024)
    hipLaunchKernelGGL((ln_elu_fwd_kernel<true>), dim3(dm_cdiv(rows, 4)), dim3(256), 0, st, rows, n, x, ldx, gamma, beta,
                       eps, y, ldy, stats, y_h);
  else
    hipLaunchKernelGGL((ln_elu_fwd_kernel<false>), dim3(dm_cdiv(rows, 4)), dim3(256), 0, st, rows, n, x, ldx, gamma, beta,
                       eps, y, ldy, stats, y_h);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// layer_norm=False (common.py:68-74 NoNorm): the activation alone.  y = ELU(x); dx = dy * ELU'(y) (ELU' from the output).
__global__ void __launch_bounds__(256) elu_fwd_kernel(int rows, int n, const float* __restrict__ x, int ldx,
                                                      float* __restrict__ y, int ldy) {
  const size_t total = (size_t)rows * n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / n, c = i % n;
    y[r * ldy + c] = dm_elu(x[r * ldx + c]);
  }
}
__global__ void __launch_bounds__(256) elu_bwd_kernel(int rows, int n, const float* __restrict__ y, int ldy,
                                                      const float* __restrict__ dy, int lddy, float* __restrict__ dx,
                                                      int lddx) {
  const size_t total = (size_t)rows * n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / n, c = i % n;
    dx[r * lddx + c] = dy[r * lddy + c] * dm_elu_grad_from_y(y[r * ldy + c]);
  }
}
int dm_elu_fwd_launch(int rows, int n, const float* x, int ldx, float* y, int ldy, hipStream_t st) {
  if (rows <= 0) return DM_OK;
  const size_t total = (size_t)rows * n;
  hipLaunchKernelGGL(elu_fwd_kernel, dim3((unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536)), dim3(256), 0,
                     st, rows, n, x, ldx, y, ldy);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
int dm_elu_bwd_launch(int rows, int n, const float* y, int ldy, const float* dy, int lddy, float* dx, int lddx,
                      hipStream_t st) {
  if (rows <= 0) return DM_OK;
  const size_t total = (size_t)rows * n;
  hipLaunchKernelGGL(elu_bwd_kernel, dim3((unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536)), dim3(256), 0,
                     st, rows, n, y, ldy, dy, lddy, dx, lddx);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// dx only (row kernel)
int dm_ln_elu_bwd_dx_launch(int rows, int n, const float* x, int ldx, const float* y, int ldy, const float* stats,
                            const float* gamma, const float* dy, int lddy, float* dx, int lddx, hipStream_t st) {
  if (rows <= 0) return DM_OK;
  if (n <= 1024)
    hipLaunchKernelGGL((ln_elu_bwd_dx_kernel<true>), dim3(dm_cdiv(rows, 4)), dim3(256), 0, st, rows, n, x, ldx, y, ldy,
                       stats, gamma, dy, lddy, dx, lddx);
  else
    hipLaunchKernelGGL((ln_elu_bwd_dx_kernel<false>), dim3(dm_cdiv(rows, 4)), dim3(256), 0, st, rows, n, x, ldx, y, ldy,
                       stats, gamma, dy, lddy, dx, lddx);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// dgamma / dbeta only (column kernel)
int dm_ln_elu_bwd_params_launch(int rows, int n, const float* x, int ldx, const float* y, int ldy, const float* stats,
                                const float* dy, int lddy, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                hipStream_t st) {
  int rpc;
  DM_REQUIRE(ws && ws_bytes >= (size_t)2 * n * sizeof(float), DM_E_WORKSPACE, "ln_bwd: workspace too small");
  const int chunks = colsum_plan(rows, ws_bytes / sizeof(float), n, 2, &rpc);
  hipLaunchKernelGGL((colsum_partial_kernel<1>), dim3(dm_cdiv(n, 64), chunks), dim3(256), 0, st, rows, n, rpc, x, ldx, y,
                     ldy, stats, dy, lddy, (float*)ws);
  DM_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3(dm_cdiv(n, 32)), dim3(256), 0, st, n, chunks, 2, (const float*)ws, dgamma,
                     dbeta);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_ln_elu_fwd(int rows, int n, const float* x, int ldx, const float* gamma, const float* beta, float eps,
                             float* y, int ldy, float* stats, void* stream) {
  DM_REQUIRE(x && gamma && beta && y && stats, DM_E_NULL, "ln_elu_fwd: null pointer");
  return dm_ln_elu_fwd_launch(rows, n, x, ldx, gamma, beta, eps, y, ldy, stats, (hipStream_t)stream);
}

extern "C" int dm_ln_elu_bwd(int rows, int n, const float* x, int ldx, const float* y, int ldy, const float* stats,
                             const float* gamma, const float* dy, int lddy, float* dx, int lddx, float* dgamma,
                             float* dbeta, void* ws, size_t ws_bytes, void* stream) {
  DM_REQUIRE(x && y && stats && gamma && dy && dx && dgamma && dbeta, DM_E_NULL, "ln_elu_bwd: null pointer");
  DM_TRY(dm_ln_elu_bwd_dx_launch(rows, n, x, ldx, y, ldy, stats, gamma, dy, lddy, dx, lddx, (hipStream_t)stream));
  return dm_ln_elu_bwd_params_launch(rows, n, x, ldx, y, ldy, stats, dy, lddy, dgamma, dbeta, ws, ws_bytes,
                                     (hipStream_t)stream);
}

extern "C" int dm_colsum(int rows, int n, const float* x, int ld, float* out, void* ws, size_t ws_bytes, void* stream) {
  DM_REQUIRE(x && out, DM_E_NULL, "colsum: null pointer");
  return dm_colsum_launch(rows, n, x, ld, out, ws, ws_bytes, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// nn.GRUCell gate math (rnn.py:48-49).  Row blocks of gi/gh: [r | z(update) | n].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dm_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ void __launch_bounds__(256) gru_gates_fwd_kernel(int rows, int D, const float* __restrict__ gi,
                                                            const float* __restrict__ gh, const float* __restrict__ h_in,
                                                            int ldh, float* __restrict__ h_out, int ldo,
                                                            float* __restrict__ h_next,
                                                            const uint8_t* __restrict__ next_reset,
                                                            float* __restrict__ h_frag, float* __restrict__ h_next_frag,
                                                            int ldg, int ldn, unsigned short* __restrict__ h_out_h) {
  const size_t total = (size_t)rows * D;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / D), d = (int)(i % D);
    const float* gir = gi + (size_t)r * ldg;        // ldg: row stride of the gate products (3*D, or 3*deter_dim for one layer
    const float* ghr = gh + (size_t)r * ldg;        //      of a GRUCellStack, rnn.py:40-67)
    const float rg = dm_sigmoid(gir[d] + ghr[d]);
    const float ug = dm_sigmoid(gir[D + d] + ghr[D + d]);
    const float ng = tanhf(gir[2 * D + d] + rg * ghr[2 * D + d]);
    const float h = h_in[(size_t)r * ldh + d];
    const float ho = (h - ng) * ug + ng;
    h_out[(size_t)r * ldo + d] = ho;
    if (h_out_h) h_out_h[(size_t)r * ldo + d] = (unsigned short)dm_f2bf(ho);      // bf16 twin, same leading dimension
    const float hn = (next_reset && next_reset[r]) ? 0.f : ho;
    if (h_next) h_next[(size_t)r * ldn + d] = hn;
    if (h_frag) h_frag[dm_frag_off(r, d)] = ho;
    if (h_next_frag) h_next_frag[dm_frag_off(r, d)] = hn;
  }
}
// Xf = fragment-major copy (dm_frag_off) of the <= 64-row block X
__global__ void __launch_bounds__(256) frag_pack_kernel(int rows, int K, const float* __restrict__ X, int ldx,
                                                        float* __restrict__ Xf) {
  const int total = rows * K;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int r = i / K, k = i % K;
    Xf[dm_frag_off(r, k)] = X[(size_t)r * ldx + k];
  }
}

__global__ void __launch_bounds__(256) gru_gates_bwd_kernel(int rows, int D, const float* __restrict__ gi,
                                                            const float* __restrict__ gh, const float* __restrict__ h_in,
                                                            int ldh, const float* __restrict__ dh_out, int lddh,
                                                            float* __restrict__ dgi, float* __restrict__ dgh,
                                                            float* __restrict__ dh_in, int lddi, int accum,
                                                            const uint8_t* __restrict__ row_zero, int ldg) {
  const size_t total = (size_t)rows * D;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / D), d = (int)(i % D);
    const size_t g0 = (size_t)r * ldg;
    const float ghn = gh[g0 + 2 * D + d];
    const float rg = dm_sigmoid(gi[g0 + d] + gh[g0 + d]);
    const float ug = dm_sigmoid(gi[g0 + D + d] + gh[g0 + D + d]);
    const float ng = tanhf(gi[g0 + 2 * D + d] + rg * ghn);
    const float h = h_in[(size_t)r * ldh + d];
    const float dh = dh_out[(size_t)r * lddh + d];
    const float dn = dh * (1.f - ug);
    const float du = dh * (h - ng);
    const float dpn = dn * (1.f - ng * ng);
    const float dpr = dpn * ghn * rg * (1.f - rg);
    const float dpu = du * ug * (1.f - ug);
    dgi[g0 + d] = dpr;
    dgi[g0 + D + d] = dpu;
    dgi[g0 + 2 * D + d] = dpn;
    dgh[g0 + d] = dpr;
    dgh[g0 + D + d] = dpu;
    dgh[g0 + 2 * D + d] = dpn * rg;
    if (dh_in) {
      const float v = (row_zero && row_zero[r]) ? 0.f : dh * ug;
      float* o = dh_in + (size_t)r * lddi + d;
      *o = accum ? *o + v : v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm GRU cells (rnn.py:95-138), one wave per row.  h' = u n + (1-u) h in both.
//   KIND 1 gru_layernorm     : r = sig(LN_r(gi_r+gh_r)), u = sig(LN_u(gi_u+gh_u)), n = tanh(LN_n(gi_n + r gh_n))
//   KIND 2 gru_layernorm_dv2 : g = LN_3D(gi+gh); r = sig(g_r), u = sig(g_u - 1), n = tanh(r g_n)        (late reset)
// gs (rows,3D): the pre-LayerNorm sums [s_r | s_u | s_n] (KIND 1) / s (KIND 2), saved for backward;
// gst (rows,6): (mean, rstd) of LN_r, LN_u, LN_n (KIND 1) / of the one LayerNorm in slots 0,1 (KIND 2).  eps = 1e-3.
// ------------------------------------------------------------------------------------------------
struct GruLnParams { const float* g[3]; const float* b[3]; };

template <int KIND>
__global__ void __launch_bounds__(256) gru_norm_fwd_kernel(int rows, int D, const float* __restrict__ gi,
                                                           const float* __restrict__ gh, const float* __restrict__ h_in,
                                                           int ldh, const GruLnParams lp, float* __restrict__ h_out, int ldo,
                                                           float* __restrict__ gs, float* __restrict__ gst,
                                                           float* __restrict__ h_next, const uint8_t* __restrict__ next_reset,
                                                           int ldg, int ldst, int ldn) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float eps = 1e-3f;
  const float* gir = gi + (size_t)row * ldg;
  const float* ghr = gh + (size_t)row * ldg;
  float* sr = gs + (size_t)row * ldg;
  const float* hr = h_in + (size_t)row * ldh;
  float st[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (KIND == 2) {
    float sum = 0.f;
    for (int j = lane; j < 3 * D; j += 64) { const float v = gir[j] + ghr[j]; sr[j] = v; sum += v; }
    const float mean = dm_wave_sum(sum) / (float)(3 * D);
    float var = 0.f;
    for (int j = lane; j < 3 * D; j += 64) { const float d = (gir[j] + ghr[j]) - mean; var += d * d; }
    const float rstd = 1.0f / sqrtf(dm_wave_sum(var) / (float)(3 * D) + eps);
    st[0] = mean; st[1] = rstd;
    for (int d = lane; d < D; d += 64) {
      const float g_r = ((gir[d] + ghr[d]) - mean) * rstd * lp.g[0][d] + lp.b[0][d];
      const float g_u = ((gir[D + d] + ghr[D + d]) - mean) * rstd * lp.g[0][D + d] + lp.b[0][D + d];
      const float g_n = ((gir[2 * D + d] + ghr[2 * D + d]) - mean) * rstd * lp.g[0][2 * D + d] + lp.b[0][2 * D + d];
      const float r = dm_sigmoid(g_r), u = dm_sigmoid(g_u - 1.0f), n = tanhf(r * g_n);
      const float ho = u * n + (1.f - u) * hr[d];
      h_out[(size_t)row * ldo + d] = ho;
      if (h_next) h_next[(size_t)row * ldn + d] = (next_reset && next_reset[row]) ? 0.f : ho;
    }
  } else {
    float s0 = 0.f, s1 = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float a = gir[d] + ghr[d], b = gir[D + d] + ghr[D + d];
      sr[d] = a; sr[D + d] = b;
      s0 += a; s1 += b;
    }
    const float m0 = dm_wave_sum(s0) / (float)D, m1 = dm_wave_sum(s1) / (float)D;
    float v0 = 0.f, v1 = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float a = (gir[d] + ghr[d]) - m0, b = (gir[D + d] + ghr[D + d]) - m1;
      v0 += a * a; v1 += b * b;
    }
    const float r0 = 1.0f / sqrtf(dm_wave_sum(v0) / (float)D + eps), r1 = 1.0f / sqrtf(dm_wave_sum(v1) / (float)D + eps);
    float s2 = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float r = dm_sigmoid(((gir[d] + ghr[d]) - m0) * r0 * lp.g[0][d] + lp.b[0][d]);
      const float c = gir[2 * D + d] + r * ghr[2 * D + d];
      sr[2 * D + d] = c;
      s2 += c;
    }
    const float m2 = dm_wave_sum(s2) / (float)D;
    float v2 = 0.f;
    for (int d = lane; d < D; d += 64) { const float c = sr[2 * D + d] - m2; v2 += c * c; }     // own writes: same lane, same address
    const float r2 = 1.0f / sqrtf(dm_wave_sum(v2) / (float)D + eps);
    st[0] = m0; st[1] = r0; st[2] = m1; st[3] = r1; st[4] = m2; st[5] = r2;
    for (int d = lane; d < D; d += 64) {
      const float u = dm_sigmoid(((gir[D + d] + ghr[D + d]) - m1) * r1 * lp.g[1][d] + lp.b[1][d]);
      const float n = tanhf((sr[2 * D + d] - m2) * r2 * lp.g[2][d] + lp.b[2][d]);
      const float ho = u * n + (1.f - u) * hr[d];
      h_out[(size_t)row * ldo + d] = ho;
      if (h_next) h_next[(size_t)row * ldn + d] = (next_reset && next_reset[row]) ? 0.f : ho;
    }
  }
  if (lane < 6) gst[(size_t)row * ldst + lane] = st[lane];
}

// dgi, dgh (rows,3D): gradients w.r.t. the two gate products; dg (rows,3D): gradients w.r.t. the LayerNorm OUTPUTS (for
// the batched gamma / beta gradients); dh_in (nullable) += mask * dh' * (1 - u).
template <int KIND>
__global__ void __launch_bounds__(256) gru_norm_bwd_kernel(int rows, int D, const float* __restrict__ gh,
                                                           const float* __restrict__ h_in, int ldh,
                                                           const float* __restrict__ gs, const float* __restrict__ gst,
                                                           const GruLnParams lp, const float* __restrict__ dh_out, int lddh,
                                                           float* __restrict__ dgi, float* __restrict__ dgh,
                                                           float* __restrict__ dg, float* __restrict__ dh_in, int lddi,
                                                           const uint8_t* __restrict__ row_zero, int ldg, int ldst) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* sr = gs + (size_t)row * ldg;
  const float* ghr = gh + (size_t)row * ldg;
  const float* hr = h_in + (size_t)row * ldh;
  const float* dhr = dh_out + (size_t)row * lddh;
  float* dgr = dg + (size_t)row * ldg;
  float* dgir = dgi + (size_t)row * ldg;
  float* dghr = dgh + (size_t)row * ldg;
  const float* st = gst + (size_t)row * ldst;
  const bool rz = row_zero && row_zero[row];
  if (KIND == 2) {
    const float mean = st[0], rstd = st[1];
    const float inv = 1.0f / (float)(3 * D);
    float c1 = 0.f, c2 = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float x_r = (sr[d] - mean) * rstd, x_u = (sr[D + d] - mean) * rstd, x_n = (sr[2 * D + d] - mean) * rstd;
      const float g_n = x_n * lp.g[0][2 * D + d] + lp.b[0][2 * D + d];
      const float r = dm_sigmoid(x_r * lp.g[0][d] + lp.b[0][d]);
      const float u = dm_sigmoid(x_u * lp.g[0][D + d] + lp.b[0][D + d] - 1.0f);
      const float n = tanhf(r * g_n);
      const float dh = dhr[d];
      const float dpn = dh * u * (1.f - n * n);
      const float d_r = dpn * g_n * r * (1.f - r), d_u = dh * (n - hr[d]) * u * (1.f - u), d_n = dpn * r;
      dgr[d] = d_r; dgr[D + d] = d_u; dgr[2 * D + d] = d_n;
      const float a = d_r * lp.g[0][d], b = d_u * lp.g[0][D + d], c = d_n * lp.g[0][2 * D + d];
      c1 += a + b + c;
      c2 += a * x_r + b * x_u + c * x_n;
      if (dh_in) dh_in[(size_t)row * lddi + d] += rz ? 0.f : dh * (1.f - u);
    }
    c1 = dm_wave_sum(c1) * inv;
    c2 = dm_wave_sum(c2) * inv;
    for (int d = lane; d < D; d += 64)          // every lane re-reads only its OWN dg writes (d, D+d, 2D+d)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int j = q * D + d;
        const float xh = (sr[j] - mean) * rstd;
        const float v = rstd * (dgr[j] * lp.g[0][j] - c1 - xh * c2);
        dgir[j] = v;
        dghr[j] = v;
      }
  } else {
    const float m0 = st[0], r0 = st[1], m1 = st[2], r1 = st[3], m2 = st[4], r2 = st[5];
    const float inv = 1.0f / (float)D;
    // pass A: LN_n backward needs the row means of dg_n gamma_n (and x xhat)
    float a1 = 0.f, a2 = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float xn = (sr[2 * D + d] - m2) * r2;
      const float u = dm_sigmoid((sr[D + d] - m1) * r1 * lp.g[1][d] + lp.b[1][d]);
      const float n = tanhf(xn * lp.g[2][d] + lp.b[2][d]);
      const float dh = dhr[d];
      const float d_n = dh * u * (1.f - n * n);
      dgr[2 * D + d] = d_n;
      const float gg = d_n * lp.g[2][d];
      a1 += gg; a2 += gg * xn;
      dgr[D + d] = dh * (n - hr[d]) * u * (1.f - u);
      if (dh_in) dh_in[(size_t)row * lddi + d] += rz ? 0.f : dh * (1.f - u);
    }
    a1 = dm_wave_sum(a1) * inv; a2 = dm_wave_sum(a2) * inv;
    // pass B: ds_n -> dgi_n, dgh_n = ds_n r, dr = ds_n gh_n -> dg_r ; row means for LN_r and LN_u
    float b1 = 0.f, b2 = 0.f, u1 = 0.f, u2 = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float xn = (sr[2 * D + d] - m2) * r2, xr = (sr[d] - m0) * r0, xu = (sr[D + d] - m1) * r1;
      const float ds_n = r2 * (dgr[2 * D + d] * lp.g[2][d] - a1 - xn * a2);
      const float r = dm_sigmoid(xr * lp.g[0][d] + lp.b[0][d]);
      dgir[2 * D + d] = ds_n;
      dghr[2 * D + d] = ds_n * r;
      const float d_r = ds_n * ghr[2 * D + d] * r * (1.f - r);
      dgr[d] = d_r;
      const float gr_ = d_r * lp.g[0][d], gu_ = dgr[D + d] * lp.g[1][d];
      b1 += gr_; b2 += gr_ * xr;
      u1 += gu_; u2 += gu_ * xu;
    }
    b1 = dm_wave_sum(b1) * inv; b2 = dm_wave_sum(b2) * inv;
    u1 = dm_wave_sum(u1) * inv; u2 = dm_wave_sum(u2) * inv;
    for (int d = lane; d < D; d += 64) {
      const float xr = (sr[d] - m0) * r0, xu = (sr[D + d] - m1) * r1;
      const float ds_r = r0 * (dgr[d] * lp.g[0][d] - b1 - xr * b2);
      const float ds_u = r1 * (dgr[D + d] * lp.g[1][d] - u1 - xu * u2);
      dgir[d] = ds_r; dghr[d] = ds_r;
      dgir[D + d] = ds_u; dghr[D + d] = ds_u;
    }
  }
}

// dgamma_q[c] = sum_rows dg[r][c] * xhat[r][c], dbeta_q[c] = sum_rows dg[r][c]  over the 3D columns; the statistics of column
// c are gst[r][2*(c / D)] (KIND 1: one LayerNorm per third) or gst[r][0] (KIND 2).  64 columns x 4 row lanes per block.
__global__ void __launch_bounds__(256) gru_norm_param_grads_kernel(int kind, int rows, int D, const float* __restrict__ gs,
                                                                   const float* __restrict__ gst, const float* __restrict__ dg,
                                                                   float* __restrict__ dgam, float* __restrict__ dbet, int ldg,
                                                                   int ldst) {
  __shared__ float red[2][4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  float a = 0.f, b = 0.f;
  if (col < 3 * D) {
    const int q = kind == 1 ? col / D : 0;
    for (int r = ry; r < rows; r += 4) {
      const float v = dg[(size_t)r * ldg + col];
      const float xh = (gs[(size_t)r * ldg + col] - gst[(size_t)r * ldst + 2 * q]) * gst[(size_t)r * ldst + 2 * q + 1];
      a += v * xh;
      b += v;
    }
  }
  red[0][ry][cx] = a; red[1][ry][cx] = b;
  __syncthreads();
  if (ry == 0 && col < 3 * D) {
    dgam[col] = (red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx]);
    dbet[col] = (red[1][0][cx] + red[1][1][cx]) + (red[1][2][cx] + red[1][3][cx]);
  }
}

int dm_gru_norm_fwd_launch(int kind, int rows, int D, const float* gi, const float* gh, const float* h_in, int ldh,
                           const float* const* ln_g, const float* const* ln_b, float* h_out, int ldo, float* gs, float* gst,
                           float* h_next, const uint8_t* next_reset, hipStream_t st, int ldg, int ldst, int ldn) {
  if (rows <= 0) return DM_OK;
  if (ldg <= 0) ldg = 3 * D;
  if (ldst <= 0) ldst = 6;
  if (ldn <= 0) ldn = D;
  GruLnParams lp;
  for (int i = 0; i < 3; ++i) { lp.g[i] = ln_g[i]; lp.b[i] = ln_b[i]; }
  if (kind == 1)
    hipLaunchKernelGGL((gru_norm_fwd_kernel<1>), dim3(dm_cdiv(rows, 4)), dim3(256), 0, st, rows, D, gi, gh, h_in, ldh, lp, h_out,
                       ldo, gs, gst, h_next, next_reset, ldg, ldst, ldn);
  else
    hipLaunchKernelGGL((gru_norm_fwd_kernel<2>), dim3(dm_cdiv(rows, 4)), dim3(256), 0, st, rows, D, gi, gh, h_in, ldh, lp, h_out,
                       ldo, gs, gst, h_next, next_reset, ldg, ldst, ldn);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
int dm_gru_norm_bwd_launch(int kind, int rows, int D, const float* gh, const float* h_in, int ldh, const float* gs,
                           const float* gst, const float* const* ln_g, const float* const* ln_b, const float* dh_out, int lddh,
                           float* dgi, float* dgh, float* dg, float* dh_in, int lddi, const uint8_t* row_zero, hipStream_t st,
                           int ldg, int ldst) {
  if (rows <= 0) return DM_OK;
  if (ldg <= 0) ldg = 3 * D;
  if (ldst <= 0) ldst = 6;
  GruLnParams lp;
  for (int i = 0; i < 3; ++i) { lp.g[i] = ln_g[i]; lp.b[i] = ln_b[i]; }
  if (kind == 1)
    hipLaunchKernelGGL((gru_norm_bwd_kernel<1>), dim3(dm_cdiv(rows, 4)), dim3(256), 0, st, rows, D, gh, h_in, ldh, gs, gst, lp,
                       dh_out, lddh, dgi, dgh, dg, dh_in, lddi, row_zero, ldg, ldst);
  else
    hipLaunchKernelGGL((gru_norm_bwd_kernel<2>), dim3(dm_cdiv(rows, 4)), dim3(256), 0, st, rows, D, gh, h_in, ldh, gs, gst, lp,
                       dh_out, lddh, dgi, dgh, dg, dh_in, lddi, row_zero, ldg, ldst);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
// dgam / dbet: 3D-float vectors laid out like the LayerNorm outputs ([reset | update | newval] thirds)
int dm_gru_norm_param_grads_launch(int kind, int rows, int D, const float* gs, const float* gst, const float* dg, float* dgam,
                                   float* dbet, hipStream_t st, int ldg, int ldst) {
  if (ldg <= 0) ldg = 3 * D;
  if (ldst <= 0) ldst = 6;
  hipLaunchKernelGGL(gru_norm_param_grads_kernel, dim3(dm_cdiv(3 * D, 64)), dim3(256), 0, st, kind, rows, D, gs, gst, dg, dgam,
                     dbet, ldg, ldst);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

static inline int ew_blocks(size_t total) {
  size_t b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// z_mlp of a ONE-HOT latent as an embedding gather-sum (rssm.py:138,162: x = z_mlp(z) + a_mlp(a)).  The forward value of a
// straight-through sample is exactly its one-hot (onehot + (p - p)), so z W^T is the sum of one row of W^T per group:
//   x[r][:] = bias + add[r][:] + sum_s Wt[s*C + idx[r][s]][:]            (row_zero[r]: the reset-masked z is 0, no rows)
//             (+ Wt2[idx2[r]][:]: a_mlp of a one-hot action the same way, for the rollout)
// S rows of n floats per output row instead of an n x S*C product (32 KB instead of a 2.05 MFLOP dot for the 32x32 latent).
// One wave per row, 4 x float4 per lane (n <= 1024, n % 4 == 0), the S gathers independent and in flight together; optional
// LayerNorm + ELU of the row in the same pass (the rollout has no use for the pre-activation), optional fragment-major copy.
__global__ void __launch_bounds__(256) z_embed_kernel(int rows, int n, int S, int C, const int32_t* __restrict__ idx,
                                                      const uint8_t* __restrict__ row_zero, const float* __restrict__ Wt,
                                                      const float* __restrict__ bias, const float* __restrict__ add,
                                                      int ldadd, const int32_t* __restrict__ idx2,
                                                      const float* __restrict__ Wt2, float* __restrict__ x, int ldx,
                                                      float* __restrict__ x_frag,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, float* __restrict__ y, int ldy,
                                                      unsigned short* __restrict__ y_h) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  // branch-free loads: a lane whose columns fall outside n reads column 0 instead and its sums are never stored (a branch
  // around a load makes hipcc drain vmcnt after every one of them)
  float4 acc[4];
  bool in[4];
  int off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = (lane + 64 * j) * 4;
    in[j] = c < n;
    off[j] = in[j] ? c : 0;
    acc[j] = bias ? *reinterpret_cast<const float4*>(bias + off[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (add) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 a = *reinterpret_cast<const float4*>(add + (size_t)row * ldadd + off[j]);
      acc[j].x += a.x; acc[j].y += a.y; acc[j].z += a.z; acc[j].w += a.w;
    }
  }
  if (idx2) {       // + a_mlp of a one-hot action: row idx2[r] of a_mlp^T
    const float* wr = Wt2 + (size_t)idx2[row] * n;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 a = *reinterpret_cast<const float4*>(wr + off[j]);
      acc[j].x += a.x; acc[j].y += a.y; acc[j].z += a.z; acc[j].w += a.w;
    }
  }
  if (!(row_zero && row_zero[row])) {
    // the row's indices are fetched 64 at a time (one per lane) and broadcast, so the S gathers do not wait on S
    // dependent index loads; 8 groups per trip keep 32 loads of a lane in flight
    const int32_t* ir = idx + (size_t)row * S;
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int mine = ir[min(s0 + lane, S - 1)];
      const int cnt = min(64, S - s0);
      int t = 0;
      for (; t + 8 <= cnt; t += 8) {
        float4 w[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float* wr = Wt + ((size_t)(s0 + t + u) * C + __shfl(mine, t + u, 64)) * n;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[u][j] = *reinterpret_cast<const float4*>(wr + off[j]);
        }
        __builtin_amdgcn_sched_barrier(0);      // all 32 loads issued before the first add waits on one
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[j].x += w[u][j].x; acc[j].y += w[u][j].y; acc[j].z += w[u][j].z; acc[j].w += w[u][j].w;
          }
      }
      for (; t < cnt; ++t) {
        const float* wr = Wt + ((size_t)(s0 + t) * C + __shfl(mine, t, 64)) * n;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 w = *reinterpret_cast<const float4*>(wr + off[j]);
          acc[j].x += w.x; acc[j].y += w.y; acc[j].z += w.z; acc[j].w += w.w;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (in[j]) {
      const int c = (lane + 64 * j) * 4;
      if (x) *reinterpret_cast<float4*>(x + (size_t)row * ldx + c) = acc[j];
      if (x_frag) *reinterpret_cast<float4*>(x_frag + dm_frag_off(row, c)) = acc[j];
    }
  if (!y) return;
  if (!gamma) {       // layer_norm=False: the activation alone
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (in[j]) {
        float4 o;
        o.x = dm_elu(acc[j].x); o.y = dm_elu(acc[j].y); o.z = dm_elu(acc[j].z); o.w = dm_elu(acc[j].w);
        *reinterpret_cast<float4*>(y + (size_t)row * ldy + (lane + 64 * j) * 4) = o;
      }
    return;
  }
  float s1 = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (in[j]) s1 += (acc[j].x + acc[j].y) + (acc[j].z + acc[j].w);
  const float mean = dm_wave_sum(s1) / (float)n;
  float s2 = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (in[j]) {
      const float dx = acc[j].x - mean, dy = acc[j].y - mean, dz = acc[j].z - mean, dw = acc[j].w - mean;
      s2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  const float rstd = 1.0f / sqrtf(dm_wave_sum(s2) / (float)n + eps);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (in[j]) {
      const int c = (lane + 64 * j) * 4;
      const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
      float4 o;
      o.x = dm_elu((acc[j].x - mean) * rstd * g.x + b.x);
      o.y = dm_elu((acc[j].y - mean) * rstd * g.y + b.y);
      o.z = dm_elu((acc[j].z - mean) * rstd * g.z + b.z);
      o.w = dm_elu((acc[j].w - mean) * rstd * g.w + b.w);
      *reinterpret_cast<float4*>(y + (size_t)row * ldy + c) = o;
      if (y_h) *reinterpret_cast<uint2*>(y_h + (size_t)row * ldy + c) = dm_pack_bf16x4(o);      // bf16 twin (common.h DmTwinScope)
    }
}
// x[r][:] = sum_e z[r][e] * Wt[e][:] over the NON-ZERO e of row r: a sparse-row product, exact for any z and cheap when z is
// a concatenation of one-hot groups (the latent part of a DreamerV2 feature row: 32 non-zeros of 1024).  One wave per
// row; the row is scanned 64 elements at a time and the rows of Wt named by the ballot of non-zeros are summed.
template <int NJ>
__global__ void __launch_bounds__(256) sparse_rows_kernel(int rows, int n, int Zc, const float* __restrict__ z, int ldz,
                                                          const float* __restrict__ Wt, float* __restrict__ x, int ldx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float4 acc[NJ];
  int off[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (lane + 64 * j) * 4;
    off[j] = c < n ? c : 0;
    acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float* zr = z + (size_t)row * ldz;
  for (int e0 = 0; e0 < Zc; e0 += 256) {          // 4 scans in flight, then their gathers
    float v[4];
    unsigned long long m[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = zr[min(e0 + 64 * u + lane, Zc - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) m[u] = __ballot(e0 + 64 * u + lane < Zc && v[u] != 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      unsigned long long mm = m[u];
      while (mm) {
        const int b = __builtin_ctzll(mm);
        mm &= mm - 1;
        const float ze = __shfl(v[u], b, 64);
        const float* wr = Wt + (size_t)(e0 + 64 * u + b) * n;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float4 w = *reinterpret_cast<const float4*>(wr + off[j]);
          acc[j].x += ze * w.x; acc[j].y += ze * w.y; acc[j].z += ze * w.z; acc[j].w += ze * w.w;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (lane + 64 * j) * 4;
    if (c < n) *reinterpret_cast<float4*>(x + (size_t)row * ldx + c) = acc[j];
  }
}
int dm_sparse_rows_launch(int rows, int n, int Zc, const float* z, int ldz, const float* Wt, float* x, int ldx,
                          hipStream_t st) {
  if (rows <= 0) return DM_OK;
  DM_REQUIRE(n >= 4 && n <= 1024 && (n & 3) == 0 && (ldx & 3) == 0 && Zc >= 1, DM_E_SHAPE, "sparse_rows: n=%d Zc=%d", n, Zc);
  const dim3 grid(dm_cdiv(rows, 4)), blk(256);
  if (n <= 256) hipLaunchKernelGGL((sparse_rows_kernel<1>), grid, blk, 0, st, rows, n, Zc, z, ldz, Wt, x, ldx);
  else if (n <= 512) hipLaunchKernelGGL((sparse_rows_kernel<2>), grid, blk, 0, st, rows, n, Zc, z, ldz, Wt, x, ldx);
  else hipLaunchKernelGGL((sparse_rows_kernel<4>), grid, blk, 0, st, rows, n, Zc, z, ldz, Wt, x, ldx);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

bool dm_z_embed_ok(int n) { return n <= 1024 && (n & 3) == 0; }
// The same sums for the <= 64-row steps of the posterior chain when only x is wanted (the LayerNorm rides in the consuming
// product's prologue there): ONE ROW PER WORKGROUP, its 4 waves split the columns (one float4 per lane for n <= 1024) and all S <= 32
// gathers of a lane are in flight at once - the one-wave-per-row form walks them in 4 dependent batches of 8, four global round
// trips on a kernel that is nothing but latency.  Per column the additions run in the same order, so x is bit-identical.
__global__ void __launch_bounds__(256) z_embed_wide_kernel(int n, int S, int C, const int32_t* __restrict__ idx,
                                                           const uint8_t* __restrict__ row_zero, const float* __restrict__ Wt,
                                                           const float* __restrict__ bias, const float* __restrict__ add, int ldadd,
                                                           const int32_t* __restrict__ idx2, const float* __restrict__ Wt2,
                                                           float* __restrict__ x, int ldx, float* __restrict__ x_frag) {
  const int row = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int c = threadIdx.x * 4;                       // this thread's 4 columns
  const bool in = c < n;
  const int off = in ? c : 0;
  float4 acc = bias ? *reinterpret_cast<const float4*>(bias + off) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (add) {
    const float4 a = *reinterpret_cast<const float4*>(add + (size_t)row * ldadd + off);
    acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
  }
  if (idx2) {
    const float4 a = *reinterpret_cast<const float4*>(Wt2 + (size_t)idx2[row] * n + off);
    acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
  }
  if (!(row_zero && row_zero[row])) {
    const int mine = idx[(size_t)row * S + min(lane, S - 1)];      // S <= 32 (host-checked): every wave holds the row's indices
    float4 w[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int sc = s < S ? s : S - 1;                  // past S: re-read the last row (not added)
      w[s] = *reinterpret_cast<const float4*>(Wt + ((size_t)sc * C + __shfl(mine, sc, 64)) * n + off);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 32; ++s)
      if (s < S) { acc.x += w[s].x; acc.y += w[s].y; acc.z += w[s].z; acc.w += w[s].w; }
  }
  if (in) {
    if (x) *reinterpret_cast<float4*>(x + (size_t)row * ldx + c) = acc;
    if (x_frag) *reinterpret_cast<float4*>(x_frag + dm_frag_off(row, c)) = acc;
  }
}

// ... and with the LayerNorm + ELU of the row behind the sums (the workgroup holds the complete row: two block reductions),
// for the chain steps whose consuming product then runs WITHOUT a LayerNorm prologue - the prologue form makes every one of
// the product's 100-230 workgroups normalise the whole 64 x n operand again.  Two-pass statistics like ln_elu_fwd_kernel.
// x (raw sums, what backward reads), y = ELU(LN(x)), y_frag (fragment-major copy of y for the skinny product), stats (mean, rstd).
__global__ void __launch_bounds__(256) z_embed_wide_ln_kernel(int n, int S, int C, const int32_t* __restrict__ idx,
                                                              const uint8_t* __restrict__ row_zero, const float* __restrict__ Wt,
                                                              const float* __restrict__ bias, const float* __restrict__ add, int ldadd,
                                                              const int32_t* __restrict__ idx2, const float* __restrict__ Wt2,
                                                              float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, float* __restrict__ y, int ldy,
                                                              float* __restrict__ y_frag, float* __restrict__ stats) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = threadIdx.x * 4;                       // this thread's 4 columns
  const bool in = c < n;
  const int off = in ? c : 0;
  float4 acc = bias ? *reinterpret_cast<const float4*>(bias + off) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (add) {
    const float4 a = *reinterpret_cast<const float4*>(add + (size_t)row * ldadd + off);
    acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
  }
  if (idx2) {
    const float4 a = *reinterpret_cast<const float4*>(Wt2 + (size_t)idx2[row] * n + off);
    acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
  }
  if (!(row_zero && row_zero[row])) {
    const int mine = idx[(size_t)row * S + min(lane, S - 1)];      // S <= 32 (host-checked): every wave holds the row's indices
    float4 w[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int sc = s < S ? s : S - 1;                  // past S: re-read the last row (not added)
      w[s] = *reinterpret_cast<const float4*>(Wt + ((size_t)sc * C + __shfl(mine, sc, 64)) * n + off);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 32; ++s)
      if (s < S) { acc.x += w[s].x; acc.y += w[s].y; acc.z += w[s].z; acc.w += w[s].w; }
  }
  if (in && x) *reinterpret_cast<float4*>(x + (size_t)row * ldx + c) = acc;
  // LayerNorm over the n columns of the row: the four waves' sums meet in LDS in a fixed order
  const float s1 = dm_wave_sum(in ? (acc.x + acc.y) + (acc.z + acc.w) : 0.f);
  if (lane == 0) red[wave] = s1;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
  const float4 d = make_float4(acc.x - mean, acc.y - mean, acc.z - mean, acc.w - mean);
  const float s2 = dm_wave_sum(in ? (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w) : 0.f);
  if (lane == 0) red[4 + wave] = s2;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[4] + red[5]) + (red[6] + red[7])) / (float)n + eps);
  if (stats && threadIdx.x == 0) { stats[2 * (size_t)row] = mean; stats[2 * (size_t)row + 1] = rstd; }
  if (in) {
    const float4 ga = make_float4(gamma[c], gamma[c + 1], gamma[c + 2], gamma[c + 3]);      // (parameter slices of a flat buffer: no 16-byte promise)
    const float4 be = make_float4(beta[c], beta[c + 1], beta[c + 2], beta[c + 3]);
    const float4 o = make_float4(dm_elu(d.x * rstd * ga.x + be.x), dm_elu(d.y * rstd * ga.y + be.y), dm_elu(d.z * rstd * ga.z + be.z),
                                 dm_elu(d.w * rstd * ga.w + be.w));
    if (y) *reinterpret_cast<float4*>(y + (size_t)row * ldy + c) = o;
    if (y_frag) *reinterpret_cast<float4*>(y_frag + dm_frag_off(row, c)) = o;
  }
}

int dm_z_embed_launch(int rows, int n, int S, int C, const int32_t* idx, const uint8_t* row_zero, const float* Wt,
                      const float* bias, const float* add, int ldadd, const int32_t* idx2, const float* Wt2, float* x, int ldx,
                      float* x_frag, const float* gamma, const float* beta, float eps, float* y, int ldy, hipStream_t st,
                      float* y_frag, float* stats) {
  if (rows <= 0) return DM_OK;
  DM_REQUIRE(!idx2 || Wt2, DM_E_NULL, "z_embed: second index list without its table");
  DM_REQUIRE(dm_z_embed_ok(n) && idx && Wt && (x || y), DM_E_SHAPE, "z_embed: n=%d needs n <= 1024, n %% 4 == 0", n);
  DM_REQUIRE((ldx & 3) == 0 && (ldy & 3) == 0 && (ldadd & 3) == 0, DM_E_SHAPE, "z_embed: leading dims must be multiples of 4");
  DM_REQUIRE(!x_frag || rows <= 64, DM_E_SHAPE, "z_embed: the fragment-major copy needs rows <= 64");
  DM_REQUIRE(!y || (gamma != nullptr) == (beta != nullptr), DM_E_NULL, "z_embed: LayerNorm gain without its bias");
  static const int no_wide = getenv("DM_Z_EMBED_NO_WIDE") ? 1 : 0;      // A/B switch
  if (!y && x && rows <= 64 && S <= 32 && !no_wide) {      // the chain's steps: latency only, one workgroup per row
    hipLaunchKernelGGL(z_embed_wide_kernel, dim3(rows), dim3(256), 0, st, n, S, C, idx, row_zero, Wt, bias, add, ldadd, idx2, Wt2, x,
                       ldx, x_frag);
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  if (y_frag) {      // the chain's steps with the LayerNorm + ELU behind the sums: one workgroup per row
    DM_REQUIRE(y && gamma && beta && rows <= 64 && S <= 32 && !x_frag, DM_E_SHAPE,
               "z_embed: the row-per-workgroup LayerNorm form needs y, gamma / beta, rows <= 64, S <= 32");
    hipLaunchKernelGGL(z_embed_wide_ln_kernel, dim3(rows), dim3(256), 0, st, n, S, C, idx, row_zero, Wt, bias, add, ldadd, idx2, Wt2, x,
                       ldx, gamma, beta, eps, y, ldy, y_frag, stats);
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  unsigned short* y_h = (y && gamma) ? dm_twin_of(y, false) : nullptr;      // bf16 mode: the LayerNorm+ELU output's twin
  if (y_h) dm_twin_mark(y);
  hipLaunchKernelGGL(z_embed_kernel, dim3(dm_cdiv(rows, 4)), dim3(256), 0, st, rows, n, S, C, idx, row_zero, Wt, bias, add,
                     ldadd, idx2, Wt2, x, ldx, x_frag, gamma, beta, eps, y, ldy, y_h);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

int dm_frag_pack_launch(int rows, int K, const float* X, int ldx, float* Xf, hipStream_t st) {
  DM_REQUIRE(rows >= 0 && rows <= 64 && K >= 1 && X && Xf, DM_E_SHAPE, "frag_pack: rows %d (<= 64), K %d", rows, K);
  if (rows == 0) return DM_OK;
  hipLaunchKernelGGL(frag_pack_kernel, dim3(ew_blocks((size_t)rows * K)), dim3(256), 0, st, rows, K, X, ldx, Xf);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
int dm_gru_gates_fwd_launch(int rows, int D, const float* gi, const float* gh, const float* h_in, int ldh, float* h_out,
                            int ldo, float* h_next, const uint8_t* next_reset, float* h_frag, float* h_next_frag,
                            hipStream_t st, int ldg, int ldn) {
  if (rows <= 0) return DM_OK;
  DM_REQUIRE((!h_frag && !h_next_frag) || rows <= 64, DM_E_SHAPE, "gru_gates_fwd: fragment-major copies need rows <= 64");
  unsigned short* h_out_h = dm_twin_of(h_out, false);      // bf16 mode: the new state's twin, written here
  if (h_out_h) dm_twin_mark(h_out);
  hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(ew_blocks((size_t)rows * D)), dim3(256), 0, st, rows, D, gi, gh, h_in,
                     ldh, h_out, ldo, h_next, next_reset, h_frag, h_next_frag, ldg > 0 ? ldg : 3 * D, ldn > 0 ? ldn : D, h_out_h);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
int dm_gru_gates_bwd_launch(int rows, int D, const float* gi, const float* gh, const float* h_in, int ldh,
                            const float* dh_out, int lddh, float* dgi, float* dgh, float* dh_in, int lddi, int accum,
                            const uint8_t* row_zero, hipStream_t st, int ldg) {
  if (rows <= 0) return DM_OK;
  hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(ew_blocks((size_t)rows * D)), dim3(256), 0, st, rows, D, gi, gh, h_in,
                     ldh, dh_out, lddh, dgi, dgh, dh_in, lddi, accum, row_zero, ldg > 0 ? ldg : 3 * D);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
extern "C" int dm_gru_gates_fwd(int rows, int D, const float* gi, const float* gh, const float* h_in, int ldh,
                                float* h_out, int ldo, void* stream) {
  DM_REQUIRE(gi && gh && h_in && h_out, DM_E_NULL, "gru_gates_fwd: null pointer");
  return dm_gru_gates_fwd_launch(rows, D, gi, gh, h_in, ldh, h_out, ldo, nullptr, nullptr, nullptr, nullptr,
                                 (hipStream_t)stream);
}
extern "C" int dm_gru_gates_bwd(int rows, int D, const float* gi, const float* gh, const float* h_in, int ldh,
                                const float* dh_out, int lddh, float* dgi, float* dgh, float* dh_in, int lddi,
                                void* stream) {
  DM_REQUIRE(gi && gh && h_in && dh_out && dgi && dgh && dh_in, DM_E_NULL, "gru_gates_bwd: null pointer");
  return dm_gru_gates_bwd_launch(rows, D, gi, gh, h_in, ldh, dh_out, lddh, dgi, dgh, dh_in, lddi, 0, nullptr,
                                 (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// Categorical sampler under the shared inverse-CDF rule (rssm.py:147-148,195-201; dreamer.py:198-200).
// One thread per (row, group).  p = exp(x-max)/sum (sequential fp32 sum), cdf = sequential cumsum,
// idx = #{k : cdf_k <= u*cdf_last}, clamped to C-1.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sample_onehot_kernel(int rows, int groups, int C, const float* __restrict__ logits,
                                                            int ldl, const float* __restrict__ u,
                                                            const int32_t* __restrict__ forced, float* __restrict__ onehot,
                                                            int ldo, int32_t* __restrict__ idx_out) {
  const int total = rows * groups;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int r = i / groups, gq = i % groups;
    const float* x = logits + (size_t)r * ldl + (size_t)gq * C;
    int idx;
    if (forced) {
      idx = forced[i];
    } else {
      float mx = x[0];
      for (int k = 1; k < C; ++k) mx = fmaxf(mx, x[k]);
      float sum = 0.f;
      for (int k = 0; k < C; ++k) sum += expf(x[k] - mx);
      float total_p = 0.f;
      for (int k = 0; k < C; ++k) total_p += expf(x[k] - mx) / sum;
      const float target = u[i] * total_p;
      float cdf = 0.f;
      idx = 0;
      for (int k = 0; k < C; ++k) {
        cdf += expf(x[k] - mx) / sum;
        idx += (cdf <= target) ? 1 : 0;
      }
      if (idx > C - 1) idx = C - 1;
    }
    float* o = onehot + (size_t)r * ldo + (size_t)gq * C;
    for (int k = 0; k < C; ++k) o[k] = (k == idx) ? 1.f : 0.f;
    if (idx_out) idx_out[i] = idx;
  }
}

// Same rule, LPG lanes per group (power of two >= C): lane k owns category k, so logits are read and the one-hot is
// written coalesced.  The sums that define the rule stay SEQUENTIAL in k (a chain of C adds fed by lane broadcasts),
// so the result is bit-identical to sample_onehot_kernel; only max (order independent) uses a butterfly.
template <int LPG>
__global__ void __launch_bounds__(256) sample_onehot_wave_kernel(int rows, int groups, int C,
                                                                 const float* __restrict__ logits, int ldl,
                                                                 const float* __restrict__ u,
                                                                 const int32_t* __restrict__ forced,
                                                                 float* __restrict__ onehot, int ldo,
                                                                 int32_t* __restrict__ idx_out) {
  constexpr int GPB = 256 / LPG;
  const int k = threadIdx.x % LPG;
  const int lane = threadIdx.x & 63;
  const int gbase = lane - k;                              // first lane of this group inside the wave
  const int total = rows * groups;
  for (int i0 = blockIdx.x * GPB; i0 < total; i0 += gridDim.x * GPB) {
    const int i = i0 + threadIdx.x / LPG;
    const bool grp = i < total;
    const bool live = grp && k < C;
    const int r = grp ? i / groups : 0, gq = grp ? i % groups : 0;
    const size_t off = (size_t)gq * C + k;
    int idx = 0;
    if (forced) {
      idx = grp ? forced[i] : 0;
    } else {
      const float x = live ? logits[(size_t)r * ldl + off] : -INFINITY;
      float mx = x;
#pragma unroll
      for (int o = LPG / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      const float e = live ? expf(x - mx) : 0.f;
      float sum = 0.f;
      for (int j = 0; j < C; ++j) sum += __shfl(e, gbase + j, 64);
      const float p = e / sum;
      float total_p = 0.f;
      for (int j = 0; j < C; ++j) total_p += __shfl(p, gbase + j, 64);
      const float target = (grp ? u[i] : 0.f) * total_p;
      float cdf = 0.f;
      for (int j = 0; j < C; ++j) {
        cdf += __shfl(p, gbase + j, 64);
        idx += (cdf <= target) ? 1 : 0;
      }
      if (idx > C - 1) idx = C - 1;
    }
    if (live) onehot[(size_t)r * ldo + off] = (k == idx) ? 1.f : 0.f;
    if (grp && k == 0 && idx_out) idx_out[i] = idx;
  }
}

// Same rule, ONE lane per group for C == 32 (the DreamerV2 latent): the 32 logits of a group are 128 contiguous
// bytes, a lane reads them as 8 float4 and all three sequential sums are register chains - no cross-lane traffic at
// all (the LPG-lane version spends ~100 ds_bpermute per group and takes 52 us on the 2500-row imagination batch).
// Operation order per group is identical to sample_onehot_kernel, so indices are bit-identical.
__global__ void __launch_bounds__(256) sample_onehot_lane32_kernel(int rows, int groups, const float* __restrict__ logits,
                                                                   int ldl, const float* __restrict__ u,
                                                                   const int32_t* __restrict__ forced,
                                                                   float* __restrict__ onehot, int ldo,
                                                                   int32_t* __restrict__ idx_out,
                                                                   float* __restrict__ z_next,
                                                                   const uint8_t* __restrict__ next_reset) {
  constexpr int C = 32;
  const int total = rows * groups;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int r = i / groups, gq = i % groups;
  int idx;
  if (forced) {
    idx = forced[i];
  } else {
    const float4* src = reinterpret_cast<const float4*>(logits + (size_t)r * ldl + (size_t)gq * C);
    float x[C];
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
      const float4 v = src[q];
      x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
    float mx = x[0];
#pragma unroll
    for (int k = 1; k < C; ++k) mx = fmaxf(mx, x[k]);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < C; ++k) { x[k] = expf(x[k] - mx); sum += x[k]; }
    float total_p = 0.f;
#pragma unroll
    for (int k = 0; k < C; ++k) { x[k] = x[k] / sum; total_p += x[k]; }
    const float target = u[i] * total_p;
    float cdf = 0.f;
    idx = 0;
#pragma unroll
    for (int k = 0; k < C; ++k) {
      cdf += x[k];
      idx += (cdf <= target) ? 1 : 0;
    }
    if (idx > C - 1) idx = C - 1;
  }
  float4* dst = reinterpret_cast<float4*>(onehot + (size_t)r * ldo + (size_t)gq * C);
#pragma unroll
  for (int q = 0; q < C / 4; ++q)
    dst[q] = make_float4(idx == 4 * q ? 1.f : 0.f, idx == 4 * q + 1 ? 1.f : 0.f, idx == 4 * q + 2 ? 1.f : 0.f,
                         idx == 4 * q + 3 ? 1.f : 0.f);
  if (z_next) {          // next step's sample input under its reset mask (rssm.py:135); rows are groups*C wide, dense
    const int keep = (next_reset && next_reset[r]) ? -1 : idx;
    float4* dn = reinterpret_cast<float4*>(z_next + ((size_t)r * groups + gq) * C);
#pragma unroll
    for (int q = 0; q < C / 4; ++q)
      dn[q] = make_float4(keep == 4 * q ? 1.f : 0.f, keep == 4 * q + 1 ? 1.f : 0.f, keep == 4 * q + 2 ? 1.f : 0.f,
                          keep == 4 * q + 3 ? 1.f : 0.f);
  }
  if (idx_out) idx_out[i] = idx;
}

// ---- Gaussian latents (stoch_discrete = 0; rssm.py:195-203, functions.py:46-56 diag_normal): a row of parameters is
// (mean[S] | raw[S]) with std = 2 sigmoid(raw) + 0.1; z = mean + std * eps (Normal.rsample, eps standard normal).
__device__ __forceinline__ float dm_gauss_std(float raw) { return 2.0f * dm_sigmoid(raw) + 0.1f; }
__global__ void __launch_bounds__(256) gauss_sample_kernel(int rows, int S, const float* __restrict__ par, int ldp,
                                                           const float* __restrict__ eps, float* __restrict__ z, int ldz,
                                                           float* __restrict__ z_next,
                                                           const uint8_t* __restrict__ next_reset) {
  const int total = rows * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int r = i / S, s = i % S;
    const float* pr = par + (size_t)r * ldp;
    const float v = pr[s] + dm_gauss_std(pr[S + s]) * eps[i];
    z[(size_t)r * ldz + s] = v;
    if (z_next) z_next[i] = (next_reset && next_reset[r]) ? 0.f : v;
  }
}
// backward of the reparameterised sample: dmean (+)= dz ; draw (+)= dz * eps * dstd/draw, eps = (z - mean) / std
__global__ void __launch_bounds__(256) gauss_sample_bwd_kernel(int rows, int S, const float* __restrict__ par, int ldp,
                                                               const float* __restrict__ z, int ldz,
                                                               const float* __restrict__ dz, int lddz,
                                                               float* __restrict__ dpar, int lddp, int accum) {
  const int total = rows * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int r = i / S, s = i % S;
    const float* pr = par + (size_t)r * ldp;
    const float sg = dm_sigmoid(pr[S + s]);
    const float sd = 2.0f * sg + 0.1f;
    const float e = (z[(size_t)r * ldz + s] - pr[s]) / sd;
    const float g = dz[(size_t)r * lddz + s];
    float* dp = dpar + (size_t)r * lddp;
    const float gm = g, gr = g * e * 2.0f * sg * (1.0f - sg);
    if (accum) { dp[s] += gm; dp[S + s] += gr; }
    else { dp[s] = gm; dp[S + s] = gr; }
  }
}
int dm_gauss_sample_bwd_launch(int rows, int S, const float* par, int ldp, const float* z, int ldz, const float* dz,
                               int lddz, float* dpar, int lddp, int accum, hipStream_t st) {
  if (rows <= 0) return DM_OK;
  hipLaunchKernelGGL(gauss_sample_bwd_kernel, dim3(ew_blocks((size_t)rows * S)), dim3(256), 0, st, rows, S, par, ldp, z, ldz,
                     dz, lddz, dpar, lddp, accum);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
// KL(N(m1,s1) || N(m2,s2)) summed over S, and the two entropies (0.5 log(2 pi e) + log std per dimension)
__global__ void __launch_bounds__(256) gauss_kl_fwd_kernel(int rows, int S, const float* __restrict__ post,
                                                           const float* __restrict__ prior, float* __restrict__ kl,
                                                           float* __restrict__ ent_post, float* __restrict__ ent_prior) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* a = post + (size_t)row * 2 * S;
  const float* b = prior + (size_t)row * 2 * S;
  float akl = 0.f, aep = 0.f, aeq = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float s1 = dm_gauss_std(a[S + s]), s2 = dm_gauss_std(b[S + s]);
    const float d = a[s] - b[s];
    akl += logf(s2 / s1) + (s1 * s1 + d * d) / (2.0f * s2 * s2) - 0.5f;
    aep += 1.4189385332046727f + logf(s1);
    aeq += 1.4189385332046727f + logf(s2);
  }
  akl = dm_wave_sum(akl); aep = dm_wave_sum(aep); aeq = dm_wave_sum(aeq);
  if (lane == 0) {
    kl[row] = akl;
    if (ent_post) ent_post[row] = aep;
    if (ent_prior) ent_prior[row] = aeq;
  }
}
__global__ void __launch_bounds__(256) gauss_kl_bwd_kernel(int rows, int S, const float* __restrict__ post,
                                                           const float* __restrict__ prior, float sp, float sq,
                                                           float* __restrict__ dpost, float* __restrict__ dprior) {
  const int total = rows * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int r = i / S, s = i % S;
    const size_t o = (size_t)r * 2 * S;
    const float g1 = dm_sigmoid(post[o + S + s]), g2 = dm_sigmoid(prior[o + S + s]);
    const float s1 = 2.0f * g1 + 0.1f, s2 = 2.0f * g2 + 0.1f;
    const float d = post[o + s] - prior[o + s];
    const float i22 = 1.0f / (s2 * s2);
    dpost[o + s] = sp * d * i22;
    dpost[o + S + s] = sp * (s1 * i22 - 1.0f / s1) * 2.0f * g1 * (1.0f - g1);
    dprior[o + s] = -sq * d * i22;
    dprior[o + S + s] = sq * (1.0f / s2 - (s1 * s1 + d * d) * i22 / s2) * 2.0f * g2 * (1.0f - g2);
  }
}

int dm_sample_onehot_launch(int rows, int groups, int C, const float* logits, int ldl, const float* u,
                            const int32_t* forced, float* onehot, int ldo, int32_t* idx, float* z_next,
                            const uint8_t* next_reset, hipStream_t st) {
  if (rows <= 0) return DM_OK;
  if (C == 0) {       // Gaussian latents: `logits` rows are (mean | raw std), `u` holds standard-normal draws, `onehot` gets z
    DM_REQUIRE(u && !forced, DM_E_NULL, "sample: Gaussian latents need their noise (and cannot be index-forced)");
    hipLaunchKernelGGL(gauss_sample_kernel, dim3(ew_blocks((size_t)rows * groups)), dim3(256), 0, st, rows, groups, logits,
                       ldl, u, onehot, ldo, z_next, next_reset);
    DM_LAUNCH_CHECK();
    if (idx && hipMemsetAsync(idx, 0, (size_t)rows * groups * sizeof(int32_t), st) != hipSuccess)
      return dm_fail(DM_E_HIP, "sample: memset failed");
    return DM_OK;
  }
  const size_t tg = (size_t)rows * groups;
  if (C == 32 && (ldl & 3) == 0 && (ldo & 3) == 0 &&
      (((uintptr_t)logits | (uintptr_t)onehot | (uintptr_t)z_next) & 15) == 0) {
    hipLaunchKernelGGL(sample_onehot_lane32_kernel, dim3((unsigned)dm_cdiv(tg, 256)), dim3(256), 0, st, rows, groups,
                       logits, ldl, u, forced, onehot, ldo, idx, z_next, next_reset);
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
#define DM_SAMPLE_WAVE(L)                                                                                              \
  hipLaunchKernelGGL((sample_onehot_wave_kernel<L>), dim3(ew_blocks(tg * L)), dim3(256), 0, st, rows, groups, C, logits, \
                     ldl, u, forced, onehot, ldo, idx)
  if (C <= 8) DM_SAMPLE_WAVE(8);
  else if (C <= 16) DM_SAMPLE_WAVE(16);
  else if (C <= 32) DM_SAMPLE_WAVE(32);
  else if (C <= 64) DM_SAMPLE_WAVE(64);
  else
    hipLaunchKernelGGL(sample_onehot_kernel, dim3(ew_blocks(tg)), dim3(256), 0, st, rows, groups, C, logits, ldl, u, forced,
                       onehot, ldo, idx);
#undef DM_SAMPLE_WAVE
  DM_LAUNCH_CHECK();
  if (z_next) DM_TRY(dm_mask_rows_launch(rows, groups * C, onehot, ldo, next_reset, z_next, groups * C, st));
  return DM_OK;
}
extern "C" int dm_sample_onehot(int rows, int groups, int C, const float* logits, int ldl, const float* u,
                                const int32_t* forced_idx, float* onehot, int ldo, int32_t* idx, void* stream) {
  DM_REQUIRE(logits && onehot && (u || forced_idx), DM_E_NULL, "sample_onehot: null pointer");
  DM_REQUIRE(C >= 0 && groups >= 1, DM_E_SHAPE, "sample_onehot: bad groups/C");       // C = 0: Gaussian latents
  return dm_sample_onehot_launch(rows, groups, C, logits, ldl, u, forced_idx, onehot, ldo, idx, nullptr, nullptr,
                                 (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// KL(post || prior) and entropies, summed over the S groups of each row (dreamer.py:326-343,369-379).
// One wave per row; lanes stride over groups.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dm_group_lse(const float* x, int C) {
  float mx = x[0];
  for (int k = 1; k < C; ++k) mx = fmaxf(mx, x[k]);
  float s = 0.f;
  for (int k = 0; k < C; ++k) s += expf(x[k] - mx);
  return mx + logf(s);
}

__global__ void __launch_bounds__(256) kl_fwd_kernel(int rows, int S, int C, const float* __restrict__ post,
                                                     const float* __restrict__ prior, float* __restrict__ kl,
                                                     float* __restrict__ ent_post, float* __restrict__ ent_prior) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float akl = 0.f, aep = 0.f, aeq = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float* a = post + ((size_t)row * S + s) * C;
    const float* b = prior + ((size_t)row * S + s) * C;
    const float la = dm_group_lse(a, C), lb = dm_group_lse(b, C);
    float gk = 0.f, ep = 0.f, eq = 0.f;
    for (int k = 0; k < C; ++k) {
      const float lp = a[k] - la, lq = b[k] - lb;
      const float p = expf(lp), q = expf(lq);
      gk += p * (lp - lq);
      ep -= p * lp;
      eq -= q * lq;
    }
    akl += gk; aep += ep; aeq += eq;
  }
  akl = dm_wave_sum(akl); aep = dm_wave_sum(aep); aeq = dm_wave_sum(aeq);
  if (lane == 0) {
    kl[row] = akl;
    if (ent_post) ent_post[row] = aep;
    if (ent_prior) ent_prior[row] = aeq;
  }
}

// dKL/dpost_k = p_k ((lp_k - lq_k) - KL_g) ; dKL/dprior_k = q_k - p_k
__global__ void __launch_bounds__(256) kl_bwd_kernel(int rows, int S, int C, const float* __restrict__ post,
                                                     const float* __restrict__ prior, float sp, float sq,
                                                     float* __restrict__ dpost, float* __restrict__ dprior) {
  const int total = rows * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const float* a = post + (size_t)i * C;
    const float* b = prior + (size_t)i * C;
    const float la = dm_group_lse(a, C), lb = dm_group_lse(b, C);
    float gk = 0.f;
    for (int k = 0; k < C; ++k) {
      const float lp = a[k] - la, lq = b[k] - lb;
      gk += expf(lp) * (lp - lq);
    }
    for (int k = 0; k < C; ++k) {
      const float lp = a[k] - la, lq = b[k] - lb;
      const float p = expf(lp), q = expf(lq);
      dpost[(size_t)i * C + k] = sp * p * ((lp - lq) - gk);
      dprior[(size_t)i * C + k] = sq * (q - p);
    }
  }
}

// ---- IWAE (iwae_samples = I > 1; dreamer.py:340-343,362-365, functions.py:97-102, rssm.py:35-41) ------------------
// Sampled KL term: loss_kl[n] = log q(z_n) - log p(z_n) with z the drawn one-hot sample.  OneHotCategorical.log_prob
// goes through value.max(-1) (indices), so no gradient flows through z itself:
//   d/dpost_k = 1[k = idx] - softmax(post)_k ,  d/dprior_k = -(1[k = idx] - softmax(prior)_k).
__global__ void __launch_bounds__(256) kl_sampled_fwd_kernel(int rows, int S, int C, const float* __restrict__ post,
                                                             const float* __restrict__ prior, const int32_t* __restrict__ idx,
                                                             float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float acc = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float* a = post + ((size_t)row * S + s) * C;
    const float* b = prior + ((size_t)row * S + s) * C;
    const int k = idx[(size_t)row * S + s];
    acc += (a[k] - dm_group_lse(a, C)) - (b[k] - dm_group_lse(b, C));
  }
  acc = dm_wave_sum(acc);
  if (lane == 0) out[row] = acc;
}
__global__ void __launch_bounds__(256) kl_sampled_bwd_kernel(int rows, int S, int C, const float* __restrict__ post,
                                                             const float* __restrict__ prior, const int32_t* __restrict__ idx,
                                                             float scale, const float* __restrict__ row_w,
                                                             float* __restrict__ dpost, float* __restrict__ dprior) {
  const int total = rows * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const float* a = post + (size_t)i * C;
    const float* b = prior + (size_t)i * C;
    const float la = dm_group_lse(a, C), lb = dm_group_lse(b, C);
    const float w = scale * (row_w ? row_w[i / S] : 1.f);
    const int kk = idx[i];
    for (int k = 0; k < C; ++k) {
      const float hot = k == kk ? 1.f : 0.f;
      dpost[(size_t)i * C + k] = w * (hot - expf(a[k] - la));
      dprior[(size_t)i * C + k] = -w * (hot - expf(b[k] - lb));
    }
  }
}
extern "C" int dm_kl_sampled_fwd(int rows, int S, int C, const float* post, const float* prior, const int32_t* idx,
                                 float* out, void* stream) {
  DM_REQUIRE(post && prior && idx && out, DM_E_NULL, "kl_sampled_fwd: null pointer");
  if (rows <= 0) return DM_OK;
  hipLaunchKernelGGL(kl_sampled_fwd_kernel, dim3(dm_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, rows, S, C, post, prior,
                     idx, out);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
extern "C" int dm_kl_sampled_bwd(int rows, int S, int C, const float* post, const float* prior, const int32_t* idx,
                                 float scale, const float* row_w, float* dpost, float* dprior, void* stream) {
  DM_REQUIRE(post && prior && idx && dpost && dprior, DM_E_NULL, "kl_sampled_bwd: null pointer");
  if (rows <= 0) return DM_OK;
  int blocks = dm_cdiv((size_t)rows * S, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(kl_sampled_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rows, S, C, post, prior, idx,
                     scale, row_w, dpost, dprior);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// Gaussian latents (stoch_discrete = 0) under IWAE: loss_kl[n] = log q(z_n) - log p(z_n) with q = N(m1, s1), p = N(m2, s2)
// diagonal (rssm.py:202-203 diag_normal, std = 2 sigmoid(raw) + 0.1) and z the REPARAMETERISED posterior sample
// (Normal.rsample: z = m1 + s1 eps), so - unlike the one-hot case above - a gradient also flows through z itself:
//   per dimension  f = -log s1 - (z - m1)^2 / (2 s1^2) + log s2 + (z - m2)^2 / (2 s2^2)
//   df/dm1 = (z - m1)/s1^2     df/ds1 = -1/s1 + (z - m1)^2/s1^3     df/dm2 = -(z - m2)/s2^2     df/ds2 = 1/s2 - (z - m2)^2/s2^3
//   df/dz  = -(z - m1)/s1^2 + (z - m2)/s2^2     (added to the sample's gradient; the BPTT pass carries it into (m1, s1))
__global__ void __launch_bounds__(256) gauss_kl_sampled_fwd_kernel(int rows, int S, const float* __restrict__ post,
                                                                   const float* __restrict__ prior, const float* __restrict__ z,
                                                                   int ldz, float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* a = post + (size_t)row * 2 * S;
  const float* b = prior + (size_t)row * 2 * S;
  float acc = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float s1 = dm_gauss_std(a[S + s]), s2 = dm_gauss_std(b[S + s]);
    const float zz = z[(size_t)row * ldz + s];
    const float d1 = (zz - a[s]) / s1, d2 = (zz - b[s]) / s2;
    acc += (-logf(s1) - 0.5f * d1 * d1) - (-logf(s2) - 0.5f * d2 * d2);
  }
  acc = dm_wave_sum(acc);
  if (lane == 0) out[row] = acc;
}
__global__ void __launch_bounds__(256) gauss_kl_sampled_bwd_kernel(int rows, int S, const float* __restrict__ post,
                                                                   const float* __restrict__ prior, const float* __restrict__ z,
                                                                   int ldz, float scale, const float* __restrict__ row_w,
                                                                   float* __restrict__ dpost, float* __restrict__ dprior,
                                                                   float* __restrict__ dz, int lddz) {
  const int total = rows * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int r = i / S, s = i % S;
    const size_t o = (size_t)r * 2 * S;
    const float g1 = dm_sigmoid(post[o + S + s]), g2 = dm_sigmoid(prior[o + S + s]);
    const float s1 = 2.0f * g1 + 0.1f, s2 = 2.0f * g2 + 0.1f;
    const float zz = z[(size_t)r * ldz + s];
    const float e1 = zz - post[o + s], e2 = zz - prior[o + s];
    const float w = scale * (row_w ? row_w[r] : 1.f);
    const float i11 = 1.0f / (s1 * s1), i22 = 1.0f / (s2 * s2);
    dpost[o + s] = w * e1 * i11;
    dpost[o + S + s] = w * (-1.0f / s1 + e1 * e1 * i11 / s1) * 2.0f * g1 * (1.0f - g1);
    dprior[o + s] = -w * e2 * i22;
    dprior[o + S + s] = w * (1.0f / s2 - e2 * e2 * i22 / s2) * 2.0f * g2 * (1.0f - g2);
    dz[(size_t)r * lddz + s] += w * (-e1 * i11 + e2 * i22);
  }
}
extern "C" int dm_kl_sampled_gauss_fwd(int rows, int S, const float* post, const float* prior, const float* z, int ldz,
                                       float* out, void* stream) {
  DM_REQUIRE(post && prior && z && out, DM_E_NULL, "kl_sampled_gauss_fwd: null pointer");
  if (rows <= 0) return DM_OK;
  hipLaunchKernelGGL(gauss_kl_sampled_fwd_kernel, dim3(dm_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, rows, S, post, prior,
                     z, ldz, out);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
extern "C" int dm_kl_sampled_gauss_bwd(int rows, int S, const float* post, const float* prior, const float* z, int ldz,
                                       float scale, const float* row_w, float* dpost, float* dprior, float* dz, int lddz,
                                       void* stream) {
  DM_REQUIRE(post && prior && z && dpost && dprior && dz, DM_E_NULL, "kl_sampled_gauss_bwd: null pointer");
  if (rows <= 0) return DM_OK;
  int blocks = dm_cdiv((size_t)rows * S, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gauss_kl_sampled_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rows, S, post, prior, z, ldz,
                     scale, row_w, dpost, dprior, dz, lddz);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// Reductions over the I samples of every (t,b): x (TB, I, W) -> out (TB, W).
//   mode 0: mean_i x        mode 2: sum_i x
//   mode 1 (W = 1): -logavgexp_i(-x) = -(logsumexp_i(-x) - log I)  (functions.py:97-102), and optionally the importance
//           weights w[tb,i] = softmax_i(-x) = d out / d x_i  (the factor every per-sample gradient of loss_model carries)
__global__ void __launch_bounds__(256) reduce_i_kernel(int TB, int I, int W, const float* __restrict__ x, int mode,
                                                       float* __restrict__ out, float* __restrict__ w_out) {
  const size_t total = (size_t)TB * W;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t tb = e / W;
    const int c = (int)(e % W);
    const float* p = x + (tb * I) * W + c;
    if (mode == 1) {
      float mx = -p[0];
      for (int i = 1; i < I; ++i) mx = fmaxf(mx, -p[(size_t)i * W]);
      float s = 0.f;
      for (int i = 0; i < I; ++i) s += expf(-p[(size_t)i * W] - mx);
      out[e] = -(mx + logf(s) - logf((float)I));
      if (w_out)
        for (int i = 0; i < I; ++i) w_out[tb * I + i] = expf(-p[(size_t)i * W] - mx) / s;
    } else {
      float s = 0.f;
      for (int i = 0; i < I; ++i) s += p[(size_t)i * W];
      out[e] = mode == 0 ? s / (float)I : s;
    }
  }
}
extern "C" int dm_reduce_i(int TB, int I, int W, const float* x, int mode, float* out, float* w_out, void* stream) {
  DM_REQUIRE(x && out, DM_E_NULL, "reduce_i: null pointer");
  DM_REQUIRE(I >= 1 && W >= 1 && (unsigned)mode <= 2 && (mode != 1 || W == 1), DM_E_SHAPE, "reduce_i: I=%d W=%d mode=%d", I, W, mode);
  if (TB <= 0) return DM_OK;
  int blocks = dm_cdiv((size_t)TB * W, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reduce_i_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, TB, I, W, x, mode, out, w_out);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// out[r] = sum_j w[j] * x_j[r]  (j < count <= 8): the per-sample model loss kl_weight*KL + image_w*.. (dreamer.py:362)
struct RowCombineArgs { const float* x[8]; float w[8]; };
__global__ void __launch_bounds__(256) combine_rows_kernel(int count, long long n, const RowCombineArgs a, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float s = 0.f;
    for (int j = 0; j < count; ++j) s += a.w[j] * a.x[j][i];
    out[i] = s;
  }
}
extern "C" int dm_combine_rows(int count, int64_t n, const float* const* x, const float* w, float* out, void* stream) {
  DM_REQUIRE(x && w && out, DM_E_NULL, "combine_rows: null pointer");
  DM_REQUIRE(count >= 1 && count <= 8, DM_E_SHAPE, "combine_rows: count %d not in [1,8]", count);
  if (n <= 0) return DM_OK;
  RowCombineArgs a;
  for (int j = 0; j < count; ++j) { a.x[j] = x[j]; a.w[j] = w[j]; }
  int blocks = dm_cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(combine_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, count, (long long)n, a, out);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// x[r, 0..n) *= w[r] * scale
__global__ void __launch_bounds__(256) scale_rows_kernel(long long rows, int n, float* __restrict__ x, int ldx,
                                                         const float* __restrict__ w, float scale) {
  const long long total = rows * n;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / n;
    x[r * ldx + e % n] *= w[r] * scale;
  }
}
extern "C" int dm_scale_rows(int64_t rows, int n, float* x, int ldx, const float* w, float scale, void* stream) {
  DM_REQUIRE(x && w, DM_E_NULL, "scale_rows: null pointer");
  if (rows <= 0 || n <= 0) return DM_OK;
  int blocks = dm_cdiv(rows * n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long long)rows, n, x, ldx, w, scale);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// straight-through: sample = onehot + (p - sg(p)) => dlogits_k = p_k (g_k - sum_j p_j g_j)
// LPG lanes per group (power of two >= C, <= 64): lane k of a group owns category k; reductions are xor-shuffles
// inside the group's lanes, so the 50-row RSSM steps cost one short wave each instead of a 96-expf serial chain.
template <int LPG>
__global__ void __launch_bounds__(256) st_softmax_bwd_kernel(int rows, int groups, int C, const float* __restrict__ logits,
                                                             int ldl, const float* __restrict__ dz, int lddz,
                                                             float* __restrict__ dlogits, int lddl, int accum) {
  constexpr int GPB = 256 / LPG;                      // groups per block
  const int k = threadIdx.x % LPG;
  const int total = rows * groups;
  for (int i0 = blockIdx.x * GPB; i0 < total; i0 += gridDim.x * GPB) {
    const int i = i0 + threadIdx.x / LPG;
    const bool live = i < total && k < C;
    const int r = live ? i / groups : 0, gq = live ? i % groups : 0;
    const size_t off = (size_t)gq * C + k;
    const float x = live ? logits[(size_t)r * ldl + off] : -INFINITY;
    const float g = live ? dz[(size_t)r * lddz + off] : 0.f;
    float mx = x;
#pragma unroll
    for (int o = LPG / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float e = live ? expf(x - mx) : 0.f;
    float se = e, sg = e * g;
#pragma unroll
    for (int o = LPG / 2; o > 0; o >>= 1) {
      se += __shfl_xor(se, o, 64);
      sg += __shfl_xor(sg, o, 64);
    }
    if (live) {
      const float p = e / se;
      const float v = p * (g - sg / se);
      float* o = dlogits + (size_t)r * lddl + off;
      *o = accum ? *o + v : v;
    }
  }
}

int dm_kl_fwd_launch(int rows, int S, int C, const float* post, const float* prior, float* kl, float* ep, float* eq,
                     hipStream_t st) {
  if (rows <= 0) return DM_OK;
  if (C == 0) {
    hipLaunchKernelGGL(gauss_kl_fwd_kernel, dim3(dm_cdiv(rows, 4)), dim3(256), 0, st, rows, S, post, prior, kl, ep, eq);
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(dm_cdiv(rows, 4)), dim3(256), 0, st, rows, S, C, post, prior, kl, ep, eq);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
int dm_kl_bwd_launch(int rows, int S, int C, const float* post, const float* prior, float sp, float sq, float* dpost,
                     float* dprior, hipStream_t st) {
  if (rows <= 0) return DM_OK;
  if (C == 0) {
    hipLaunchKernelGGL(gauss_kl_bwd_kernel, dim3(ew_blocks((size_t)rows * S)), dim3(256), 0, st, rows, S, post, prior, sp, sq,
                       dpost, dprior);
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(ew_blocks((size_t)rows * S)), dim3(256), 0, st, rows, S, C, post, prior, sp, sq,
                     dpost, dprior);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
int dm_st_softmax_bwd_launch(int rows, int groups, int C, const float* logits, int ldl, const float* dz, int lddz,
                             float* dlogits, int lddl, int accum, hipStream_t st) {
  if (rows <= 0) return DM_OK;
  DM_REQUIRE(C >= 1 && C <= 64, DM_E_SHAPE, "st_softmax_bwd: C=%d not in [1,64]", C);
  const size_t tg = (size_t)rows * groups;
  if (C <= 8) hipLaunchKernelGGL((st_softmax_bwd_kernel<8>), dim3(ew_blocks(tg * 8)), dim3(256), 0, st, rows, groups, C, logits, ldl, dz, lddz, dlogits, lddl, accum);
  else if (C <= 16) hipLaunchKernelGGL((st_softmax_bwd_kernel<16>), dim3(ew_blocks(tg * 16)), dim3(256), 0, st, rows, groups, C, logits, ldl, dz, lddz, dlogits, lddl, accum);
  else if (C <= 32) hipLaunchKernelGGL((st_softmax_bwd_kernel<32>), dim3(ew_blocks(tg * 32)), dim3(256), 0, st, rows, groups, C, logits, ldl, dz, lddz, dlogits, lddl, accum);
  else hipLaunchKernelGGL((st_softmax_bwd_kernel<64>), dim3(ew_blocks(tg * 64)), dim3(256), 0, st, rows, groups, C, logits, ldl, dz, lddz, dlogits, lddl, accum);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
extern "C" int dm_kl_balance_fwd(int rows, int S, int C, const float* post, const float* prior, float* kl,
                                 float* ent_post, float* ent_prior, void* stream) {
  DM_REQUIRE(post && prior && kl, DM_E_NULL, "kl_fwd: null pointer");
  return dm_kl_fwd_launch(rows, S, C, post, prior, kl, ent_post, ent_prior, (hipStream_t)stream);
}
extern "C" int dm_kl_balance_bwd(int rows, int S, int C, const float* post, const float* prior, float scale_post,
                                 float scale_prior, float* dpost, float* dprior, void* stream) {
  DM_REQUIRE(post && prior && dpost && dprior, DM_E_NULL, "kl_bwd: null pointer");
  return dm_kl_bwd_launch(rows, S, C, post, prior, scale_post, scale_prior, dpost, dprior, (hipStream_t)stream);
}
extern "C" int dm_st_softmax_bwd(int rows, int groups, int C, const float* logits, int ldl, const float* dz, int lddz,
                                 float* dlogits, int lddl, int accum, void* stream) {
  DM_REQUIRE(logits && dz && dlogits, DM_E_NULL, "st_softmax_bwd: null pointer");
  return dm_st_softmax_bwd_launch(rows, groups, C, logits, ldl, dz, lddz, dlogits, lddl, accum, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// reset masks (rssm.py:41,134-135)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mask_rows_kernel(int rows, int n, const float* __restrict__ x, int ldx,
                                                        const uint8_t* __restrict__ reset, float* __restrict__ y, int ldy) {
  const size_t total = (size_t)rows * n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / n), c = (int)(i % n);
    const float m = reset[r] ? 0.f : 1.f;
    y[(size_t)r * ldy + c] = x[(size_t)r * ldx + c] * m;
  }
}
// both recurrent-state halves in one launch: y1 = x1 * m (n1 columns), y2 = x2 * m (n2 columns)
__global__ void __launch_bounds__(256) mask_rows2_kernel(int rows, int n1, const float* __restrict__ x1, int ldx1,
                                                         float* __restrict__ y1, int ldy1, int n2,
                                                         const float* __restrict__ x2, int ldx2, float* __restrict__ y2,
                                                         int ldy2, const uint8_t* __restrict__ reset) {
  const int n = n1 + n2;
  const size_t total = (size_t)rows * n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / n), c = (int)(i % n);
    const float m = reset[r] ? 0.f : 1.f;
    if (c < n1) y1[(size_t)r * ldy1 + c] = x1[(size_t)r * ldx1 + c] * m;
    else y2[(size_t)r * ldy2 + (c - n1)] = x2[(size_t)r * ldx2 + (c - n1)] * m;
  }
}
int dm_mask_rows2_launch(int rows, int n1, const float* x1, int ldx1, float* y1, int ldy1, int n2, const float* x2,
                         int ldx2, float* y2, int ldy2, const uint8_t* reset, hipStream_t st) {
  if (rows <= 0) return DM_OK;
  hipLaunchKernelGGL(mask_rows2_kernel, dim3(ew_blocks((size_t)rows * (n1 + n2))), dim3(256), 0, st, rows, n1, x1, ldx1, y1,
                     ldy1, n2, x2, ldx2, y2, ldy2, reset);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
int dm_mask_rows_launch(int rows, int n, const float* x, int ldx, const uint8_t* reset, float* y, int ldy, hipStream_t st) {
  if (rows <= 0 || n <= 0) return DM_OK;
  hipLaunchKernelGGL(mask_rows_kernel, dim3(ew_blocks((size_t)rows * n)), dim3(256), 0, st, rows, n, x, ldx, reset, y, ldy);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
extern "C" int dm_mask_rows(int rows, int n, const float* x, int ldx, const uint8_t* reset, float* y, int ldy,
                            void* stream) {
  DM_REQUIRE(x && reset && y, DM_E_NULL, "mask_rows: null pointer");
  return dm_mask_rows_launch(rows, n, x, ldx, reset, y, ldy, (hipStream_t)stream);
}

// y = dy * ELU'(yact)
__global__ void __launch_bounds__(256) mul_elu_grad_kernel(size_t n, const float* __restrict__ dy,
                                                           const float* __restrict__ yact, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = dy[i] * dm_elu_grad_from_y(yact[i]);
}
int dm_mul_elu_grad_launch(size_t n, const float* dy, const float* yact, float* out, hipStream_t st) {
  if (n == 0) return DM_OK;
  hipLaunchKernelGGL(mul_elu_grad_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, n, dy, yact, out);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// ------------------------------------------------------------------------------------------------
// dense-head loss epilogues (decoders.py:257-319)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) head_loss_kernel(int kind, int rows, const float* __restrict__ out,
                                                        const float* __restrict__ target, float scale, float loss_const,
                                                        float* __restrict__ loss, float* __restrict__ dout,
                                                        float* __restrict__ mean_out) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256) {
    const float o = out[i];
    const float y = target ? target[i] : 0.f;
    if (kind == 0) {
      const float d = o - y;
      if (loss) loss[i] = 0.5f * d * d + loss_const;
      if (dout) dout[i] = scale * d;
      if (mean_out) mean_out[i] = o;
    } else {
      // -Bernoulli(logits=o).log_prob(y) = max(o,0) - o*y + log1p(exp(-|o|))
      const float sg = dm_sigmoid(o);
      if (loss) loss[i] = fmaxf(o, 0.f) - o * y + log1pf(expf(-fabsf(o)));
      if (dout) dout[i] = scale * (sg - y);
      if (mean_out) mean_out[i] = sg;
    }
  }
}
extern "C" int dm_head_loss(int kind, int rows, const float* out, const float* target, float scale, float loss_const,
                            float* loss, float* dout, float* mean_out, void* stream) {
  DM_REQUIRE(out, DM_E_NULL, "head_loss: null pointer");
  DM_REQUIRE(kind == 0 || kind == 1, DM_E_SHAPE, "head_loss: kind %d", kind);
  if (rows <= 0) return DM_OK;
  hipLaunchKernelGGL(head_loss_kernel, dim3(ew_blocks(rows)), dim3(256), 0, (hipStream_t)stream, kind, rows, out, target,
                     scale, loss_const, loss, dout, mean_out);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// ------------------------------------------------------------------------------------------------
// GAE reverse scan + reality weight, one thread per dream column (a2c.py:81-108)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gae_kernel(int H, int M, float gamma, float lambda, const float* __restrict__ reward,
                                                  const float* __restrict__ terminal, const float* __restrict__ value_t,
                                                  float* __restrict__ advantage, float* __restrict__ advantage_gae,
                                                  float* __restrict__ value_target, float* __restrict__ weight) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float agae = 0.f;
  for (int t = H - 1; t >= 0; --t) {
    const float v0 = value_t[(size_t)t * M + m];
    const float v1 = value_t[(size_t)(t + 1) * M + m];
    const float r1 = reward[(size_t)(t + 1) * M + m];
    const float nt1 = 1.0f - terminal[(size_t)(t + 1) * M + m];
    const float adv = -v0 + r1 + gamma * nt1 * v1;
    agae = (t == H - 1) ? adv : adv + lambda * gamma * nt1 * agae;
    advantage[(size_t)t * M + m] = adv;
    advantage_gae[(size_t)t * M + m] = agae;
    value_target[(size_t)t * M + m] = agae + v0;
  }
  float cs = 0.f;
  for (int t = 0; t < H; ++t) {
    cs += logf(1.0f - terminal[(size_t)t * M + m]);
    weight[(size_t)t * M + m] = expf(cs);
  }
}
extern "C" int dm_gae_losses(int H, int M, float gamma, float lambda, const float* reward, const float* terminal,
                             const float* value_t, float* advantage, float* advantage_gae, float* value_target,
                             float* weight, void* stream) {
  DM_REQUIRE(reward && terminal && value_t && advantage && advantage_gae && value_target && weight, DM_E_NULL,
             "gae: null pointer");
  DM_REQUIRE(H >= 1 && M >= 1, DM_E_SHAPE, "gae: H=%d M=%d", H, M);
  hipLaunchKernelGGL(gae_kernel, dim3(dm_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, H, M, gamma, lambda, reward,
                     terminal, value_t, advantage, advantage_gae, value_target, weight);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// actor (reinforce, OneHotCategorical) rows (a2c.py:119-130)
__global__ void __launch_bounds__(256) actor_loss_kernel(int rows, int A, const float* __restrict__ logits,
                                                         const int32_t* __restrict__ act_idx,
                                                         const float* __restrict__ adv, const float* __restrict__ weight,
                                                         float ent_w, float scale, float* __restrict__ loss,
                                                         float* __restrict__ entropy, float* __restrict__ dlogits) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256) {
    const float* x = logits + (size_t)i * A;
    const float l = dm_group_lse(x, A);
    float ent = 0.f;
    for (int k = 0; k < A; ++k) {
      const float lp = x[k] - l;
      ent -= expf(lp) * lp;
    }
    const int a = act_idx[i];
    const float lpa = x[a] - l;
    const float w = weight[i], ad = adv[i];
    if (loss) loss[i] = (-lpa * ad - ent_w * ent) * w;
    if (entropy) entropy[i] = ent;
    if (dlogits) {
      const float sw = scale * w;
      for (int k = 0; k < A; ++k) {
        const float lp = x[k] - l;
        const float p = expf(lp);
        const float dpol = -ad * ((k == a ? 1.f : 0.f) - p);
        const float dent = ent_w * p * (lp + ent);
        dlogits[(size_t)i * A + k] = sw * (dpol + dent);
      }
    }
  }
}
extern "C" int dm_actor_loss(int rows, int A, const float* logits, const int32_t* act_idx, const float* adv_gae,
                             const float* weight, float ent_w, float scale, float* loss, float* entropy, float* dlogits,
                             void* stream) {
  DM_REQUIRE(logits && act_idx && adv_gae && weight, DM_E_NULL, "actor_loss: null pointer");
  if (rows <= 0) return DM_OK;
  hipLaunchKernelGGL(actor_loss_kernel, dim3(ew_blocks(rows)), dim3(256), 0, (hipStream_t)stream, rows, A, logits,
                     act_idx, adv_gae, weight, ent_w, scale, loss, entropy, dlogits);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

__global__ void __launch_bounds__(256) critic_loss_kernel(int rows, const float* __restrict__ value,
                                                          const float* __restrict__ vt, const float* __restrict__ weight,
                                                          float scale, float* __restrict__ loss, float* __restrict__ dvalue) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256) {
    const float d = vt[i] - value[i];
    const float w = weight[i];
    if (loss) loss[i] = 0.5f * d * d * w;
    if (dvalue) dvalue[i] = -scale * d * w;
  }
}
extern "C" int dm_critic_loss(int rows, const float* value, const float* value_target, const float* weight, float scale,
                              float* loss, float* dvalue, void* stream) {
  DM_REQUIRE(value && value_target && weight, DM_E_NULL, "critic_loss: null pointer");
  if (rows <= 0) return DM_OK;
  hipLaunchKernelGGL(critic_loss_kernel, dim3(ew_blocks(rows)), dim3(256), 0, (hipStream_t)stream, rows, value,
                     value_target, weight, scale, loss, dvalue);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// ------------------------------------------------------------------------------------------------
// multi-array sums: one block per array, deterministic order
// ------------------------------------------------------------------------------------------------
struct MultiSumArgs {
  const float* x[32];
  const float* center[32];
  long long n[32];
  float scale[32];
  int mode[32];
};
__global__ void __launch_bounds__(256) multi_sum_kernel(const MultiSumArgs a, float* __restrict__ out) {
  __shared__ float red[4];
  const int item = blockIdx.x;
  const float* x = a.x[item];
  const long long n = a.n[item];
  float s = 0.f;
  if (a.mode[item] == 0) {
    for (long long i = threadIdx.x; i < n; i += 256) s += x[i];
  } else {      // modes 1, 2: squared deviations from a device scalar
    const float c = a.center[item][0];
    for (long long i = threadIdx.x; i < n; i += 256) {
      const float d = x[i] - c;
      s += d * d;
    }
  }
  s = dm_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (red[0] + red[1] + red[2] + red[3]) * a.scale[item];
    out[item] = a.mode[item] == 2 ? sqrtf(v) : v;
  }
}
extern "C" int dm_multi_sum(int count, const dm_reduce_item* items, float* out, void* stream) {
  DM_REQUIRE(items && out, DM_E_NULL, "multi_sum: null pointer");
  DM_REQUIRE(count >= 1 && count <= 32, DM_E_SHAPE, "multi_sum: count %d not in [1,32]", count);
  MultiSumArgs a;
  for (int i = 0; i < count; ++i) {
    DM_REQUIRE(items[i].x || items[i].n == 0, DM_E_NULL, "multi_sum: item %d null", i);
    DM_REQUIRE(items[i].mode == 0 || ((items[i].mode == 1 || items[i].mode == 2) && items[i].center), DM_E_SHAPE,
               "multi_sum: item %d bad mode", i);
    a.x[i] = items[i].x;
    a.center[i] = items[i].center;
    a.n[i] = items[i].n;
    a.scale[i] = items[i].scale;
    a.mode[i] = items[i].mode;
  }
  hipLaunchKernelGGL(multi_sum_kernel, dim3(count), dim3(256), 0, (hipStream_t)stream, a, out);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

struct CombineArgs { float w[16]; };
__global__ void combine_kernel(int count, const float* __restrict__ x, const CombineArgs a, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < count; ++i) s += a.w[i] * x[i];
    out[0] = s;
  }
}
extern "C" int dm_combine(int count, const float* x, const float* w, float* out, void* stream) {
  DM_REQUIRE(x && w && out, DM_E_NULL, "combine: null pointer");
  DM_REQUIRE(count >= 1 && count <= 16, DM_E_SHAPE, "combine: count %d not in [1,16]", count);
  CombineArgs a;
  for (int i = 0; i < count; ++i) a.w[i] = w[i];
  hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, count, x, a, out);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// ------------------------------------------------------------------------------------------------
// Continuous actors (functions.py:59-78, a2c.py:43-55,119-130; DMC config).  params row = [mean_raw (A) | std_raw (A)].
//   kind 1  tanh_normal : mean = 5 tanh(m/5), std = softplus(s) + 0.1, action = tanh(N(mean, std)); entropy is the base
//                         Normal's (the reference's own "HACK", functions.py:77)
//   kind 2  normal_tanh : mean = tanh(m), std = sigmoid(s) + 0.01, action = N(mean, std) (no squashing)
// The sample uses explicit standard-normal noise eps: x = mean + std * eps (torch.normal restated).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dm_softplus(float v) { return v > 20.f ? v : log1pf(expf(v)); }

__device__ __forceinline__ void dm_cont_params(int kind, float m, float s, float* mean, float* std, float* dmean_dm,
                                               float* dstd_ds) {
  if (kind == 1) {
    const float t = tanhf(m / 5.f);
    *mean = 5.f * t;
    *dmean_dm = 1.f - t * t;
    *std = dm_softplus(s) + 0.1f;
    *dstd_ds = dm_sigmoid(s);
  } else {
    const float t = tanhf(m);
    *mean = t;
    *dmean_dm = 1.f - t * t;
    const float sg = dm_sigmoid(s);
    *std = sg + 0.01f;
    *dstd_ds = sg * (1.f - sg);
  }
}

__global__ void __launch_bounds__(256) sample_continuous_kernel(int kind, int rows, int A, const float* __restrict__ params,
                                                                const float* __restrict__ eps, float* __restrict__ action) {
  const int total = rows * A;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int r = i / A, a = i % A;
    float mean, std, d0, d1;
    dm_cont_params(kind, params[(size_t)r * 2 * A + a], params[(size_t)r * 2 * A + A + a], &mean, &std, &d0, &d1);
    const float x = mean + std * eps[i];
    action[i] = kind == 1 ? tanhf(x) : x;
  }
}
extern "C" int dm_sample_continuous(int kind, int rows, int A, const float* params, const float* eps, float* action,
                                    void* stream) {
  DM_REQUIRE(params && eps && action, DM_E_NULL, "sample_continuous: null pointer");
  DM_REQUIRE(kind == 1 || kind == 2, DM_E_SHAPE, "sample_continuous: kind %d", kind);
  if (rows <= 0) return DM_OK;
  hipLaunchKernelGGL(sample_continuous_kernel, dim3(ew_blocks((size_t)rows * A)), dim3(256), 0, (hipStream_t)stream, kind,
                     rows, A, params, eps, action);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
int dm_sample_continuous_launch(int kind, int rows, int A, const float* params, const float* eps, float* action,
                                hipStream_t st) {
  return dm_sample_continuous(kind, rows, A, params, eps, action, (void*)st);
}

// reinforce loss rows for continuous actors: loss[r] = (-log pi(a) * adv - ent_w * H) * w ; dparams = scale * dloss/dparams.
// log pi(a) for kind 1 = sum_a [ logN(x; mean, std) - 2 (log 2 - x - softplus(-2x)) ], x = atanh(a)   (TanhTransform)
__global__ void __launch_bounds__(256) actor_loss_continuous_kernel(int kind, int rows, int A,
                                                                    const float* __restrict__ params,
                                                                    const float* __restrict__ actions,
                                                                    const float* __restrict__ adv,
                                                                    const float* __restrict__ weight, float ent_w,
                                                                    float scale, float* __restrict__ loss,
                                                                    float* __restrict__ entropy,
                                                                    float* __restrict__ dparams) {
  const float LOG_SQRT_2PI = 0.9189385332046727f;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
    const float w = weight[r], ad = adv[r];
    float logp = 0.f, ent = 0.f;
    for (int a = 0; a < A; ++a) {
      float mean, std, dm_, ds_;
      dm_cont_params(kind, params[(size_t)r * 2 * A + a], params[(size_t)r * 2 * A + A + a], &mean, &std, &dm_, &ds_);
      const float y = actions[(size_t)r * A + a];
      const float x = kind == 1 ? atanhf(y) : y;
      const float z = (x - mean) / std;
      logp += -0.5f * z * z - logf(std) - LOG_SQRT_2PI;
      if (kind == 1) logp -= 2.f * (0.6931471805599453f - x - dm_softplus(-2.f * x));
      ent += 0.5f + LOG_SQRT_2PI + logf(std);
      if (dparams) {
        // d(-logp*ad - ent_w*ent)/dmean = -ad * z/std ; /dstd = -ad * (z*z - 1)/std - ent_w / std
        const float dmean = -ad * z / std;
        const float dstd = -ad * (z * z - 1.f) / std - ent_w / std;
        dparams[(size_t)r * 2 * A + a] = scale * w * dmean * dm_;
        dparams[(size_t)r * 2 * A + A + a] = scale * w * dstd * ds_;
      }
    }
    if (loss) loss[r] = (-logp * ad - ent_w * ent) * w;
    if (entropy) entropy[r] = ent;
  }
}
extern "C" int dm_actor_loss_continuous(int kind, int rows, int A, const float* params, const float* actions,
                                        const float* adv_gae, const float* weight, float ent_w, float scale, float* loss,
                                        float* entropy, float* dparams, void* stream) {
  DM_REQUIRE(params && actions && adv_gae && weight, DM_E_NULL, "actor_loss_continuous: null pointer");
  DM_REQUIRE(kind == 1 || kind == 2, DM_E_SHAPE, "actor_loss_continuous: kind %d", kind);
  if (rows <= 0) return DM_OK;
  hipLaunchKernelGGL(actor_loss_continuous_kernel, dim3(ew_blocks(rows)), dim3(256), 0, (hipStream_t)stream, kind, rows, A,
                     params, actions, adv_gae, weight, ent_w, scale, loss, entropy, dparams);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// ---- fp32 -> bf16 (RNE) copies of up to 8 buffers in one launch: the weight twins of a bf16-mode call (common.h DmTwinScope)
struct CvtArgs { DmCvtSeg seg[8]; size_t start[9]; };      // start[i]: first 4-element group of segment i in the flat group index
__global__ void __launch_bounds__(256) to_bf16_multi_kernel(const CvtArgs a, int count) {
  const size_t total = a.start[count];
  for (size_t gidx = (size_t)blockIdx.x * 256 + threadIdx.x; gidx < total; gidx += (size_t)gridDim.x * 256) {
    int sidx = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) sidx += (i < count && gidx >= a.start[i]) ? 1 : 0;
    const DmCvtSeg sg = a.seg[sidx];
    const size_t e = (gidx - a.start[sidx]) * 4;
    if (e + 4 <= sg.n && ((((uintptr_t)sg.src) & 15) == 0) && ((((uintptr_t)sg.dst) & 7) == 0)) {
      const float4 v = *reinterpret_cast<const float4*>(sg.src + e);
      *reinterpret_cast<uint2*>(sg.dst + e) = dm_pack_bf16x4(v);
    } else {
      for (size_t j = e; j < sg.n && j < e + 4; ++j) sg.dst[j] = (unsigned short)dm_f2bf(sg.src[j]);
    }
  }
}
int dm_to_bf16_multi_launch(const DmCvtSeg* segs, int count, hipStream_t st) {
  DM_REQUIRE(count >= 0 && count <= 8, DM_E_SHAPE, "to_bf16: %d segments (max 8)", count);
  CvtArgs a;
  size_t groups = 0;
  int n = 0;
  for (int i = 0; i < count; ++i) {
    if (!segs[i].src || !segs[i].dst || segs[i].n == 0) continue;
    a.seg[n] = segs[i];
    a.start[n] = groups;
    groups += (segs[i].n + 3) / 4;
    ++n;
  }
  if (n == 0) return DM_OK;
  for (int i = n; i <= 8; ++i) a.start[i] = groups;
  a.start[n] = groups;
  hipLaunchKernelGGL(to_bf16_multi_kernel, dim3(ew_blocks(groups)), dim3(256), 0, st, a, n);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
