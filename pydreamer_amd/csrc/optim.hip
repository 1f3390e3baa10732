// Flat-buffer optimizer kernels: global grad-norm + clip coefficient, in-place scaling, AdamW, parameter copy.
// Reference: Dreamer.grad_clip / init_optimizers (dreamer.py:60-87) = torch clip_grad_norm_ + torch.optim.AdamW
// (lr, eps given; betas (0.9,0.999) and weight_decay 0.01 are the torch defaults), ActorCritic.update_critic_target
// (a2c.py:151-152).  All HBM-streaming: every kernel reads/writes each element once with 16-byte accesses.
#include "common.h"

__global__ void __launch_bounds__(256) sumsq_partial_kernel(const float* __restrict__ g, long long n,
                                                            float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  const long long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = g4[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    s += v * v;
  }
  s = dm_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) norm_final_kernel(const float* __restrict__ partial, int nblocks, float max_norm,
                                                         float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
  s = dm_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    const float coef = max_norm / (norm + 1e-6f);
    out[0] = norm;
    out[1] = coef < 1.0f ? coef : 1.0f;
  }
}

extern "C" int dm_multi_tensor_norm_clip(const float* grad, int64_t n, float max_norm, float* norm_out, void* ws,
                                         size_t ws_bytes, void* stream) {
  DM_REQUIRE(grad && norm_out, DM_E_NULL, "norm_clip: null pointer");
  DM_REQUIRE(((uintptr_t)grad & 15) == 0, DM_E_SHAPE, "norm_clip: grad buffer must be 16-byte aligned");
  int blocks = dm_cdiv(n > 0 ? n : 1, 256 * 16);
  if (blocks > 1024) blocks = 1024;
  DM_REQUIRE(ws && (size_t)blocks * sizeof(float) <= ws_bytes, DM_E_WORKSPACE, "norm_clip: workspace too small");
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grad, (long long)n, (float*)ws);
  DM_LAUNCH_CHECK();
  hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, blocks, max_norm,
                     norm_out);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// x *= *coef   (clip_grad_norm_ scales the gradients in place)
__global__ void __launch_bounds__(256) scale_inplace_kernel(float* __restrict__ x, long long n,
                                                            const float* __restrict__ coef) {
  const float c = coef[0];
  if (c == 1.0f) return;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] *= c;
}
extern "C" int dm_scale_inplace(float* x, int64_t n, const float* coef, void* stream) {
  DM_REQUIRE(x && coef, DM_E_NULL, "scale_inplace: null pointer");
  if (n <= 0) return DM_OK;
  int blocks = dm_cdiv(n, 256 * 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(scale_inplace_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)n, coef);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// torch.optim.AdamW single-tensor math (decoupled weight decay, bias-corrected):
//   p *= 1 - lr*wd ; m += (g-m)(1-b1) ; v = v*b2 + (1-b2) g^2 ; p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n,
                                                    float decay, float one_minus_b1, float b2, float one_minus_b2,
                                                    float step_size, float bc2_sqrt, float eps,
                                                    const float* __restrict__ clip_coef) {
  const float c = clip_coef ? clip_coef[0] : 1.0f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * c;
    float pi = p[i] * decay;
    float mi = m[i];
    mi = mi + (gi - mi) * one_minus_b1;
    const float vi = v[i] * b2 + one_minus_b2 * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

extern "C" int dm_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, const float* clip_coef,
                             void* stream) {
  DM_REQUIRE(param && grad && exp_avg && exp_avg_sq, DM_E_NULL, "adamw: null pointer");
  DM_REQUIRE(step >= 1, DM_E_SHAPE, "adamw: step %d < 1", step);
  if (n <= 0) return DM_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const float decay = (float)(1.0 - (double)lr * (double)weight_decay);
  int blocks = dm_cdiv(n, 256 * 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     (long long)n, decay, (float)(1.0 - (double)beta1), beta2, (float)(1.0 - (double)beta2), step_size,
                     bc2_sqrt, eps, clip_coef);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_copy_params(float* dst, const float* src, int64_t n, void* stream) {
  DM_REQUIRE(dst && src, DM_E_NULL, "copy_params: null pointer");
  if (n <= 0) return DM_OK;
  hipError_t e = hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) return dm_fail(DM_E_HIP, "copy_params: %s", hipGetErrorString(e));
  return DM_OK;
}

// y = a*x + b*y
__global__ void __launch_bounds__(256) axpby_kernel(long long n, float a, const float* __restrict__ x, float b,
                                                    float* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    y[i] = a * x[i] + (b == 0.f ? 0.f : b * y[i]);
}
extern "C" int dm_axpby(int64_t n, float a, const float* x, float b, float* y, void* stream) {
  DM_REQUIRE(x && y, DM_E_NULL, "axpby: null pointer");
  if (n <= 0) return DM_OK;
  int blocks = dm_cdiv(n, 256 * 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(axpby_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, a, x, b, y);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
