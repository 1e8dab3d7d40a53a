// Train-step scalar work: the 4-term loss of the reference trainer and its output gradients
// (uformerWM/audio_uformer_stft.py:463-482), fused AdamW over the flat parameter arena
// (:252-255, torch.optim.AdamW semantics), stochastic-depth keep scales (timm DropPath as used at
// model.py:916,1016-1017) and the device-side step counter that makes all of it hipGraph-replayable.
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// acc[0] += sum (a-b)^2 over n ; used for the three MSE terms ; acc[1] (optional) += sum c^2
__global__ __launch_bounds__(256) void sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ c, int64_t n, float* acc0,
                                                     float* acc1) {
  __shared__ float red[4];
  float s0 = 0.f, s1 = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = a[i] - b[i];
    s0 += d * d;
    if (c) s1 += c[i] * c[i];
  }
  s0 = block_sum256(s0, red);
  if (threadIdx.x == 0) atomicAdd(acc0, s0);
  if (c) {
    s1 = block_sum256(s1, red);
    if (threadIdx.x == 0) atomicAdd(acc1, s1);
  }
}

__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ stft_new, const float* __restrict__ x,
                                                        const float* __restrict__ noise, const float* __restrict__ wm_pred,
                                                        const float* __restrict__ wm, const float* __restrict__ msg,
                                                        const float* __restrict__ acc, float* __restrict__ losses,
                                                        float* __restrict__ d_stft_new, float* __restrict__ d_noise,
                                                        float* __restrict__ d_wm_pred, float* __restrict__ d_wm,
                                                        int B, int64_t n1, int64_t n2) {
  const float nrm = sqrtf(acc[3]);
  const float nn = nrm / (float)B;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const float l1 = acc[0] / (float)n1, l2 = acc[1] / (float)n2, l3 = acc[2] / (float)n2;
    const float l4 = (nn - 1.0f) * (nn - 1.0f);
    losses[0] = l1; losses[1] = l2; losses[2] = l3; losses[3] = l4;
    losses[4] = ((l1 + l2) + l3) + l4;
    losses[5] = nrm;
  }
  const float k1 = 2.0f / (float)n1, k2 = 2.0f / (float)n2;
  const float k4 = nrm > 0.f ? 2.0f * (nn - 1.0f) / ((float)B * nrm) : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n1; i += (int64_t)gridDim.x * 256) {
    if (d_stft_new) d_stft_new[i] = k1 * (stft_new[i] - x[i]);
    if (d_noise) d_noise[i] = k4 * noise[i];
    if (i < n2) {
      const float m = msg[i];
      if (d_wm_pred) d_wm_pred[i] = k2 * (wm_pred[i] - m);
      if (d_wm) d_wm[i] = k2 * (wm[i] - m);
    }
  }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, const float* __restrict__ lr_ptr, float b1, float b2,
                             float eps, float wd, const int64_t* __restrict__ step, const float* __restrict__ gscale) {
  const float lr = *lr_ptr;
  const float t = (float)(*step);
  const float bc1 = 1.0f - powf(b1, t), bc2s = sqrtf(1.0f - powf(b2, t));
  const float gs = gscale ? *gscale : 1.0f;
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gs;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2s + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

__global__ void step_inc_kernel(int64_t* step) { *step += 1; }
__global__ void step_copy_kernel(const int64_t* src, int64_t* dst) { *dst = *src; }
__global__ void zero_small_kernel(float* p, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = 0.f;
}
// clears the gradient arena / accumulation targets (a kernel node, for the same reason as zero_small_kernel)
__global__ __launch_bounds__(256) void zero_kernel(uint4* __restrict__ p, int64_t n16, float* __restrict__ tail, int ntail) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) p[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}

__global__ void droppath_kernel(float* __restrict__ scales, const float* __restrict__ rates, int nslots, int B,
                                uint64_t seed, const int64_t* __restrict__ step) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nslots * B) return;
  const float rate = rates[i / B];
  float s = 1.0f;
  if (rate > 0.f) {
    const uint64_t st = step ? (uint64_t)*step : 0ull;
    uint32_t r[4];
    philox4x32((uint32_t)i, 0x44505448u, (uint32_t)st, (uint32_t)(st >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float keep = 1.0f - rate;
    s = u01(r[0]) < keep ? 1.0f / keep : 0.f;
  }
  scales[i] = s;
}

inline unsigned grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (unsigned)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int wm_loss_fwd_bwd(const float* stft_new, const float* x, const float* noise, const float* wm_pred,
                               const float* wm, const float* msg, float* losses, float* d_stft_new, float* d_noise,
                               float* d_wm_pred, float* d_wm, int32_t B, float* scratch, void* stream) {
  if (!stft_new || !x || !noise || !wm_pred || !wm || !msg || !losses || !scratch || B <= 0) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t n1 = (int64_t)B * 2 * 128 * 128, n2 = (int64_t)B * 32 * 32;
  // accumulators are cleared by a kernel, not hipMemsetAsync: a 16-byte memset NODE inside a captured hipGraph
  // stopped clearing them from the second replay on (ROCm 7.2, observed on MI355X), a kernel node replays reliably
  hipLaunchKernelGGL(zero_small_kernel, dim3(1), dim3(64), 0, st, scratch, 8);
  hipLaunchKernelGGL(sqdiff_kernel, dim3(grid_for(n1)), dim3(256), 0, st, stft_new, x, noise, n1, scratch + 0, scratch + 3);
  hipLaunchKernelGGL(sqdiff_kernel, dim3(grid_for(n2)), dim3(256), 0, st, wm_pred, msg, (const float*)nullptr, n2,
                     scratch + 1, (float*)nullptr);
  hipLaunchKernelGGL(sqdiff_kernel, dim3(grid_for(n2)), dim3(256), 0, st, wm, msg, (const float*)nullptr, n2,
                     scratch + 2, (float*)nullptr);
  hipLaunchKernelGGL(loss_grad_kernel, dim3(grid_for(n1)), dim3(256), 0, st, stft_new, x, noise, wm_pred, wm, msg,
                     scratch, losses, d_stft_new, d_noise, d_wm_pred, d_wm, B, n1, n2);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_zero(float* p, int64_t n, void* stream) {
  if (!p || n <= 0 || (reinterpret_cast<uintptr_t>(p) & 15)) return WM_EINVAL;
  const int64_t n16 = n / 4;
  int64_t blocks = (n16 + 255) / 256;
  blocks = blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks);
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<uint4*>(p), n16, p + n16 * 4, (int)(n - n16 * 4));
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_adamw(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, float beta1,
                        float beta2, float eps, float wd, const int64_t* step, const float* grad_scale, void* stream) {
  if (!p || !g || !m || !v || !lr || !step || n <= 0) return WM_EINVAL;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n, lr,
                     beta1, beta2, eps, wd, step, grad_scale);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_step_inc(int64_t* step, void* stream) {
  if (!step) return WM_EINVAL;
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), step);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_step_copy(const int64_t* src, int64_t* dst, void* stream) {
  if (!src || !dst) return WM_EINVAL;
  hipLaunchKernelGGL(step_copy_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), src, dst);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_droppath_scales(float* scales, const float* rates, int32_t nslots, int32_t B, uint64_t seed,
                                  const int64_t* step, void* stream) {
  if (!scales || !rates || nslots <= 0 || B <= 0) return WM_EINVAL;
  hipLaunchKernelGGL(droppath_kernel, dim3((nslots * B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     scales, rates, nslots, B, seed, step);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
