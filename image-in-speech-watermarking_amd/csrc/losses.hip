// The loss callables of the reference's uformerWM/losses.py as device kernels with their input gradients:
//   CharbonnierLoss.forward (:41-52)   mean(sqrt((x - y)^2 + eps^2))
//   TVLoss.forward          (:20-37)   weight * 2 * (sum dh^2 / count_h + sum dw^2 / count_w) / batch
//   tv_loss                 (:8-18)    reg * sum((dw^2 + dh^2)^beta over the (H-1) x (W-1) interior corner) / numel
// x is NCHW fp32 (planes = N * C).  Every kernel ADDS its loss into *loss (the caller zeroes it) and writes
// d loss / d x scaled by `gscale` (the upstream gradient) when dx is given; the gradient is a gather over the (at most
// three) difference terms an element takes part in, so no atomics and a fixed summation order per element.
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

__global__ __launch_bounds__(256) void charbonnier_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t n,
                                                          float eps2, float* loss, float* __restrict__ dx,
                                                          float* __restrict__ dy, float gscale) {
  __shared__ float red[4];
  const float inv = 1.0f / (float)n;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = x[i] - y[i], r = sqrtf(d * d + eps2);
    s += r;
    const float g = gscale * inv * d / r;
    if (dx) dx[i] = g;
    if (dy) dy[i] = -g;
  }
  s = block_sum256(s, red);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, s * inv);
}

// TVLoss: sum over vertical / horizontal neighbour differences squared
__global__ __launch_bounds__(256) void tvloss_kernel(const float* __restrict__ x, int planes, int H, int W, float kh, float kw,
                                                     float* loss, float* __restrict__ dx, float gscale) {
  __shared__ float red[4];
  const int64_t n = (int64_t)planes * H * W;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int w = (int)(i % W), h = (int)((i / W) % H);
    const float v = x[i];
    float g = 0.f;
    if (h + 1 < H) { const float d = x[i + W] - v; s += kh * d * d; g -= 2.f * kh * d; }
    if (w + 1 < W) { const float d = x[i + 1] - v; s += kw * d * d; g -= 2.f * kw * d; }
    if (h > 0) g += 2.f * kh * (v - x[i - W]);
    if (w > 0) g += 2.f * kw * (v - x[i - 1]);
    if (dx) dx[i] = gscale * g;
  }
  s = block_sum256(s, red);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, s);
}

// tv_loss: t(h, w) = (x[h, w+1] - x[h, w])^2 + (x[h+1, w] - x[h, w])^2 for h < H-1, w < W-1; loss = k * sum t^beta
__device__ __forceinline__ float tv_t(const float* x, int64_t i, int W) {
  const float a = x[i + 1] - x[i], b = x[i + W] - x[i];
  return a * a + b * b;
}
__global__ __launch_bounds__(256) void tv_pow_kernel(const float* __restrict__ x, int planes, int H, int W, float beta, float k,
                                                     float* loss, float* __restrict__ dx, float gscale) {
  __shared__ float red[4];
  const int64_t n = (int64_t)planes * H * W;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int w = (int)(i % W), h = (int)((i / W) % H);
    float g = 0.f;
    if (h + 1 < H && w + 1 < W) {                               // own corner term: d t / d x[h, w] = -2 (a + b)
      const float a = x[i + 1] - x[i], b = x[i + W] - x[i], t = a * a + b * b;
      s += powf(t, beta);
      g += beta * powf(t, beta - 1.f) * (-2.f * (a + b));
    }
    if (w > 0 && h + 1 < H) {                                   // term of (h, w-1): + 2 a
      const float t = tv_t(x, i - 1, W);
      g += beta * powf(t, beta - 1.f) * 2.f * (x[i] - x[i - 1]);
    }
    if (h > 0 && w + 1 < W) {                                   // term of (h-1, w): + 2 b
      const float t = tv_t(x, i - W, W);
      g += beta * powf(t, beta - 1.f) * 2.f * (x[i] - x[i - W]);
    }
    if (dx) dx[i] = gscale * k * g;
  }
  s = block_sum256(s, red);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, k * s);
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256); }

}  // namespace

extern "C" int wm_charbonnier_fwd_bwd(const float* x, const float* y, int64_t n, float eps, float* loss, float* dx, float* dy,
                                      float gscale, void* stream) {
  if (!x || !y || n <= 0) return WM_EINVAL;
  hipLaunchKernelGGL(charbonnier_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n, eps * eps,
                     loss, dx, dy, gscale);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_tvloss_fwd_bwd(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, float weight, float* loss,
                                 float* dx, float gscale, void* stream) {
  if (!x || N <= 0 || C <= 0 || H < 2 || W < 2) return WM_EINVAL;
  const float kh = weight * 2.f / ((float)C * (H - 1) * W) / (float)N, kw = weight * 2.f / ((float)C * H * (W - 1)) / (float)N;
  hipLaunchKernelGGL(tvloss_kernel, dim3(grid_for((int64_t)N * C * H * W)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     N * C, H, W, kh, kw, loss, dx, gscale);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_tv_pow_fwd_bwd(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, float beta, float reg_coeff,
                                 float* loss, float* dx, float gscale, void* stream) {
  if (!x || N <= 0 || C <= 0 || H < 2 || W < 2) return WM_EINVAL;
  const float k = reg_coeff / ((float)N * C * H * W);
  hipLaunchKernelGGL(tv_pow_kernel, dim3(grid_for((int64_t)N * C * H * W)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     N * C, H, W, beta, k, loss, dx, gscale);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
