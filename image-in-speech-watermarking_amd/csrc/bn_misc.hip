// Layers of the two small CNN models on the path besides UformerAudio: ModelA (model.py:3000-3066, trained by
// train_modelA.py:423-445) and the HiDDeN decoder (hidden/model/decoder.py:6-40, conv_bn_relu.py:3-18):
// BatchNorm2d (+ fused activation), Dropout and the plain MSE terms of their losses.  NCHW fp32.  These models have
// 18 k / 0.3 M parameters ("API-compat check, not a performance target"): kernels are simple and
// bandwidth-shaped (one workgroup per channel for the statistics, grid-stride elementwise passes).
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

__device__ __forceinline__ float bn_act(float v, int act, float slope) {
  return act == 1 ? fmaxf(v, 0.f) : (act == 2 ? (v > 0.f ? v : slope * v) : (act == 3 ? 1.0f / (1.0f + __expf(-v)) : v));
}
__device__ __forceinline__ float bn_act_grad(float y, int act, float slope) {     // through the activation's OUTPUT
  return act == 1 ? (y > 0.f ? 1.f : 0.f) : (act == 2 ? (y > 0.f ? 1.f : slope) : (act == 3 ? y * (1.f - y) : 1.f));
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// one workgroup per channel: batch mean and biased variance (two passes: the second on centred values), running stats
__global__ __launch_bounds__(1024) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ mean,
                                                        float* __restrict__ invstd, float* __restrict__ rmean,
                                                        float* __restrict__ rvar, int B, int C, int HW, float eps,
                                                        float momentum) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const int64_t n = (int64_t)B * HW;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += x[((i / HW) * C + c) * HW + i % HW];
  const float mu = block_sum(s, red) / (float)n;
  float q = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const float d = x[((i / HW) * C + c) * HW + i % HW] - mu;
    q += d * d;
  }
  const float var = block_sum(q, red) / (float)n;
  if (threadIdx.x == 0) {
    mean[c] = mu;
    invstd[c] = rsqrtf(var + eps);
    if (rmean) {
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * (float)n / (float)(n > 1 ? n - 1 : 1);
    }
  }
}
// eval mode: statistics from the running buffers
__global__ void bn_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar,
                                     float* __restrict__ mean, float* __restrict__ invstd, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { mean[c] = rmean[c]; invstd[c] = rsqrtf(rvar[c] + eps); }
}
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ y,
                                int C, int HW, int64_t total, int act, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    y[i] = bn_act((x[i] - mean[c]) * invstd[c] * gamma[c] + beta[c], act, slope);
  }
}
// one workgroup per channel: sum g and sum g * xhat with g = dy * act'(y); accumulated into dgamma / dbeta
__global__ __launch_bounds__(1024) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, float* __restrict__ sums,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int B,
                                                             int C, int HW, int act, float slope) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  const int64_t n = (int64_t)B * HW;
  const float mu = mean[c], is = invstd[c];
  float s1 = 0.f, s2 = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const int64_t o = ((i / HW) * C + c) * HW + i % HW;
    const float g = dy[o] * bn_act_grad(y[o], act, slope);
    s1 += g;
    s2 += g * (x[o] - mu) * is;
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    sums[c] = s1;
    sums[C + c] = s2;
    if (dbeta) dbeta[c] += s1;
    if (dgamma) dgamma[c] += s2;
  }
}
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ sums, float* __restrict__ dx,
                                    int B, int C, int HW, int64_t total, int act, float slope, int training) {
  const float inv_n = 1.0f / ((float)B * HW);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    const float g = dy[i] * bn_act_grad(y[i], act, slope);
    const float xh = (x[i] - mean[c]) * invstd[c];
    dx[i] = training ? gamma[c] * invstd[c] * (g - sums[c] * inv_n - xh * sums[C + c] * inv_n) : gamma[c] * invstd[c] * g;
  }
}

// y = x * keep_mask / (1 - p); the mask is a pure function of (seed, offset, index): the backward calls it on dy
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float p, uint64_t seed,
                               uint64_t offset) {
  const float keep = 1.0f - p;
  for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 * 4 < n; i4 += (int64_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4x32((uint32_t)i4, (uint32_t)(i4 >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed,
               (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t i = i4 * 4 + q;
      if (i < n) y[i] = (p <= 0.f || u01(r[q]) < keep) ? x[i] / keep : 0.f;
    }
  }
}

// loss += mean((a - b)^2); da = gscale * 2 (a - b) / n, db = -da (each optional)
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                  float* __restrict__ loss, float* __restrict__ da, float* __restrict__ db,
                                                  float gscale) {
  __shared__ float red[4];
  const float k = 2.0f * gscale / (float)n;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    s += d * d;
    if (da) da[i] = k * d;
    if (db) db[i] = -k * d;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, s / (float)n);
}

// dst[b][c0 + l][:] = scale * v[b][l]  (message planes of the HiDDeN encoder, backward of the global average pool)
__global__ __launch_bounds__(256) void plane_fill_kernel(const float* __restrict__ v, float* __restrict__ dst, int L, int Ctot,
                                                         int c0, int HW, float scale) {
  const int bl = blockIdx.x, b = bl / L, l = bl - b * L;
  const float x = v[bl] * scale;
  float* d = dst + ((int64_t)b * Ctot + c0 + l) * HW;
  for (int i = threadIdx.x; i < HW; i += 256) d[i] = x;
}
// v[b][l] = scale * sum over the plane src[b][c0 + l][:]  (global average pool, backward of the message planes)
__global__ __launch_bounds__(256) void plane_sum_kernel(const float* __restrict__ src, float* __restrict__ v, int L, int Ctot,
                                                        int c0, int HW, float scale) {
  __shared__ float red[4];
  const int bl = blockIdx.x, b = bl / L, l = bl - b * L;
  const float* s = src + ((int64_t)b * Ctot + c0 + l) * HW;
  float a = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) a += s[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) v[bl] = a * scale;
}
// BCEWithLogitsLoss(x, target) with one target for all n logits: loss += mean(max(x, 0) - x t + log(1 + exp(-|x|)));
// dx = gscale * (sigmoid(x) - t) / n
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ x, int n, float t, float* __restrict__ loss,
                                                         float* __restrict__ dx, float gscale) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = x[i], e = __expf(-fabsf(v));
    s += fmaxf(v, 0.f) - v * t + log1pf(e);
    if (dx) {
      const float sig = v >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
      dx[i] = gscale * (sig - t) / (float)n;
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, s / (float)n);
}

// Linear head of the discriminator (a few outputs on [M, K] pooled features): y[m][n] = x[m] . w[n] + b[n]
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ b, float* __restrict__ y, int M, int N,
                                                               int K) {
  __shared__ float red[4];
  const int m = blockIdx.x / N, n = blockIdx.x - m * N;
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) s += x[(int64_t)m * K + k] * w[(int64_t)n * K + k];
  s = block_sum(s, red);
  if (threadIdx.x == 0) y[blockIdx.x] = s + (b ? b[n] : 0.f);
}
// dx[m][k] = sum_n dy[m][n] w[n][k];  dw[n][k] += sum_m dy[m][n] x[m][k];  db[n] += sum_m dy[m][n]  (thread = one k)
__global__ __launch_bounds__(256) void linear_small_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ w, float* __restrict__ dx,
                                                               float* __restrict__ dw, float* __restrict__ db, int M, int N,
                                                               int K) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < K) {
    if (dx)
      for (int m = 0; m < M; ++m) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dy[m * N + n] * w[(int64_t)n * K + k];
        dx[(int64_t)m * K + k] = s;
      }
    if (dw)
      for (int n = 0; n < N; ++n) {
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += dy[m * N + n] * x[(int64_t)m * K + k];
        dw[(int64_t)n * K + k] += s;
      }
  }
  if (db && k < N) {
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += dy[m * N + k];
    db[k] += s;
  }
}

inline unsigned ew_grid(int64_t total) {
  const int64_t b = (total + 255) / 256;
  return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int wm_batchnorm2d_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float* y, float* save_mean, float* save_invstd, int32_t B, int32_t C,
                                  int32_t HW, float eps, float momentum, int32_t training, int32_t act, float slope,
                                  void* stream) {
  if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || B <= 0 || C <= 0 || HW <= 0) return WM_EINVAL;
  if (!training && (!running_mean || !running_var)) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (training)
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C), dim3(1024), 0, st, x, save_mean, save_invstd, running_mean, running_var, B, C, HW,
                       eps, momentum);
  else
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 63) / 64), dim3(64), 0, st, running_mean, running_var, save_mean,
                       save_invstd, C, eps);
  WM_CHECK_LAUNCH();
  const int64_t total = (int64_t)B * C * HW;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(total)), dim3(256), 0, st, x, gamma, beta, save_mean, save_invstd, y, C, HW,
                     total, act, slope);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_batchnorm2d_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* save_mean,
                                  const float* save_invstd, float* dx, float* dgamma, float* dbeta, float* sums /* [2C] */,
                                  int32_t B, int32_t C, int32_t HW, int32_t training, int32_t act, float slope, void* stream) {
  if (!dy || !x || !y || !gamma || !save_mean || !save_invstd || !dx || !sums || B <= 0 || C <= 0 || HW <= 0) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C), dim3(1024), 0, st, dy, x, y, save_mean, save_invstd, sums, dgamma, dbeta, B, C,
                     HW, act, slope);
  WM_CHECK_LAUNCH();
  const int64_t total = (int64_t)B * C * HW;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(total)), dim3(256), 0, st, dy, x, y, gamma, save_mean, save_invstd, sums, dx,
                     B, C, HW, total, act, slope, training);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
  if (!x || !y || n <= 0 || p < 0.f || p >= 1.f) return WM_EINVAL;
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n, p, seed,
                     offset);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_mse_fwd_bwd(const float* a, const float* b, int64_t n, float* loss, float* da, float* db, float gscale,
                              void* stream) {
  if (!a || !b || n <= 0) return WM_EINVAL;
  hipLaunchKernelGGL(mse_kernel, dim3(ew_grid(n) > 512 ? 512 : ew_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, n,
                     loss, da, db, gscale);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_plane_fill(const float* v, float* dst, int32_t B, int32_t L, int32_t Ctot, int32_t c0, int32_t HW, float scale,
                             void* stream) {
  if (!v || !dst || B <= 0 || L <= 0 || c0 < 0 || c0 + L > Ctot || HW <= 0) return WM_EINVAL;
  hipLaunchKernelGGL(plane_fill_kernel, dim3(B * L), dim3(256), 0, static_cast<hipStream_t>(stream), v, dst, L, Ctot, c0, HW, scale);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_plane_sum(const float* src, float* v, int32_t B, int32_t L, int32_t Ctot, int32_t c0, int32_t HW, float scale,
                            void* stream) {
  if (!src || !v || B <= 0 || L <= 0 || c0 < 0 || c0 + L > Ctot || HW <= 0) return WM_EINVAL;
  hipLaunchKernelGGL(plane_sum_kernel, dim3(B * L), dim3(256), 0, static_cast<hipStream_t>(stream), src, v, L, Ctot, c0, HW, scale);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_bce_logits_fwd_bwd(const float* logits, int32_t n, float target, float* loss, float* dlogits, float gscale,
                                     void* stream) {
  if (!logits || n <= 0 || (!loss && !dlogits)) return WM_EINVAL;
  hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), logits, n, target, loss, dlogits,
                     gscale);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_linear_small_fwd(const float* x, const float* w, const float* b, float* y, int32_t M, int32_t N, int32_t K,
                                   void* stream) {
  if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || (int64_t)M * N > 1 << 20) return WM_EINVAL;
  hipLaunchKernelGGL(linear_small_fwd_kernel, dim3(M * N), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, b, y, M, N, K);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_linear_small_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int32_t M,
                                   int32_t N, int32_t K, void* stream) {
  if (!dy || !x || !w || M <= 0 || N <= 0 || K <= 0 || N > K) return WM_EINVAL;
  hipLaunchKernelGGL(linear_small_bwd_kernel, dim3((K + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), dy, x, w, dx,
                     dw, db, M, N, K);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
