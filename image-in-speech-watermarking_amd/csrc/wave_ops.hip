// Whole-utterance low-pass and resampling attacks of the evaluation loop (reference
// uformerWM/audio_test.py:637 'low_pass' -> audio_attack.py:21-30 and :644 'resampling' -> :71-83) on [U, L] waveforms of
// any length.  Same operators as the in-loop kernels of attack.hip (scipy conventions: butter(8, 0.5) + filtfilt with odd
// padding 27 and steady-state initial conditions, expanded to 128 FIR taps; resample_poly by 2 with the 41-tap Kaiser
// half-band filter — the reference's librosa kernel is parity-unpinned), but tiled over global memory instead of keeping a
// row in LDS: a workgroup computes 2048 consecutive outputs from a staged window of the row.
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

constexpr int PADLEN = 27, J = 128, CH = 2048, RS_HALF = 20;

// sample q of the odd extension of x (length L): q in [0, L + 2 * PADLEN)
__device__ __forceinline__ float odd_ext(const float* __restrict__ x, int64_t L, int64_t q) {
  if (q < PADLEN) return 2.0f * x[0] - x[PADLEN - q];
  if (q < PADLEN + L) return x[q - PADLEN];
  return 2.0f * x[L - 1] - x[L - 2 - (q - PADLEN - L)];
}

// PASS 0 (causal):      y[n] = sum_{j <= min(n, J-1)} h[j] e[n - j] + e[0] tail[min(n, J-1)],  e = odd extension of the row,
//                       n < N = L + 54, written to tmp[u][n]
// PASS 1 (anti-causal): z[n] = sum_{j <= min(N-1-n, J-1)} h[j] y[n + j] + y[N-1] tail[min(N-1-n, J-1)]; out[u][i] = z[i + 27]
template <int PASS>
__global__ __launch_bounds__(256) void wave_fir_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t L,
                                                       const float* __restrict__ taps) {
  __shared__ float win[CH + J];
  __shared__ float h[J], tail[J];
  const int tid = threadIdx.x, u = blockIdx.y;
  const int64_t N = L + 2 * PADLEN, n0 = (int64_t)blockIdx.x * CH;
  if (tid < J) { h[tid] = taps[tid]; tail[tid] = taps[J + tid]; }
  const float* row = in + (int64_t)u * (PASS == 0 ? L : N);
  // window: samples [n0 - (J-1), n0 + CH) for the causal pass, [n0, n0 + CH + J - 1) for the anti-causal one
  const int64_t base = PASS == 0 ? n0 - (J - 1) : n0;
  for (int i = tid; i < CH + J - 1; i += 256) {
    const int64_t q = base + i;
    float v = 0.f;
    if (q >= 0 && q < N) v = PASS == 0 ? odd_ext(row, L, q) : row[q];
    win[i] = v;
  }
  __syncthreads();
  const float edge = PASS == 0 ? odd_ext(row, L, 0) : row[N - 1];
  for (int k = tid; k < CH; k += 256) {
    const int64_t n = n0 + k;
    if (n >= N) break;
    float acc = 0.f;
    if (PASS == 0) {
      const float* w = win + k + (J - 1);                      // w[-j] = e[n - j]; zeros left of the row start
#pragma unroll 8
      for (int j = 0; j < J; ++j) acc += h[j] * w[-j];
      const int64_t room = n;
      acc += edge * tail[room < J - 1 ? room : J - 1];
      out[(int64_t)u * N + n] = acc;
    } else {
      const float* w = win + k;                                // w[j] = y[n + j]; zeros beyond the row end
#pragma unroll 8
      for (int j = 0; j < J; ++j) acc += h[j] * w[j];
      const int64_t room = N - 1 - n;
      acc += edge * tail[room < J - 1 ? room : J - 1];
      const int64_t i = n - PADLEN;
      if (i >= 0 && i < L) out[(int64_t)u * L + i] = acc;
    }
  }
}

// half-band resampling by 2: centre tap + ten symmetric pairs at odd offsets (see attack.hip)
// UP = false: y[m] = h0 x[2m] + sum_k hk (x[2m - 2k - 1] + x[2m + 2k + 1]), m < Ld
// UP = true : z[n] = 2 h0 y[n / 2] (n even), 2 sum_k hk (y[c - k] + y[c + 1 + k]), c = n / 2 (n odd), n < L
template <bool UP>
__global__ __launch_bounds__(256) void wave_resample_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t Lin,
                                                            int64_t Lout, const float* __restrict__ taps) {
  __shared__ float hk[11];
  if (threadIdx.x < 10) hk[threadIdx.x] = taps[RS_HALF + 2 * threadIdx.x + 1];
  if (threadIdx.x == 10) hk[10] = taps[RS_HALF];
  __syncthreads();
  const int u = blockIdx.y;
  const float* x = in + (int64_t)u * Lin;
  const float h0 = hk[10];
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < Lout; o += (int64_t)gridDim.x * 256) {
    float acc;
    if (!UP) {
      const int64_t c = 2 * o;
      acc = h0 * x[c];
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const int64_t lo = c - 2 * k - 1, hi = c + 2 * k + 1;
        acc += hk[k] * ((lo >= 0 ? x[lo] : 0.f) + (hi < Lin ? x[hi] : 0.f));
      }
    } else if (!(o & 1)) {
      acc = 2.0f * h0 * x[o >> 1];
    } else {
      const int64_t c = o >> 1;
      acc = 0.f;
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const int64_t lo = c - k, hi = c + 1 + k;
        acc += hk[k] * ((lo >= 0 ? x[lo] : 0.f) + (hi < Lin ? x[hi] : 0.f));
      }
      acc *= 2.0f;
    }
    out[(int64_t)u * Lout + o] = acc;
  }
}

}  // namespace

extern "C" int wm_wave_lowpass(const float* wave, float* out, float* tmp, int64_t L, int32_t U, const float* lp_h, int32_t lp_n,
                               void* stream) {
  if (!wave || !out || !tmp || !lp_h || lp_n != J || U <= 0 || L <= PADLEN + 1 || L > ((int64_t)1 << 31)) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t N = L + 2 * PADLEN;
  const dim3 grid((unsigned)((N + CH - 1) / CH), U);
  hipLaunchKernelGGL(wave_fir_kernel<0>, grid, dim3(256), 0, st, wave, tmp, L, lp_h);
  hipLaunchKernelGGL(wave_fir_kernel<1>, grid, dim3(256), 0, st, tmp, out, L, lp_h);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_wave_resample(const float* wave, float* out, float* tmp, int64_t L, int32_t U, const float* rs_h, int32_t rs_n,
                                void* stream) {
  if (!wave || !out || !tmp || !rs_h || rs_n != 2 * RS_HALF + 1 || U <= 0 || L < 2 || L > ((int64_t)1 << 31)) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t Ld = (L + 1) / 2;
  hipLaunchKernelGGL(wave_resample_kernel<false>, dim3((unsigned)((Ld + 1023) / 1024 > 512 ? 512 : (Ld + 1023) / 1024), U),
                     dim3(256), 0, st, wave, tmp, L, Ld, rs_h);
  hipLaunchKernelGGL(wave_resample_kernel<true>, dim3((unsigned)((L + 1023) / 1024 > 512 ? 512 : (L + 1023) / 1024), U), dim3(256),
                     0, st, tmp, out, Ld, L, rs_h);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
