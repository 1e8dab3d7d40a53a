// Phase-vocoder attacks of the evaluation loop (reference uformerWM/audio_attack.py:195-231 time_scaling =
// librosa.effects.time_stretch(y, rate), :233-239 pitch_scaling = librosa.effects.pitch_shift(y, sr=16000, n_steps=-6)).
// librosa is not in the reference tree (third-party, version unpinned; the positional call at :199 implies < 0.10) and is
// not installed here: the kernels restate its published algorithm and are PARITY UNPINNED (oracle/vocoder.py is the numpy
// restatement they are tested against):
//   stft        n_fft 2048, hop 512, periodic Hann window, centre = True with reflect padding        (librosa.core.stft)
//   vocoder     magnitude interpolated between neighbouring frames, phase advanced by the measured    (librosa.core.phase_vocoder)
//               per-bin frequency deviation, output frames at steps of `rate`
//   istft       Hann-windowed inverse frames, overlap-add, division by the window sum of squares,      (librosa.core.istft)
//               centre trim, length fixed to round(L / rate)
//   resample    band-limited sinc interpolation from a Kaiser-windowed filter table (the layout of     (resampy, 'kaiser_best')
//               resampy's table: 64 zero crossings, 512 samples per crossing, linear interpolation)
// Outside the train step: direct DFTs against an LDS twiddle table (2048 is a power of two, but one utterance is ~300
// frames: the work is a few hundred microseconds next to tens of milliseconds of model passes).
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

constexpr int NF = 2048, HOPV = 512, NB = NF / 2 + 1;
constexpr float TWO_PI = 6.28318530717958647692f;

__device__ __forceinline__ float hann(int n) { return 0.5f - 0.5f * cospif(2.0f * (float)n / (float)NF); }

// spec[u][k][t] (re, im) = sum_n hann[n] ypad[512 t + n] exp(-2 pi i k n / 2048); ypad = reflect-padded by 1024
__global__ __launch_bounds__(256) void pv_stft_kernel(const float* __restrict__ wave, int64_t L, float2* __restrict__ spec, int T,
                                                      int Tp) {
  __shared__ float xw[NF];
  __shared__ float2 tw[NF];
  const int tid = threadIdx.x, t = blockIdx.x, u = blockIdx.y;
  const float* y = wave + (int64_t)u * L;
  for (int n = tid; n < NF; n += 256) {
    int64_t i = (int64_t)HOPV * t + n - NF / 2;
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
    xw[n] = (i >= 0 && i < L) ? y[i] * hann(n) : 0.f;
    float sn, cs;
    sincospif(2.0f * (float)n / (float)NF, &sn, &cs);
    tw[n] = make_float2(cs, sn);
  }
  __syncthreads();
  for (int k = tid; k < NB; k += 256) {
    float re = 0.f, im = 0.f;
    int idx = 0;
    for (int n = 0; n < NF; ++n) {
      const float2 w = tw[idx];
      const float x = xw[n];
      re = fmaf(x, w.x, re);
      im = fmaf(-x, w.y, im);
      idx = (idx + k) & (NF - 1);
    }
    spec[((int64_t)u * NB + k) * Tp + t] = make_float2(re, im);
  }
}

// one thread per (utterance, bin): the sequential phase accumulation of librosa.core.phase_vocoder
__global__ __launch_bounds__(256) void pv_stretch_kernel(const float2* __restrict__ spec, int Tp, int T, double rate,
                                                         float2* __restrict__ out, int Ts) {
  const int k = blockIdx.x * 256 + threadIdx.x, u = blockIdx.y;
  if (k >= NB) return;
  const float2* D = spec + ((int64_t)u * NB + k) * Tp;         // columns T, T + 1 are zero
  float2* o = out + ((int64_t)u * NB + k) * Ts;
  const float phi = (float)(3.14159265358979323846 * (double)HOPV * (double)k / (double)(NB - 1));   // linspace(0, pi hop, 1025)
  float acc = atan2f(D[0].y, D[0].x);
  for (int s = 0; s < Ts; ++s) {
    const double step = (double)s * rate;
    const int c0 = (int)step;
    const float alpha = (float)(step - floor(step));
    const float2 a = D[c0], b = D[c0 + 1];
    const float mag = (1.0f - alpha) * hypotf(a.x, a.y) + alpha * hypotf(b.x, b.y);
    float sn, cs;
    sincosf(acc, &sn, &cs);
    o[s] = make_float2(mag * cs, mag * sn);
    float dphase = atan2f(b.y, b.x) - atan2f(a.y, a.x) - phi;
    dphase = dphase - TWO_PI * rintf(dphase / TWO_PI);
    acc += phi + dphase;
  }
}

// frames[u][t][n] = hann[n] irfft(spec[u][:, t])[n]
__global__ __launch_bounds__(256) void pv_iframes_kernel(const float2* __restrict__ spec, int Ts, float* __restrict__ frames) {
  __shared__ float2 X[NB];
  __shared__ float2 tw[NF];
  const int tid = threadIdx.x, t = blockIdx.x, u = blockIdx.y;
  for (int k = tid; k < NB; k += 256) X[k] = spec[((int64_t)u * NB + k) * Ts + t];
  for (int n = tid; n < NF; n += 256) {
    float sn, cs;
    sincospif(2.0f * (float)n / (float)NF, &sn, &cs);
    tw[n] = make_float2(cs, sn);
  }
  __syncthreads();
  float* dst = frames + ((int64_t)u * Ts + t) * NF;
  for (int n = tid; n < NF; n += 256) {
    float acc = X[0].x + ((n & 1) ? -X[NB - 1].x : X[NB - 1].x);   // DC and Nyquist bins: real parts only
    float s2 = 0.f;
    int idx = n & (NF - 1);
    for (int k = 1; k < NB - 1; ++k) {
      const float2 w = tw[idx], x = X[k];
      s2 = fmaf(x.x, w.x, s2);
      s2 = fmaf(-x.y, w.y, s2);
      idx = (idx + n) & (NF - 1);
    }
    dst[n] = (acc + 2.0f * s2) * (1.0f / (float)NF) * hann(n);
  }
}

// overlap-add, division by the window sum of squares, centre trim, fixed length
__global__ __launch_bounds__(256) void pv_ola_kernel(const float* __restrict__ frames, int Ts, float* __restrict__ out, int64_t Lout) {
  const int u = blockIdx.y;
  const int64_t expected = NF + (int64_t)HOPV * (Ts - 1);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < Lout; i += (int64_t)gridDim.x * 256) {
    const int64_t q = i + NF / 2;
    float v = 0.f;
    if (q < expected) {
      float sum = 0.f, norm = 0.f;
      const int64_t thi = q / HOPV;
#pragma unroll
      for (int j = 0; j < NF / HOPV; ++j) {
        const int64_t t = thi - j;
        const int64_t n = q - HOPV * t;
        if (t >= 0 && t < Ts && n < NF) {
          sum += frames[((int64_t)u * Ts + t) * NF + n];
          const float w = hann((int)n);
          norm += w * w;
        }
      }
      v = norm > 1.17549435e-38f ? sum / norm : sum;
    }
    out[(int64_t)u * Lout + i] = v;
  }
}

// resampy's interpolation loop: out[t] = sum over the taps of the scaled filter around time t / ratio
__global__ __launch_bounds__(256) void sinc_resample_kernel(const float* __restrict__ in, int64_t Lin, float* __restrict__ out,
                                                            int64_t Lout, double ratio, const float* __restrict__ win,
                                                            const float* __restrict__ delta, int nwin, int precision) {
  const int u = blockIdx.y;
  const float* x = in + (int64_t)u * Lin;
  const double scale = ratio < 1.0 ? ratio : 1.0, inc = 1.0 / ratio;
  const int index_step = (int)(scale * precision);
  const float gain = ratio < 1.0 ? (float)ratio : 1.0f;         // resampy scales the window when decimating
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < Lout; t += (int64_t)gridDim.x * 256) {
    const double treg = (double)t * inc;
    const int64_t n = (int64_t)treg;
    float acc = 0.f;
    if (n < Lin) {
      double frac = scale * (treg - (double)n);
      double index_frac = frac * precision;
      int offset = (int)index_frac;
      float eta = (float)(index_frac - offset);
      int64_t imax = (nwin - offset) / index_step;
      if (imax > n + 1) imax = n + 1;
      for (int64_t i = 0; i < imax; ++i) {
        const int w = offset + (int)i * index_step;
        acc += (win[w] + eta * delta[w]) * gain * x[n - i];
      }
      frac = scale - frac;
      index_frac = frac * precision;
      offset = (int)index_frac;
      eta = (float)(index_frac - offset);
      int64_t kmax = (nwin - offset) / index_step;
      if (kmax > Lin - n - 1) kmax = Lin - n - 1;
      for (int64_t k = 0; k < kmax; ++k) {
        const int w = offset + (int)k * index_step;
        acc += (win[w] + eta * delta[w]) * gain * x[n + k + 1];
      }
    }
    out[(int64_t)u * Lout + t] = acc;
  }
}

}  // namespace

extern "C" int32_t wm_pv_frames(int64_t L) { return (int32_t)(1 + L / HOPV); }
extern "C" int32_t wm_pv_steps(int64_t L, double rate) {
  const int T = wm_pv_frames(L);
  int s = 0;
  while ((double)s * rate < (double)T) ++s;                    // len(np.arange(0, T, rate))
  return s;
}

// spec: [U, 1025, T + 2] complex (the two columns past T zeroed here); stretched: [U, 1025, Ts] complex;
// frames: [U, Ts, 2048] floats; out: [U, Lout]
extern "C" int wm_time_stretch(const float* wave, int64_t L, int32_t U, double rate, void* spec, void* stretched, float* frames,
                               float* out, int64_t Lout, void* stream) {
  if (!wave || !spec || !stretched || !frames || !out || U <= 0 || L <= NF / 2 || L > ((int64_t)1 << 30) || !(rate > 0.05 && rate < 20.0) ||
      Lout <= 0)
    return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int T = wm_pv_frames(L), Tp = T + 2, Ts = wm_pv_steps(L, rate);
  if (hipMemsetAsync(spec, 0, sizeof(float2) * (size_t)U * NB * Tp, st) != hipSuccess) return 1000;
  hipLaunchKernelGGL(pv_stft_kernel, dim3(T, U), dim3(256), 0, st, wave, L, static_cast<float2*>(spec), T, Tp);
  hipLaunchKernelGGL(pv_stretch_kernel, dim3((NB + 255) / 256, U), dim3(256), 0, st, static_cast<const float2*>(spec), Tp, T, rate,
                     static_cast<float2*>(stretched), Ts);
  hipLaunchKernelGGL(pv_iframes_kernel, dim3(Ts, U), dim3(256), 0, st, static_cast<const float2*>(stretched), Ts, frames);
  hipLaunchKernelGGL(pv_ola_kernel, dim3((unsigned)((Lout + 1023) / 1024 > 1024 ? 1024 : (Lout + 1023) / 1024), U), dim3(256), 0, st,
                     frames, Ts, out, Lout);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_sinc_resample(const float* in, int64_t Lin, int32_t U, double ratio, const float* win, const float* delta,
                                int32_t nwin, int32_t precision, float* out, int64_t Lout, void* stream) {
  if (!in || !win || !delta || !out || U <= 0 || Lin <= 0 || Lout <= 0 || nwin <= 0 || precision <= 0 || !(ratio > 0.01 && ratio < 100.0))
    return WM_EINVAL;
  hipLaunchKernelGGL(sinc_resample_kernel, dim3((unsigned)((Lout + 1023) / 1024 > 1024 ? 1024 : (Lout + 1023) / 1024), U), dim3(256),
                     0, static_cast<hipStream_t>(stream), in, Lin, out, Lout, ratio, win, delta, nwin, precision);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
