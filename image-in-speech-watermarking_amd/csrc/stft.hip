// Data-side and evaluation-side transforms on whole utterances (the in-graph n_fft = 255 round trip of the train step
// lives in stft_mfma.hip): the data-prep STFT (256 / 128), the whole-utterance STFT (255 / 63) cut into tiles, the
// stitching iSTFT, and the state-free evaluation attacks.  torch conventions throughout: rectangular window,
// center=True with reflect padding, no normalisation; tiles are [2, 128 bins, 128 frames] (re | im planes).
// 255 = 3*5*17 is not a power of two: these (outside the train step) evaluate a direct real DFT against a 255-entry
// cos/sin table held in LDS, the twiddle index k*n mod 255 advanced incrementally.
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

constexpr int NFFT = 255, HOP = 63, NBIN = 128, NFRAME = 128, PADL = 127;
constexpr int CHUNK = 512;                             // output samples per workgroup (synthesis)
constexpr int MAXF = 16;                               // frames staged per synthesis chunk
constexpr int TG = 16;                                 // frames per workgroup (analysis)
constexpr int SPAN = HOP * (TG - 1) + NFFT;            // samples staged per analysis workgroup

__device__ __forceinline__ int frames_covering(int q, int& tlo, int nframe = NFRAME) {   // returns count, sets first frame
  int thi = q / HOP;
  if (thi > nframe - 1) thi = nframe - 1;
  int lo = q - (NFFT - 1);
  tlo = lo <= 0 ? 0 : (lo + HOP - 1) / HOP;
  return thi - tlo + 1;
}

// sum over frames covering OLA position q of sum_k w_k (Xr cos - Xi sin).  Twiddles and spectrum values are held as
// (cos, sin) / (re, im) pairs so that one 8-byte LDS read feeds two multiply-adds (the loop is LDS-issue bound).
template <bool ADJ>
__device__ __forceinline__ float synth_at(int q, const float2* Xs, int t0, const float2* tw, int nframe = NFRAME) {
  int tlo;
  const int cnt = frames_covering(q, tlo, nframe);
  float total = 0.f;
  for (int f = 0; f < cnt; ++f) {
    const int t = tlo + f, n = q - HOP * t, tt = t - t0;
    const float acc = Xs[tt].x;                           // k = 0: cos = 1 (imag of DC never contributes)
    int idx = n;                                          // k = 1
    float a2 = 0.f, a3 = 0.f;
#pragma unroll 4
    for (int k = 1; k < NBIN; ++k) {
      const float2 x = Xs[k * MAXF + tt], w = tw[idx];
      a2 = fmaf(x.x, w.x, a2);
      a3 = fmaf(x.y, w.y, a3);
      idx += n;
      if (idx >= NFFT) idx -= NFFT;
    }
    a2 -= a3;
    total += ADJ ? (acc + a2) : (acc + 2.0f * a2);
  }
  return ADJ ? total : total / (float)(NFFT * cnt);
}

// ---------------------------------------------------------------------------------------------------------------
// Data-prep STFT (row A1): torch.stft(n_fft=256, hop=128, win=256 rectangular, center=True, reflect) of whole
// utterances, last bin dropped, frames zero-padded to the reference's tile count, written straight as
// [tile][re|im][bin][frame] tiles (audio_test.py:465-488,507-519).  One workgroup = 16 frames x 128 bins of one
// utterance: the 16 frames' 2176 + 128 samples (hop 128, so consecutive frames share half their samples) and the
// 256-entry twiddle table live in LDS; a thread owns one bin of 8 frames, so each (plane, bin) row leaves as two
// 16-byte stores.  Direct DFT: 8.4 MFLOP per tile, the kernel is bound by its 131 KB of output per tile.
__global__ __launch_bounds__(256) void stft256_tiles_kernel(const float* __restrict__ wave, float* __restrict__ tiles,
                                                            int64_t L, int nframes, int tiles_per_utt, float scale) {
  __shared__ float xs[16 * 128 + 128];
  __shared__ float tw[256];
  const int tid = threadIdx.x;
  const int groups = tiles_per_utt * 8;                       // 16-frame groups per utterance
  const int u = blockIdx.x / groups, gq = blockIdx.x % groups, f0 = gq * 16;
  tw[tid] = cospif(tid * (1.0f / 128.0f));
  const float* w = wave + (int64_t)u * L;
  for (int i = tid; i < 16 * 128 + 128; i += 256) {
    int64_t p = (int64_t)f0 * 128 + i - 128;                  // sample index before centre padding
    if (p < 0) p = -p;                                        // reflect (no edge repeat), pad = 128 < L
    if (p >= L) p = 2 * (L - 1) - p;
    xs[i] = (p >= 0 && p < L) ? w[p] : 0.f;
  }
  __syncthreads();
  const int k = tid & 127, half = tid >> 7;
  float re[8], im[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float* x = xs + (half * 8 + j) * 128;
    float ar = 0.f, ai = 0.f;
    int idx = 0;
#pragma unroll 8
    for (int n = 0; n < 256; ++n) {
      const float v = x[n];
      ar = fmaf(v, tw[idx], ar);                              // cos(2 pi k n / 256)
      ai = fmaf(v, tw[(idx + 64) & 255], ai);                 // -sin(2 pi k n / 256) = cos(2 pi (kn + 64) / 256)
      idx = (idx + k) & 255;
    }
    const bool live = f0 + half * 8 + j < nframes;
    re[j] = live ? ar * scale : 0.f;
    im[j] = live ? ai * scale : 0.f;
  }
  const int tile = f0 >> 7, fl = (f0 & 127) + half * 8;
  float* o = tiles + (((int64_t)(u * tiles_per_utt + tile) * 2) * 128 + k) * 128 + fl;
  *reinterpret_cast<float4*>(o) = make_float4(re[0], re[1], re[2], re[3]);
  *reinterpret_cast<float4*>(o + 4) = make_float4(re[4], re[5], re[6], re[7]);
  *reinterpret_cast<float4*>(o + 128 * 128) = make_float4(im[0], im[1], im[2], im[3]);
  *reinterpret_cast<float4*>(o + 128 * 128 + 4) = make_float4(im[4], im[5], im[6], im[7]);
}

// ---------------------------------------------------------------------------------------------------------------
// Whole-utterance pair of the evaluation path (rows N1 / N2): SpeechDataTest.prepare_data (audio_test.py:313-343) and
// the re-analysis after the attack (:676-688) are torch.stft(n_fft=255) of a waveform of any length cut into
// [2,128,128] tiles; reconstruct_audio (:594-600) concatenates the encoded tiles along the frame axis (the last one
// trimmed to len_last_clip frames) and calls torch.istft(n_fft=255, length=L).  Same direct real DFT as above with
// the frame count and the waveform length as run-time values; frame t lives in tile t / 128, column t % 128.
__global__ __launch_bounds__(256) void istft255_stitch_kernel(const float* __restrict__ tiles, float* __restrict__ wave,
                                                             int n_tiles, int nframe, int64_t L, float out_scale) {
  __shared__ float2 Xs[NBIN * MAXF];
  __shared__ float2 tw[NFFT + 1];
  const int tid = threadIdx.x, u = blockIdx.y;
  const int64_t i0 = (int64_t)blockIdx.x * CHUNK, i1 = i0 + CHUNK < L ? i0 + CHUNK : L;
  const int64_t ola = NFFT + (int64_t)HOP * (nframe - 1);      // torch pads with zeros beyond the overlap-add length
  float* dst = wave + (int64_t)u * L;
  if (nframe <= 0 || i0 + PADL >= ola) {
    for (int64_t i = i0 + tid; i < i1; i += 256) dst[i] = 0.f;
    return;
  }
  const int qlo = (int)i0 + PADL;
  int qhi = (int)(i1 - 1) + PADL;
  if (qhi > ola - 1) qhi = (int)(ola - 1);
  int t0;
  frames_covering(qlo, t0, nframe);
  int t1 = qhi / HOP;
  if (t1 > nframe - 1) t1 = nframe - 1;
  const int nf = t1 - t0 + 1;                                  // <= MAXF: (CHUNK + NFFT - 1) / HOP + 1 = 14
  for (int m = tid; m < NFFT; m += 256) {
    double sn, cs;
    sincospi(2.0 * (double)m / (double)NFFT, &sn, &cs);
    tw[m] = make_float2((float)cs, (float)sn);
  }
  const float* src = tiles + (int64_t)u * n_tiles * 2 * NBIN * 128;
  for (int idx = tid; idx < NBIN * MAXF; idx += 256) {
    const int k = idx / MAXF, tt = idx % MAXF, t = t0 + tt;
    float2 v = make_float2(0.f, 0.f);
    if (tt < nf) {
      const float* p = src + ((int64_t)(t >> 7) * 2 * NBIN + k) * 128 + (t & 127);
      v = make_float2(p[0], p[NBIN * 128]);
    }
    Xs[idx] = v;
  }
  __syncthreads();
  for (int64_t i = i0 + tid; i < i1; i += 256)
    dst[i] = i + PADL < ola ? synth_at<false>((int)i + PADL, Xs, t0, tw, nframe) * out_scale : 0.f;
}

__global__ __launch_bounds__(256) void stft255_tiles_kernel(const float* __restrict__ wave, float* __restrict__ tiles,
                                                            int64_t L, int nframe, int tiles_per_utt, float scale) {
  __shared__ float fs[SPAN];
  __shared__ float2 tw[NFFT + 1];
  const int tid = threadIdx.x;
  const int groups = tiles_per_utt * (128 / TG);
  const int u = blockIdx.x / groups, tg0 = (blockIdx.x % groups) * TG;
  for (int m = tid; m < NFFT; m += 256) {
    double sn, cs;
    sincospi(2.0 * (double)m / (double)NFFT, &sn, &cs);
    tw[m] = make_float2((float)cs, (float)sn);
  }
  const float* src = wave + (int64_t)u * L;
  for (int s = tid; s < SPAN; s += 256) {
    int64_t i = (int64_t)HOP * tg0 + s - PADL;
    if (i < 0) i = -i;                                         // reflect, PADL < L
    if (i >= L) i = 2 * (L - 1) - i;
    fs[s] = (i >= 0 && i < L) ? src[i] : 0.f;                  // frames past the utterance are zeroed below anyway
  }
  __syncthreads();
  const int k = tid & (NBIN - 1), half = tid >> 7;
  float re[TG / 2], im[TG / 2];
#pragma unroll
  for (int f = 0; f < TG / 2; ++f) { re[f] = 0.f; im[f] = 0.f; }
  const float* base = fs + HOP * (half * (TG / 2));
  int idx = 0;
  for (int n = 0; n < NFFT; ++n) {
    const float2 w = tw[idx];
#pragma unroll
    for (int f = 0; f < TG / 2; ++f) {
      const float x = base[HOP * f + n];
      re[f] += x * w.x;
      im[f] -= x * w.y;
    }
    idx += k;
    if (idx >= NFFT) idx -= NFFT;
  }
  const int f0 = tg0 + half * (TG / 2), tile = f0 >> 7;
  float* dst = tiles + (((int64_t)(u * tiles_per_utt + tile) * 2) * NBIN + k) * 128 + (f0 & 127);
#pragma unroll
  for (int f = 0; f < TG / 2; ++f) {
    const bool live = f0 + f < nframe;
    dst[f] = live ? re[f] * scale : 0.f;
    dst[NBIN * 128 + f] = live ? im[f] * scale : 0.f;
  }
}

// Whole-utterance waveform attacks of the evaluation loop that need no filter state (audio_attack.py:55-58 amplitude
// scaling; awgn :99-125 with the noise drawn from Philox instead of numpy's generator): power pass + apply pass.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  const float m = sqrtf(-2.0f * logf(u01(a)));
  float sn, cs;
  sincospif(2.0f * u01(b), &sn, &cs);
  z0 = m * cs;
  z1 = m * sn;
}
// one workgroup per utterance, fixed summation order: the noise level (and so the attacked waveform) is reproducible
__global__ __launch_bounds__(1024) void wave_power_kernel(const float* __restrict__ wave, float* __restrict__ power, int64_t L) {
  __shared__ float red[16];
  const float* w = wave + (int64_t)blockIdx.x * L;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < L; i += 1024) s += w[i] * w[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += red[i];
    power[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(256) void wave_attack_kernel(const float* __restrict__ wave, float* __restrict__ out,
                                                          const float* __restrict__ power, int64_t L, const WmWaveAttack a) {
  const int u = blockIdx.y, flags = a.flags;
  float sigma = 0.f;
  if (flags & 2) sigma = sqrtf(power[u] / (float)L * exp10f(-a.snr_db / 10.0f));
  const float amp = (flags & 8) ? a.amp : 1.0f;
  const float* w = wave + (int64_t)u * L;
  float* o = out + (int64_t)u * L;
  for (int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x; i4 * 4 < L; i4 += (int64_t)gridDim.x * 256) {
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (flags & 2) {
      uint32_t r[4];
      philox4x32((uint32_t)i4, (uint32_t)(i4 >> 32), (uint32_t)u, 0x57a7u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
      box_muller(r[0], r[1], z[0], z[1]);
      box_muller(r[2], r[3], z[2], z[3]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t i = i4 * 4 + q;
      if (i >= L) continue;
      float v = (w[i] + sigma * z[q]) * amp;
      if ((flags & 16) && i >= a.echo_delay) v += a.echo_gain * w[i - a.echo_delay];       // audio_attack.py:33-52
      if (flags & 64) v = rintf(fminf(fmaxf(v, -1.0f), 1.0f) * 127.0f) * (1.0f / 128.0f);   // PCM_U8 write / read (:85-96)
      o[i] = v;
    }
  }
}
// jittering_2 (audio_attack.py:176-193): `count` uniformly drawn sample positions are set to zero
__global__ void wave_jitter_kernel(float* __restrict__ out, int64_t L, int count, uint64_t seed) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, u = blockIdx.y;
  if (j >= count) return;
  uint32_t r[4];
  philox4x32((uint32_t)j, 0u, (uint32_t)u, 0x6a17u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  const int64_t idx = (int64_t)(((uint64_t)r[0] * (uint64_t)L) >> 32);
  out[(int64_t)u * L + idx] = 0.f;
}

}  // namespace

extern "C" int32_t wm_stft256_tiles_per_utterance(int64_t L) {
  const int64_t frames = 1 + L / 128;
  return (int32_t)(frames / 128 + 1);                         // pad = 128 - frames % 128 (audio_test.py:476-477): always >= 1 tile
}

extern "C" int wm_stft256_tiles(const float* wave, int64_t L, int32_t U, float* tiles, float scale, void* stream) {
  if (!wave || !tiles || U <= 0 || L <= 128) return WM_EINVAL;
  const int nframes = (int)(1 + L / 128), tpu = wm_stft256_tiles_per_utterance(L);
  hipLaunchKernelGGL(stft256_tiles_kernel, dim3(U * tpu * 8), dim3(256), 0, static_cast<hipStream_t>(stream), wave, tiles, L,
                     nframes, tpu, scale);
  WM_CHECK_LAUNCH();
  return WM_OK;
}


extern "C" int32_t wm_stft255_tiles_per_utterance(int64_t L) {
  const int64_t frames = 1 + (L - 1) / HOP;                   // centre padding adds 2 * 127 = n_fft - 1 samples
  return (int32_t)(frames / 128 + 1);                         // len_pad = 128 - frames % 128 (audio_test.py:318-319)
}
extern "C" int32_t wm_stft255_frames(int64_t L) { return (int32_t)(1 + (L - 1) / HOP); }

extern "C" int wm_stft255_tiles(const float* wave, int64_t L, int32_t U, float* tiles, float scale, void* stream) {
  if (!wave || !tiles || U <= 0 || L <= PADL || L > (int64_t)1 << 30) return WM_EINVAL;
  const int nframe = wm_stft255_frames(L), tpu = wm_stft255_tiles_per_utterance(L);
  hipLaunchKernelGGL(stft255_tiles_kernel, dim3(U * tpu * (128 / TG)), dim3(256), 0, static_cast<hipStream_t>(stream), wave,
                     tiles, L, nframe, tpu, scale);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_istft255_stitch(const float* tiles, int32_t n_tiles, int32_t last_len, int32_t U, float* wave, int64_t L,
                                  float out_scale, void* stream) {
  if (!tiles || !wave || n_tiles <= 0 || last_len < 0 || last_len > 128 || U <= 0 || L <= 0 || L > (int64_t)1 << 30)
    return WM_EINVAL;
  const int nframe = (n_tiles - 1) * 128 + last_len;
  hipLaunchKernelGGL(istft255_stitch_kernel, dim3((unsigned)((L + CHUNK - 1) / CHUNK), U), dim3(256), 0,
                     static_cast<hipStream_t>(stream), tiles, wave, n_tiles, nframe, L, out_scale);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_wave_attack(const WmWaveAttack* a, const float* wave, float* out, int64_t L, int32_t U, float* power,
                              void* stream) {
  if (!a || !wave || !out || L <= 0 || L > (int64_t)1 << 31 || U <= 0 || (a->flags & ~(2 | 8 | 16 | 32 | 64)) ||
      ((a->flags & 2) && !power) || ((a->flags & 16) && a->echo_delay <= 0) || ((a->flags & 32) && a->jitter_count < 0))
    return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned gx = (unsigned)((L + 4095) / 4096 > 256 ? 256 : (L + 4095) / 4096);
  if (a->flags & 2) {
    hipLaunchKernelGGL(wave_power_kernel, dim3(U), dim3(1024), 0, st, wave, power, L);
    WM_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(wave_attack_kernel, dim3(gx, U), dim3(256), 0, st, wave, out, power, L, *a);
  WM_CHECK_LAUNCH();
  if ((a->flags & 32) && a->jitter_count > 0) {
    hipLaunchKernelGGL(wave_jitter_kernel, dim3((a->jitter_count + 255) / 256, U), dim3(256), 0, st, out, L, a->jitter_count,
                       a->seed);
    WM_CHECK_LAUNCH();
  }
  return WM_OK;
}
