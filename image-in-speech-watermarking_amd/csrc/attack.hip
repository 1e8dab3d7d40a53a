// Fused waveform attacks between the in-graph iSTFT and STFT (hook point reference model.py:2461-2462).
// The four distortions BASELINE.json names (reference uformerWM/audio_attack.py):
//   resampling 16k -> 8k -> 16k (:71-83)   polyphase Kaiser-windowed-sinc, 41 taps (scipy resample_poly
//                                           conventions; the reference's librosa/resampy kernel is not pinned)
//   awgn(snr)                 (:99-125)     per-waveform power, Philox/Box-Muller noise
//   low_pass_filter           (:21-30)      butter(8, 0.5) + filtfilt(odd padding 27, steady-state zi), evaluated
//                                           as the equivalent pair of causal / anti-causal 128-tap FIR passes
//                                           (impulse response truncated at 5e-12)
//   amplitude_scaling(factor) (:55-58)
// One workgroup owns one waveform and keeps it in LDS through the whole chain (8002 samples), so the chain costs one
// HBM read and one HBM write per sample.  Every stage is linear except AWGN's signal-dependent sigma; the backward
// kernel applies the exact adjoints in reverse order.
//
// The two 128-tap passes are 2 M multiply-adds per row: as scalar loops they were 90 % of the kernel (round 1: 110 us
// forward, 163 us backward per launch).  They run on the matrix cores instead: with the row cut into blocks of 32
// samples, y[32 a + r] = sum_d sum_r' e[32 (a - d) + r'] h[32 d + r - r'] is a [blocks x 160] x [160 x 32] product
// whose right factor is a constant banded Toeplitz matrix (wm_attack_fir_tables lays it out in MFMA operand order) and
// whose left factor is the row itself, read as it lies; split-precision bf16 (hi*hi + hi*lo + lo*hi, fp32
// accumulate) keeps the result within 1e-5 of the fp32 loop.  The row is held in LDS in blocks of 32 samples padded
// to 36 words so that the 16-byte operand reads of 32 lanes (one block each) fall on distinct banks.
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

constexpr int NT = 1024;               // 16 waves: the LDS-latency-bound stages (resampler, extension, reductions) overlap better
constexpr int PADLEN = 27;                 // scipy filtfilt default: 3 * max(len(a), len(b)) = 3 * 9
constexpr int RS_HALF = 20;                // resample_poly half length for rate 2
constexpr int NBLK = 252, LDB = 36;        // 252 blocks of 32 samples (8064 >= 8002 + 2 * 27), 36 words apart
constexpr int BUF = NBLK * LDB;
constexpr int FIR_J = 128, FIR_KS = 10;    // taps of the MFMA path; k steps of 16 over 5 block shifts x 32 samples
constexpr int FIR_TAB = 2 * 2 * FIR_KS * 64 * 8;   // [causal | anti-causal][hi | lo][k step][lane][8] bf16

__device__ __forceinline__ int ix(int n) { return (n >> 5) * LDB + (n & 31); }

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) s += red[i];
  return s;
}

// ---- 16k -> 8k -> 16k resampling ---------------------------------------------------------------------------------
// The prototype h (41 taps, Kaiser-windowed sinc at half band) is symmetric and zero at every even offset but the
// centre (sinc(n / 2) = 0), so each output needs the centre tap and ten symmetric pairs at odd offsets:
//   down  y[m] = h0 x[2m] + sum_k hk (x[2m - (2k+1)] + x[2m + (2k+1)])                         m < Ld = ceil(L / 2)
//   up    z[n] = 2 h0 y[n/2] (n even),  2 sum_k hk (y[(n - 2k - 1) / 2] + y[(n + 2k + 1) / 2])  (n odd)
// with x, y zero outside [0, L) / [0, Ld) (scipy.signal.resample_poly conventions).  Interior outputs take the
// unguarded path; the few whose taps reach over an edge take the guarded one.  `LIN` buffers are plainly indexed,
// the others use the blocked layout ix().
template <bool BLOCKED> __device__ __forceinline__ int at(int n) { return BLOCKED ? ix(n) : n; }

// y[m], m < Ld, from x (length L)
template <bool IN_BLOCKED, bool OUT_BLOCKED>
__device__ __forceinline__ void rs_down(const float* in, float* out, const float* h, int L, int Ld, int tid) {
  float hk[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) hk[k] = h[RS_HALF + 2 * k + 1];
  const float h0 = h[RS_HALF];
  for (int m = tid; m < Ld; m += NT) {
    const int c = 2 * m;
    float acc = h0 * in[at<IN_BLOCKED>(c)];
    if (c >= 19 && c + 19 < L) {
#pragma unroll
      for (int k = 0; k < 10; ++k) acc += hk[k] * (in[at<IN_BLOCKED>(c - 2 * k - 1)] + in[at<IN_BLOCKED>(c + 2 * k + 1)]);
    } else {
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const int lo = c - 2 * k - 1, hi = c + 2 * k + 1;
        acc += hk[k] * ((lo >= 0 ? in[at<IN_BLOCKED>(lo)] : 0.f) + (hi < L ? in[at<IN_BLOCKED>(hi)] : 0.f));
      }
    }
    out[at<OUT_BLOCKED>(m)] = acc;
  }
}
// z[n], n < L, from y (length Ld).  STORE(n, value) places the result (LDS in either layout, or global memory)
template <bool IN_BLOCKED, typename STORE>
__device__ __forceinline__ void rs_up(const float* in, const float* h, int L, int Ld, int tid, STORE store) {
  float hk[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) hk[k] = h[RS_HALF + 2 * k + 1];
  const float h0 = h[RS_HALF];
  for (int n = tid; n < L; n += NT) {
    float acc;
    if (!(n & 1)) {
      acc = h0 * in[at<IN_BLOCKED>(n >> 1)];
    } else {
      const int c = n >> 1;                                    // pairs y[c - k], y[c + 1 + k]
      acc = 0.f;
      if (c >= 9 && c + 10 < Ld) {
#pragma unroll
        for (int k = 0; k < 10; ++k) acc += hk[k] * (in[at<IN_BLOCKED>(c - k)] + in[at<IN_BLOCKED>(c + 1 + k)]);
      } else {
#pragma unroll
        for (int k = 0; k < 10; ++k) {
          const int lo = c - k, hi = c + 1 + k;
          acc += hk[k] * ((lo >= 0 ? in[at<IN_BLOCKED>(lo)] : 0.f) + (hi < Ld ? in[at<IN_BLOCKED>(hi)] : 0.f));
        }
      }
    }
    store(n, 2.0f * acc);
  }
}
// The adjoints: up^T has the shape of `down` (times 2) on a length-L input with Ld outputs, down^T the shape of `up`
// (without the 2) from Ld inputs to L outputs.
template <bool IN_BLOCKED, bool OUT_BLOCKED>
__device__ __forceinline__ void rs_up_T(const float* g, float* out, const float* h, int L, int Ld, int tid) {
  float hk[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) hk[k] = h[RS_HALF + 2 * k + 1];
  const float h0 = h[RS_HALF];
  for (int m = tid; m < Ld; m += NT) {
    const int c = 2 * m;                                       // c < L always (Ld = ceil(L / 2))
    float acc = h0 * g[at<IN_BLOCKED>(c)];
    if (c >= 19 && c + 19 < L) {
#pragma unroll
      for (int k = 0; k < 10; ++k) acc += hk[k] * (g[at<IN_BLOCKED>(c - 2 * k - 1)] + g[at<IN_BLOCKED>(c + 2 * k + 1)]);
    } else {
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const int lo = c - 2 * k - 1, hi = c + 2 * k + 1;
        acc += hk[k] * ((lo >= 0 ? g[at<IN_BLOCKED>(lo)] : 0.f) + (hi < L ? g[at<IN_BLOCKED>(hi)] : 0.f));
      }
    }
    out[at<OUT_BLOCKED>(m)] = 2.0f * acc;
  }
}
template <bool IN_BLOCKED, typename STORE>
__device__ __forceinline__ void rs_down_T(const float* g, const float* h, int L, int Ld, int tid, STORE store) {
  float hk[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) hk[k] = h[RS_HALF + 2 * k + 1];
  const float h0 = h[RS_HALF];
  for (int i = tid; i < L; i += NT) {
    float acc;
    if (!(i & 1)) {
      acc = h0 * g[at<IN_BLOCKED>(i >> 1)];
    } else {
      const int c = i >> 1;                                    // y indices (i -+ (2k+1)) / 2 = c - k, c + 1 + k
      acc = 0.f;
      if (c >= 9 && c + 10 < Ld) {
#pragma unroll
        for (int k = 0; k < 10; ++k) acc += hk[k] * (g[at<IN_BLOCKED>(c - k)] + g[at<IN_BLOCKED>(c + 1 + k)]);
      } else {
#pragma unroll
        for (int k = 0; k < 10; ++k) {
          const int lo = c - k, hi = c + 1 + k;
          acc += hk[k] * ((lo >= 0 ? g[at<IN_BLOCKED>(lo)] : 0.f) + (hi < Ld ? g[at<IN_BLOCKED>(hi)] : 0.f));
        }
      }
    }
    store(i, acc);
  }
}

// ---- the FIR passes ------------------------------------------------------------------------------------------------
// causal:       y[n] = sum_{j <= min(n, J-1)}       h[j] e[n - j] + edge * tail[min(n, J-1)]
// anti-causal:  y[n] = sum_{j <= min(N-1-n, J-1)}   h[j] e[n + j] + edge * tail[min(N-1-n, J-1)]
// (edge = the sample the steady-state initial condition repeats; the adjoint passes call these with edge = 0)
template <bool CAUSAL>
__device__ __forceinline__ void fir_valu(const float* e, float* y, const float* h, const float* tail, int N, int J, float edge,
                                         int tid) {
  for (int n = tid; n < NBLK * 32; n += NT) {
    float acc = 0.f;
    if (n < N) {
      const int room = CAUSAL ? n : N - 1 - n, jm = room < J - 1 ? room : J - 1;
      acc = edge * tail[jm];
      for (int j = 0; j <= jm; ++j) acc += h[j] * e[ix(CAUSAL ? n - j : n + j)];
    }
    y[ix(n)] = acc;
  }
}

__device__ __forceinline__ void split8(const float4& x0, const float4& x1, bf16x8& hi, bf16x8& lo) {
  const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h_ = (__bf16)v[j];
    hi[j] = h_;
    lo[j] = (__bf16)(v[j] - (float)h_);
  }
}

// One wave = 32 blocks (1024 samples) of the row; e has zeros from N up to NBLK * 32 (the caller pads).  J = 128.
template <bool CAUSAL>
__device__ __forceinline__ void fir_mfma(const float* e, float* y, const __bf16* tab, const float* tail, int N, float edge,
                                         int tid) {
  const int lane = tid & 63, wv = tid >> 6, r = lane & 31, hh = lane >> 5;
  if (wv >= (NBLK + 31) / 32) return;                         // 8 row tiles of 32 blocks: waves 8 .. 15 sit this stage out
  const __bf16* t = tab + (CAUSAL ? 0 : 2 * FIR_KS * 512) + lane * 8;   // (the LDS copy the kernel made at its start)
  bf16x8 Bh[FIR_KS], Bl[FIR_KS];
#pragma unroll
  for (int ks = 0; ks < FIR_KS; ++ks) {
    Bh[ks] = *reinterpret_cast<const bf16x8*>(t + ks * 512);
    Bl[ks] = *reinterpret_cast<const bf16x8*>(t + (FIR_KS + ks) * 512);
  }
  const int a = 32 * wv + r;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
  for (int ks = 0; ks < FIR_KS; ++ks) {
    const int d = ks >> 1, src = CAUSAL ? a - d : a + d;
    const bool ok = src >= 0 && src < NBLK;
    const float* p = e + (ok ? src : 0) * LDB + 16 * (ks & 1) + 8 * hh;
    float4 x0 = *reinterpret_cast<const float4*>(p), x1 = *reinterpret_cast<const float4*>(p + 4);
    if (!ok) { x0 = make_float4(0.f, 0.f, 0.f, 0.f); x1 = x0; }
    bf16x8 ah, al;
    split8(x0, x1, ah, al);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, Bh[ks], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, Bl[ks], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, Bh[ks], acc, 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int blk = 32 * wv + mma32_row(q, hh);
    if (blk < NBLK) {
      const int n = 32 * blk + r;
      float v = 0.f;
      if (n < N) {
        const int room = CAUSAL ? n : N - 1 - n;
        v = acc[q] + edge * tail[room < FIR_J - 1 ? room : FIR_J - 1];
      }
      y[blk * LDB + r] = v;
    }
  }
}

template <bool CAUSAL>
__device__ __forceinline__ void fir_pass(const float* e, float* y, const float* h, const float* tail, const __bf16* tab, int N,
                                         int J, float edge, int tid) {
  if (tab) fir_mfma<CAUSAL>(e, y, tab, tail, N, edge, tid);
  else fir_valu<CAUSAL>(e, y, h, tail, N, J, edge, tid);
}

// B[k][col r] of k step ks = (block shift d = ks >> 1, half = ks & 1): k-local r' = 16 half + 8 (lane >> 5) + j
__global__ __launch_bounds__(256) void fir_table_kernel(const float* __restrict__ h, __bf16* __restrict__ tab) {
  const int id = blockIdx.x * 256 + threadIdx.x;              // (direction, k step, lane, j)
  if (id >= 2 * FIR_KS * 64 * 8) return;
  const int j = id & 7, lane = (id >> 3) & 63, ks = (id >> 9) % FIR_KS, dir = id / (FIR_KS * 512);
  const int r = lane & 31, rp = 16 * (ks & 1) + 8 * (lane >> 5) + j, d = ks >> 1;
  const int tap = dir == 0 ? 32 * d + r - rp : 32 * d + rp - r;
  const float v = (tap >= 0 && tap < FIR_J) ? h[tap] : 0.f;
  const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
  __bf16* o = tab + dir * 2 * FIR_KS * 512 + (ks * 64 + lane) * 8 + j;
  o[0] = hi;
  o[FIR_KS * 512] = lo;
}

__device__ __forceinline__ void gauss4(uint64_t seed, uint64_t step, uint32_t row, uint32_t blk, float z[4]) {
  uint32_t r[4];
  philox4x32(blk, row, (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
  // Box-Muller on the hardware transcendentals (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32 take turns): ~1e-6 of the libm
  // values, a quarter of their instructions; the stream is defined by THIS function (wm_awgn_noise replays it)
  const float u0 = u01(r[0]), u1 = u01(r[1]), u2 = u01(r[2]), u3 = u01(r[3]);
  const float m0 = __builtin_amdgcn_sqrtf(-1.38629436111989f * __builtin_amdgcn_logf(u0));     // -2 ln 2 log2(u)
  const float m1 = __builtin_amdgcn_sqrtf(-1.38629436111989f * __builtin_amdgcn_logf(u2));
  z[0] = m0 * __builtin_amdgcn_cosf(u1); z[1] = m0 * __builtin_amdgcn_sinf(u1);
  z[2] = m1 * __builtin_amdgcn_cosf(u3); z[3] = m1 * __builtin_amdgcn_sinf(u3);
}

// jittering_2: the j-th zeroed position of a row; its own Philox stream (key differs from the AWGN stream's)
__device__ __forceinline__ int jitter_index(uint64_t seed, uint64_t step, uint32_t row, uint32_t j, int L) {
  uint32_t r[4];
  philox4x32(j, row, (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed ^ 0x6a17c0deu, (uint32_t)(seed >> 32), r);
  return (int)(((uint64_t)r[0] * (uint64_t)L) >> 32);
}

// Copies the MFMA operand image of the low-pass taps (40 KB) from global memory into LDS at the start of a kernel: the two FIR
// passes then take their B fragments from LDS instead of waiting for 20 L2 loads each.  The first barrier of the kernel
// orders the copy.  Returns NULL when the scalar path is to be used.
__device__ __forceinline__ const __bf16* stage_fir_table(const WmAttack& a, int J, __bf16* dst, int tid) {
  if (!(a.flags & 4) || !a.lp_mfma || J != FIR_J) return nullptr;
  const uint4* src = static_cast<const uint4*>(a.lp_mfma);
  uint4* d = reinterpret_cast<uint4*>(dst);
  for (int i = tid; i < FIR_TAB / 8; i += NT) d[i] = src[i];
  return dst;
}

__global__ __launch_bounds__(NT) void attack_fwd_kernel(const WmAttack a, const float* __restrict__ wave,
                                                        float* __restrict__ out, float* __restrict__ sigma_out,
                                                        int L) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* A = sm; float* Bf = sm + BUF; float* hs = sm + 2 * BUF;   // hs: lp_h[J] | tail[J] | rs_h[41] | red[8]
  const int tid = threadIdx.x, row = blockIdx.x, J = a.lp_n;
  float* tails = hs + J; float* rsh = hs + 2 * J; float* red = rsh + 48;
  const __bf16* tab = stage_fir_table(a, J, reinterpret_cast<__bf16*>(sm + 2 * BUF + 2 * 256 + 64), tid);
  if (a.flags & 4) for (int i = tid; i < 2 * J; i += NT) hs[i] = a.lp_h[i];
  if (a.flags & 1) for (int i = tid; i < 2 * RS_HALF + 1; i += NT) rsh[i] = a.rs_h[i];
  float* cur = A; float* oth = Bf;
  int off = 0;
  if (a.flags & 1) {                                           // plain layout while resampling, blocked from then on
    const int Ld = (L + 1) / 2;
    for (int i = tid; i < L; i += NT) oth[i] = wave[(int64_t)row * L + i];
    __syncthreads();
    rs_down<false, false>(oth, cur, rsh, L, Ld, tid);
    __syncthreads();
    rs_up<false>(cur, rsh, L, Ld, tid, [&](int n, float v) { oth[ix(n)] = v; });
    __syncthreads();
    float* t = cur; cur = oth; oth = t;
  } else {
    for (int i = tid; i < L; i += NT) cur[ix(i)] = wave[(int64_t)row * L + i];
    __syncthreads();
  }
  if (a.flags & 2) {
    float p = 0.f;
    for (int i = tid; i < L; i += NT) p += cur[ix(i)] * cur[ix(i)];
    p = block_sum(p, red, tid) / (float)L;
    const float sigma = sqrtf(p * exp10f(-a.snr_db * 0.1f));
    const uint64_t step = a.step ? (uint64_t)*a.step : 0ull;
    for (int q = tid; 4 * q < L; q += NT) {
      float z[4];
      gauss4(a.seed, step, (uint32_t)row, (uint32_t)q, z);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < L) cur[ix(4 * q + e)] += sigma * z[e];
    }
    if (tid == 0 && sigma_out) sigma_out[row] = sigma;
    __syncthreads();
  }
  if (a.flags & 4) {
    const int N = L + 2 * PADLEN;
    const float x0 = cur[ix(0)], xl = cur[ix(L - 1)];
    for (int q = tid; q < NBLK * 32; q += NT) {                // odd extension; zeros from N to the end of the last block
      float v = 0.f;
      if (q < PADLEN) v = 2.0f * x0 - cur[ix(PADLEN - q)];
      else if (q < PADLEN + L) v = cur[ix(q - PADLEN)];
      else if (q < N) v = 2.0f * xl - cur[ix(L - 2 - (q - PADLEN - L))];
      oth[ix(q)] = v;
    }
    __syncthreads();
    fir_pass<true>(oth, cur, hs, tails, tab, N, J, oth[ix(0)], tid);
    __syncthreads();
    fir_pass<false>(cur, oth, hs, tails, tab, N, J, cur[ix(N - 1)], tid);
    __syncthreads();
    float* t = cur; cur = oth; oth = t;
    off = PADLEN;
  }
  const float amp = (a.flags & 8) ? a.amp : 1.0f;
  if (!(a.flags & (16 | 32 | 64))) {
    for (int i = tid; i < L; i += NT) out[(int64_t)row * L + i] = amp * cur[ix(off + i)];
    return;
  }
  // row N3 stages, after the amplitude scale: echo addition -> jittering_2 -> 8-bit requantisation (oth: plain indexing)
  const int d = (a.flags & 16) ? a.echo_delay : L;
  for (int i = tid; i < L; i += NT) {
    float v = amp * cur[ix(off + i)];
    if (i >= d) v += a.echo_gain * amp * cur[ix(off + i - d)];
    oth[i] = v;
  }
  __syncthreads();
  if (a.flags & 32) {
    const uint64_t step = a.step ? (uint64_t)*a.step : 0ull;
    for (int j = tid; j < a.jitter_count; j += NT) oth[jitter_index(a.seed, step, (uint32_t)row, (uint32_t)j, L)] = 0.f;
    __syncthreads();
  }
  for (int i = tid; i < L; i += NT) {
    float v = oth[i];
    if (a.flags & 64) v = rintf(fminf(fmaxf(v, -1.0f), 1.0f) * 127.0f) * (1.0f / 128.0f);
    out[(int64_t)row * L + i] = v;
  }
}

__global__ __launch_bounds__(NT) void attack_bwd_kernel(const WmAttack a, const float* __restrict__ wave,
                                                        const float* __restrict__ dout,
                                                        const float* __restrict__ sigma_in,
                                                        float* __restrict__ dwave, int L) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* A = sm; float* Bf = sm + BUF; float* Cx = sm + 2 * BUF; float* hs = sm + 3 * BUF;
  const int tid = threadIdx.x, row = blockIdx.x, J = a.lp_n;
  float* tails = hs + J; float* rsh = hs + 2 * J; float* red = rsh + 48;
  const __bf16* tab = stage_fir_table(a, J, reinterpret_cast<__bf16*>(sm + 3 * BUF + 2 * 256 + 64), tid);
  if (a.flags & 4) for (int i = tid; i < 2 * J; i += NT) hs[i] = a.lp_h[i];
  if (a.flags & 1) for (int i = tid; i < 2 * RS_HALF + 1; i += NT) rsh[i] = a.rs_h[i];
  const int Ld = (L + 1) / 2;
  // x1 = input of the AWGN stage (needed for d sigma / d x)
  if (a.flags & 2) {
    if (a.flags & 1) {
      for (int i = tid; i < L; i += NT) A[i] = wave[(int64_t)row * L + i];
      __syncthreads();
      rs_down<false, false>(A, Bf, rsh, L, Ld, tid);
      __syncthreads();
      rs_up<false>(Bf, rsh, L, Ld, tid, [&](int n, float v) { Cx[ix(n)] = v; });
    } else {
      for (int i = tid; i < L; i += NT) Cx[ix(i)] = wave[(int64_t)row * L + i];
    }
  }
  __syncthreads();
  const float amp = (a.flags & 8) ? a.amp : 1.0f;
  float* g = A; float* oth = Bf;
  // adjoint of the row-N3 stages (requantisation: straight-through; jitter: the zeroed samples pass no gradient;
  // echo: g[i] += gain * g[i + delay]); the result replaces dout below (oth: plain indexing here)
  const bool tail_stages = (a.flags & (16 | 32)) != 0;
  if (tail_stages) {
    for (int i = tid; i < L; i += NT) oth[i] = dout[(int64_t)row * L + i];
    __syncthreads();
    if (a.flags & 32) {
      const uint64_t step = a.step ? (uint64_t)*a.step : 0ull;
      for (int j = tid; j < a.jitter_count; j += NT) oth[jitter_index(a.seed, step, (uint32_t)row, (uint32_t)j, L)] = 0.f;
      __syncthreads();
    }
  }
  const int ed = (a.flags & 16) ? a.echo_delay : L;
  auto dsrc = [&](int i) -> float {
    if (!tail_stages) return dout[(int64_t)row * L + i];
    float v = oth[i];
    if (i + ed < L) v += a.echo_gain * oth[i + ed];
    return v;
  };
  if (a.flags & 4) {
    const int N = L + 2 * PADLEN;
    for (int q = tid; q < NBLK * 32; q += NT)
      g[ix(q)] = (q >= PADLEN && q < PADLEN + L) ? amp * dsrc(q - PADLEN) : 0.f;
    __syncthreads();
    // adjoint of the anti-causal pass: causal correlation + tail term onto the last sample
    // (oth held the N3 scratch in plain indexing until here: dsrc is not used again)
    fir_pass<true>(g, oth, hs, tails, tab, N, J, 0.f, tid);
    float t2 = 0.f;
    for (int r = tid; r < J && r < N; r += NT) t2 += tails[r] * g[ix(N - 1 - r)];
    t2 = block_sum(t2, red, tid);
    // samples further than J-1 from the end all see tail[J-1]
    float t2b = 0.f;
    for (int n = tid; n < N - J; n += NT) t2b += g[ix(n)];
    t2b = block_sum(t2b, red, tid) * tails[J - 1];
    if (tid == 0) oth[ix(N - 1)] += t2 + t2b;
    __syncthreads();
    // adjoint of the causal pass
    fir_pass<false>(oth, g, hs, tails, tab, N, J, 0.f, tid);
    float t1 = 0.f;
    for (int n = tid; n < J && n < N; n += NT) t1 += tails[n] * oth[ix(n)];
    t1 = block_sum(t1, red, tid);
    float t1b = 0.f;
    for (int n = J + tid; n < N; n += NT) t1b += oth[ix(n)];
    t1b = block_sum(t1b, red, tid) * tails[J - 1];
    if (tid == 0) g[ix(0)] += t1 + t1b;
    __syncthreads();
    // adjoint of the odd extension
    float sl = 0.f, sr = 0.f;
    if (tid < PADLEN) { sl = g[ix(tid)]; sr = g[ix(PADLEN + L + tid)]; }
    sl = block_sum(sl, red, tid);
    sr = block_sum(sr, red, tid);
    for (int i = tid; i < L; i += NT) {
      float v = g[ix(PADLEN + i)];
      if (i == 0) v += 2.0f * sl;
      if (i >= 1 && i <= PADLEN) v -= g[ix(PADLEN - i)];
      if (i == L - 1) v += 2.0f * sr;
      if (i >= L - 1 - PADLEN && i <= L - 2) v -= g[ix(PADLEN + L + (L - 2 - i))];
      oth[ix(i)] = v;
    }
    __syncthreads();
    float* t = g; g = oth; oth = t;
  } else {
    for (int i = tid; i < L; i += NT) g[ix(i)] = amp * dsrc(i);
    __syncthreads();
  }
  if (a.flags & 2) {
    const float sigma = sigma_in[row];
    const uint64_t step = a.step ? (uint64_t)*a.step : 0ull;
    float dot = 0.f;
    for (int q = tid; 4 * q < L; q += NT) {
      float z[4];
      gauss4(a.seed, step, (uint32_t)row, (uint32_t)q, z);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < L) dot += g[ix(4 * q + e)] * z[e];
    }
    dot = block_sum(dot, red, tid);
    const float k = sigma > 0.f ? dot * exp10f(-a.snr_db * 0.1f) / ((float)L * sigma) : 0.f;
    for (int i = tid; i < L; i += NT) g[ix(i)] += k * Cx[ix(i)];
    __syncthreads();
  }
  if (a.flags & 1) {
    rs_up_T<true, false>(g, oth, rsh, L, Ld, tid);
    __syncthreads();
    rs_down_T<false>(oth, rsh, L, Ld, tid, [&](int i, float v) { dwave[(int64_t)row * L + i] = v; });
    return;
  }
  for (int i = tid; i < L; i += NT) dwave[(int64_t)row * L + i] = g[ix(i)];
}

// the AWGN stage's standard-normal stream by itself (z[row][i] as attack_fwd_kernel draws it for (seed, step))
__global__ __launch_bounds__(256) void awgn_noise_kernel(uint64_t seed, const int64_t* __restrict__ step_p, float* __restrict__ z,
                                                         int L) {
  const int row = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
  if (4 * q >= L) return;
  const uint64_t step = step_p ? (uint64_t)*step_p : 0ull;
  float v[4];
  gauss4(seed, step, (uint32_t)row, (uint32_t)q, v);
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (4 * q + e < L) z[(int64_t)row * L + 4 * q + e] = v[e];
}

bool args_ok(const WmAttack* a, int B, int L) {
  if (!a || B <= 0 || L < 64 || L + 2 * PADLEN > NBLK * 32) return false;
  if ((a->flags & 4) && (!a->lp_h || a->lp_n < 2 || a->lp_n > 256)) return false;
  if ((a->flags & 4) && a->lp_mfma && (reinterpret_cast<uintptr_t>(a->lp_mfma) & 15)) return false;
  if ((a->flags & 1) && (!a->rs_h || a->rs_n != 2 * RS_HALF + 1)) return false;
  if ((a->flags & 16) && a->echo_delay <= 0) return false;
  if ((a->flags & 32) && a->jitter_count < 0) return false;
  if (a->flags & ~127) return false;
  return true;
}
size_t smem_bytes(int nbuf, int J) { (void)J; return sizeof(float) * ((size_t)nbuf * BUF + 2 * 256 + 64) + (size_t)FIR_TAB * 2; }

}  // namespace

extern "C" int64_t wm_attack_fir_table_bytes(void) { return (int64_t)FIR_TAB * 2; }

extern "C" int wm_attack_fir_tables(const float* lp_h, int32_t lp_n, void* tables, void* stream) {
  if (!lp_h || !tables || lp_n != FIR_J || (reinterpret_cast<uintptr_t>(tables) & 15)) return WM_EINVAL;
  hipLaunchKernelGGL(fir_table_kernel, dim3(2 * FIR_KS * 64 * 8 / 256), dim3(256), 0, static_cast<hipStream_t>(stream), lp_h,
                     static_cast<__bf16*>(tables));
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_attack_fwd(const WmAttack* a, const float* wave, float* out, float* sigma_out, int32_t B,
                             int32_t L, void* stream) {
  if (!args_ok(a, B, L) || !wave || !out) return WM_EINVAL;
  if ((a->flags & 2) && !sigma_out) return WM_EINVAL;
  const size_t sm = smem_bytes(2, a->lp_n > 0 ? a->lp_n : 1);
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attack_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)smem_bytes(2, 256));
    done = true;
  }
  hipLaunchKernelGGL(attack_fwd_kernel, dim3(B), dim3(NT), sm, static_cast<hipStream_t>(stream), *a, wave, out,
                     sigma_out, L);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_attack_bwd(const WmAttack* a, const float* wave, const float* dout, const float* sigma,
                             float* dwave, int32_t B, int32_t L, void* stream) {
  if (!args_ok(a, B, L) || !wave || !dout || !dwave) return WM_EINVAL;
  if ((a->flags & 2) && !sigma) return WM_EINVAL;
  const size_t sm = smem_bytes(3, a->lp_n > 0 ? a->lp_n : 1);
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attack_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)smem_bytes(3, 256));
    done = true;
  }
  hipLaunchKernelGGL(attack_bwd_kernel, dim3(B), dim3(NT), sm, static_cast<hipStream_t>(stream), *a, wave, dout, sigma,
                     dwave, L);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_awgn_noise(uint64_t seed, const int64_t* step, float* z, int32_t B, int32_t L, void* stream) {
  if (!z || B <= 0 || L <= 0) return WM_EINVAL;
  hipLaunchKernelGGL(awgn_noise_kernel, dim3((L + 1023) / 1024, B), dim3(256), 0, static_cast<hipStream_t>(stream), seed, step,
                     z, L);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
