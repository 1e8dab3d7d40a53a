// In-graph STFT round trip of the watermark model on the matrix cores (reference model.py:2458 torch.istft and :2463
// torch.stft, both n_fft = 255 with every other argument defaulted): 128 one-sided bins, hop 63, rectangular window of
// 255, center=True with reflect padding, no normalisation.
//   spec  : [B, 2, 128 bins, 128 frames]  (re | im planes, as the model holds it)      wave : [B, 8002]
//
// 255 = 3 * 5 * 17 has no radix-2 butterfly, and a direct DFT is O(N^2) scalar work (round 1: 132 us per launch).
// The real DFT is folded instead: with c[k][n] = cos(2 pi k n / 255), s[k][n] = sin(2 pi k n / 255) for k, n in [0, 128)
//   synthesis  E[n] = sum_k Xr[k] c[k][n],  O[n] = sum_k Xi[k] s[k][n],  frame[n] = E - O, frame[255 - n] = E + O
//   analysis   u[n] = x[n] + x[255 - n], v[n] = x[n] - x[255 - n],  Re[k] = sum_n u[n] c[k][n], Im[k] = -sum_n v[n] s[k][n]
// i.e. two [128 x 128] x [128 x frames] products per tile and direction with ONE symmetric pair of tables, which run on
// v_mfma_f32_32x32x16_bf16 in split precision: x = hi + lo (two bf16), table = hi + lo, product = hi*hi + hi*lo + lo*hi
// (the dropped lo*lo term is 2^-18 relative; measured max error 2e-6 of full scale against torch.stft / istft, the
// fp32 MFMA form would run at 1/16 of the rate).  Overlap-add, envelope normalisation, centre trim, reflect padding
// and their adjoints are epilogues / prologues on the LDS image, so a tile costs one read of its input and one write of
// its output (algorithmic 163 080 B per tile and transform).
//
// The tables live in a caller-owned device buffer (wm_stft255_tables) already laid out in MFMA operand order: a wave's
// A fragment of a k-step is one coalesced 1 KB load from L2, no LDS staging.
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int NFFT = 255, HOP = 63, NBIN = 128, NFRAME = 128, WLEN = 8002, PADL = 127;
constexpr int OLA_LEN = NFFT + HOP * (NFRAME - 1);     // 8256 = 131 * 63 + 3
constexpr int FT = 32;                                 // frames per workgroup
constexpr int SLOTS = 28;                              // output slots (63 samples each) a synthesis workgroup owns: FT - 4
constexpr int SYN_GROUPS = 5;                          // 5 * 28 = 140 >= 132 slots
constexpr int LDF = 257;                               // frame image row stride (floats): odd -> conflict-free column writes
constexpr int SPAN = HOP * (FT - 1) + NFFT;            // 2208 samples staged per analysis workgroup
constexpr int LDU = 136;                               // folded-frame image row stride (bf16): 272 B = 4 mod 64 words
constexpr int TAB_ELEMS = 2 * 2 * 4 * 8 * 64 * 8;      // [cos|sin][hi|lo][row tile][k step][lane][8]

__device__ __forceinline__ int frames_covering(int q) {
  int thi = q / HOP;
  if (thi > NFRAME - 1) thi = NFRAME - 1;
  const int lo = q - (NFFT - 1);
  const int tlo = lo <= 0 ? 0 : (lo + HOP - 1) / HOP;
  return thi - tlo + 1;
}

__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

// table entry (tbl, part) of row r, column c, in operand order for v_mfma_f32_32x32x16_bf16 A fragments:
// lane (r & 31, h) of row tile r >> 5 holds columns 16 ks + 8 h + j
__global__ __launch_bounds__(256) void stft_table_kernel(__bf16* __restrict__ tab) {
  const int id = blockIdx.x * 256 + threadIdx.x;              // one (row tile, k step, lane, j) per thread
  if (id >= 4 * 8 * 64 * 8) return;
  const int j = id & 7, lane = (id >> 3) & 63, ks = (id >> 9) & 7, i = id >> 12;
  const int r = 32 * i + (lane & 31), c = 16 * ks + 8 * (lane >> 5) + j;
  double sn, cs;
  sincospi(2.0 * (double)((r * c) % NFFT) / (double)NFFT, &sn, &cs);
  __bf16 h, l;
  split_bf16((float)cs, h, l);
  tab[(0 * 2 + 0) * 16384 + id] = h;
  tab[(0 * 2 + 1) * 16384 + id] = l;
  split_bf16((float)sn, h, l);
  tab[(1 * 2 + 0) * 16384 + id] = h;
  tab[(1 * 2 + 1) * 16384 + id] = l;
}

struct TabFrags { bf16x8 ch, cl, sh, sl; };
__device__ __forceinline__ TabFrags load_tab(const __bf16* __restrict__ tab, int i, int ks, int lane) {
  const int off = ((i * 8 + ks) * 64 + lane) * 8;
  TabFrags t;
  t.ch = *reinterpret_cast<const bf16x8*>(tab + 0 * 16384 + off);
  t.cl = *reinterpret_cast<const bf16x8*>(tab + 1 * 16384 + off);
  t.sh = *reinterpret_cast<const bf16x8*>(tab + 2 * 16384 + off);
  t.sl = *reinterpret_cast<const bf16x8*>(tab + 3 * 16384 + off);
  return t;
}

// B operand (k = 16 ks + 8 h + j, column = lane & 31) out of an as-it-lies [k][32] bf16 LDS image (64-byte rows: the
// four rows a 32-lane half touches fall on disjoint bank quarters)
__device__ __forceinline__ bf16x8 tr_frag32(const __bf16* img, int ks, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const __bf16* a = img + (16 * ks + 8 * (g >> 1) + q) * FT + 16 * (g & 1) + 4 * p;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * FT));
  bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return f;
}

__device__ __forceinline__ void mma3(f32x16& acc, const bf16x8& th, const bf16x8& tl, const bf16x8& xh, const bf16x8& xl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th, xh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th, xl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tl, xh, acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// ADJ = false : wave = istft(spec)        frames weighted (1 | 2) / 255, overlap-add, / envelope, centre trim
// ADJ = true  : dwave = stft^T(dspec)     unweighted overlap-add, reflect padding folded back onto the samples
// grid (5, B): workgroup j owns OLA slots [28 j, 28 j + 28) (slot s = positions [63 s, 63 s + 63)) and computes the 32
// frames [28 j - 4, 28 j + 28) that reach them, so no sum crosses a workgroup (frames outside [0, 128) are zero).
template <bool ADJ>
__global__ __launch_bounds__(256) void synth_mfma_kernel(const float* __restrict__ spec, float* __restrict__ wave,
                                                         const __bf16* __restrict__ tab, int B) {
  __shared__ __attribute__((aligned(16))) char smem[FT * LDF * 4 + (ADJ ? SLOTS * HOP * 4 : 0)];
  __bf16* Xs = reinterpret_cast<__bf16*>(smem);               // [re|im][hi|lo][128 k][32 f] bf16 = 32 KB
  float* Fs = reinterpret_cast<float*>(smem);                 // [32 f][257] floats, after the products (aliases Xs)
  float* Ob = reinterpret_cast<float*>(smem + FT * LDF * 4);  // ADJ: this workgroup's padded-domain sums
  // Blocks are dealt round-robin to the 8 XCDs.  The 32-frame windows of a tile's five workgroups start at frame 28 j - 4:
  // their 128-byte row segments straddle two cache lines each, and on five different L2s every line was fetched from
  // HBM twice (PMC: 2.04 x the spectrum).  Linear id -> (tile, group) so that the groups of one tile share an XCD.
  const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3, grp = sq % SYN_GROUPS, b = (sq / SYN_GROUPS) * 8 + xcd;
  if (b >= B) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int s0 = SLOTS * grp, tf0 = s0 - 4;                   // first slot / first frame (a multiple of 4, may be -4)
  const float* src = spec + (int64_t)b * 2 * NBIN * NFRAME;
  // stage the 256 (plane, bin) rows x 32 frames: 8 float4 loads per thread, all issued before the first use
  float4 v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int id = tid + 256 * q, row = id >> 3, t = tf0 + 4 * (id & 7);
    v[q] = (t >= 0 && t < NFRAME) ? *reinterpret_cast<const float4*>(src + row * NFRAME + t) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int id = tid + 256 * q, row = id >> 3, k = row & (NBIN - 1), c4 = 4 * (id & 7);
    const float sc = ADJ ? 1.0f : (k == 0 ? 1.0f : 2.0f) / (float)NFFT;
    const float e[4] = {v[q].x * sc, v[q].y * sc, v[q].z * sc, v[q].w * sc};
    bf16x4 hi, lo;
#pragma unroll
    for (int u = 0; u < 4; ++u) { __bf16 h_, l_; split_bf16(e[u], h_, l_); hi[u] = h_; lo[u] = l_; }
    __bf16* dst = Xs + ((row >> 7) * 2) * (NBIN * FT) + k * FT + c4;
    *reinterpret_cast<bf16x4*>(dst) = hi;
    *reinterpret_cast<bf16x4*>(dst + NBIN * FT) = lo;
  }
  __syncthreads();
  // wave w: output samples n in [32 w, 32 w + 32) of the folded half, all 32 frames
  f32x16 E, O;
#pragma unroll
  for (int e = 0; e < 16; ++e) { E[e] = 0.f; O[e] = 0.f; }
  TabFrags tb = load_tab(tab, w, 0, lane);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    TabFrags nx = tb;
    if (ks + 1 < 8) nx = load_tab(tab, w, ks + 1, lane);
    const bf16x8 rh = tr_frag32(Xs + 0 * NBIN * FT, ks, lane), rl = tr_frag32(Xs + 1 * NBIN * FT, ks, lane);
    const bf16x8 ih = tr_frag32(Xs + 2 * NBIN * FT, ks, lane), il = tr_frag32(Xs + 3 * NBIN * FT, ks, lane);
    mma3(E, tb.ch, tb.cl, rh, rl);
    mma3(O, tb.sh, tb.sl, ih, il);
    tb = nx;
  }
  __syncthreads();                                            // every wave is done with Xs: the frame image takes its place
  {
    const int f = lane & 31, h = lane >> 5;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = 32 * w + mma32_row(e, h);
      Fs[f * LDF + n] = E[e] - O[e];
      if (n >= 1) Fs[f * LDF + NFFT - n] = E[e] + O[e];
    }
  }
  __syncthreads();
  float* dst = wave + (int64_t)b * WLEN;
  for (int ql = tid; ql < SLOTS * HOP; ql += 256) {
    const int q = HOP * s0 + ql;
    float sum = 0.f;
    if (q < OLA_LEN) {
      const int s = q / HOP, r = q - HOP * s;
#pragma unroll
      for (int jj = 0; jj < 5; ++jj) {
        const int t = s - jj, m = HOP * jj + r;
        if (m < NFFT && t >= 0 && t < NFRAME) sum += Fs[(t - tf0) * LDF + m];
      }
    }
    if (ADJ) {
      Ob[ql] = sum;
    } else {
      const int i = q - PADL;
      if (i >= 0 && i < WLEN) dst[i] = sum / (float)frames_covering(q);
    }
  }
  if (ADJ) {
    __syncthreads();
    for (int ql = tid; ql < SLOTS * HOP; ql += 256) {
      const int i = HOP * s0 + ql - PADL;
      if (i < 0 || i >= WLEN) continue;
      float val = Ob[ql];
      if (i >= 1 && i <= PADL) val += Ob[PADL - i - HOP * s0];                       // left reflection (workgroup 0 holds both)
      if (i >= WLEN - 1 - PADL && i <= WLEN - 2) val += Ob[2 * (WLEN - 1) - i + PADL - HOP * s0];   // right (workgroup 4)
      dst[i] = val;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// ADJ = false : spec = stft(wave)          reflect padding
// ADJ = true  : dspec = istft^T(dwave)     zero padding, 1 / envelope on the samples, (1 | 2) / 255 on the bins
// grid (4, B): workgroup = 32 frames x 128 bins x (re, im); wave w owns bins [32 w, 32 w + 32).
template <bool ADJ>
__global__ __launch_bounds__(256) void analysis_mfma_kernel(const float* __restrict__ wave, float* __restrict__ spec,
                                                            const __bf16* __restrict__ tab) {
  __shared__ __attribute__((aligned(16))) float xs[SPAN + 8];
  __shared__ __attribute__((aligned(16))) __bf16 UV[4 * FT * LDU];          // [u|v][hi|lo][32 t][136]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.y, t0 = FT * blockIdx.x;
  const float* src = wave + (int64_t)b * WLEN;
  for (int s = tid; s < SPAN; s += 256) {
    const int q = HOP * t0 + s;                               // padded / overlap-add coordinate
    int i = q - PADL;
    float val;
    if (ADJ) {
      val = (i >= 0 && i < WLEN) ? src[i] / (float)frames_covering(q) : 0.f;
    } else {
      if (i < 0) i = -i;
      if (i >= WLEN) i = 2 * (WLEN - 1) - i;
      val = src[i];
    }
    xs[s] = val;
  }
  __syncthreads();
  // folded frames u[n] = x[n] + x[255 - n], v[n] = x[n] - x[255 - n] (u[0] = x[0], v[0] = 0), split, [t][n] images.
  // lane -> frame (stride 63 words between lanes: conflict-free reads), wave / half-wave -> two chunks of 8 samples.
  // NOTE (measured, MI355X / ROCm 7.2): with SLP vectorisation on, the folds below compile to v_pk_add_f32 on register
  // pairs that ds_read2_b32 has just delivered, and with a second workgroup resident on the CU the SUM (not the
  // difference, computed a few instructions earlier from the same registers) came out wrong in lanes 48-63 of the last
  // chunk about once per 5 launches at B = 96.  This file is therefore built with -fno-slp-vectorize (csrc/Makefile);
  // tests/test_kernels_gpu.py::test_stft_round_trip_is_deterministic_under_load holds it.
  {
    const int t = tid & 31, c0 = tid >> 5;
    const float* fr = xs + HOP * t;
    float a[2][8], m[2][8];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = 8 * (c0 + 8 * it) + j;
        a[it][j] = fr[n];
        m[it][j] = fr[n ? NFFT - n : 0];
      }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int n0 = 8 * (c0 + 8 * it);
      bf16x8 uh, ul, vh, vl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool dc = n0 + j == 0;
        __bf16 h_, l_;
        split_bf16(dc ? a[it][j] : a[it][j] + m[it][j], h_, l_); uh[j] = h_; ul[j] = l_;
        split_bf16(dc ? 0.f : a[it][j] - m[it][j], h_, l_); vh[j] = h_; vl[j] = l_;
      }
      __bf16* d = UV + t * LDU + n0;
      *reinterpret_cast<bf16x8*>(d) = uh;
      *reinterpret_cast<bf16x8*>(d + FT * LDU) = ul;
      *reinterpret_cast<bf16x8*>(d + 2 * FT * LDU) = vh;
      *reinterpret_cast<bf16x8*>(d + 3 * FT * LDU) = vl;
    }
  }
  __syncthreads();
  f32x16 RE, IM;
#pragma unroll
  for (int e = 0; e < 16; ++e) { RE[e] = 0.f; IM[e] = 0.f; }
  const int r = lane & 31, h = lane >> 5;
  const __bf16* brow = UV + r * LDU + 8 * h;
  TabFrags tb = load_tab(tab, w, 0, lane);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    TabFrags nx = tb;
    if (ks + 1 < 8) nx = load_tab(tab, w, ks + 1, lane);
    const bf16x8 uh = *reinterpret_cast<const bf16x8*>(brow + 16 * ks);
    const bf16x8 ul = *reinterpret_cast<const bf16x8*>(brow + 16 * ks + FT * LDU);
    const bf16x8 vh = *reinterpret_cast<const bf16x8*>(brow + 16 * ks + 2 * FT * LDU);
    const bf16x8 vl = *reinterpret_cast<const bf16x8*>(brow + 16 * ks + 3 * FT * LDU);
    mma3(RE, tb.ch, tb.cl, uh, ul);
    mma3(IM, tb.sh, tb.sl, vh, vl);
    tb = nx;
  }
  float* dst = spec + (int64_t)b * 2 * NBIN * NFRAME + t0 + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int k = 32 * w + mma32_row(e, h);
    const float wk = ADJ ? (k == 0 ? 1.0f : 2.0f) / (float)NFFT : 1.0f;
    dst[k * NFRAME] = RE[e] * wk;
    dst[(NBIN + k) * NFRAME] = (ADJ && k == 0) ? 0.f : -IM[e] * wk;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int64_t wm_stft255_table_bytes(void) { return (int64_t)TAB_ELEMS * 2; }

extern "C" int wm_stft255_tables(void* tables, void* stream) {
  if (!tables || !aligned16(tables)) return WM_EINVAL;
  hipLaunchKernelGGL(stft_table_kernel, dim3(4 * 8 * 64 * 8 / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<__bf16*>(tables));
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_istft255_fwd(const float* spec, float* wave, const void* tables, int32_t B, void* stream) {
  if (!spec || !wave || !tables || B <= 0 || !aligned16(spec) || !aligned16(tables)) return WM_EINVAL;
  hipLaunchKernelGGL(synth_mfma_kernel<false>, dim3(8 * ((B + 7) / 8) * SYN_GROUPS), dim3(256), 0, static_cast<hipStream_t>(stream),
                     spec, wave, static_cast<const __bf16*>(tables), B);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_stft255_bwd(const float* dspec, float* dwave, const void* tables, int32_t B, void* stream) {
  if (!dspec || !dwave || !tables || B <= 0 || !aligned16(dspec) || !aligned16(tables)) return WM_EINVAL;
  hipLaunchKernelGGL(synth_mfma_kernel<true>, dim3(8 * ((B + 7) / 8) * SYN_GROUPS), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dspec, dwave, static_cast<const __bf16*>(tables), B);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_stft255_fwd(const float* wave, float* spec, const void* tables, int32_t B, void* stream) {
  if (!wave || !spec || !tables || B <= 0 || !aligned16(tables)) return WM_EINVAL;
  hipLaunchKernelGGL(analysis_mfma_kernel<false>, dim3(NFRAME / FT, B), dim3(256), 0, static_cast<hipStream_t>(stream), wave,
                     spec, static_cast<const __bf16*>(tables));
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_istft255_bwd(const float* dwave, float* dspec, const void* tables, int32_t B, void* stream) {
  if (!dwave || !dspec || !tables || B <= 0 || !aligned16(tables)) return WM_EINVAL;
  hipLaunchKernelGGL(analysis_mfma_kernel<true>, dim3(NFRAME / FT, B), dim3(256), 0, static_cast<hipStream_t>(stream), dwave,
                     dspec, static_cast<const __bf16*>(tables));
  WM_CHECK_LAUNCH();
  return WM_OK;
}
