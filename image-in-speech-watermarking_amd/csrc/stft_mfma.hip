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
// v_mfma_f32_32x32x16_bf16 in split precision, NS bf16 terms per operand:
//   NS = 2 (the bf16 compute mode): x = hi + lo, table = hi + lo, product = hi*hi + hi*lo + lo*hi (operands carry 16 significant
//           bits, the dropped lo*lo term is 2^-18 relative; measured max error 2e-6 of full scale against torch.stft / istft)
//   NS = 3 (the fp32 / split compute modes, whose outputs are held to 1e-4 of the CPU oracle BEHIND the extractor network):
//           x = hi + mid + lo (24 bits), product = hh + hm + mh + mm + hl + lh (dropped terms 2^-24 relative): fp32-grade, at
//           twice the MFMA work of a path that is 0.1 % of the step
// (the fp32 MFMA form would run at 1/16 of the rate).  Overlap-add, envelope normalisation, centre trim, reflect padding
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
constexpr int TAB_PLANE = 4 * 8 * 64 * 8;              // [row tile][k step][lane][8]
constexpr int tab_elems(int NS) { return 2 * NS * TAB_PLANE; }   // [cos|sin][term][row tile][k step][lane][8]

__device__ __forceinline__ int frames_covering(int q) {
  int thi = q / HOP;
  if (thi > NFRAME - 1) thi = NFRAME - 1;
  const int lo = q - (NFFT - 1);
  const int tlo = lo <= 0 ? 0 : (lo + HOP - 1) / HOP;
  return thi - tlo + 1;
}

template <int NS>
__device__ __forceinline__ void split_terms(float x, __bf16 (&p)[NS]) {
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    p[i] = (__bf16)x;
    x -= (float)p[i];                                          // exact: the remainder fits fp32
  }
}

// table entry (tbl, term) of row r, column c, in operand order for v_mfma_f32_32x32x16_bf16 A fragments:
// lane (r & 31, h) of row tile r >> 5 holds columns 16 ks + 8 h + j
template <int NS>
__global__ __launch_bounds__(256) void stft_table_kernel(__bf16* __restrict__ tab) {
  const int id = blockIdx.x * 256 + threadIdx.x;              // one (row tile, k step, lane, j) per thread
  if (id >= TAB_PLANE) return;
  const int j = id & 7, lane = (id >> 3) & 63, ks = (id >> 9) & 7, i = id >> 12;
  const int r = 32 * i + (lane & 31), c = 16 * ks + 8 * (lane >> 5) + j;
  double sn, cs;
  sincospi(2.0 * (double)((r * c) % NFFT) / (double)NFFT, &sn, &cs);
  __bf16 t[NS];
  split_terms<NS>((float)cs, t);
#pragma unroll
  for (int u = 0; u < NS; ++u) tab[(0 * NS + u) * TAB_PLANE + id] = t[u];
  split_terms<NS>((float)sn, t);
#pragma unroll
  for (int u = 0; u < NS; ++u) tab[(1 * NS + u) * TAB_PLANE + id] = t[u];
}

template <int NS> struct TabFrags { bf16x8 c[NS], s[NS]; };
template <int NS>
__device__ __forceinline__ TabFrags<NS> load_tab(const __bf16* __restrict__ tab, int i, int ks, int lane) {
  const int off = ((i * 8 + ks) * 64 + lane) * 8;
  TabFrags<NS> t;
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    t.c[u] = *reinterpret_cast<const bf16x8*>(tab + (0 * NS + u) * TAB_PLANE + off);
    t.s[u] = *reinterpret_cast<const bf16x8*>(tab + (1 * NS + u) * TAB_PLANE + off);
  }
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

// acc += (sum t) (sum x) without the terms below 2^-8NS: the small products first
template <int NS>
__device__ __forceinline__ void mma_split(f32x16& acc, const bf16x8 (&t)[NS], const bf16x8 (&x)[NS]) {
  if constexpr (NS == 3) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[0], x[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[2], x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[1], x[1], acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[0], x[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[1], x[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[0], x[0], acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// ADJ = false : wave = istft(spec)        frames weighted (1 | 2) / 255, overlap-add, / envelope, centre trim
// ADJ = true  : dwave = stft^T(dspec)     unweighted overlap-add, reflect padding folded back onto the samples
// Work item (tile, j): OLA slots [28 j, 28 j + 28) (slot s = positions [63 s, 63 s + 63)) and the 32 frames [28 j - 4, 28 j + 28)
// that reach them, so no sum crosses an item (frames outside [0, 128) are zero); five items per tile.
// Persistent workgroups (round 3): a wave keeps the table fragments of its 32 output rows for all eight k steps in 128
// registers and walks a strided list of items, the next item's spectrum columns in flight during the products.  Before, a
// workgroup fetched a k step's fragments from L2 one step ahead of their use — eight dependent round trips that made up most of
// its 20 us life, and 1280 such workgroups met 1024 resident slots at B = 256.
template <bool ADJ, int NS>
__global__ __launch_bounds__(256, NS == 2 ? 2 : 1) void synth_mfma_kernel(const float* __restrict__ spec, float* __restrict__ wave,
                                                                          const __bf16* __restrict__ tab, int B, int nitems) {
  constexpr int IMG = FT * LDF * 4 > 2 * NS * NBIN * FT * 2 ? FT * LDF * 4 : 2 * NS * NBIN * FT * 2;
  __shared__ __attribute__((aligned(16))) char smem[IMG + (ADJ ? SLOTS * HOP * 4 : 0)];
  __bf16* Xs = reinterpret_cast<__bf16*>(smem);               // [re|im][term][128 k][32 f] bf16 = 32 KB (NS = 2)
  float* Fs = reinterpret_cast<float*>(smem);                 // [32 f][257] floats, after the products (aliases Xs)
  float* Ob = reinterpret_cast<float*>(smem + IMG);           // ADJ: this item's padded-domain sums
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  TabFrags<NS> tb[8];                                         // NS = 3: 192 registers, one workgroup per CU
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) tb[ks] = load_tab<NS>(tab, w, ks, lane);
  // Blocks are dealt round-robin to the 8 XCDs.  The 32-frame windows of a tile's five items start at frame 28 j - 4: their
  // 128-byte row segments straddle two cache lines each, and on five different L2s every line was fetched from HBM twice
  // (PMC: 2.04 x the spectrum).  Item id -> (tile, group) so that the groups of one tile share an XCD (the grid is a multiple of
  // 8, so an item's id mod 8 is its workgroup's XCD).
  auto tile_of = [&](int item, int& b, int& grp) {
    const int xcd = item & 7, sq = item >> 3;
    grp = sq % SYN_GROUPS;
    b = (sq / SYN_GROUPS) * 8 + xcd;
  };
  // the 256 (plane, bin) rows x 32 frames of an item: 8 float4 loads per thread, all issued before the first use
  float4 v[8];
  auto fetch = [&](int item) {
    int b, grp;
    tile_of(item, b, grp);
    const int tf0 = SLOTS * grp - 4;
    const float* src = spec + (int64_t)(b < B ? b : 0) * 2 * NBIN * NFRAME;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int id = tid + 256 * q, row = id >> 3, t = tf0 + 4 * (id & 7);
      v[q] = (t >= 0 && t < NFRAME) ? *reinterpret_cast<const float4*>(src + row * NFRAME + t) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  int item = blockIdx.x;
  if (item < nitems) fetch(item);
  for (; item < nitems; item += gridDim.x) {
    int b, grp;
    tile_of(item, b, grp);
    const int s0 = SLOTS * grp;                               // first slot; first frame s0 - 4 (a multiple of 4, may be -4)
    __syncthreads();                                          // the previous item's frame image has been read
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int id = tid + 256 * q, row = id >> 3, k = row & (NBIN - 1), c4 = 4 * (id & 7);
      const float sc = ADJ ? 1.0f : (k == 0 ? 1.0f : 2.0f) / (float)NFFT;
      const float e[4] = {v[q].x * sc, v[q].y * sc, v[q].z * sc, v[q].w * sc};
      bf16x4 pt[NS];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        __bf16 t_[NS];
        split_terms<NS>(e[u], t_);
#pragma unroll
        for (int z = 0; z < NS; ++z) pt[z][u] = t_[z];
      }
      __bf16* dst = Xs + ((row >> 7) * NS) * (NBIN * FT) + k * FT + c4;
#pragma unroll
      for (int z = 0; z < NS; ++z) *reinterpret_cast<bf16x4*>(dst + z * NBIN * FT) = pt[z];
    }
    __syncthreads();
    if (item + (int)gridDim.x < nitems) fetch(item + gridDim.x);   // next item: in flight during the products and the overlap-add
    if (b >= B) continue;                                     // (padding item of a batch that is no multiple of 8; uniform)
    // wave w: output samples n in [32 w, 32 w + 32) of the folded half, all 32 frames
    f32x16 E, O;
#pragma unroll
    for (int e = 0; e < 16; ++e) { E[e] = 0.f; O[e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      bf16x8 re[NS], im[NS];
#pragma unroll
      for (int z = 0; z < NS; ++z) {
        re[z] = tr_frag32(Xs + z * NBIN * FT, ks, lane);
        im[z] = tr_frag32(Xs + (NS + z) * NBIN * FT, ks, lane);
      }
      mma_split<NS>(E, tb[ks].c, re);
      mma_split<NS>(O, tb[ks].s, im);
    }
    __syncthreads();                                          // every wave is done with Xs: the frame image takes its place
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
    // overlap-add: five frames reach a position (frames outside [0, 128) hold zeros); the five reads are issued unconditionally
    // at clamped addresses (a read under a branch waits for the one before it: 35 dependent LDS round trips per thread)
    for (int ql = tid; ql < SLOTS * HOP; ql += 256) {
      const int q = HOP * s0 + ql;
      const int sl = ql / HOP, r = ql - HOP * sl;             // slot within the item, position within the slot
      float t5[5];
#pragma unroll
      for (int jj = 0; jj < 5; ++jj) {
        const int m = HOP * jj + r;
        t5[jj] = Fs[(sl + 4 - jj) * LDF + (m < NFFT ? m : 0)];
      }
      float sum = 0.f;
#pragma unroll
      for (int jj = 0; jj < 5; ++jj) sum += (HOP * jj + r < NFFT) ? t5[jj] : 0.f;
      if (q >= OLA_LEN) sum = 0.f;
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
        if (i >= 1 && i <= PADL) val += Ob[PADL - i - HOP * s0];                       // left reflection (item 0 holds both)
        if (i >= WLEN - 1 - PADL && i <= WLEN - 2) val += Ob[2 * (WLEN - 1) - i + PADL - HOP * s0];   // right (item 4)
        dst[i] = val;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// ADJ = false : spec = stft(wave)          reflect padding
// ADJ = true  : dspec = istft^T(dwave)     zero padding, 1 / envelope on the samples, (1 | 2) / 255 on the bins
// Work item = 32 frames x 128 bins x (re, im) of one tile (four per tile); wave w owns bins [32 w, 32 w + 32).  Persistent
// workgroups with the table fragments in registers and the next item's samples in flight, as in synth_mfma_kernel.
template <bool ADJ, int NS>
__global__ __launch_bounds__(256, NS == 2 ? 2 : 1) void analysis_mfma_kernel(const float* __restrict__ wave, float* __restrict__ spec,
                                                                             const __bf16* __restrict__ tab, int nitems) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];                 // analysis_smem(NS) bytes (70 KB with three terms)
  float (*xs)[SPAN + 8] = reinterpret_cast<float (*)[SPAN + 8]>(dsm);        // [2]: this item's samples | the next item's
  __bf16* UV = reinterpret_cast<__bf16*>(dsm + 2 * (SPAN + 8) * 4);          // [u|v][term][32 t][136]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  TabFrags<NS> tb[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) tb[ks] = load_tab<NS>(tab, w, ks, lane);
  constexpr int NPV = (SPAN + 255) / 256;                      // 9 samples per thread
  float pv[NPV];
  auto fetch = [&](int item) {
    const int b = item >> 2, t0 = FT * (item & 3);
    const float* src = wave + (int64_t)b * WLEN;
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
      const int s = tid + 256 * u;
      const int q = HOP * t0 + (s < SPAN ? s : SPAN - 1);       // padded / overlap-add coordinate
      int i = q - PADL;
      float val;
      if (ADJ) {
        val = (i >= 0 && i < WLEN) ? src[i] / (float)frames_covering(q) : 0.f;
      } else {
        if (i < 0) i = -i;
        if (i >= WLEN) i = 2 * (WLEN - 1) - i;
        val = src[i];
      }
      pv[u] = val;
    }
  };
  auto stage = [&](float* buf) {
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
      const int s = tid + 256 * u;
      if (s < SPAN) buf[s] = pv[u];
    }
  };
  int item = blockIdx.x;
  if (item >= nitems) return;
  fetch(item);
  stage(xs[0]);
  int cur = 0;
  // (the NEXT item's samples are staged before this item's 32 stores per lane are issued: see synth_mfma_kernel)
  for (; item < nitems; item += gridDim.x, cur ^= 1) {
    const int b = item >> 2, t0 = FT * (item & 3);
    const bool more = item + (int)gridDim.x < nitems;
    __syncthreads();                                          // the samples are staged; the previous item's folded frames consumed
    if (more) fetch(item + gridDim.x);
    // folded frames u[n] = x[n] + x[255 - n], v[n] = x[n] - x[255 - n] (u[0] = x[0], v[0] = 0), split, [t][n] images.
    // lane -> frame (stride 63 words between lanes: conflict-free reads), wave / half-wave -> two chunks of 8 samples.
    // NOTE (measured, MI355X / ROCm 7.2): with SLP vectorisation on, the folds below compile to v_pk_add_f32 with a
    // half-crossing operand selector, which returns wrong values in lanes 16-31 / 48-63 (scripts/experiments/
    // fold_hazard).  The library is built with -fno-slp-vectorize and scripts/check_isa.py rejects the form;
    // tests/test_kernels_gpu.py::test_stft_round_trip_is_deterministic_under_load holds it.
    {
      const int t = tid & 31, c0 = tid >> 5;
      const float* fr = xs[cur] + HOP * t;
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
        bf16x8 ut[NS], vt[NS];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool dc = n0 + j == 0;
          __bf16 t_[NS];
          split_terms<NS>(dc ? a[it][j] : a[it][j] + m[it][j], t_);
#pragma unroll
          for (int z = 0; z < NS; ++z) ut[z][j] = t_[z];
          split_terms<NS>(dc ? 0.f : a[it][j] - m[it][j], t_);
#pragma unroll
          for (int z = 0; z < NS; ++z) vt[z][j] = t_[z];
        }
        __bf16* d = UV + t * LDU + n0;
#pragma unroll
        for (int z = 0; z < NS; ++z) {
          *reinterpret_cast<bf16x8*>(d + z * FT * LDU) = ut[z];
          *reinterpret_cast<bf16x8*>(d + (NS + z) * FT * LDU) = vt[z];
        }
      }
    }
    __syncthreads();
    f32x16 RE, IM;
#pragma unroll
    for (int e = 0; e < 16; ++e) { RE[e] = 0.f; IM[e] = 0.f; }
    const int r = lane & 31, h = lane >> 5;
    const __bf16* brow = UV + r * LDU + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      bf16x8 ut[NS], vt[NS];
#pragma unroll
      for (int z = 0; z < NS; ++z) {
        ut[z] = *reinterpret_cast<const bf16x8*>(brow + 16 * ks + z * FT * LDU);
        vt[z] = *reinterpret_cast<const bf16x8*>(brow + 16 * ks + (NS + z) * FT * LDU);
      }
      mma_split<NS>(RE, tb[ks].c, ut);
      mma_split<NS>(IM, tb[ks].s, vt);
    }
    if (more) stage(xs[cur ^ 1]);                             // (the other buffer: its reader, the previous fold, is long done)
    float* dst = spec + (int64_t)b * 2 * NBIN * NFRAME + t0 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = 32 * w + mma32_row(e, h);
      const float wk = ADJ ? (k == 0 ? 1.0f : 2.0f) / (float)NFFT : 1.0f;
      __builtin_nontemporal_store(RE[e] * wk, dst + k * NFRAME);
      __builtin_nontemporal_store((ADJ && k == 0) ? 0.f : -IM[e] * wk, dst + (NBIN + k) * NFRAME);
    }
  }
}

// persistent grids: two 256-thread workgroups per CU (tables in registers: 2 waves per SIMD; one with three terms), a multiple of
// 8 (XCD affinity)
inline int persistent_grid(int nitems, int terms) {
  const int cap = terms == 2 ? 512 : 256;
  int g = nitems < cap ? nitems : cap;
  return (g + 7) / 8 * 8;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int analysis_smem(int NS) { return 2 * (SPAN + 8) * 4 + 2 * NS * FT * LDU * 2; }

template <bool ADJ, int NS>
int launch_synth(const float* in, float* out, const void* tables, int B, hipStream_t st) {
  const int nitems = 8 * ((B + 7) / 8) * SYN_GROUPS;
  hipLaunchKernelGGL((synth_mfma_kernel<ADJ, NS>), dim3(persistent_grid(nitems, NS)), dim3(256), 0, st, in, out,
                     static_cast<const __bf16*>(tables), B, nitems);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
template <bool ADJ, int NS>
int launch_analysis(const float* in, float* out, const void* tables, int B, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(analysis_mfma_kernel<ADJ, NS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              analysis_smem(NS));
    attr = true;
  }
  const int nitems = (NFRAME / FT) * B;
  hipLaunchKernelGGL((analysis_mfma_kernel<ADJ, NS>), dim3(persistent_grid(nitems, NS)), dim3(256), analysis_smem(NS), st, in, out,
                     static_cast<const __bf16*>(tables), nitems);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

}  // namespace

// terms: 2 (hi + lo operands: the bf16 compute mode) or 3 (hi + mid + lo: fp32-grade, the fp32 / split modes); the tables buffer
// must have been filled with the same value
extern "C" int64_t wm_stft255_table_bytes(int32_t terms) { return terms == 2 || terms == 3 ? (int64_t)tab_elems(terms) * 2 : 0; }

extern "C" int wm_stft255_tables(void* tables, int32_t terms, void* stream) {
  if (!tables || !aligned16(tables) || (terms != 2 && terms != 3)) return WM_EINVAL;
  if (terms == 2)
    hipLaunchKernelGGL(stft_table_kernel<2>, dim3(TAB_PLANE / 256), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<__bf16*>(tables));
  else
    hipLaunchKernelGGL(stft_table_kernel<3>, dim3(TAB_PLANE / 256), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<__bf16*>(tables));
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_istft255_fwd(const float* spec, float* wave, const void* tables, int32_t terms, int32_t B, void* stream) {
  if (!spec || !wave || !tables || B <= 0 || !aligned16(spec) || !aligned16(tables) || (terms != 2 && terms != 3)) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return terms == 2 ? launch_synth<false, 2>(spec, wave, tables, B, st) : launch_synth<false, 3>(spec, wave, tables, B, st);
}
extern "C" int wm_stft255_bwd(const float* dspec, float* dwave, const void* tables, int32_t terms, int32_t B, void* stream) {
  if (!dspec || !dwave || !tables || B <= 0 || !aligned16(dspec) || !aligned16(tables) || (terms != 2 && terms != 3)) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return terms == 2 ? launch_synth<true, 2>(dspec, dwave, tables, B, st) : launch_synth<true, 3>(dspec, dwave, tables, B, st);
}
extern "C" int wm_stft255_fwd(const float* wave, float* spec, const void* tables, int32_t terms, int32_t B, void* stream) {
  if (!wave || !spec || !tables || B <= 0 || !aligned16(tables) || (terms != 2 && terms != 3)) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return terms == 2 ? launch_analysis<false, 2>(wave, spec, tables, B, st) : launch_analysis<false, 3>(wave, spec, tables, B, st);
}
extern "C" int wm_istft255_bwd(const float* dwave, float* dspec, const void* tables, int32_t terms, int32_t B, void* stream) {
  if (!dwave || !dspec || !tables || B <= 0 || !aligned16(tables) || (terms != 2 && terms != 3)) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return terms == 2 ? launch_analysis<true, 2>(dwave, dspec, tables, B, st) : launch_analysis<true, 3>(dwave, dspec, tables, B, st);
}
