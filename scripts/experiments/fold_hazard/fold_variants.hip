// Discriminating experiment for the wrong-sum hazard of round 2 (DESIGN.md "A wrong-result hazard under co-residency").
//
// The failing build of analysis_mfma_kernel (csrc/stft_mfma.hip, SLP vectorisation on) differs from the passing one
// (-fno-slp-vectorize) in TWO ways in its fold region (ISA diff, profiles/r03_fold_hazard_isa.md):
//   P  the folds are v_pk_add_f32 ... op_sel on register pairs delivered by ds_read2_b32;
//   W  the ds_read2_b32 of chunk 1 are issued right behind the four ds_write_b128 of chunk 0 and RETURN INTO THE DATA
//      REGISTERS OF THOSE STORES (v[4:5], v[6:7], v[10:11], v[14:15], v[18:19]); the passing build issues every read
//      of both chunks before the first store.
// This file holds the fold + MFMA + store of that kernel (forward transform only) in three source orders so that the two
// factors can be separated; it is compiled twice (with and without -fno-slp-vectorize) by run.sh:
//   VAR 0  source as in round 2 (reads of both chunks, then fold + store per chunk; the compiler is free to sink reads)
//   VAR 1  a scheduling barrier after the reads: every LDS read is issued (and waited for) before the first store
//   VAR 2  chunk by chunk: read, fold, store chunk 0, then read, fold, store chunk 1 (register reuse invited)
// Second pass (what in P is at fault), same skeleton:
//   VAR 3  VAR 0 with `s_waitcnt lgkmcnt(0)` + 32 idle issue slots between the last LDS read and the first fold (asm with a
//          memory clobber: no read can sink below it): are the packed adds wrong even when every LDS return is long complete?
//   VAR 4  the sum / difference pairs written by hand as v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0] (the half swap the SLP
//          vectoriser emits for the reversed m pair), everything else scalar
//   VAR 5  as VAR 4 with the m pair swapped by two moves first and a plain v_pk_add_f32 (no op_sel)
// Test infrastructure only; nothing here is linked into libwm_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

namespace {
constexpr int NFFT = 255, HOP = 63, NBIN = 128, NFRAME = 128, WLEN = 8002, PADL = 127;
constexpr int FT = 32, SPAN = HOP * (FT - 1) + NFFT, LDU = 136;

__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
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
__device__ __forceinline__ void mma3(f32x16& acc, const bf16x8& th, const bf16x8& tl, const bf16x8& xh, const bf16x8& xl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th, xh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(th, xl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tl, xh, acc, 0, 0, 0);
}
__device__ __forceinline__ int mma32_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ void fold_store(const float (&a)[8], const float (&m)[8], int n0, __bf16* d) {
  bf16x8 uh, ul, vh, vl;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool dc = n0 + j == 0;
    __bf16 h_, l_;
    split_bf16(dc ? a[j] : a[j] + m[j], h_, l_); uh[j] = h_; ul[j] = l_;
    split_bf16(dc ? 0.f : a[j] - m[j], h_, l_); vh[j] = h_; vl[j] = l_;
  }
  *reinterpret_cast<bf16x8*>(d) = uh;
  *reinterpret_cast<bf16x8*>(d + FT * LDU) = ul;
  *reinterpret_cast<bf16x8*>(d + 2 * FT * LDU) = vh;
  *reinterpret_cast<bf16x8*>(d + 3 * FT * LDU) = vl;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool OPSEL>
__device__ __forceinline__ void fold_store_pk(const float (&a)[8], const float (&m)[8], int n0, __bf16* d) {
  // m[j] = fr[255 - n0 - j]: the pair (m[j + 1], m[j]) lies at ascending addresses, as ds_read2_b32 delivers it
  bf16x8 uh, ul, vh, vl;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    f32x2 ap = {a[j], a[j + 1]}, mr = {m[j + 1], m[j]}, su, di;
    if constexpr (OPSEL) {
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(su) : "v"(ap), "v"(mr));
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(di) : "v"(ap), "v"(mr));
    } else {
      f32x2 mm = {mr[1], mr[0]};
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(su) : "v"(ap), "v"(mm));
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(di) : "v"(ap), "v"(mm));
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool dc = n0 + j + u == 0;
      __bf16 h_, l_;
      split_bf16(dc ? a[j + u] : su[u], h_, l_); uh[j + u] = h_; ul[j + u] = l_;
      split_bf16(dc ? 0.f : di[u], h_, l_); vh[j + u] = h_; vl[j + u] = l_;
    }
  }
  *reinterpret_cast<bf16x8*>(d) = uh;
  *reinterpret_cast<bf16x8*>(d + FT * LDU) = ul;
  *reinterpret_cast<bf16x8*>(d + 2 * FT * LDU) = vh;
  *reinterpret_cast<bf16x8*>(d + 3 * FT * LDU) = vl;
}

template <int VAR>
__global__ __launch_bounds__(256) void analysis_var_kernel(const float* __restrict__ wave, float* __restrict__ spec,
                                                           const __bf16* __restrict__ tab) {
  __shared__ __attribute__((aligned(16))) float xs[SPAN + 8];
  __shared__ __attribute__((aligned(16))) __bf16 UV[4 * FT * LDU];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.y, t0 = FT * blockIdx.x;
  const float* src = wave + (int64_t)b * WLEN;
  for (int s = tid; s < SPAN; s += 256) {
    int i = HOP * t0 + s - PADL;
    if (i < 0) i = -i;
    if (i >= WLEN) i = 2 * (WLEN - 1) - i;
    xs[s] = src[i];
  }
  __syncthreads();
  {
    const int t = tid & 31, c0 = tid >> 5;
    const float* fr = xs + HOP * t;
    if constexpr (VAR == 2) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        float a[8], m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int n = 8 * (c0 + 8 * it) + j;
          a[j] = fr[n];
          m[j] = fr[n ? NFFT - n : 0];
        }
        fold_store(a, m, 8 * (c0 + 8 * it), UV + t * LDU + 8 * (c0 + 8 * it));
      }
    } else {
      float a[2][8], m[2][8];
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int n = 8 * (c0 + 8 * it) + j;
          a[it][j] = fr[n];
          m[it][j] = fr[n ? NFFT - n : 0];
        }
      if constexpr (VAR == 1) {
        // all 32 values in registers before anything is folded or stored: nothing may move across this point
        __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (VAR == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        if constexpr (VAR == 4) fold_store_pk<true>(a[it], m[it], 8 * (c0 + 8 * it), UV + t * LDU + 8 * (c0 + 8 * it));
        else if constexpr (VAR == 5) fold_store_pk<false>(a[it], m[it], 8 * (c0 + 8 * it), UV + t * LDU + 8 * (c0 + 8 * it));
        else fold_store(a[it], m[it], 8 * (c0 + 8 * it), UV + t * LDU + 8 * (c0 + 8 * it));
      }
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
    dst[k * NFRAME] = RE[e];
    dst[(NBIN + k) * NFRAME] = -IM[e];
  }
}
}  // namespace

extern "C" int fold_run(int var, const float* wave, float* spec, const void* tables, int B, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const __bf16* tab = static_cast<const __bf16*>(tables);
  const dim3 grid(NFRAME / FT, B), blk(256);
  if (var == 0) hipLaunchKernelGGL(analysis_var_kernel<0>, grid, blk, 0, st, wave, spec, tab);
  else if (var == 1) hipLaunchKernelGGL(analysis_var_kernel<1>, grid, blk, 0, st, wave, spec, tab);
  else if (var == 2) hipLaunchKernelGGL(analysis_var_kernel<2>, grid, blk, 0, st, wave, spec, tab);
  else if (var == 3) hipLaunchKernelGGL(analysis_var_kernel<3>, grid, blk, 0, st, wave, spec, tab);
  else if (var == 4) hipLaunchKernelGGL(analysis_var_kernel<4>, grid, blk, 0, st, wave, spec, tab);
  else if (var == 5) hipLaunchKernelGGL(analysis_var_kernel<5>, grid, blk, 0, st, wave, spec, tab);
  else return 22;
  return (int)hipGetLastError();
}
