// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the watermark train step.
// Wave = 64 lanes everywhere; MFMA tiles are 32x32 (f32 accumulators, 16 regs per lane).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

// which GEMM kernel family the last wm_gemm call of this host thread launched (profiling only: wm_last_gemm_family())
enum { WM_FAM_GENERIC = 0, WM_FAM_NT = 1, WM_FAM_TN = 2, WM_FAM_NT_PP = 3, WM_FAM_TN_PP = 4, WM_FAM_NT_BIG = 5, WM_FAM_TN_BIG = 6 };
extern thread_local int wm_gemm_family_tls;

#define WM_OK 0
#define WM_EINVAL 22

#define WM_CHECK_LAUNCH()                          \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return 1000 + (int)e__; \
  } while (0)

// ---- activations -------------------------------------------------------------------------
// nn.GELU() is the erf form.  libm's erff costs several hundred (divergent) instructions per element and was the
// single largest VALU cost of the step (GELU sits in GEMM prologues/epilogues and in both depthwise-conv kernels),
// so erf is evaluated with Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, two orders below the 1e-4 parity bound):
//   erf(z) = sign(z) * (1 - (a1 t + ... + a5 t^5) exp(-z^2)),  t = 1 / (1 + p |z|)
// With z = x/sqrt(2) the exponential exp(-x^2/2) is shared by gelu'(x) = Phi(x) + x phi(x).
__device__ __forceinline__ float phi_tail_f(float x, float e) {          // 0.5 * erfc(|x|/sqrt2), e = exp(-x^2/2)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, fabsf(x), 1.0f));   // v_rcp_f32 (1 ulp), not the IEEE division sequence
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  return 0.5f * p * t * e;
}
__device__ __forceinline__ float gelu_f(float x) {          // x * Phi(x)
  const float e = __expf(-0.5f * x * x);
  const float q = phi_tail_f(x, e);                          // Phi(-|x|)
  return x * (x >= 0.f ? 1.0f - q : q);
}
__device__ __forceinline__ float gelu_grad_f(float x) {     // Phi(x) + x * phi(x)
  const float e = __expf(-0.5f * x * x);
  const float q = phi_tail_f(x, e);
  return (x >= 0.f ? 1.0f - q : q) + x * 0.39894228040143268f * e;
}

// ---- LDS element types: float (exact "parity" mode) or bf16 (fast mode) --------------------
template <typename LT> struct LdsT;
template <> struct LdsT<float> {
  static constexpr int PAD = 4;                       // row stride 36 floats = 144 B (16-B aligned)
  __device__ static __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  }
  __device__ static __forceinline__ void store1(float* p, float a) { *p = a; }
};
template <> struct LdsT<__bf16> {
  static constexpr int PAD = 8;                       // row stride 40 bf16 = 80 B (16-B aligned)
  __device__ static __forceinline__ void store4(__bf16* p, float a, float b, float c, float d) {
    bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
    *reinterpret_cast<bf16x4*>(p) = v;
  }
  __device__ static __forceinline__ void store1(__bf16* p, float a) { *p = (__bf16)a; }
};

// ---- HBM storage types: float, or bf16 for the wide activations in bf16 mode -------------------------
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) {
  bf16x4 t = {(__bf16)a, (__bf16)b, (__bf16)0.f, (__bf16)0.f};
  return reinterpret_cast<const uint2&>(t).x;
}
// Streaming stores (the `nt` bit): activations and gradients written by one kernel are read back by a later launch, long
// after they left the caches (hundreds of MB per tensor at bench batches); marking the store non-temporal keeps the lines
// from displacing the operands that ARE re-read (weights, halo rows) on their way to HBM.
typedef __attribute__((ext_vector_type(2))) unsigned int nt_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int nt_u32x4;
typedef __attribute__((ext_vector_type(4))) float nt_f32x4;
__device__ __forceinline__ void stream_store(float* p, float4 v) {
  __builtin_nontemporal_store((nt_f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f32x4*>(p));
}
__device__ __forceinline__ void stream_store(__bf16* p, uint2 v) {
  __builtin_nontemporal_store((nt_u32x2){v.x, v.y}, reinterpret_cast<nt_u32x2*>(p));
}
__device__ __forceinline__ void stream_store(__bf16* p, uint4 v) {
  __builtin_nontemporal_store((nt_u32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<nt_u32x4*>(p));
}
__device__ __forceinline__ void stream_store(__bf16* p, uint32_t v) { __builtin_nontemporal_store(v, reinterpret_cast<uint32_t*>(p)); }
__device__ __forceinline__ void stream_store(float* p, float2 v) {
  typedef __attribute__((ext_vector_type(2))) float nt_f32x2;
  __builtin_nontemporal_store((nt_f32x2){v.x, v.y}, reinterpret_cast<nt_f32x2*>(p));
}
template <typename T> struct IO;
template <> struct IO<float> {
  static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void st4(float* p, float4 v) { stream_store(p, v); }
  static __device__ __forceinline__ float ld1(const float* p) { return *p; }
  static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct IO<__bf16> {
  static __device__ __forceinline__ float4 ld4(const __bf16* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y));
  }
  static __device__ __forceinline__ void st4(__bf16* p, float4 v) {
    stream_store(p, make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)));
  }
  static __device__ __forceinline__ float ld1(const __bf16* p) { return (float)*p; }
  static __device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }
};

// acc(32x32) += A(32 x kcount) * B(kcount x 32).
// a_row / b_row point at THIS LANE's row (tile_row0 + (lane & 31)) of a K-contiguous LDS image;
// B is held "n-major" (Bs[n][k]) so both operands are read the same way.
// f32: v_mfma_f32_32x32x2_f32, lane half h supplies k = 8s + 4h + j (any k permutation shared by
//      A and B is a valid summation order).  bf16: v_mfma_f32_32x32x16_bf16, lane supplies
//      k = 16s + 8h .. +7.
template <typename LT>
__device__ __forceinline__ void mma32(f32x16& acc, const LT* a_row, const LT* b_row, int kcount, int h);

template <>
__device__ __forceinline__ void mma32<float>(f32x16& acc, const float* a_row, const float* b_row,
                                             int kcount, int h) {
#pragma unroll 4
  for (int k = 0; k < kcount; k += 8) {
    const float4 a = *reinterpret_cast<const float4*>(a_row + k + 4 * h);
    const float4 b = *reinterpret_cast<const float4*>(b_row + k + 4 * h);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
}
template <>
__device__ __forceinline__ void mma32<__bf16>(f32x16& acc, const __bf16* a_row, const __bf16* b_row,
                                              int kcount, int h) {
#pragma unroll 4
  for (int k = 0; k < kcount; k += 16) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_row + k + 8 * h);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(b_row + k + 8 * h);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
}
// C/D layout of every 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
__device__ __forceinline__ int mma32_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Counter-based RNG (Philox4x32-10) so that random draws are a pure function of
// (seed, step, stream, index): replayable inside hipGraphs and identical across ranks/launch shapes.
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(uint32_t x) { return (x >> 8) * (1.0f / 16777216.0f) + (0.5f / 16777216.0f); }
