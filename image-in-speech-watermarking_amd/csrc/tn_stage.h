// Shared by gemm_tn.hip (weight gradients) and gemm.hip (dX: row-contiguous weight operand): tiles whose HBM layout is
// [k][columns] are copied to LDS as they lie and read back through ds_read_b64_tr_b16 as MFMA 32x32x16 fragments.
#pragma once
#include "common.h"

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int ld, int col0, int kstep, int lane) {
  // MFMA 32x32x16 operand for 32 columns starting at col0, k = 16*kstep .. +15 (lane l: column l&31, k = 8*(l>>5) + j)
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const __bf16* a = tile + (16 * kstep + 8 * (g >> 1) + q) * ld + col0 + 16 * (g & 1) + 4 * p;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * ld));
  bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return f;
}

// stage a [KT][R] tile (row-contiguous source) into LDS [KT][R + 32] bf16
template <typename ST, int R, int KT> struct TnStage;
template <int R, int KT> struct TnStage<__bf16, R, KT> {
  static constexpr int CPR = R / 8, NCH = KT * CPR, PT = (NCH + 255) / 256;     // 16-byte chunks
  uint4 v[PT];
  __device__ __forceinline__ void load(const void* p, int64_t ld, int c0, int cmax, int k0, int kend, int tid) {
    const __bf16* src = static_cast<const __bf16*>(p);
#pragma unroll
    for (int q = 0; q < PT; ++q) {
      const int ch = tid + 256 * q, kr = ch / CPR, c = c0 + 8 * (ch % CPR);
      uint4 x = make_uint4(0, 0, 0, 0);
      if (ch < NCH && k0 + kr < kend) {
        const __bf16* s = src + (int64_t)(k0 + kr) * ld + c;
        if (c + 7 < cmax) {
          x = *reinterpret_cast<const uint4*>(s);
        } else {
          uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (c + j < cmax) w[j >> 1] |= (uint32_t)reinterpret_cast<const uint16_t*>(s)[j] << (16 * (j & 1));
          x = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
      v[q] = x;
    }
  }
  __device__ __forceinline__ void store(__bf16* t, int op, const float* kscale, int kps, int k0, int tid) const {
    constexpr int LD = R + 32;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
      const int ch = tid + 256 * q, kr = ch / CPR, cl = 8 * (ch % CPR);
      if (ch < NCH) {
        uint4 x = v[q];
        if (op || kscale) {
          const float sc = kscale ? kscale[(k0 + kr) / kps] : 1.f;
          uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float a = bf_lo(w[j]), b = bf_hi(w[j]);
            if (op) { a = gelu_f(a); b = gelu_f(b); }
            w[j] = pack_bf2(a * sc, b * sc);
          }
          x = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *reinterpret_cast<uint4*>(t + kr * LD + cl) = x;
      }
    }
  }
};
template <int R, int KT> struct TnStage<float, R, KT> {
  static constexpr int CPR = R / 4, NCH = KT * CPR, PT = (NCH + 255) / 256;     // float4 chunks
  float4 v[PT];
  __device__ __forceinline__ void load(const void* p, int64_t ld, int c0, int cmax, int k0, int kend, int tid) {
    const float* src = static_cast<const float*>(p);
#pragma unroll
    for (int q = 0; q < PT; ++q) {
      const int ch = tid + 256 * q, kr = ch / CPR, c = c0 + 4 * (ch % CPR);
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ch < NCH && k0 + kr < kend) {
        const float* s = src + (int64_t)(k0 + kr) * ld + c;
        if (c + 3 < cmax) {
          x = *reinterpret_cast<const float4*>(s);
        } else {
          float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c + j < cmax) e[j] = s[j];
          x = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
      v[q] = x;
    }
  }
  __device__ __forceinline__ void store(__bf16* t, int op, const float* kscale, int kps, int k0, int tid) const {
    constexpr int LD = R + 32;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
      const int ch = tid + 256 * q, kr = ch / CPR, cl = 4 * (ch % CPR);
      if (ch < NCH) {
        float4 x = v[q];
        if (op) { x.x = gelu_f(x.x); x.y = gelu_f(x.y); x.z = gelu_f(x.z); x.w = gelu_f(x.w); }
        const float sc = kscale ? kscale[(k0 + kr) / kps] : 1.f;
        *reinterpret_cast<uint2*>(t + kr * LD + cl) = make_uint2(pack_bf2(x.x * sc, x.y * sc), pack_bf2(x.z * sc, x.w * sc));
      }
    }
  }
};

