// Epilogues shared by the two schedules of the 256 x 256 tile GEMMs (gemm_big.hip: two barriers per 64-deep k tile;
// gemm_pp.hip: four 32-deep stages, the two wave groups half a slot apart).  A wave (wm, wn) of the 2 x 4 arrangement holds
// rows [128 wm, +128) x columns [64 wn, +64) of the tile as acc[4][2] (32 x 32 accumulators of v_mfma_f32_32x32x16_bf16).
#pragma once
#include "common.h"
#include "../../include/wm_kernels.h"

namespace bigtile {

constexpr int LDC = 256 + 4;                        // fp32 staging row of the epilogue

//   EPI 0: C(bf16) = (acc + bias) * sc          EPI 1: C(fp32) = (acc + bias) * sc + res          EPI 2: C(bf16) = acc * sc * gaux(bf16)
//   EPI 4: EPI 1 + LayerNorm of the finished rows (N == 256: the 64 threads = one wave of a staging row hold all of it), as in
//          gemm_fast.hip (host: M % 256 == 0, so no row of a tile is outside the matrix)
// smem: at least 64 * LDC floats, free of other use (every wave has passed a barrier after its last operand read)
template <int EPI>
__device__ __forceinline__ void nt_epilogue(f32x16 (&acc)[4][2], const WmGemm& g, int m0, int n0, char* smem, int tid, int wm, int wn,
                                            int r, int h) {
  // ---- epilogue: four passes of 64 rows (tile row i of every wave) through a [64][260] fp32 image ----
  float* Cs = reinterpret_cast<float*>(smem);
  const int c4 = tid & 63, rg = tid >> 6, n = n0 + 4 * c4;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.bias) bias4 = *reinterpret_cast<const float4*>(g.bias + n);
#pragma unroll                                                  // (a run-time index into acc[][] would put it in scratch memory)
  for (int i = 0; i < 4; ++i) {
    if (i) __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        Cs[(wm * 32 + mma32_row(e, h)) * LDC + wn * 64 + j * 32 + r] = acc[i][j][e];
    __syncthreads();
    // rows rg + 8 q of the image; q < 4: the wm = 0 half (global rows m0 + 32 i + ..), q >= 4: the wm = 1 half (+ 128)
    int64_t grow[8];
    float sc[2] = {1.f, 1.f};
#pragma unroll
    for (int q = 0; q < 8; ++q) grow[q] = m0 + (q >> 2) * 128 + 32 * i + ((rg + 8 * q) & 31);
    const bool ok0 = m0 + 32 * i < g.M, ok1 = m0 + 128 + 32 * i < g.M;     // M % 32 == 0 (host): a 32-row block is in or out
    if (g.rowscale) {
      if (ok0) sc[0] = g.rowscale[(m0 + 32 * i) / g.rows_per_scale];
      if (ok1) sc[1] = g.rowscale[(m0 + 128 + 32 * i) / g.rows_per_scale];
    }
    uint2 ga[8];
    float4 re[8];
    if constexpr (EPI == 2) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q < 4 ? ok0 : ok1) ga[q] = *reinterpret_cast<const uint2*>(static_cast<const __bf16*>(g.gaux) + grow[q] * g.ldg + n);
    }
    if constexpr (EPI == 1 || EPI == 4) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q < 4 ? ok0 : ok1) re[q] = *reinterpret_cast<const float4*>(g.res + grow[q] * g.ldr + n);
    }
    if constexpr (EPI == 4) {
      const float4 gm = *reinterpret_cast<const float4*>(g.ln_gamma + n), bt = *reinterpret_cast<const float4*>(g.ln_beta + n);
      float4 v[8];
      float s1[8], s2[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float s = sc[q >> 2];
        float4 x = *reinterpret_cast<const float4*>(Cs + (rg + 8 * q) * LDC + 4 * c4);
        x.x = (x.x + bias4.x) * s + re[q].x; x.y = (x.y + bias4.y) * s + re[q].y;
        x.z = (x.z + bias4.z) * s + re[q].z; x.w = (x.w + bias4.w) * s + re[q].w;
        stream_store(static_cast<float*>(g.C) + grow[q] * g.ldc + n, x);
        v[q] = x;
        s1[q] = (x.x + x.y) + (x.z + x.w);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int q = 0; q < 8; ++q) s1[q] += __shfl_xor(s1[q], o);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float mu = s1[q] * (1.0f / 256.0f);
        s1[q] = mu;
        v[q].x -= mu; v[q].y -= mu; v[q].z -= mu; v[q].w -= mu;
        s2[q] = (v[q].x * v[q].x + v[q].y * v[q].y) + (v[q].z * v[q].z + v[q].w * v[q].w);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int q = 0; q < 8; ++q) s2[q] += __shfl_xor(s2[q], o);
      const int64_t img = (int64_t)g.ln_hw * g.ln_hw;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float rs = 1.0f / sqrtf(s2[q] * (1.0f / 256.0f) + 1e-5f);
        float4 o = make_float4(v[q].x * rs * gm.x + bt.x, v[q].y * rs * gm.y + bt.y, v[q].z * rs * gm.z + bt.z, v[q].w * rs * gm.w + bt.w);
        if (g.ln_mod) {
          const int t = (int)(grow[q] % img), yy = t / g.ln_hw, xx = t - yy * g.ln_hw;
          const int pi = (yy - g.ln_shift + g.ln_hw) % g.ln_hw, pj = (xx - g.ln_shift + g.ln_hw) % g.ln_hw;
          const float4 mm = *reinterpret_cast<const float4*>(g.ln_mod + (int64_t)((pi & 7) * 8 + (pj & 7)) * 256 + n);
          o.x += mm.x; o.y += mm.y; o.z += mm.z; o.w += mm.w;
        }
        stream_store(static_cast<__bf16*>(g.ln_out) + grow[q] * 256 + n, make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w)));
        if (c4 == 0) { g.ln_mean[grow[q]] = s1[q]; g.ln_rstd[grow[q]] = rs; }
      }
      continue;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (!(q < 4 ? ok0 : ok1)) continue;
      const float s = sc[q >> 2];
      float4 v = *reinterpret_cast<const float4*>(Cs + (rg + 8 * q) * LDC + 4 * c4);
      if constexpr (EPI == 1) {
        v.x = (v.x + bias4.x) * s + re[q].x; v.y = (v.y + bias4.y) * s + re[q].y;
        v.z = (v.z + bias4.z) * s + re[q].z; v.w = (v.w + bias4.w) * s + re[q].w;
        stream_store(static_cast<float*>(g.C) + grow[q] * g.ldc + n, v);
      } else {
        if constexpr (EPI == 0) { v.x = (v.x + bias4.x) * s; v.y = (v.y + bias4.y) * s; v.z = (v.z + bias4.z) * s; v.w = (v.w + bias4.w) * s; }
        else { v.x *= s * bf_lo(ga[q].x); v.y *= s * bf_hi(ga[q].x); v.z *= s * bf_lo(ga[q].y); v.w *= s * bf_hi(ga[q].y); }
        stream_store(static_cast<__bf16*>(g.C) + grow[q] * g.ldc + n, make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)));
      }
    }
  }
}

// split-K slice (or, for a single k range, the running sum in C) of the weight-gradient kernels: plain stores, 128-byte row
// segments per half wave
__device__ __forceinline__ void tn_store(f32x16 (&acc)[4][2], const WmGemm& g, float* part, int bz, int m0, int n0, int wm, int wn,
                                         int r, int h) {
  float* const dst = part ? part + (int64_t)bz * g.M * g.N : static_cast<float*>(g.C);
  const int64_t ldd = part ? g.N : g.ldc;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float* d = dst + (int64_t)(m0 + wm * 128 + i * 32 + 4 * h) * ldd + n0 + wn * 64 + j * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (part) *d = acc[i][j][e]; else *d += acc[i][j][e];
        d += ((e & 3) == 3) ? 5 * ldd : ldd;                     // rows (e & 3) + 8 (e >> 2) + 4 h
      }
    }
}

}  // namespace bigtile
