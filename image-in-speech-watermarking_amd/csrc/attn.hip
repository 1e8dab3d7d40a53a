// 8x8-window multi-head self-attention (head dim 32) on token-major [B*hw*hw, 3C] q|k|v.
// Reference: WindowAttention.forward (model.py:523-551) together with the cyclic shift, window
// partition/reverse (model.py:987-1013) and the shifted-window mask (model.py:954-972), all of
// which are folded into index arithmetic here: nothing is rolled or re-laid-out in HBM.
//
// One wave (= one 64-thread workgroup) owns one (window, head): 64 queries x 64 keys x d=32.
// QK^T, PV and the five backward products run on 32x32 MFMA tiles out of LDS; softmax is
// row-per-lane (64 lanes <-> 64 query rows).
#include <cstdlib>
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

constexpr int LDS_S = 68;   // fp32 score rows (16-B aligned so the fp32 image can feed MFMA A-operands)

template <typename LT> struct AttnDims {
  static constexpr int LDK = 32 + LdsT<LT>::PAD;   // [64 tokens][32 + pad]
  static constexpr int LDT = 64 + LdsT<LT>::PAD;   // [32 or 64][64 + pad]
  static constexpr int SMALL = (64 * LDK > 32 * LDT) ? 64 * LDK : 32 * LDT;   // elements
};

struct WinGeom {
  int hw, shift, nw, C, heads;
  __device__ __forceinline__ int64_t token_row(int b, int wy, int wx, int p) const {
    int yy = wy * 8 + (p >> 3) + shift, xx = wx * 8 + (p & 7) + shift;
    if (yy >= hw) yy -= hw;
    if (xx >= hw) xx -= hw;
    return (int64_t)b * hw * hw + yy * hw + xx;
  }
  __device__ __forceinline__ int region1(int ii) const { return ii < hw - 8 ? 0 : (ii < hw - shift ? 1 : 2); }
  __device__ __forceinline__ int region(int wy, int wx, int p) const {
    return 3 * region1(wy * 8 + (p >> 3)) + region1(wx * 8 + (p & 7));
  }
};

__device__ __forceinline__ int rel_index(int i, int j) {
  return ((i >> 3) - (j >> 3) + 7) * 15 + ((i & 7) - (j & 7) + 7);
}

// dst[tok][0..31] = scale * src[row(tok)][col0 .. col0+31]
template <typename LT, typename ST>
__device__ __forceinline__ void stage_rows(LT* dst, const ST* __restrict__ src, int64_t ld, int col0,
                                           const WinGeom& g, int b, int wy, int wx, float scale, int lane) {
  constexpr int LDK = AttnDims<LT>::LDK;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int idx = lane + 64 * it, tok = idx >> 3, f4 = idx & 7;
    const float4 v = IO<ST>::ld4(src + g.token_row(b, wy, wx, tok) * ld + col0 + 4 * f4);
    LdsT<LT>::store4(dst + tok * LDK + 4 * f4, v.x * scale, v.y * scale, v.z * scale, v.w * scale);
  }
}
// dst[e][tok] = scale * src[row(tok)][col0 + e]
template <typename LT, typename ST>
__device__ __forceinline__ void stage_cols(LT* dst, const ST* __restrict__ src, int64_t ld, int col0,
                                           const WinGeom& g, int b, int wy, int wx, float scale, int lane) {
  constexpr int LDT = AttnDims<LT>::LDT;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int idx = lane + 64 * it, tok = idx & 63, f4 = idx >> 6;
    const float4 v = IO<ST>::ld4(src + g.token_row(b, wy, wx, tok) * ld + col0 + 4 * f4);
    LdsT<LT>::store1(dst + (4 * f4 + 0) * LDT + tok, v.x * scale);
    LdsT<LT>::store1(dst + (4 * f4 + 1) * LDT + tok, v.y * scale);
    LdsT<LT>::store1(dst + (4 * f4 + 2) * LDT + tok, v.z * scale);
    LdsT<LT>::store1(dst + (4 * f4 + 3) * LDT + tok, v.w * scale);
  }
}

// S[i][j] = (Qs K^T)[i][j] + bias(i,j) + mask(i,j), written to the fp32 LDS image.
template <typename LT>
__device__ __forceinline__ void scores_to_lds(float* S, const LT* Qs, const LT* Ks, const float* __restrict__ table,
                                              const WinGeom& g, int wy, int wx, int head, int lane) {
  constexpr int LDK = AttnDims<LT>::LDK;
  const int r = lane & 31, h = lane >> 5;
  const bool masked = g.shift > 0 && (wy == g.nw - 1 || wx == g.nw - 1);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      mma32<LT>(acc, Qs + (ti * 32 + r) * LDK, Ks + (tj * 32 + r) * LDK, 32, h);
      const int j = tj * 32 + r;
      const int rj = masked ? g.region(wy, wx, j) : 0;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = ti * 32 + mma32_row(e, h);
        float v = acc[e] + table[rel_index(i, j) * g.heads + head];
        if (masked && g.region(wy, wx, i) != rj) v += -100.0f;
        S[i * LDS_S + j] = v;
      }
    }
  }
}

// Row-per-lane softmax of the LDS score image into registers.
__device__ __forceinline__ float softmax_row(const float* S, int lane, float (&p)[64]) {
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 64; ++j) { p[j] = S[lane * LDS_S + j]; mx = fmaxf(mx, p[j]); }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 64; ++j) { p[j] = expf(p[j] - mx); sum += p[j]; }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int j = 0; j < 64; ++j) p[j] *= inv;
  return mx + logf(sum);                                      // log-sum-exp of the row (kept for the fast backward)
}

template <typename LT, typename ST>
__global__ __launch_bounds__(64) void attn_fwd_kernel(const ST* __restrict__ qkv, const float* __restrict__ table,
                                                      ST* __restrict__ out, float* __restrict__ lse, WinGeom g, int B) {
  constexpr int LDK = AttnDims<LT>::LDK, LDT = AttnDims<LT>::LDT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LT* Qs = reinterpret_cast<LT*>(smem);                       // [64][LDK]; later P [64][LDT] over Qs+Ks
  LT* Ks = Qs + 64 * LDK;
  LT* Vt = Ks + 64 * LDK;                                     // [32][LDT]
  float* S = reinterpret_cast<float*>(Vt + 32 * LDT);         // [64][LDS_S]
  LT* P = Qs;
  static_assert(64 * LDT <= 2 * 64 * LDK, "P must fit over Q|K");

  const int lane = threadIdx.x;
  const int head = blockIdx.y;
  const int w = blockIdx.x;
  const int b = w / (g.nw * g.nw), wr = w - b * g.nw * g.nw, wy = wr / g.nw, wx = wr - wy * g.nw;
  const int64_t ld = 3 * (int64_t)g.C;
  const float scale = 0.17677669529663687f;                   // 32^-0.5 (model.py:489)

  stage_rows<LT, ST>(Qs, qkv, ld, head * 32, g, b, wy, wx, scale, lane);
  stage_rows<LT, ST>(Ks, qkv, ld, g.C + head * 32, g, b, wy, wx, 1.0f, lane);
  stage_cols<LT, ST>(Vt, qkv, ld, 2 * g.C + head * 32, g, b, wy, wx, 1.0f, lane);
  __syncthreads();
  scores_to_lds<LT>(S, Qs, Ks, table, g, wy, wx, head, lane);
  __syncthreads();
  float p[64];
  const float row_lse = softmax_row(S, lane, p);
  if (lse) lse[g.token_row(b, wy, wx, lane) * g.heads + head] = row_lse;
#pragma unroll
  for (int j = 0; j < 64; j += 4) LdsT<LT>::store4(P + lane * LDT + j, p[j], p[j + 1], p[j + 2], p[j + 3]);
  __syncthreads();
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    mma32<LT>(acc, P + (ti * 32 + r) * LDT, Vt + r * LDT, 64, h);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = ti * 32 + mma32_row(e, h);
      IO<ST>::st1(out + g.token_row(b, wy, wx, i) * g.C + head * 32 + r, acc[e]);
    }
  }
}

template <typename LT, typename ST>
__global__ __launch_bounds__(64) void attn_bwd_kernel(const ST* __restrict__ qkv, const float* __restrict__ table,
                                                      const ST* __restrict__ dout, ST* __restrict__ dqkv,
                                                      float* __restrict__ dtable, WinGeom g, int B) {
  constexpr int LDK = AttnDims<LT>::LDK, LDT = AttnDims<LT>::LDT, SMALL = AttnDims<LT>::SMALL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LT* T0 = reinterpret_cast<LT*>(smem);
  LT* T1 = T0 + SMALL;
  LT* T2 = T1 + SMALL;
  float* S = reinterpret_cast<float*>(T2 + SMALL);            // fp32 [64][LDS_S]; later Pt (LT [64][LDT])
  float* DP = S + 64 * LDS_S;                                 // fp32 [64][LDS_S]; later dS (LT [64][LDT])
  LT* dSt = reinterpret_cast<LT*>(DP + 64 * LDS_S);           // LT [64][LDT]
  LT* Pt = reinterpret_cast<LT*>(S);
  LT* dS = reinterpret_cast<LT*>(DP);
  static_assert(sizeof(LT) * 64 * LDT <= sizeof(float) * 64 * LDS_S, "overlay must fit");

  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y;
  const int64_t ld = 3 * (int64_t)g.C;
  const float scale = 0.17677669529663687f;
  const int nwin = B * g.nw * g.nw;

  float accb[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) accb[j] = 0.f;

  for (int w = blockIdx.x; w < nwin; w += gridDim.x) {
    const int b = w / (g.nw * g.nw), wr = w - b * g.nw * g.nw, wy = wr / g.nw, wx = wr - wy * g.nw;
    __syncthreads();
    // P1: S = softmax input
    stage_rows<LT, ST>(T0, qkv, ld, head * 32, g, b, wy, wx, scale, lane);
    stage_rows<LT, ST>(T1, qkv, ld, g.C + head * 32, g, b, wy, wx, 1.0f, lane);
    __syncthreads();
    scores_to_lds<LT>(S, T0, T1, table, g, wy, wx, head, lane);
    __syncthreads();
    float p[64];
    softmax_row(S, lane, p);
    // P2: dP = dO V^T
    stage_rows<LT, ST>(T0, dout, (int64_t)g.C, head * 32, g, b, wy, wx, 1.0f, lane);
    stage_rows<LT, ST>(T1, qkv, ld, 2 * g.C + head * 32, g, b, wy, wx, 1.0f, lane);
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        mma32<LT>(acc, T0 + (ti * 32 + r) * LDK, T1 + (tj * 32 + r) * LDK, 32, h);
#pragma unroll
        for (int e = 0; e < 16; ++e) DP[(ti * 32 + mma32_row(e, h)) * LDS_S + tj * 32 + r] = acc[e];
      }
    }
    __syncthreads();
    // P3: dS = P * (dP - rowsum(dP * P)); emit dS, dS^T, P^T as MFMA operands
    {
      float dp[64];
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < 64; ++j) { dp[j] = DP[lane * LDS_S + j]; dot += dp[j] * p[j]; }
      __syncthreads();                      // every lane holds its S / dP rows before the overlays are written
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        dp[j] = p[j] * (dp[j] - dot);
        accb[j] += dp[j];
        LdsT<LT>::store1(dSt + j * LDT + lane, dp[j]);
        LdsT<LT>::store1(Pt + j * LDT + lane, p[j]);
      }
#pragma unroll
      for (int j = 0; j < 64; j += 4) LdsT<LT>::store4(dS + lane * LDT + j, dp[j], dp[j + 1], dp[j + 2], dp[j + 3]);
    }
    // P4: dQ = scale * dS K ; dK = dS^T Qs ; dV = P^T dO
    stage_cols<LT, ST>(T0, qkv, ld, g.C + head * 32, g, b, wy, wx, 1.0f, lane);       // K^T  [e][j]
    stage_cols<LT, ST>(T1, qkv, ld, head * 32, g, b, wy, wx, scale, lane);             // Qs^T [e][i]
    stage_cols<LT, ST>(T2, dout, (int64_t)g.C, head * 32, g, b, wy, wx, 1.0f, lane);   // dO^T [e][i]
    __syncthreads();
#pragma unroll
    for (int which = 0; which < 3; ++which) {
      const LT* A = which == 0 ? dS : (which == 1 ? dSt : Pt);
      const LT* Bm = which == 0 ? T0 : (which == 1 ? T1 : T2);
      const float osc = which == 0 ? scale : 1.0f;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        mma32<LT>(acc, A + (ti * 32 + r) * LDT, Bm + r * LDT, 64, h);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int t = ti * 32 + mma32_row(e, h);
          IO<ST>::st1(dqkv + g.token_row(b, wy, wx, t) * ld + which * g.C + head * 32 + r, acc[e] * osc);
        }
      }
    }
  }
  // relative-position-bias gradient: fold 64x64 -> 225 bins in LDS, then one atomic per bin
  __syncthreads();
  float* bins = reinterpret_cast<float*>(T0);
  for (int t = lane; t < 225; t += 64) bins[t] = 0.f;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 64; ++j) atomicAdd(&bins[rel_index(lane, j)], accb[j]);
  __syncthreads();
  for (int t = lane; t < 225; t += 64) atomicAdd(dtable + t * g.heads + head, bins[t]);
}


// ---------------------------------------------------------------------------------------------------------------
// Fast backward (bf16 MFMA mode).  Scores are recomputed with the KEY on the MFMA lane (S = Q K^T, columns = keys), so
//   * P = exp(S - lse[i]) needs no row reduction (the forward saved the log-sum-exp per query row),
//   * rowsum(dP * P) = rowsum(dO * O) = delta[i] is a per-row dot product of two 32-wide rows,
//   * the P and dS accumulator tiles ARE the B operands of dV^T = dO^T P and dK^T = Qs^T dS (accumulator-as-operand,
//     k order permuted: element j of lane half h is row 16s + 8(j>>2) + 4h + (j&3), matched by the transposing LDS
//     read that fetches the A operand), and only dS crosses LDS once, for dQ = dS K.
// No fp32 score images in LDS, no 64-element register rows: ~27 KB of LDS and one wave per (window, head).
__device__ __forceinline__ bf16x8 tr_frag_attn(const __bf16* tile, int ld, int rowbase, int lane, bool perm) {
  // as-it-lies [k rows][32 columns] image -> MFMA operand (lane l: column l&31).  perm == false: k = rowbase + 8h + j;
  // perm == true: k = rowbase + 8(j>>2) + 4h + (j&3)  (the order an accumulator tile presents its rows in)
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, h = g >> 1;
  const int r0 = rowbase + (perm ? 4 * h : 8 * h) + q;
  const __bf16* a = tile + r0 * ld + 16 * (g & 1) + 4 * p;
  typedef __attribute__((address_space(3))) bf16x4 lds4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4*)a);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4*)(a + (perm ? 8 : 4) * ld));
  bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return f;
}

template <typename ST>
__global__ __launch_bounds__(64) void attn_bwd_fast_kernel(const ST* __restrict__ qkv, const float* __restrict__ table,
                                                           const ST* __restrict__ out, const float* __restrict__ lse_g,
                                                           const ST* __restrict__ dout, ST* __restrict__ dqkv,
                                                           float* __restrict__ dtable, WinGeom g, int B) {
  constexpr int LDK = 40, LDD = 72;
  __shared__ __attribute__((aligned(16))) __bf16 Qs[64 * LDK], Ks[64 * LDK], Vs[64 * LDK], Os[64 * LDK];
  __shared__ __attribute__((aligned(16))) __bf16 DS[32 * LDD];
  __shared__ float lse_s[64], delta_s[64], bias_s[232], bins[232];
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y;
  const int64_t ld = 3 * (int64_t)g.C;
  const float scale = 0.17677669529663687f;
  const int nwin = B * g.nw * g.nw;
  for (int t = lane; t < 225; t += 64) { bias_s[t] = table[t * g.heads + head]; bins[t] = 0.f; }
  float accb[2][2][16];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
      for (int e = 0; e < 16; ++e) accb[a][b2][e] = 0.f;

  for (int w = blockIdx.x; w < nwin; w += gridDim.x) {
    const int b = w / (g.nw * g.nw), wr = w - b * g.nw * g.nw, wy = wr / g.nw, wx = wr - wy * g.nw;
    const bool masked = g.shift > 0 && (wy == g.nw - 1 || wx == g.nw - 1);
    __syncthreads();
    stage_rows<__bf16, ST>(Qs, qkv, ld, head * 32, g, b, wy, wx, scale, lane);
    stage_rows<__bf16, ST>(Ks, qkv, ld, g.C + head * 32, g, b, wy, wx, 1.0f, lane);
    stage_rows<__bf16, ST>(Vs, qkv, ld, 2 * g.C + head * 32, g, b, wy, wx, 1.0f, lane);
    stage_rows<__bf16, ST>(Os, dout, (int64_t)g.C, head * 32, g, b, wy, wx, 1.0f, lane);
    {
      const int64_t trow = g.token_row(b, wy, wx, lane);
      const ST* orow = out + trow * g.C + head * 32;
      const ST* drow = dout + trow * g.C + head * 32;
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 32; e += 4) {
        const float4 ov = IO<ST>::ld4(orow + e), dv = IO<ST>::ld4(drow + e);
        d += ov.x * dv.x + ov.y * dv.y + ov.z * dv.z + ov.w * dv.w;
      }
      delta_s[lane] = d;
      lse_s[lane] = lse_g[trow * g.heads + head];
    }
    __syncthreads();
    f32x16 dVt[2], dKt[2];
#pragma unroll
    for (int kj = 0; kj < 2; ++kj)
#pragma unroll
      for (int e = 0; e < 16; ++e) { dVt[kj][e] = 0.f; dKt[kj][e] = 0.f; }

#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      f32x16 sacc[2], pacc[2];
#pragma unroll
      for (int kj = 0; kj < 2; ++kj) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { sacc[kj][e] = 0.f; pacc[kj][e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 aq = *reinterpret_cast<const bf16x8*>(Qs + (32 * qi + r) * LDK + 16 * ks + 8 * h);
          const bf16x8 bk = *reinterpret_cast<const bf16x8*>(Ks + (32 * kj + r) * LDK + 16 * ks + 8 * h);
          sacc[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, bk, sacc[kj], 0, 0, 0);
          const bf16x8 ad = *reinterpret_cast<const bf16x8*>(Os + (32 * qi + r) * LDK + 16 * ks + 8 * h);
          const bf16x8 bv = *reinterpret_cast<const bf16x8*>(Vs + (32 * kj + r) * LDK + 16 * ks + 8 * h);
          pacc[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, bv, pacc[kj], 0, 0, 0);
        }
      }
      // P and dS in place (sacc -> P, pacc -> dS)
#pragma unroll
      for (int kj = 0; kj < 2; ++kj) {
        const int j = 32 * kj + r;
        const int rj = masked ? g.region(wy, wx, j) : 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = 32 * qi + mma32_row(e, h);
          float sv = sacc[kj][e] + bias_s[rel_index(i, j)];
          if (masked && g.region(wy, wx, i) != rj) sv += -100.0f;
          const float pv = __expf(sv - lse_s[i]);
          const float dsv = pv * (pacc[kj][e] - delta_s[i]);
          sacc[kj][e] = pv;
          pacc[kj][e] = dsv;
          accb[qi][kj][e] += dsv;
        }
      }
      // dV^T += dO^T P ; dK^T += Qs^T dS   (A through the permuted transposing read, B = the accumulator tiles)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 adt = tr_frag_attn(Os, LDK, 32 * qi + 16 * ks, lane, true);
        const bf16x8 aqt = tr_frag_attn(Qs, LDK, 32 * qi + 16 * ks, lane, true);
#pragma unroll
        for (int kj = 0; kj < 2; ++kj) {
          bf16x8 bp, bd;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) { bp[jj] = (__bf16)sacc[kj][8 * ks + jj]; bd[jj] = (__bf16)pacc[kj][8 * ks + jj]; }
          dVt[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(adt, bp, dVt[kj], 0, 0, 0);
          dKt[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aqt, bd, dKt[kj], 0, 0, 0);
        }
      }
      // dQ = scale * dS K : dS through LDS (rows = the 32 queries of this tile)
#pragma unroll
      for (int kj = 0; kj < 2; ++kj)
#pragma unroll
        for (int e = 0; e < 16; ++e) DS[mma32_row(e, h) * LDD + 32 * kj + r] = (__bf16)pacc[kj][e];
      __syncthreads();
      f32x16 dq;
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(DS + r * LDD + 16 * ks + 8 * h);
        const bf16x8 bkt = tr_frag_attn(Ks, LDK, 16 * ks, lane, false);
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bkt, dq, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = 32 * qi + mma32_row(e, h);
        IO<ST>::st1(dqkv + g.token_row(b, wy, wx, i) * ld + head * 32 + r, dq[e] * scale);
      }
      __syncthreads();
    }
    // dK, dV: [e][key] accumulators -> LDS [key][e] -> one 64-byte row per lane
    __bf16* OTk = Qs;
    __bf16* OTv = Os;
#pragma unroll
    for (int kj = 0; kj < 2; ++kj)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        OTk[(32 * kj + r) * LDK + mma32_row(e, h)] = (__bf16)dKt[kj][e];
        OTv[(32 * kj + r) * LDK + mma32_row(e, h)] = (__bf16)dVt[kj][e];
      }
    __syncthreads();
    {
      const int64_t trow = g.token_row(b, wy, wx, lane);
      ST* kdst = dqkv + trow * ld + g.C + head * 32;
      ST* vdst = dqkv + trow * ld + 2 * g.C + head * 32;
#pragma unroll
      for (int e = 0; e < 32; e += 4) {
        const bf16x4 kv = *reinterpret_cast<const bf16x4*>(OTk + lane * LDK + e);
        const bf16x4 vv = *reinterpret_cast<const bf16x4*>(OTv + lane * LDK + e);
        IO<ST>::st4(kdst + e, make_float4((float)kv[0], (float)kv[1], (float)kv[2], (float)kv[3]));
        IO<ST>::st4(vdst + e, make_float4((float)vv[0], (float)vv[1], (float)vv[2], (float)vv[3]));
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int qi = 0; qi < 2; ++qi)
#pragma unroll
    for (int kj = 0; kj < 2; ++kj)
#pragma unroll
      for (int e = 0; e < 16; ++e) atomicAdd(&bins[rel_index(32 * qi + mma32_row(e, h), 32 * kj + r)], accb[qi][kj][e]);
  __syncthreads();
  for (int t = lane; t < 225; t += 64) atomicAdd(dtable + t * g.heads + head, bins[t]);
}

template <typename LT> size_t fwd_smem() {
  return sizeof(LT) * (2 * 64 * AttnDims<LT>::LDK + 32 * AttnDims<LT>::LDT) + sizeof(float) * 64 * LDS_S;
}
template <typename LT> size_t bwd_smem() {
  return sizeof(LT) * (3 * AttnDims<LT>::SMALL + 64 * AttnDims<LT>::LDT) + sizeof(float) * 2 * 64 * LDS_S;
}

bool geom_ok(const void* a, const void* b, int B, int hw, int C, int shift) {
  return a && b && B > 0 && hw >= 8 && (hw & 7) == 0 && C >= 32 && (C & 31) == 0 && shift >= 0 && shift < 8 &&
         !(shift > 0 && hw <= 8);
}



template <typename LT, typename ST>
int attn_fwd_launch(const void* qkv, const float* table, void* out, float* lse, const WinGeom& g, int B, hipStream_t st) {
  dim3 grid(B * g.nw * g.nw, g.heads);
  hipLaunchKernelGGL((attn_fwd_kernel<LT, ST>), grid, dim3(64), fwd_smem<LT>(), st, static_cast<const ST*>(qkv), table,
                     static_cast<ST*>(out), lse, g, B);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
template <typename LT, typename ST>
int attn_bwd_launch(const void* qkv, const float* table, const void* dout, void* dqkv, float* dtable, const WinGeom& g,
                    int B, hipStream_t st) {
  const int nwin = B * g.nw * g.nw;
  int chunks = 2048 / g.heads;
  if (chunks < 64) chunks = 64;
  if (chunks > nwin) chunks = nwin;
  static bool attr_done = false;
  if (!attr_done) {      // > 64 KiB of dynamic LDS needs an explicit opt-in
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<LT, ST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_smem<LT>());
    attr_done = true;
  }
  hipLaunchKernelGGL((attn_bwd_kernel<LT, ST>), dim3(chunks, g.heads), dim3(64), bwd_smem<LT>(), st,
                     static_cast<const ST*>(qkv), table, static_cast<const ST*>(dout), static_cast<ST*>(dqkv), dtable, g, B);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

template <typename ST>
int attn_bwd_fast_launch(const void* qkv, const float* table, const void* out, const float* lse, const void* dout,
                         void* dqkv, float* dtable, const WinGeom& g, int B, hipStream_t st) {
  const int nwin = B * g.nw * g.nw;
  int chunks = 4096 / g.heads;
  if (chunks < 64) chunks = 64;
  if (chunks > nwin) chunks = nwin;
  hipLaunchKernelGGL((attn_bwd_fast_kernel<ST>), dim3(chunks, g.heads), dim3(64), 0, st, static_cast<const ST*>(qkv), table,
                     static_cast<const ST*>(out), lse, static_cast<const ST*>(dout), static_cast<ST*>(dqkv), dtable, g, B);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

}  // namespace

extern "C" int wm_winattn_fwd(const void* qkv, const float* bias_table, void* out, float* lse, int32_t io_bf16,
                              int32_t B, int32_t hw, int32_t C, int32_t shift, int32_t mode, void* stream) {
  if (!geom_ok(qkv, bias_table, B, hw, C, shift) || !out || (io_bf16 && mode != 1)) return WM_EINVAL;
  WinGeom g{hw, shift, hw / 8, C, C / 32};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 0) return attn_fwd_launch<float, float>(qkv, bias_table, out, lse, g, B, st);
  if (mode == 1 && io_bf16) return attn_fwd_launch<__bf16, __bf16>(qkv, bias_table, out, lse, g, B, st);
  if (mode == 1) return attn_fwd_launch<__bf16, float>(qkv, bias_table, out, lse, g, B, st);
  return WM_EINVAL;
}

extern "C" int wm_winattn_bwd(const void* qkv, const float* bias_table, const void* out, const float* lse,
                              const void* dout, void* dqkv, float* dbias_table, int32_t io_bf16, int32_t B, int32_t hw,
                              int32_t C, int32_t shift, int32_t mode, void* stream) {
  if (!geom_ok(qkv, bias_table, B, hw, C, shift) || !dout || !dqkv || !dbias_table || (io_bf16 && mode != 1))
    return WM_EINVAL;
  WinGeom g{hw, shift, hw / 8, C, C / 32};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 1 && out && lse && !getenv("WM_NO_FAST_ATTN")) {      // forward output + row log-sum-exp available
    if (io_bf16) return attn_bwd_fast_launch<__bf16>(qkv, bias_table, out, lse, dout, dqkv, dbias_table, g, B, st);
    return attn_bwd_fast_launch<float>(qkv, bias_table, out, lse, dout, dqkv, dbias_table, g, B, st);
  }
  if (mode == 0) return attn_bwd_launch<float, float>(qkv, bias_table, dout, dqkv, dbias_table, g, B, st);
  if (mode == 1 && io_bf16) return attn_bwd_launch<__bf16, __bf16>(qkv, bias_table, dout, dqkv, dbias_table, g, B, st);
  if (mode == 1) return attn_bwd_launch<__bf16, float>(qkv, bias_table, dout, dqkv, dbias_table, g, B, st);
  return WM_EINVAL;
}
