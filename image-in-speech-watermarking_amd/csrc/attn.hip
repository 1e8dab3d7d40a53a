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

// bf16-storage fast path (attn_fast.hip)
int wm_attn_fast_fwd(const __bf16* qkv, const float* table, __bf16* out, float* lse, int B, int hw, int C, int shift,
                     hipStream_t st);
int wm_attn_fast_bwd(const __bf16* qkv, const float* table, const __bf16* out, const float* lse, const __bf16* dout,
                     __bf16* dqkv, float* dtable, float* scratch, WmReduceJob* defer, int B, int hw, int C, int shift,
                     hipStream_t st);
// compute_mode "split" (attn_split.hip): fp32 storage, split-precision bf16 MFMA products
int wm_attn_split_fwd(const float* qkv, const float* table, float* out, float* lse, int B, int hw, int C, int shift, hipStream_t st);
int wm_attn_split_bwd(const float* qkv, const float* table, const float* out, const float* lse, const float* dout, float* dqkv,
                      float* dtable, float* scratch, WmReduceJob* defer, int B, int hw, int C, int shift, hipStream_t st);

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

}  // namespace

extern "C" int wm_winattn_fwd(const void* qkv, const float* bias_table, void* out, float* lse, int32_t io_bf16,
                              int32_t B, int32_t hw, int32_t C, int32_t shift, int32_t mode, void* stream) {
  if (!geom_ok(qkv, bias_table, B, hw, C, shift) || !out || (io_bf16 && mode != 1)) return WM_EINVAL;
  WinGeom g{hw, shift, hw / 8, C, C / 32};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 0) return attn_fwd_launch<float, float>(qkv, bias_table, out, lse, g, B, st);
  if (mode == 1 && io_bf16)
    return wm_attn_fast_fwd(static_cast<const __bf16*>(qkv), bias_table, static_cast<__bf16*>(out), lse, B, hw, C, shift, st);
  if (mode == 1) return attn_fwd_launch<__bf16, float>(qkv, bias_table, out, lse, g, B, st);
  if (mode == 2)                                               // fp32 storage, split-precision products
    return wm_attn_split_fwd(static_cast<const float*>(qkv), bias_table, static_cast<float*>(out), lse, B, hw, C, shift, st);
  return WM_EINVAL;
}

extern "C" int64_t wm_winattn_bwd_scratch_floats(void) { return 1024 * 232; }

extern "C" int wm_winattn_bwd(const void* qkv, const float* bias_table, const void* out, const float* lse,
                              const void* dout, void* dqkv, float* dbias_table, float* scratch, WmReduceJob* defer,
                              int32_t io_bf16, int32_t B, int32_t hw, int32_t C, int32_t shift, int32_t mode,
                              void* stream) {
  if (defer) *defer = WmReduceJob{};                           // count 0 unless the fast path defers its bias-table reduction
  if (!geom_ok(qkv, bias_table, B, hw, C, shift) || !dout || !dqkv || !dbias_table || (io_bf16 && mode != 1))
    return WM_EINVAL;
  WinGeom g{hw, shift, hw / 8, C, C / 32};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (mode == 1 && io_bf16 && out && lse)                 // forward output + row log-sum-exp available: attn_fast.hip
    return wm_attn_fast_bwd(static_cast<const __bf16*>(qkv), bias_table, static_cast<const __bf16*>(out), lse,
                            static_cast<const __bf16*>(dout), static_cast<__bf16*>(dqkv), dbias_table, scratch, defer, B, hw, C, shift, st);
  if (mode == 2 && out && lse)
    return wm_attn_split_bwd(static_cast<const float*>(qkv), bias_table, static_cast<const float*>(out), lse,
                             static_cast<const float*>(dout), static_cast<float*>(dqkv), dbias_table, scratch, defer, B, hw, C, shift, st);
  if (mode == 0 || mode == 2) return attn_bwd_launch<float, float>(qkv, bias_table, dout, dqkv, dbias_table, g, B, st);
  if (mode == 1 && io_bf16) return attn_bwd_launch<__bf16, __bf16>(qkv, bias_table, dout, dqkv, dbias_table, g, B, st);
  if (mode == 1) return attn_bwd_launch<__bf16, float>(qkv, bias_table, dout, dqkv, dbias_table, g, B, st);
  return WM_EINVAL;
}
