// The attention half of a LeWin block as ONE window-local kernel per direction, for the blocks with C <= 128 (bf16 mode):
//   x1 = x + DropPath(proj(WindowAttention(norm1(x) (+ modulator))))  and  xn2 = norm2(x1)
// reference LeWinTransformerBlock.forward, model.py:982-1015 (norm1, roll, window partition, with_pos modulator :996-997,
// attention :998, window reverse, roll back, first residual :1015-1016) and WindowAttention.forward, model.py:523-551.
//
// Everything between the residual stream and itself is local to an 8 x 8 window (shifted or not), so a workgroup (one wave per
// head) keeps the window on chip: x rows -> LayerNorm in registers -> xn1 (bf16, LDS) -> q | k | v of the wave's head on the
// matrix cores (weights of the head resident in registers for the whole persistent window loop) -> scores, softmax and P V in
// registers (as csrc/attn_fast.hip) -> O of all heads in LDS -> projection (+ bias) * DropPath + x -> x1 rows out, and the
// second LayerNorm of those rows (the operand of the LeFF half) in the same pass.  The un-fused chain moves 34 C bytes per
// token (LayerNorm 6 C, qkv 8 C, attention 8 C, projection + norm2 12 C); this kernel reads x (4 C) and writes x1 + xn2
// (6 C), plus what the backward keeps: row statistics and the softmax log-sum-exp (backward recomputes q, k, v and O from x).
#include "common.h"
#include "../../include/wm_kernels.h"
#include "attn_geom.h"

namespace {
using namespace attnk;

constexpr float LN_EPS = 1e-5f;

template <int C> struct AH {
  static constexpr int HEADS = C / 32, T = 64 * HEADS;
  static constexpr int XS = C + 8;                  // bf16 row stride of the C-wide LDS images (16-byte aligned rows)
  static constexpr int FS = C + 4;                  // fp32 row stride of the staged projection rows
  static constexpr int CPT = C / 4;                 // 16-byte fp32 chunks per token = lanes that share a token
  static constexpr int KS = C / 16;                 // 16-deep k steps over the channel dimension
  // LDS (bytes): xn1 image | per head Q tile | per head K, V tiles (later the staged projection rows) | per head bias table,
  // q | k | v bias, mask regions
  static constexpr int XN_B = 64 * XS * 2, TILE_B = 64 * LDK * 2;
  static constexpr int KV_B = 2 * HEADS * TILE_B;
  static constexpr int SMEM = XN_B + HEADS * TILE_B + KV_B + HEADS * (232 + 96 + 64) * 4;
  static_assert(64 * FS * 4 <= KV_B, "the staged projection rows reuse the K / V tiles");
};

template <int L>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = L / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int ks) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (__bf16)a[8 * ks + j];
  return f;
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) { return make_uint2(pack_bf2(a, b), pack_bf2(c, d)); }
__device__ __forceinline__ void wave_fence() { asm volatile("" ::: "memory"); }     // LDS traffic of ONE wave is in order
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LayerNorm of the 8 row chunks a thread holds (token tid / CPT + 8 k, channels 4 (tid % CPT) .. + 3)
template <int C>
__device__ __forceinline__ void ln_rows(const float4 (&v)[8], float (&mu)[8], float (&rs)[8]) {
  constexpr int L = AH<C>::CPT;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mu[k] = group_sum<L>((v[k].x + v[k].y) + (v[k].z + v[k].w)) * (1.0f / C);
    const float a = v[k].x - mu[k], b = v[k].y - mu[k], c = v[k].z - mu[k], d = v[k].w - mu[k];
    rs[k] = 1.0f / sqrtf(group_sum<L>((a * a + b * b) + (c * c + d * d)) * (1.0f / C) + LN_EPS);
  }
}

// ------------------------------------------------------------------------------------------------------------ forward
// SAVE: also write xn1, q | k | v and the attention output (the operands of the un-fused backward kernels)
template <int C, bool SAVE>
__global__ __launch_bounds__(AH<C>::T, C == 128 ? 1 : 2) void attn_half_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ g1, const float* __restrict__ b1, const float* __restrict__ mod,
    const __bf16* __restrict__ wqkv, const float* __restrict__ bqkv, const float* __restrict__ table,
    const __bf16* __restrict__ wproj, const float* __restrict__ bproj, const float* __restrict__ s1,
    const float* __restrict__ g2, const float* __restrict__ b2, float* __restrict__ x1, __bf16* __restrict__ xn2,
    float* __restrict__ mean1, float* __restrict__ rstd1, float* __restrict__ mean2, float* __restrict__ rstd2,
    float* __restrict__ lse_g, __bf16* __restrict__ xn1_g, __bf16* __restrict__ qkv_g, __bf16* __restrict__ o_g, Geom g, int nwin) {
  typedef AH<C> A;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* const XN = reinterpret_cast<__bf16*>(smem);
  __bf16* const QT = XN + 64 * A::XS;                                 // [head][64][LDK]: q, later the head's attention output
  __bf16* const KV = QT + A::HEADS * 64 * LDK;                        // [head][k | v][64][LDK]
  float* const X1s = reinterpret_cast<float*>(KV);                    // staged projection rows [64][FS] (K / V are dead by then)
  float* const bias_all = reinterpret_cast<float*>(KV + 2 * A::HEADS * 64 * LDK);
  const int tid0 = threadIdx.x, head = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  float* const bias_s = bias_all + head * (232 + 96 + 64);            // relative-position bias of this head
  float* const bq_s = bias_s + 232;                                   // q | k | v bias of this head
  int* const reg_s = reinterpret_cast<int*>(bq_s + 96);               // shift-mask region of the window's 64 positions
  __bf16* const Qs = QT + head * 64 * LDK;
  __bf16* const Ks = KV + head * 2 * 64 * LDK;
  __bf16* const Vs = Ks + 64 * LDK;
  bf16x8 wq[3][A::KS], wp[A::KS];
  float bp;
  float4 gm1, bt1, gm2, bt2;
  {
    const int lane = tid0 & 63, r = lane & 31, h = lane >> 5, part = tid0 % A::CPT;
    for (int t = lane; t < 225; t += 64) bias_s[t] = table[t * A::HEADS + head];
    for (int t = lane; t < 96; t += 64) bq_s[t] = bqkv[(t >> 5) * C + 32 * head + (t & 31)];
    // this head's rows of the q | k | v weight and of the projection weight as MFMA operands, for the whole window loop
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int ks = 0; ks < A::KS; ++ks)
        wq[t][ks] = *reinterpret_cast<const bf16x8*>(wqkv + (int64_t)(t * C + 32 * head + r) * C + 16 * ks + 8 * h);
#pragma unroll
    for (int ks = 0; ks < A::KS; ++ks) wp[ks] = *reinterpret_cast<const bf16x8*>(wproj + (int64_t)(32 * head + r) * C + 16 * ks + 8 * h);
    bp = bproj[32 * head + r];
    gm1 = *reinterpret_cast<const float4*>(g1 + 4 * part); bt1 = *reinterpret_cast<const float4*>(b1 + 4 * part);
    gm2 = *reinterpret_cast<const float4*>(g2 + 4 * part); bt2 = *reinterpret_cast<const float4*>(b2 + 4 * part);
  }

  const int w0 = blockIdx.x;
  if (w0 >= nwin) return;
  int b, wy, wx;
  float4 xv[8], xnext[8];
  win_pos(w0, g.nw, b, wy, wx);
#pragma unroll
  for (int k = 0; k < 8; ++k)
    xnext[k] = *reinterpret_cast<const float4*>(x + (int64_t)g.token_row(b, wy, wx, tid0 / A::CPT + 8 * k) * C + 4 * (tid0 % A::CPT));

  for (int w = w0; w < nwin; w += gridDim.x) {
    // per-thread indices from an OPAQUE copy of the thread id in every window: hipcc otherwise hoists the loop-invariant parts
    // of every address in the body out of the persistent loop and keeps them in registers through all phases
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int part = tid % A::CPT, trow = tid / A::CPT;               // chunk k of this thread: token trow + 8 k, channels 4 part ..
    win_pos(w, g.nw, b, wy, wx);
    int64_t rowoff[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      rowoff[k] = (int64_t)g.token_row(b, wy, wx, trow + 8 * k);
      xv[k] = xnext[k];
    }
    // ---- norm1 (+ modulator) -> xn1 image
    {
      float mu[8], rs[8];
      ln_rows<C>(xv, mu, rs);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float4 o;
        o.x = (xv[k].x - mu[k]) * rs[k] * gm1.x + bt1.x;
        o.y = (xv[k].y - mu[k]) * rs[k] * gm1.y + bt1.y;
        o.z = (xv[k].z - mu[k]) * rs[k] * gm1.z + bt1.z;
        o.w = (xv[k].w - mu[k]) * rs[k] * gm1.w + bt1.w;
        if (mod) {
          const float4 m = *reinterpret_cast<const float4*>(mod + (trow + 8 * k) * C + 4 * part);
          o.x += m.x; o.y += m.y; o.z += m.z; o.w += m.w;
        }
        const uint2 pk = pack4(o.x, o.y, o.z, o.w);
        *reinterpret_cast<uint2*>(XN + (trow + 8 * k) * A::XS + 4 * part) = pk;
        if (SAVE) stream_store(xn1_g + rowoff[k] * C + 4 * part, pk);
        if (part == 0) { mean1[rowoff[k]] = mu[k]; rstd1[rowoff[k]] = rs[k]; }
      }
    }
    if (lane < 64) reg_s[lane] = g.region(wy, wx, lane);
    wg_barrier();                                                     // xn1 complete; the previous window's staged rows are consumed
    {                                                                 // the next window's rows fly during this window's products
      const int wn = w + (int)gridDim.x < nwin ? w + (int)gridDim.x : w;
      int nb, nwy, nwx;
      win_pos(wn, g.nw, nb, nwy, nwx);
#pragma unroll
      for (int k = 0; k < 8; ++k) xnext[k] = *reinterpret_cast<const float4*>(x + (int64_t)g.token_row(nb, nwy, nwx, trow + 8 * k) * C + 4 * part);
    }
    // ---- q | k | v of this head: [channel on the accumulator rows][token on the lane] -> [token][32] tiles
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      bf16x8 xf[A::KS];
#pragma unroll
      for (int ks = 0; ks < A::KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(XN + (32 * tt + r) * A::XS + 16 * ks + 8 * h);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[t][ks], xf[ks], acc, 0, 0, 0);
        __bf16* tile = t == 0 ? Qs : (t == 1 ? Ks : Vs);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float4 bb = *reinterpret_cast<const float4*>(bq_s + 32 * t + 8 * g4 + 4 * h);
          *reinterpret_cast<uint2*>(tile + (32 * tt + r) * LDK + 8 * g4 + 4 * h) =
              pack4(acc[4 * g4] + bb.x, acc[4 * g4 + 1] + bb.y, acc[4 * g4 + 2] + bb.z, acc[4 * g4 + 3] + bb.w);
        }
      }
    }
    wave_fence();
    if (SAVE) {                                                       // q | k | v rows of this head, 16-byte pieces
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int p = (lane >> 2) + 16 * it;
        __bf16* dst = qkv_g + (int64_t)g.token_row(b, wy, wx, p) * (3 * C) + 32 * head + 8 * (lane & 3);
        stream_store(dst, *reinterpret_cast<const uint4*>(Qs + p * LDK + 8 * (lane & 3)));
        stream_store(dst + C, *reinterpret_cast<const uint4*>(Ks + p * LDK + 8 * (lane & 3)));
        stream_store(dst + 2 * C, *reinterpret_cast<const uint4*>(Vs + p * LDK + 8 * (lane & 3)));
      }
    }
    // ---- window attention of this head (csrc/attn_fast.hip: S^T = K Q^T, softmax on the lane's query, O^T = V^T P^T)
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      const int i = 32 * qi + r;
      f32x16 s[2];
#pragma unroll
      for (int kj = 0; kj < 2; ++kj) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[kj][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          s[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ks, 32 * kj + r, ks, h), row_frag(Qs, i, ks, h), s[kj], 0, 0, 0);
      }
      const int regi = reg_s[i];
      const int ib = (i >> 3) * 15 + (i & 7) + 112;
      float m = -INFINITY;
#pragma unroll
      for (int kj = 0; kj < 2; ++kj)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int j = 32 * kj + mma32_row(e, h);
          float sv = fmaf(s[kj][e], kScale, bias_s[ib - ((j >> 3) * 15 + (j & 7))]);
          sv += (reg_s[j] != regi) ? -100.0f : 0.f;
          s[kj][e] = sv;
          m = fmaxf(m, sv);
        }
      m = fmaxf(m, __shfl_xor(m, 32));
      float sum = 0.f;
#pragma unroll
      for (int kj = 0; kj < 2; ++kj)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pv_ = __expf(s[kj][e] - m);
          s[kj][e] = pv_;
          sum += pv_;
        }
      sum += __shfl_xor(sum, 32);
      if (h == 0) lse_g[(int64_t)g.token_row(b, wy, wx, i) * A::HEADS + head] = m + __logf(sum);
      f32x16 o;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
      for (int kj = 0; kj < 2; ++kj)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vs, 32 * kj + 16 * ks, lane, true), acc_frag(s[kj], ks), o, 0, 0, 0);
      const float inv = 1.0f / sum;
      wave_fence();                                                   // Q rows of this query tile are consumed: they take O
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<uint2*>(Qs + i * LDK + 8 * g4 + 4 * h) =
            pack4(o[4 * g4] * inv, o[4 * g4 + 1] * inv, o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv);
    }
    wave_fence();
    if (SAVE) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int p = (lane >> 2) + 16 * it;
        stream_store(o_g + (int64_t)g.token_row(b, wy, wx, p) * C + 32 * head + 8 * (lane & 3),
                     *reinterpret_cast<const uint4*>(Qs + p * LDK + 8 * (lane & 3)));
      }
    }
    wg_barrier();                                                     // every head's O is in its tile; K / V are dead
    // ---- projection: this wave's 32 output channels of all 64 tokens, (acc + bias) * DropPath -> staged rows
    {
      const float sc = s1 ? s1[b] : 1.0f;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(QT + (ks >> 1) * 64 * LDK, 32 * tt + r, ks & 1, h), wp[ks], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) X1s[(32 * tt + mma32_row(e, h)) * A::FS + 32 * head + r] = (acc[e] + bp) * sc;
      }
    }
    wg_barrier();
    // ---- x1 = x + branch, norm2(x1): rows as 16-byte chunks
    {
      float4 y[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 p = *reinterpret_cast<const float4*>(X1s + (trow + 8 * k) * A::FS + 4 * part);
        y[k] = make_float4(xv[k].x + p.x, xv[k].y + p.y, xv[k].z + p.z, xv[k].w + p.w);
        stream_store(x1 + rowoff[k] * C + 4 * part, y[k]);
      }
      float mu[8], rs[8];
      ln_rows<C>(y, mu, rs);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        stream_store(xn2 + rowoff[k] * C + 4 * part,
                     pack4((y[k].x - mu[k]) * rs[k] * gm2.x + bt2.x, (y[k].y - mu[k]) * rs[k] * gm2.y + bt2.y,
                           (y[k].z - mu[k]) * rs[k] * gm2.z + bt2.z, (y[k].w - mu[k]) * rs[k] * gm2.w + bt2.w));
        if (part == 0) { mean2[rowoff[k]] = mu[k]; rstd2[rowoff[k]] = rs[k]; }
      }
    }
  }
}

bool ok(int B, int hw, int C) { return B > 0 && hw >= 8 && (hw & 7) == 0 && (C == 32 || C == 64 || C == 128); }

template <int C> constexpr int fwd_wgs_per_cu() { return C == 32 ? 6 : C == 64 ? 3 : 1; }

template <int C>
int launch_fwd(const float* x, const float* g1, const float* b1, const float* mod, const void* wqkv, const float* bqkv,
               const float* table, const void* wproj, const float* bproj, const float* s1, const float* g2, const float* b2,
               float* x1, void* xn2, float* mean1, float* rstd1, float* mean2, float* rstd2, float* lse, void* xn1, void* qkv,
               void* o, int B, int hw, int shift, hipStream_t st) {
  typedef AH<C> A;
  Geom g{hw, shift, hw / 8, C, C / 32};
  const int nwin = B * g.nw * g.nw;
  int grid = 256 * fwd_wgs_per_cu<C>();
  if (grid > nwin) grid = nwin;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_half_fwd_kernel<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, A::SMEM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_half_fwd_kernel<C, false>), hipFuncAttributeMaxDynamicSharedMemorySize, A::SMEM);
    attr = true;
  }
  if (xn1 && qkv && o)
    hipLaunchKernelGGL((attn_half_fwd_kernel<C, true>), dim3(grid), dim3(A::T), A::SMEM, st, x, g1, b1, mod,
                       static_cast<const __bf16*>(wqkv), bqkv, table, static_cast<const __bf16*>(wproj), bproj, s1, g2, b2, x1,
                       static_cast<__bf16*>(xn2), mean1, rstd1, mean2, rstd2, lse, static_cast<__bf16*>(xn1),
                       static_cast<__bf16*>(qkv), static_cast<__bf16*>(o), g, nwin);
  else
    hipLaunchKernelGGL((attn_half_fwd_kernel<C, false>), dim3(grid), dim3(A::T), A::SMEM, st, x, g1, b1, mod,
                       static_cast<const __bf16*>(wqkv), bqkv, table, static_cast<const __bf16*>(wproj), bproj, s1, g2, b2, x1,
                       static_cast<__bf16*>(xn2), mean1, rstd1, mean2, rstd2, lse, static_cast<__bf16*>(nullptr),
                       static_cast<__bf16*>(nullptr), static_cast<__bf16*>(nullptr), g, nwin);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

}  // namespace

extern "C" int32_t wm_attn_half_fusable(int32_t hw, int32_t C) { return ok(1, hw, C) ? 1 : 0; }

extern "C" int wm_attn_half_fwd(const float* x, const float* gamma1, const float* beta1, const float* modulator, const void* wqkv,
                                const float* bqkv, const float* bias_table, const void* wproj, const float* bproj,
                                const float* rowscale, const float* gamma2, const float* beta2, float* x1, void* xn2,
                                float* mean1, float* rstd1, float* mean2, float* rstd2, float* lse, void* xn1, void* qkv,
                                void* o, int32_t B, int32_t hw, int32_t C, int32_t shift, void* stream) {
  if (!x || !gamma1 || !beta1 || !wqkv || !bqkv || !bias_table || !wproj || !bproj || !gamma2 || !beta2 || !x1 || !xn2 || !mean1 ||
      !rstd1 || !mean2 || !rstd2 || !lse || !ok(B, hw, C) || shift < 0 || shift >= 8)
    return WM_EINVAL;
  if ((xn1 || qkv || o) && !(xn1 && qkv && o)) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (C == 32)
    return launch_fwd<32>(x, gamma1, beta1, modulator, wqkv, bqkv, bias_table, wproj, bproj, rowscale, gamma2, beta2, x1, xn2, mean1,
                          rstd1, mean2, rstd2, lse, xn1, qkv, o, B, hw, shift, st);
  if (C == 64)
    return launch_fwd<64>(x, gamma1, beta1, modulator, wqkv, bqkv, bias_table, wproj, bproj, rowscale, gamma2, beta2, x1, xn2, mean1,
                          rstd1, mean2, rstd2, lse, xn1, qkv, o, B, hw, shift, st);
  return launch_fwd<128>(x, gamma1, beta1, modulator, wqkv, bqkv, bias_table, wproj, bproj, rowscale, gamma2, beta2, x1, xn2, mean1,
                         rstd1, mean2, rstd2, lse, xn1, qkv, o, B, hw, shift, st);
}
