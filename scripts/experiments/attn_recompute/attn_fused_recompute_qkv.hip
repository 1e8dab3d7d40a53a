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
// SAVE (bits): also write 1 xn1, 2 q | k | v, 4 the attention output (operands of the backward: the fused one reads xn1 and o and
// recomputes q | k | v, the un-fused launches read all three)
template <int C, int SAVE>
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
        if (SAVE & 1) stream_store(xn1_g + rowoff[k] * C + 4 * part, pk);
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
    if (SAVE & 2) {                                                   // q | k | v rows of this head, 16-byte pieces
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
    if (SAVE & 4) {
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

// ----------------------------------------------------------------------------------------------------------- backward
// The block's whole residual-stream backward between the two weight-gradient products, per window:
//   dx1  = dy + norm2'(dxn2)                       (LayerNorm backward of norm2, rows in registers; dgamma2 / dbeta2 summed)
//   dO   = (dx1 * DropPath) Wproj                  (this head's 32 columns, matrix cores)
//   dq | dk | dv = window-attention backward        (as csrc/attn_fast.hip: P from the saved log-sum-exp, delta = dO . O)
//   dxn1 = dq|dk|dv Wqkv                           (this wave's 32 channels, all heads' gradients from LDS)
//   dx   = dx1 + norm1'(dxn1)                      (dgamma1 / dbeta1 / modulator gradient summed)
// It replaces wm_layernorm_bwd (norm2), the projection's input-gradient product, wm_winattn_bwd, the q | k | v input-gradient
// product and wm_layernorm_bwd (norm1): 60 C bytes per token in five launches -> 36 C in one (dxn2, dy, x1, q|k|v, o, x read;
// the bf16 copies of dx1 * DropPath and dq|dk|dv written for the two weight-gradient products, dx and its bf16 copy written).
template <int C> struct AB {
  static constexpr int HEADS = C / 32, T = 64 * HEADS, XS = C + 8, FS = C + 4, CPT = C / 4, KS = C / 16;
  static constexpr int TILE = 64 * LDK;                              // elements of one [64][LDK] bf16 tile
  // LDS: DX image | xn1 image | Q tiles | K tiles | V tiles | dO tiles | scratch tiles (dO + scratch later take the staged dxn1
  //      rows) | per head: bias table (232 floats), q | k | v bias (96 floats), row info (64 float4)
  static constexpr int SMEM = 2 * 64 * XS * 2 + 5 * HEADS * TILE * 2 + HEADS * (232 * 4 + 96 * 4 + 64 * 16);
  static_assert(64 * FS * 4 <= 2 * HEADS * TILE * 2, "the staged dxn1 rows reuse the dO / scratch tiles");
  static_assert(8 * 4 * C * 4 <= 64 * XS * 2, "the LayerNorm parameter sums reuse the DX image");
};

__device__ __forceinline__ float dot8(u32x4 a, u32x4 b) {
  return bf_lo(a.x) * bf_lo(b.x) + bf_hi(a.x) * bf_hi(b.x) + bf_lo(a.y) * bf_lo(b.y) + bf_hi(a.y) * bf_hi(b.y) +
         bf_lo(a.z) * bf_lo(b.z) + bf_hi(a.z) * bf_hi(b.z) + bf_lo(a.w) * bf_lo(b.w) + bf_hi(a.w) * bf_hi(b.w);
}
// chunk c = lane + 64 it of a [64 tokens][32] head tile: token (lane >> 2) + 16 it, 16-byte part lane & 3
__device__ __forceinline__ void chunk_rows(int (&row)[4], const Geom& g, int b, int wy, int wx, int lane) {
  const int xx = g.wrap(wx * 8 + ((lane >> 2) & 7) + g.shift);
#pragma unroll
  for (int it = 0; it < 4; ++it) row[it] = (b * g.hw + g.wrap(wy * 8 + (lane >> 5) + 2 * it + g.shift)) * g.hw + xx;
}
__device__ __forceinline__ void tile_load(u32x4 (&c)[4], const __bf16* __restrict__ src, int64_t ld, int col, const int (&row)[4]) {
#pragma unroll
  for (int it = 0; it < 4; ++it) c[it] = *reinterpret_cast<const u32x4*>(src + row[it] * ld + col);
}
__device__ __forceinline__ void tile_to_lds(__bf16* dst, const u32x4 (&c)[4], int lane) {
#pragma unroll
  for (int it = 0; it < 4; ++it) *reinterpret_cast<u32x4*>(dst + ((lane >> 2) + 16 * it) * LDK + 8 * (lane & 3)) = c[it];
}

template <int C, bool MOD>
__global__ __launch_bounds__(AB<C>::T) void attn_half_bwd_kernel(
    const __bf16* __restrict__ dxn2, const float* __restrict__ dy, const float* __restrict__ x1, const float* __restrict__ mean2,
    const float* __restrict__ rstd2, const float* __restrict__ g2, const float* __restrict__ s1, const float* __restrict__ x,
    const float* __restrict__ mean1, const float* __restrict__ rstd1, const float* __restrict__ g1, const __bf16* __restrict__ xn1,
    const __bf16* __restrict__ o_g, const float* __restrict__ lse_g, const float* __restrict__ table,
    const __bf16* __restrict__ wproj, const __bf16* __restrict__ wqkv, const float* __restrict__ bqkv,
    const __bf16* __restrict__ wqkvT, const float* __restrict__ cscale,
    __bf16* __restrict__ dx1h, __bf16* __restrict__ dqkv, float* __restrict__ dx, __bf16* __restrict__ dxh,
    float* __restrict__ part, Geom g, int nwin) {
  typedef AB<C> A;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* const DX = reinterpret_cast<__bf16*>(smem);                 // (dx1 * DropPath) rows, bf16: operand of the dO product
  __bf16* const XN = DX + 64 * A::XS;                                 // xn1 rows (saved by the forward): operand of the q | k | v product
  __bf16* const TQ = XN + 64 * A::XS;                                 // [kind q, k, v, dO, scratch][head][64][LDK]
  float* const DXN = reinterpret_cast<float*>(TQ + 3 * A::HEADS * A::TILE);   // staged dxn1 rows [64][FS] over the dO | scratch tiles
  float* const tail = reinterpret_cast<float*>(TQ + 5 * A::HEADS * A::TILE);
  const int tid0 = threadIdx.x, head = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  float* const bias_s = tail + head * (232 + 96);
  float* const bq_s = bias_s + 232;                                   // q | k | v bias of this head
  float4* const rowinfo = reinterpret_cast<float4*>(tail + A::HEADS * (232 + 96)) + head * 64;   // lse, delta = dO . O, mask region, -
  __bf16* const Qs = TQ + (0 * A::HEADS + head) * A::TILE;
  __bf16* const Ks = TQ + (1 * A::HEADS + head) * A::TILE;
  __bf16* const Vs = TQ + (2 * A::HEADS + head) * A::TILE;
  __bf16* const Ds = TQ + (3 * A::HEADS + head) * A::TILE;
  __bf16* const Xs = TQ + (4 * A::HEADS + head) * A::TILE;
  const int64_t ld3 = 3 * (int64_t)C;

  // this head's / this wave's slices of the two weights as MFMA operands of the TRANSPOSED products (gathered once):
  //   dO^T[c'][token] = sum_n Wproj[n][32 head + c'] dx1h[token][n]          (A operand: row c', k = n)
  //   dxn1[token][c]  = sum_j dqkv[token][j] Wqkv[j][32 head + c]            (B operand: k = j, column c; from wqkvT, per window)
  bf16x8 wpT[A::KS];
  float4 gm1, gm2;
  {
    const int lane = tid0 & 63, r = lane & 31, h = lane >> 5;
    for (int t = lane; t < 225; t += 64) bias_s[t] = table[t * A::HEADS + head];
    for (int t = lane; t < 96; t += 64) bq_s[t] = bqkv[(t >> 5) * C + 32 * head + (t & 31)];
#pragma unroll
    for (int ks = 0; ks < A::KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) wpT[ks][j] = wproj[(int64_t)(16 * ks + 8 * h + j) * C + 32 * head + r];
    gm1 = *reinterpret_cast<const float4*>(g1 + 4 * (tid0 % A::CPT));
    gm2 = *reinterpret_cast<const float4*>(g2 + 4 * (tid0 % A::CPT));
  }
  // relative-position-bias gradient of this head: the (query block, key block) pairs (0,0) and (1,1) cover the same offsets
  float accb[3][16];                                                  // [0]: qi == kj, [1]: (qi, kj) = (1, 0), [2]: (0, 1)
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[a][e] = 0.f;
  float4 ag2 = make_float4(0.f, 0.f, 0.f, 0.f), ab2 = ag2, ag1 = ag2, ab1 = ag2;   // LayerNorm parameter gradients (this lane's 4 channels)
  float4 dm[MOD ? 8 : 1];                                             // modulator gradient (this lane's 8 window positions x 4 channels)
#pragma unroll
  for (int k = 0; k < (MOD ? 8 : 1); ++k) dm[k] = make_float4(0.f, 0.f, 0.f, 0.f);

  const int w0 = blockIdx.x;                                          // (grid <= nwin)
  int nrow[4], nb, nwy, nwx;
  u32x4 po[4];
  uint2 pxn[8];                                                       // next window's xn1 rows (8-byte chunks: 4 channels)
  float plse;
  {
    const int lane = tid0 & 63, col = 32 * head + 8 * (lane & 3);
    win_pos(w0, g.nw, nb, nwy, nwx);
    chunk_rows(nrow, g, nb, nwy, nwx, lane);
    tile_load(po, o_g, C, col, nrow);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      pxn[k] = *reinterpret_cast<const uint2*>(xn1 + (int64_t)g.token_row(nb, nwy, nwx, tid0 / A::CPT + 8 * k) * C + 4 * (tid0 % A::CPT));
    plse = lse_g[(int64_t)g.token_row(nb, nwy, nwx, lane) * A::HEADS + head];
  }

  for (int w = w0; w < nwin; w += gridDim.x) {
    int tid = tid0;                                                   // (opaque per window: see the forward kernel)
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int part_ = tid % A::CPT, trow = tid / A::CPT;
    const int col = 32 * head + 8 * (lane & 3);
    int b, wy, wx, row[4];
    win_pos(w, g.nw, b, wy, wx);
    const float sc = s1 ? s1[b] : 1.0f, cs = cscale ? cscale[b] : 1.0f;
#pragma unroll
    for (int it = 0; it < 4; ++it) row[it] = nrow[it];
    // ---- norm2 backward on the rows: dx1 = dy + rstd2 (g - mean(g) - xhat mean(g xhat)), g = dxn2 * gamma2.  dx1 (fp32) waits in
    // the dx buffer for the end of the window (32 registers through the attention core would cost more than the L2 round trip)
    {
      int64_t rowoff[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) rowoff[k] = (int64_t)g.token_row(b, wy, wx, trow + 8 * k);
      uint2 dn[8];
      float4 dyv[8], xv[8];
      float mu[8], rs[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        dn[k] = *reinterpret_cast<const uint2*>(dxn2 + rowoff[k] * C + 4 * part_);
        dyv[k] = *reinterpret_cast<const float4*>(dy + rowoff[k] * C + 4 * part_);
        xv[k] = *reinterpret_cast<const float4*>(x1 + rowoff[k] * C + 4 * part_);
        mu[k] = mean2[rowoff[k]];
        rs[k] = rstd2[rowoff[k]];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 dv = make_float4(bf_lo(dn[k].x), bf_hi(dn[k].x), bf_lo(dn[k].y), bf_hi(dn[k].y));
        const float4 xh = make_float4((xv[k].x - mu[k]) * rs[k], (xv[k].y - mu[k]) * rs[k], (xv[k].z - mu[k]) * rs[k], (xv[k].w - mu[k]) * rs[k]);
        const float4 gg = make_float4(dv.x * gm2.x, dv.y * gm2.y, dv.z * gm2.z, dv.w * gm2.w);
        ag2.x += dv.x * xh.x; ag2.y += dv.y * xh.y; ag2.z += dv.z * xh.z; ag2.w += dv.w * xh.w;
        ab2.x += dv.x; ab2.y += dv.y; ab2.z += dv.z; ab2.w += dv.w;
        const float m1 = group_sum<A::CPT>((gg.x + gg.y) + (gg.z + gg.w)) * (1.0f / C);
        const float m2 = group_sum<A::CPT>((gg.x * xh.x + gg.y * xh.y) + (gg.z * xh.z + gg.w * xh.w)) * (1.0f / C);
        float4 d;
        d.x = dyv[k].x + rs[k] * (gg.x - m1 - xh.x * m2);
        d.y = dyv[k].y + rs[k] * (gg.y - m1 - xh.y * m2);
        d.z = dyv[k].z + rs[k] * (gg.z - m1 - xh.z * m2);
        d.w = dyv[k].w + rs[k] * (gg.w - m1 - xh.w * m2);
        *reinterpret_cast<float4*>(dx + rowoff[k] * C + 4 * part_) = d;
        const uint2 pk2 = pack4(d.x * sc, d.y * sc, d.z * sc, d.w * sc);
        *reinterpret_cast<uint2*>(DX + (trow + 8 * k) * A::XS + 4 * part_) = pk2;
        stream_store(dx1h + rowoff[k] * C + 4 * part_, pk2);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<uint2*>(XN + (trow + 8 * k) * A::XS + 4 * part_) = pxn[k];
    wg_barrier();                                                     // DX and xn1 complete; the previous window's staged rows are consumed
    // ---- q | k | v of this head again (the forward keeps xn1, not q | k | v): [channel on the accumulator rows][token on the lane]
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      bf16x8 wq[A::KS];                                               // this head's 32 rows of the q / k / v weight (L2-resident, per window)
#pragma unroll
      for (int ks = 0; ks < A::KS; ++ks) wq[ks] = *reinterpret_cast<const bf16x8*>(wqkv + (int64_t)(t * C + 32 * head + r) * C + 16 * ks + 8 * h);
      __bf16* tile = t == 0 ? Qs : (t == 1 ? Ks : Vs);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[ks], *reinterpret_cast<const bf16x8*>(XN + (32 * tt + r) * A::XS + 16 * ks + 8 * h), acc, 0, 0, 0);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float4 bb = *reinterpret_cast<const float4*>(bq_s + 32 * t + 8 * g4 + 4 * h);
          *reinterpret_cast<uint2*>(tile + (32 * tt + r) * LDK + 8 * g4 + 4 * h) =
              pack4(acc[4 * g4] + bb.x, acc[4 * g4 + 1] + bb.y, acc[4 * g4 + 2] + bb.z, acc[4 * g4 + 3] + bb.w);
        }
      }
    }
    rowinfo[lane].x = plse;
    rowinfo[lane].z = __int_as_float(g.region(wy, wx, lane));
    // ---- dO of this head: [channel on the accumulator rows][token on the lane] -> [token][32] tile
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < A::KS; ++ks)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wpT[ks], *reinterpret_cast<const bf16x8*>(DX + (32 * tt + r) * A::XS + 16 * ks + 8 * h), acc, 0, 0, 0);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<uint2*>(Ds + (32 * tt + r) * LDK + 8 * g4 + 4 * h) = pack4(acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
    }
    wave_fence();
#pragma unroll
    for (int it = 0; it < 4; ++it) {                                  // delta = dO . O per query (O from the forward, as 16-byte chunks)
      const u32x4 dch = *reinterpret_cast<const u32x4*>(Ds + ((lane >> 2) + 16 * it) * LDK + 8 * (lane & 3));
      float d = dot8(dch, po[it]);
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      if ((lane & 3) == 0) rowinfo[(lane >> 2) + 16 * it].y = d;
    }
    wave_fence();
    // ---- attention backward of this head (csrc/attn_fast.hip); dK / dV of a key block overwrite its K / V rows, dQ the Q tile
    f32x16 dq[2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[qi][e] = 0.f;
#pragma unroll
    for (int kj = 0; kj < 2; ++kj) {
      const int j = 32 * kj + r;
      const int regj = __float_as_int(rowinfo[j].z);
      const int jb = 112 - ((j >> 3) * 15 + (j & 7));
      f32x16 dVt, dKt;
#pragma unroll
      for (int e = 0; e < 16; ++e) { dVt[e] = 0.f; dKt[e] = 0.f; }
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {
        f32x16 s, p;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; p[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Qs, 32 * qi + r, ks, h), row_frag(Ks, j, ks, h), s, 0, 0, 0);
          p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ds, 32 * qi + r, ks, h), row_frag(Vs, j, ks, h), p, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = 32 * qi + mma32_row(e, h);
          const float4 info = rowinfo[i];
          float sv = fmaf(s[e], kScale, bias_s[jb + (i >> 3) * 15 + (i & 7)]);
          sv += (__float_as_int(info.z) != regj) ? -100.0f : 0.f;
          const float pv_ = __expf(sv - info.x);
          const float dsv = pv_ * (p[e] - info.y);
          s[e] = pv_;
          p[e] = dsv;
          accb[qi == kj ? 0 : (qi ? 1 : 2)][e] += dsv;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          dVt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ds, 32 * qi + 16 * ks, lane, true), acc_frag(s, ks), dVt, 0, 0, 0);
          dKt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qs, 32 * qi + 16 * ks, lane, true), acc_frag(p, ks), dKt, 0, 0, 0);
        }
        wave_fence();
#pragma unroll
        for (int e = 0; e < 16; ++e) Xs[mma32_row(e, h) * LDK + r] = (__bf16)p[e];          // dS tile [query][key]
        wave_fence();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          dq[qi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Xs, r, ks, h), tr_frag(Ks, 32 * kj + 16 * ks, lane, false), dq[qi], 0, 0, 0);
      }
      // dK, dV of keys 32 kj .. + 31: [e][key] accumulators -> the block's own K / V rows (no reader is left) -> 16-byte chunks
      wave_fence();
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        *reinterpret_cast<uint2*>(Ks + j * LDK + 8 * g4 + 4 * h) =
            pack4(dKt[4 * g4] * kScale, dKt[4 * g4 + 1] * kScale, dKt[4 * g4 + 2] * kScale, dKt[4 * g4 + 3] * kScale);
        *reinterpret_cast<uint2*>(Vs + j * LDK + 8 * g4 + 4 * h) = pack4(dVt[4 * g4], dVt[4 * g4 + 1], dVt[4 * g4 + 2], dVt[4 * g4 + 3]);
      }
      wave_fence();
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int rl = 32 * kj + (lane >> 2) + 16 * t;
        __bf16* dst = dqkv + row[2 * kj + t] * ld3 + col;
        stream_store(dst + C, *reinterpret_cast<const uint4*>(Ks + rl * LDK + 8 * (lane & 3)));
        stream_store(dst + 2 * C, *reinterpret_cast<const uint4*>(Vs + rl * LDK + 8 * (lane & 3)));
      }
    }
    wave_fence();
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)                                    // dQ: [query][e] accumulators (lane = e) -> the Q tile
#pragma unroll
      for (int e = 0; e < 16; ++e) Qs[(32 * qi + mma32_row(e, h)) * LDK + r] = (__bf16)(dq[qi][e] * kScale);
    wave_fence();
#pragma unroll
    for (int it = 0; it < 4; ++it)
      stream_store(dqkv + row[it] * ld3 + col, *reinterpret_cast<const uint4*>(Qs + ((lane >> 2) + 16 * it) * LDK + 8 * (lane & 3)));
    {                                                                 // next window's tiles (the last one re-reads itself: no branch):
      const int wn = w + (int)gridDim.x < nwin ? w + (int)gridDim.x : w;   // requested here, where the core's accumulators are dead
      win_pos(wn, g.nw, nb, nwy, nwx);
      chunk_rows(nrow, g, nb, nwy, nwx, lane);
      tile_load(po, o_g, C, col, nrow);
#pragma unroll
      for (int k = 0; k < 8; ++k) pxn[k] = *reinterpret_cast<const uint2*>(xn1 + (int64_t)g.token_row(nb, nwy, nwx, trow + 8 * k) * C + 4 * part_);
      plse = lse_g[(int64_t)g.token_row(nb, nwy, nwx, lane) * A::HEADS + head];
    }
    // this wave's 32 columns of Wqkv as B operands (k = the 3 C rows of dq | dk | dv) from the TRANSPOSED copy [C][3 C]: requested per
    // window (L2-resident) instead of held in 24 - 96 registers through the attention core
    bf16x8 wqT[3 * A::KS];
#pragma unroll
    for (int kk = 0; kk < 3 * A::KS; ++kk) wqT[kk] = *reinterpret_cast<const bf16x8*>(wqkvT + (int64_t)(32 * head + r) * (3 * C) + 16 * kk + 8 * h);
    wg_barrier();                                                     // every head's dq | dk | dv tile is complete; dO / scratch are dead
    // ---- dxn1, this wave's 32 channels: k runs over dq | dk | dv of all heads
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 3 * A::KS; ++kk) {
        const int t = kk / A::KS, rem = kk % A::KS;
        const __bf16* tile = TQ + (t * A::HEADS + (rem >> 1)) * A::TILE;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tile, 32 * tt + r, rem & 1, h), wqT[kk], acc, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) DXN[(32 * tt + mma32_row(e, h)) * A::FS + 32 * head + r] = acc[e];
    }
    wg_barrier();
    // ---- norm1 backward on the rows: dx = dx1 + rstd1 (g - mean(g) - xhat mean(g xhat)), g = dxn1 * gamma1
    {
      float4 xv[8], dx1r[8];
      float mu[8], rs[8];
      int64_t rowoff[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        rowoff[k] = (int64_t)g.token_row(b, wy, wx, trow + 8 * k);
        xv[k] = *reinterpret_cast<const float4*>(x + rowoff[k] * C + 4 * part_);
        dx1r[k] = *reinterpret_cast<const float4*>(dx + rowoff[k] * C + 4 * part_);   // written by this very thread above
        mu[k] = mean1[rowoff[k]];
        rs[k] = rstd1[rowoff[k]];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 dv = *reinterpret_cast<const float4*>(DXN + (trow + 8 * k) * A::FS + 4 * part_);
        const float4 xh = make_float4((xv[k].x - mu[k]) * rs[k], (xv[k].y - mu[k]) * rs[k], (xv[k].z - mu[k]) * rs[k], (xv[k].w - mu[k]) * rs[k]);
        const float4 gg = make_float4(dv.x * gm1.x, dv.y * gm1.y, dv.z * gm1.z, dv.w * gm1.w);
        ag1.x += dv.x * xh.x; ag1.y += dv.y * xh.y; ag1.z += dv.z * xh.z; ag1.w += dv.w * xh.w;
        ab1.x += dv.x; ab1.y += dv.y; ab1.z += dv.z; ab1.w += dv.w;
        if (MOD) { dm[k].x += dv.x; dm[k].y += dv.y; dm[k].z += dv.z; dm[k].w += dv.w; }
        const float m1 = group_sum<A::CPT>((gg.x + gg.y) + (gg.z + gg.w)) * (1.0f / C);
        const float m2 = group_sum<A::CPT>((gg.x * xh.x + gg.y * xh.y) + (gg.z * xh.z + gg.w * xh.w)) * (1.0f / C);
        float4 d;
        d.x = dx1r[k].x + rs[k] * (gg.x - m1 - xh.x * m2);
        d.y = dx1r[k].y + rs[k] * (gg.y - m1 - xh.y * m2);
        d.z = dx1r[k].z + rs[k] * (gg.z - m1 - xh.z * m2);
        d.w = dx1r[k].w + rs[k] * (gg.w - m1 - xh.w * m2);
        stream_store(dx + rowoff[k] * C + 4 * part_, d);
        if (dxh) stream_store(dxh + rowoff[k] * C + 4 * part_, pack4(d.x * cs, d.y * cs, d.z * cs, d.w * cs));
      }
    }
  }
  // ---- parameter gradients of this workgroup -> its slot of `part`: [head][232] bias bins | dgamma2 dbeta2 dgamma1 dbeta1 [4][C] | dmod [64][C]
  const int lane = tid0 & 63, r = lane & 31, h = lane >> 5;
  float* const slot = part + (int64_t)blockIdx.x * (A::HEADS * 232 + 4 * C + (MOD ? 64 * C : 0));
  wg_barrier();
  {                                                                   // LayerNorm parameters: fold the 8 row groups through LDS
    float* red = reinterpret_cast<float*>(DX);                        // [8 row groups][4][C]
    const int part_ = tid0 % A::CPT, trow = tid0 / A::CPT;
    *reinterpret_cast<float4*>(red + (trow * 4 + 0) * C + 4 * part_) = ag2;
    *reinterpret_cast<float4*>(red + (trow * 4 + 1) * C + 4 * part_) = ab2;
    *reinterpret_cast<float4*>(red + (trow * 4 + 2) * C + 4 * part_) = ag1;
    *reinterpret_cast<float4*>(red + (trow * 4 + 3) * C + 4 * part_) = ab1;
    wg_barrier();
    for (int i = tid0; i < 4 * C; i += A::T) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += red[q * 4 * C + i];
      slot[A::HEADS * 232 + i] = sum;
    }
    if (MOD) {
#pragma unroll
      for (int k = 0; k < 8; ++k) *reinterpret_cast<float4*>(slot + A::HEADS * 232 + 4 * C + (trow + 8 * k) * C + 4 * part_) = dm[k];
    }
  }
  // relative-position-bias gradient: registers -> private LDS bins (read-add-write rounds, as csrc/attn_fast.hip) -> the slot
  float* scr = reinterpret_cast<float*>(Qs);                          // 8 copies x 232 floats in this head's Q | (next) tiles: use Q and K of this head
  float* scr2 = reinterpret_cast<float*>(Ks);                         // (a tile holds 64 * 40 * 2 B = 1280 floats: 4 copies of 232 fit twice over)
  auto bin = [&](int cpy) -> float* { return (cpy < 4 ? scr : scr2) + (cpy & 3) * 232; };
  for (int t = lane; t < 4 * 232; t += 64) { scr[t] = 0.f; scr2[t] = 0.f; }
  wave_fence();
#pragma unroll
  for (int kb = 0; kb < 48; kb += 4) {
    float cur[4];
    float* adr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int grp = (kb + u) >> 4, e = (kb + u) & 15;
      const int i = 32 * (grp == 1) + mma32_row(e, h), j = 32 * (grp == 2) + r;
      adr[u] = bin(2 * u + h) + ((i >> 3) - (j >> 3) + 7) * 15 + ((i & 7) - (j & 7) + 7);
      cur[u] = *adr[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int grp = (kb + u) >> 4, e = (kb + u) & 15;
      *adr[u] = cur[u] + accb[grp][e];
    }
    wave_fence();
  }
  for (int t = lane; t < 225; t += 64) {
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) sum += bin(c)[t];
    slot[head * 232 + t] = sum;
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
  const int save = (xn1 ? 1 : 0) | (qkv ? 2 : 0) | (o ? 4 : 0);
#define WM_AF(S)                                                                                                                    \
  {                                                                                                                                 \
    static bool attr = false;                                                                                                       \
    if (!attr) {                                                                                                                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_half_fwd_kernel<C, S>), hipFuncAttributeMaxDynamicSharedMemorySize, A::SMEM); \
      attr = true;                                                                                                                  \
    }                                                                                                                               \
    hipLaunchKernelGGL((attn_half_fwd_kernel<C, S>), dim3(grid), dim3(A::T), A::SMEM, st, x, g1, b1, mod,                           \
                       static_cast<const __bf16*>(wqkv), bqkv, table, static_cast<const __bf16*>(wproj), bproj, s1, g2, b2, x1,      \
                       static_cast<__bf16*>(xn2), mean1, rstd1, mean2, rstd2, lse, static_cast<__bf16*>(xn1),                       \
                       static_cast<__bf16*>(qkv), static_cast<__bf16*>(o), g, nwin);                                                \
  }
  if (save == 7) WM_AF(7) else if (save == 5) WM_AF(5) else if (save == 0) WM_AF(0) else return WM_EINVAL;
#undef WM_AF
  WM_CHECK_LAUNCH();
  return WM_OK;
}

template <int C> constexpr int bwd_wgs_per_cu() { return C == 32 ? 4 : C == 64 ? 2 : 1; }
template <int C>
int bwd_grid(int nwin) { const int gsz = 256 * bwd_wgs_per_cu<C>(); return gsz < nwin ? gsz : nwin; }
int bwd_grid_for(int C, int nwin) { return C == 32 ? bwd_grid<32>(nwin) : C == 64 ? bwd_grid<64>(nwin) : bwd_grid<128>(nwin); }

template <int C, bool MOD>
int launch_bwd(const void* dxn2, const float* dy, const float* x1, const float* mean2, const float* rstd2, const float* g2, const float* s1,
               const float* x, const float* mean1, const float* rstd1, const float* g1, const void* xn1, const void* o, const float* lse,
               const float* table, const void* wproj, const void* wqkv, const float* bqkv, const void* wqkvT, const float* cscale, void* dx1h,
               void* dqkv, float* dx, void* dxh, float* part, int B, int hw, int shift, hipStream_t st) {
  typedef AB<C> A;
  Geom g{hw, shift, hw / 8, C, C / 32};
  const int nwin = B * g.nw * g.nw;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_half_bwd_kernel<C, MOD>), hipFuncAttributeMaxDynamicSharedMemorySize, A::SMEM);
    attr = true;
  }
  hipLaunchKernelGGL((attn_half_bwd_kernel<C, MOD>), dim3(bwd_grid<C>(nwin)), dim3(A::T), A::SMEM, st, static_cast<const __bf16*>(dxn2), dy, x1,
                     mean2, rstd2, g2, s1, x, mean1, rstd1, g1, static_cast<const __bf16*>(xn1), static_cast<const __bf16*>(o), lse, table,
                     static_cast<const __bf16*>(wproj), static_cast<const __bf16*>(wqkv), bqkv, static_cast<const __bf16*>(wqkvT), cscale,
                     static_cast<__bf16*>(dx1h),
                     static_cast<__bf16*>(dqkv), dx, static_cast<__bf16*>(dxh), part, g, nwin);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

}  // namespace

extern "C" int32_t wm_attn_half_fusable(int32_t hw, int32_t C) { return ok(1, hw, C) ? 1 : 0; }

extern "C" int64_t wm_attn_half_bwd_scratch_floats(int32_t B, int32_t hw, int32_t C, int32_t has_mod) {
  if (!ok(B, hw, C)) return 0;
  const int nwin = B * (hw / 8) * (hw / 8);
  return (int64_t)bwd_grid_for(C, nwin) * ((C / 32) * 232 + 4 * C + (has_mod ? 64 * C : 0));
}

extern "C" int wm_attn_half_bwd(const void* dxn2, const float* dy, const float* x1, const float* mean2, const float* rstd2,
                                const float* gamma2, const float* rowscale, const float* x, const float* mean1, const float* rstd1,
                                const float* gamma1, const void* xn1, const void* o, const float* lse, const float* bias_table,
                                const void* wproj, const void* wqkv, const float* bqkv, const void* wqkvT, const float* copy_scale,
                                void* dx1h, void* dqkv, float* dx,
                                void* dxh, float* dgamma2, float* dbeta2, float* dgamma1, float* dbeta1, float* dmod, float* dbias_table,
                                float* scratch, WmReduceJob* defer, int32_t B, int32_t hw, int32_t C, int32_t shift, void* stream) {
  if (!dxn2 || !dy || !x1 || !mean2 || !rstd2 || !gamma2 || !x || !mean1 || !rstd1 || !gamma1 || !xn1 || !o || !lse || !bias_table ||
      !wproj || !wqkv || !bqkv || !wqkvT || !dx1h || !dqkv || !dx || !dgamma2 || !dbeta2 || !dgamma1 || !dbeta1 || !dbias_table || !scratch ||
      !ok(B, hw, C) || shift < 0 || shift >= 8)
    return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc;
#define WM_AB(CC, MM) rc = launch_bwd<CC, MM>(dxn2, dy, x1, mean2, rstd2, gamma2, rowscale, x, mean1, rstd1, gamma1, xn1, o, lse, bias_table, \
                                              wproj, wqkv, bqkv, wqkvT, copy_scale, dx1h, dqkv, dx, dxh, scratch, B, hw, shift, st)
  if (C == 32) { if (dmod) WM_AB(32, true); else WM_AB(32, false); }
  else if (C == 64) { if (dmod) WM_AB(64, true); else WM_AB(64, false); }
  else { if (dmod) WM_AB(128, true); else WM_AB(128, false); }
#undef WM_AB
  if (rc != WM_OK) return rc;
  const int heads = C / 32, nwin = B * (hw / 8) * (hw / 8), count = bwd_grid_for(C, nwin);
  const int64_t per = heads * 232 + 4 * C + (dmod ? 64 * C : 0);
  WmReduceJob jobs[6] = {};
  jobs[0].src = scratch; jobs[0].pstride = per; jobs[0].src_ld = 232; jobs[0].count = count; jobs[0].rows = heads; jobs[0].cols = 225;
  jobs[0].dst = dbias_table; jobs[0].dst_rs = 1; jobs[0].dst_cs = heads;
  float* dsts[4] = {dgamma2, dbeta2, dgamma1, dbeta1};
  for (int i = 0; i < 4; ++i) {
    WmReduceJob& jb = jobs[1 + i];
    jb.src = scratch + heads * 232 + i * C; jb.pstride = per; jb.src_ld = C; jb.count = count; jb.rows = 1; jb.cols = C;
    jb.dst = dsts[i]; jb.dst_rs = C; jb.dst_cs = 1;
  }
  if (dmod) {
    WmReduceJob& jb = jobs[5];
    jb.src = scratch + heads * 232 + 4 * C; jb.pstride = per; jb.src_ld = C; jb.count = count; jb.rows = 64; jb.cols = C;
    jb.dst = dmod; jb.dst_rs = C; jb.dst_cs = 1;
  }
  if (defer) {
    for (int i = 0; i < 6; ++i) defer[i] = jobs[i];
    return WM_OK;
  }
  return wm_reduce_jobs(jobs, dmod ? 6 : 5, stream);
}


extern "C" int wm_attn_half_fwd(const float* x, const float* gamma1, const float* beta1, const float* modulator, const void* wqkv,
                                const float* bqkv, const float* bias_table, const void* wproj, const float* bproj,
                                const float* rowscale, const float* gamma2, const float* beta2, float* x1, void* xn2,
                                float* mean1, float* rstd1, float* mean2, float* rstd2, float* lse, void* xn1, void* qkv,
                                void* o, int32_t B, int32_t hw, int32_t C, int32_t shift, void* stream) {
  if (!x || !gamma1 || !beta1 || !wqkv || !bqkv || !bias_table || !wproj || !bproj || !gamma2 || !beta2 || !x1 || !xn2 || !mean1 ||
      !rstd1 || !mean2 || !rstd2 || !lse || !ok(B, hw, C) || shift < 0 || shift >= 8)
    return WM_EINVAL;
  if (!((xn1 && qkv && o) || (xn1 && !qkv && o) || (!xn1 && !qkv && !o))) return WM_EINVAL;      // all, (xn1, o), or none
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
