// Window attention of compute_mode "split": fp32 storage (q|k|v, out, gradients as the fp32 mode holds them), every product on
// v_mfma_f32_32x32x16_bf16 in split precision — x = hi + lo (two bf16), product = hi*hi + hi*lo + lo*hi, fp32 accumulate, 2^-17
// relative per term (the stft_mfma.hip / gemm_fast.hip SPLIT scheme) — instead of the fp32-input kernels of attn.hip.
// Reference semantics: WindowAttention.forward (model.py:523-551) + shift / window partition / mask (model.py:954-1013).
//
// Same dataflow as attn_fast.hip (one wave per (window, head), persistent over a strided window list, scores in registers,
// forward S^T = K Q^T so that the softmax is a register reduction, backward S = Q K^T with the saved row log-sum-exp, bias
// gradient reduced in registers across windows), with three differences that fp32 storage forces:
//  * a token's head row is 128 bytes = 8 chunks of 16; a tile is 8 chunks per lane, converted to a hi and a lo bf16 image in LDS
//    on arrival (no register prefetch of the next window: this is the parity mode, the tiles would take 160 registers);
//  * P, dS (accumulators used as operands) are split in registers; dS crosses LDS as a hi and a lo image;
//  * results leave through fp32 LDS images so that every global store is a 16-byte chunk of a 128-byte head row.
#include "attn_geom.h"

namespace {
using namespace attnk;

constexpr int LDO = 36;                                        // fp32 staging row (floats): 144 B, 16-byte aligned

struct HiLo { bf16x8 hi, lo; };

__device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) {
  const float e[4] = {v.x, v.y, v.z, v.w};
  float h[4], l[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const __bf16 hb = (__bf16)e[u];
    h[u] = (float)hb;
    l[u] = e[u] - h[u];
  }
  hi = make_uint2(pack_bf2(h[0], h[1]), pack_bf2(h[2], h[3]));
  lo = make_uint2(pack_bf2(l[0], l[1]), pack_bf2(l[2], l[3]));
}

// accumulator rows 8 ks .. 8 ks + 7 of a 32 x 32 result as the B operand of the next product, split
__device__ __forceinline__ HiLo acc_split(const f32x16& a, int ks) {
  HiLo f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = a[8 * ks + j];
    const __bf16 hb = (__bf16)x;
    f.hi[j] = hb;
    f.lo[j] = (__bf16)(x - (float)hb);
  }
  return f;
}

__device__ __forceinline__ f32x16 mma3(f32x16 acc, const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  return acc;
}

// chunk (lane, it) of a [64 tokens][32 floats] tile: token (lane >> 3) + 8 it (window row it, column lane >> 3), floats 4 (lane & 7) ..
__device__ __forceinline__ void token_rows(int (&row)[8], const Geom& g, int b, int wy, int wx, int lane) {
  const int xx = g.wrap(wx * 8 + (lane >> 3) + g.shift);
#pragma unroll
  for (int it = 0; it < 8; ++it) row[it] = (b * g.hw + g.wrap(wy * 8 + it + g.shift)) * g.hw + xx;
}
__device__ __forceinline__ void tile_load(float4 (&c)[8], const float* __restrict__ src, int64_t ld, int col, const int (&row)[8]) {
#pragma unroll
  for (int it = 0; it < 8; ++it) c[it] = *reinterpret_cast<const float4*>(src + row[it] * ld + col);
}
__device__ __forceinline__ void tile_to_lds(__bf16* hi, __bf16* lo, const float4 (&c)[8], int lane) {
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    uint2 h2, l2;
    split4(c[it], h2, l2);
    const int o = ((lane >> 3) + 8 * it) * LDK + 4 * (lane & 7);
    *reinterpret_cast<uint2*>(hi + o) = h2;
    *reinterpret_cast<uint2*>(lo + o) = l2;
  }
}

// ------------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(64) void attn_fwd_split_kernel(const float* __restrict__ qkv, const float* __restrict__ table,
                                                            float* __restrict__ out, float* __restrict__ lse_g, Geom g,
                                                            int nwin, int nchunk) {
  __shared__ __attribute__((aligned(16))) __bf16 tiles[6 * 64 * LDK];
  __bf16 *Qh = tiles, *Ql = tiles + 64 * LDK, *Kh = tiles + 2 * 64 * LDK, *Kl = tiles + 3 * 64 * LDK, *Vh = tiles + 4 * 64 * LDK,
         *Vl = tiles + 5 * 64 * LDK;
  float* Os = reinterpret_cast<float*>(tiles);                 // [64][LDO] fp32 = 9 KB over Qh | Ql (consumed per query tile)
  __shared__ float bias_s[232];
  __shared__ int reg_s[64];
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int v = virtual_block(blockIdx.x, gridDim.x);
  const int head = v % g.heads, chunk = v / g.heads;
  const int64_t ld = 3 * (int64_t)g.C;
  const int col = head * 32 + 4 * (lane & 7);
  for (int t = lane; t < 225; t += 64) bias_s[t] = table[t * g.heads + head];
  for (int w = chunk; w < nwin; w += nchunk) {
    int row[8], b, wy, wx;
    win_pos(w, g.nw, b, wy, wx);
    token_rows(row, g, b, wy, wx, lane);
    float4 cq[8], ck[8], cv[8];
    tile_load(cq, qkv, ld, col, row);
    tile_load(ck, qkv, ld, g.C + col, row);
    tile_load(cv, qkv, ld, 2 * g.C + col, row);
    __syncthreads();                                           // the previous window's output image has been stored
    tile_to_lds(Qh, Ql, cq, lane);
    tile_to_lds(Kh, Kl, ck, lane);
    tile_to_lds(Vh, Vl, cv, lane);
    reg_s[lane] = g.region(wy, wx, lane);
    __syncthreads();
    f32x16 o[2];
    float inv[2];
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
          s[kj] = mma3(s[kj], row_frag(Kh, 32 * kj + r, ks, h), row_frag(Kl, 32 * kj + r, ks, h), row_frag(Qh, i, ks, h),
                       row_frag(Ql, i, ks, h));
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
          const float pv_ = expf(s[kj][e] - m);
          s[kj][e] = pv_;
          sum += pv_;
        }
      sum += __shfl_xor(sum, 32);
      if (lse_g && h == 0) lse_g[(int64_t)g.token_row(b, wy, wx, i) * g.heads + head] = m + logf(sum);
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qi][e] = 0.f;
#pragma unroll
      for (int kj = 0; kj < 2; ++kj)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const HiLo p = acc_split(s[kj], ks);
          o[qi] = mma3(o[qi], tr_frag(Vh, 32 * kj + 16 * ks, lane, true), tr_frag(Vl, 32 * kj + 16 * ks, lane, true), p.hi, p.lo);
        }
      inv[qi] = 1.0f / sum;
    }
    __syncthreads();                                           // Q, K, V images consumed: the fp32 output image takes Q's place
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<float4*>(Os + (32 * qi + r) * LDO + 8 * g4 + 4 * h) =
            make_float4(o[qi][4 * g4] * inv[qi], o[qi][4 * g4 + 1] * inv[qi], o[qi][4 * g4 + 2] * inv[qi], o[qi][4 * g4 + 3] * inv[qi]);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it)
      stream_store(out + row[it] * (int64_t)g.C + col, *reinterpret_cast<const float4*>(Os + ((lane >> 3) + 8 * it) * LDO + 4 * (lane & 7)));
  }
}

// ----------------------------------------------------------------------------------------------------------- backward
__global__ __launch_bounds__(64) void attn_bwd_split_kernel(const float* __restrict__ qkv, const float* __restrict__ table,
                                                            const float* __restrict__ out, const float* __restrict__ lse_g,
                                                            const float* __restrict__ dout, float* __restrict__ dqkv,
                                                            float* __restrict__ dtable, float* __restrict__ part, Geom g, int nwin,
                                                            int nchunk) {
  __shared__ __attribute__((aligned(16))) __bf16 tiles[10 * 64 * LDK];      // Q K V dO (hi | lo each), dS hi | lo
  __bf16 *Qh = tiles, *Ql = tiles + 64 * LDK, *Kh = tiles + 2 * 64 * LDK, *Kl = tiles + 3 * 64 * LDK, *Vh = tiles + 4 * 64 * LDK,
         *Vl = tiles + 5 * 64 * LDK, *Dh = tiles + 6 * 64 * LDK, *Dl = tiles + 7 * 64 * LDK, *Xh = tiles + 8 * 64 * LDK,
         *Xl = tiles + 9 * 64 * LDK;
  float* Xf = reinterpret_cast<float*>(Xh);                    // [64][LDO] fp32 staging of dK | dV (32 rows each) and of dQ: 9 KB over Xh | Xl
  __shared__ float4 rowinfo[64];                               // lse, delta = dO . O, mask region (int bits), -
  __shared__ float bias_s[232];
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int v = virtual_block(blockIdx.x, gridDim.x);
  const int head = v % g.heads, chunk = v / g.heads;
  const int64_t ld = 3 * (int64_t)g.C;
  const int col = head * 32 + 4 * (lane & 7);
  for (int t = lane; t < 225; t += 64) bias_s[t] = table[t * g.heads + head];
  float accb[2][2][16];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
      for (int e = 0; e < 16; ++e) accb[a][b2][e] = 0.f;

  for (int w = chunk; w < nwin; w += nchunk) {
    int row[8], b, wy, wx;
    win_pos(w, g.nw, b, wy, wx);
    token_rows(row, g, b, wy, wx, lane);
    __syncthreads();                                           // the previous window's last staging image has been stored
    {
      float4 cq[8], ck[8], cv[8];
      tile_load(cq, qkv, ld, col, row);
      tile_load(ck, qkv, ld, g.C + col, row);
      tile_load(cv, qkv, ld, 2 * g.C + col, row);
      tile_to_lds(Qh, Ql, cq, lane);
      tile_to_lds(Kh, Kl, ck, lane);
      tile_to_lds(Vh, Vl, cv, lane);
    }
    {
      float4 cd[8], co[8];
      tile_load(cd, dout, g.C, col, row);
      tile_load(co, out, g.C, col, row);
      tile_to_lds(Dh, Dl, cd, lane);
#pragma unroll
      for (int it = 0; it < 8; ++it) {                         // delta = dO . O of token (lane >> 3) + 8 it: 8 lanes per token
        float d = (cd[it].x * co[it].x + cd[it].y * co[it].y) + (cd[it].z * co[it].z + cd[it].w * co[it].w);
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        d += __shfl_xor(d, 4);
        if ((lane & 7) == 0) rowinfo[(lane >> 3) + 8 * it].y = d;
      }
    }
    rowinfo[lane].x = lse_g[(int64_t)g.token_row(b, wy, wx, lane) * g.heads + head];
    rowinfo[lane].z = __int_as_float(g.region(wy, wx, lane));
    __syncthreads();
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
          s = mma3(s, row_frag(Qh, 32 * qi + r, ks, h), row_frag(Ql, 32 * qi + r, ks, h), row_frag(Kh, j, ks, h), row_frag(Kl, j, ks, h));
          p = mma3(p, row_frag(Dh, 32 * qi + r, ks, h), row_frag(Dl, 32 * qi + r, ks, h), row_frag(Vh, j, ks, h), row_frag(Vl, j, ks, h));
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = 32 * qi + mma32_row(e, h);
          const float4 info = rowinfo[i];
          float sv = fmaf(s[e], kScale, bias_s[jb + (i >> 3) * 15 + (i & 7)]);
          sv += (__float_as_int(info.z) != regj) ? -100.0f : 0.f;
          const float pv_ = expf(sv - info.x);
          const float dsv = pv_ * (p[e] - info.y);
          s[e] = pv_;
          p[e] = dsv;
          accb[qi][kj][e] += dsv;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const HiLo ps = acc_split(s, ks), ds = acc_split(p, ks);
          dVt = mma3(dVt, tr_frag(Dh, 32 * qi + 16 * ks, lane, true), tr_frag(Dl, 32 * qi + 16 * ks, lane, true), ps.hi, ps.lo);
          dKt = mma3(dKt, tr_frag(Qh, 32 * qi + 16 * ks, lane, true), tr_frag(Ql, 32 * qi + 16 * ks, lane, true), ds.hi, ds.lo);
        }
        __syncthreads();                                       // the dS image (or a staging image over it) is free
#pragma unroll
        for (int e = 0; e < 16; ++e) {                         // dS tile [query][key], split
          const __bf16 hb = (__bf16)p[e];
          Xh[mma32_row(e, h) * LDK + r] = hb;
          Xl[mma32_row(e, h) * LDK + r] = (__bf16)(p[e] - (float)hb);
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          dq[qi] = mma3(dq[qi], row_frag(Xh, r, ks, h), row_frag(Xl, r, ks, h), tr_frag(Kh, 32 * kj + 16 * ks, lane, false),
                        tr_frag(Kl, 32 * kj + 16 * ks, lane, false));
      }
      // dK, dV of keys 32 kj .. 32 kj + 31: [e][key] accumulators -> fp32 LDS [key][e] -> 16-byte chunks
      __syncthreads();
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        *reinterpret_cast<float4*>(Xf + r * LDO + 8 * g4 + 4 * h) =
            make_float4(dKt[4 * g4] * kScale, dKt[4 * g4 + 1] * kScale, dKt[4 * g4 + 2] * kScale, dKt[4 * g4 + 3] * kScale);
        *reinterpret_cast<float4*>(Xf + (32 + r) * LDO + 8 * g4 + 4 * h) = make_float4(dVt[4 * g4], dVt[4 * g4 + 1], dVt[4 * g4 + 2], dVt[4 * g4 + 3]);
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; ++t) {                            // keys 32 kj + (lane >> 3) + 8 t = token slot 4 kj + t of row[]
        const int rl = (lane >> 3) + 8 * t;
        float* dst = dqkv + row[4 * kj + t] * ld + col;
        stream_store(dst + g.C, *reinterpret_cast<const float4*>(Xf + rl * LDO + 4 * (lane & 7)));
        stream_store(dst + 2 * g.C, *reinterpret_cast<const float4*>(Xf + (32 + rl) * LDO + 4 * (lane & 7)));
      }
    }
    // dQ: [query][e] accumulators (lane = e) -> fp32 LDS -> 16-byte chunks
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int e = 0; e < 16; ++e) Xf[(32 * qi + mma32_row(e, h)) * LDO + r] = dq[qi][e] * kScale;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it)
      stream_store(dqkv + row[it] * ld + col, *reinterpret_cast<const float4*>(Xf + ((lane >> 3) + 8 * it) * LDO + 4 * (lane & 7)));
  }
  // relative-position-bias gradient: as in attn_fast.hip (private bin copies filled by read-add-write rounds)
  __syncthreads();
  float* scr = reinterpret_cast<float*>(tiles);                // 8 copies x 232 floats
  for (int t = lane; t < 8 * 232; t += 64) scr[t] = 0.f;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) accb[0][0][e] += accb[1][1][e];                              // same (dy, dx) per lane
#pragma unroll
  for (int kb = 0; kb < 48; kb += 4) {
    float cur[4];
    int idx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int grp = (kb + u) >> 4, e = (kb + u) & 15;
      const int i = 32 * (grp == 1) + mma32_row(e, h), j = 32 * (grp == 2) + r;
      idx[u] = (2 * u + h) * 232 + ((i >> 3) - (j >> 3) + 7) * 15 + ((i & 7) - (j & 7) + 7);
      cur[u] = scr[idx[u]];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int grp = (kb + u) >> 4, e = (kb + u) & 15;
      scr[idx[u]] = cur[u] + accb[grp == 1][grp == 2][e];
    }
    __syncthreads();
  }
  for (int t = lane; t < 225; t += 64) {
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) sum += scr[c * 232 + t];
    if (part) part[((int64_t)chunk * g.heads + head) * 232 + t] = sum;
    else atomicAdd(dtable + t * g.heads + head, sum);
  }
}

}  // namespace

int wm_attn_split_fwd(const float* qkv, const float* table, float* out, float* lse, int B, int hw, int C, int shift, hipStream_t st) {
  Geom g{hw, shift, hw / 8, C, C / 32};
  const int nwin = B * g.nw * g.nw;
  int blocks, nchunk;
  grid_for(nwin, g.heads, 4, blocks, nchunk);                  // 31 KB of LDS per wave: 4 - 5 resident per CU
  hipLaunchKernelGGL(attn_fwd_split_kernel, dim3(blocks), dim3(64), 0, st, qkv, table, out, lse, g, nwin, nchunk);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

int wm_attn_split_bwd(const float* qkv, const float* table, const float* out, const float* lse, const float* dout, float* dqkv,
                      float* dtable, float* scratch, WmReduceJob* defer, int B, int hw, int C, int shift, hipStream_t st) {
  Geom g{hw, shift, hw / 8, C, C / 32};
  const int nwin = B * g.nw * g.nw;
  int blocks, nchunk;
  grid_for(nwin, g.heads, 3, blocks, nchunk);                  // 52 KB of LDS per wave: 3 resident per CU
  hipLaunchKernelGGL(attn_bwd_split_kernel, dim3(blocks), dim3(64), 0, st, qkv, table, out, lse, dout, dqkv, dtable, scratch, g,
                     nwin, nchunk);
  WM_CHECK_LAUNCH();
  if (defer) {                                                 // one entry: dtable[t][head] out of part[chunk][head][232]
    WmReduceJob jb{};
    if (scratch) {
      jb.src = scratch; jb.pstride = (int64_t)g.heads * 232; jb.src_ld = 232; jb.count = nchunk; jb.rows = g.heads; jb.cols = 225;
      jb.dst = dtable; jb.dst_rs = 1; jb.dst_cs = g.heads;
    }
    *defer = jb;
    if (scratch) return WM_OK;
  }
  if (scratch) return wm_attn_bias_reduce(scratch, dtable, g.heads, nchunk, st);
  return WM_OK;
}
