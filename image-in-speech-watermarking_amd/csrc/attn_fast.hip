// Window attention, bf16-storage fast path (mode 1 with bf16 q|k|v / out / gradients).
// Reference semantics: WindowAttention.forward (model.py:523-551) + shift / window partition / mask (model.py:954-1013),
// as in attn.hip; this file only changes HOW the same numbers are produced:
//
//  * one wave owns one (window, head) at a time and walks a strided list of windows (persistent grid sized to the
//    number of resident waves), so the relative-position-bias gradient is reduced in registers across windows;
//  * every global access is a 16-byte chunk of a token's 64-byte head row; the chunks of the NEXT window are loaded
//    into registers before the current window's MFMAs start (the compiler does not hoist loads over LDS stores by
//    itself: the previous kernels waited on every single load);
//  * scores never touch LDS.  Forward computes S^T = K Q^T (query on the lane) so the row softmax is a register
//    reduction plus one cross-half exchange, and exp(S^T) is directly the B operand of O^T = V^T P^T.  Backward
//    computes S = Q K^T (key on the lane): P = exp(S - lse[i]) needs no reduction (the forward saved the row
//    log-sum-exp), rowsum(dP * P) = dO . O, and the P / dS accumulators are the B operands of dV^T = dO^T P and
//    dK^T = Q^T dS; only dS crosses LDS once (for dQ = dS K).
//    Accumulator-as-operand: k slot (h, j) of a 16-deep step s carries row 16s + 8(j>>2) + 4h + (j&3); the A operand is
//    fetched with the matching permuted transposing LDS read (ds_read_b64_tr_b16).
#include "common.h"
#include "../../include/wm_kernels.h"

#include "attn_geom.h"

namespace {
using namespace attnk;

__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int ks) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (__bf16)a[8 * ks + j];
  return f;
}
__device__ __forceinline__ float dot8(u32x4 a, u32x4 b) {
  return bf_lo(a.x) * bf_lo(b.x) + bf_hi(a.x) * bf_hi(b.x) + bf_lo(a.y) * bf_lo(b.y) + bf_hi(a.y) * bf_hi(b.y) +
         bf_lo(a.z) * bf_lo(b.z) + bf_hi(a.z) * bf_hi(b.z) + bf_lo(a.w) * bf_lo(b.w) + bf_hi(a.w) * bf_hi(b.w);
}

// Chunk c = lane + 64 * it of a [64 tokens][32] tile: token (lane >> 2) + 16 * it, 16-byte part lane & 3.
// (plain arrays + always-executed loads: aggregates assigned under a condition end up in scratch memory)
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

// ------------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(64) void attn_fwd_fast_kernel(const __bf16* __restrict__ qkv, const float* __restrict__ table,
                                                           __bf16* __restrict__ out, float* __restrict__ lse_g, Geom g,
                                                           int nwin, int nchunk) {
  __shared__ __attribute__((aligned(16))) __bf16 Qs[64 * LDK], Ks[64 * LDK], Vs[64 * LDK];
  __shared__ float bias_s[232];
  __shared__ int reg_s[64];
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int v = virtual_block(blockIdx.x, gridDim.x);
  const int head = v % g.heads, chunk = v / g.heads;
  const int64_t ld = 3 * (int64_t)g.C;
  const int col = head * 32 + 8 * (lane & 3);
  for (int t = lane; t < 225; t += 64) bias_s[t] = table[t * g.heads + head];

  if (chunk >= nwin) return;
  int nrow[4], row[4], b, wy, wx, nb, nwy, nwx;
  u32x4 pq[4], pk[4], pv[4];
  win_pos(chunk, g.nw, nb, nwy, nwx);
  chunk_rows(nrow, g, nb, nwy, nwx, lane);
  tile_load(pq, qkv, ld, col, nrow);
  tile_load(pk, qkv, ld, g.C + col, nrow);
  tile_load(pv, qkv, ld, 2 * g.C + col, nrow);
  for (int w = chunk; w < nwin; w += nchunk) {
    b = nb, wy = nwy, wx = nwx;
#pragma unroll
    for (int it = 0; it < 4; ++it) row[it] = nrow[it];
    __syncthreads();
    tile_to_lds(Qs, pq, lane);
    tile_to_lds(Ks, pk, lane);
    tile_to_lds(Vs, pv, lane);
    reg_s[lane] = g.region(wy, wx, lane);
    __syncthreads();
    {                                                         // next window (the last one re-reads itself: no branch)
      const int wn = w + nchunk < nwin ? w + nchunk : w;
      win_pos(wn, g.nw, nb, nwy, nwx);
      chunk_rows(nrow, g, nb, nwy, nwx, lane);
      tile_load(pq, qkv, ld, col, nrow);
      tile_load(pk, qkv, ld, g.C + col, nrow);
      tile_load(pv, qkv, ld, 2 * g.C + col, nrow);
    }
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
      if (lse_g && h == 0) lse_g[(int64_t)g.token_row(b, wy, wx, i) * g.heads + head] = m + __logf(sum);
      f32x16 o;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
      for (int kj = 0; kj < 2; ++kj)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vs, 32 * kj + 16 * ks, lane, true), acc_frag(s[kj], ks), o, 0, 0, 0);
      const float inv = 1.0f / sum;
      __syncthreads();                                        // Q rows of this query tile are consumed: reuse them for O
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<uint2*>(Qs + i * LDK + 8 * g4 + 4 * h) =
            make_uint2(pack_bf2(o[4 * g4] * inv, o[4 * g4 + 1] * inv), pack_bf2(o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv));
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it)
      stream_store(out + row[it] * (int64_t)g.C + col, *reinterpret_cast<const uint4*>(Qs + ((lane >> 2) + 16 * it) * LDK + 8 * (lane & 3)));
  }
}

// ----------------------------------------------------------------------------------------------------------- backward
__global__ __launch_bounds__(64) void attn_bwd_fast_kernel(const __bf16* __restrict__ qkv, const float* __restrict__ table,
                                                           const __bf16* __restrict__ out, const float* __restrict__ lse_g,
                                                           const __bf16* __restrict__ dout, __bf16* __restrict__ dqkv,
                                                           float* __restrict__ dtable, float* __restrict__ part, Geom g, int nwin, int nchunk) {
  __shared__ __attribute__((aligned(16))) __bf16 tiles[5 * 64 * LDK];
  __bf16 *Qs = tiles, *Ks = tiles + 64 * LDK, *Vs = tiles + 2 * 64 * LDK, *Ds = tiles + 3 * 64 * LDK, *Xs = tiles + 4 * 64 * LDK;
  __shared__ float4 rowinfo[64];                              // lse, delta = dO . O, mask region (int bits), -
  __shared__ float bias_s[232];
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int v = virtual_block(blockIdx.x, gridDim.x);
  const int head = v % g.heads, chunk = v / g.heads;
  const int64_t ld = 3 * (int64_t)g.C;
  const int col = head * 32 + 8 * (lane & 3);
  for (int t = lane; t < 225; t += 64) bias_s[t] = table[t * g.heads + head];
  float accb[2][2][16];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
      for (int e = 0; e < 16; ++e) accb[a][b2][e] = 0.f;

  int nrow[4], row[4], wy, wx, nb, nwy, nwx;
  u32x4 pq[4], pk[4], pv[4], pd[4], po[4];
  float plse;
  const int w0 = chunk < nwin ? chunk : nwin - 1;             // (a block without work still loads valid addresses)
  win_pos(w0, g.nw, nb, nwy, nwx);
  chunk_rows(nrow, g, nb, nwy, nwx, lane);
  tile_load(pq, qkv, ld, col, nrow);
  tile_load(pk, qkv, ld, g.C + col, nrow);
  tile_load(pv, qkv, ld, 2 * g.C + col, nrow);
  tile_load(pd, dout, g.C, col, nrow);
  tile_load(po, out, g.C, col, nrow);
  plse = lse_g[(int64_t)g.token_row(nb, nwy, nwx, lane) * g.heads + head];
  for (int w = chunk; w < nwin; w += nchunk) {
    wy = nwy, wx = nwx;
#pragma unroll
    for (int it = 0; it < 4; ++it) row[it] = nrow[it];
    __syncthreads();
    tile_to_lds(Qs, pq, lane);
    tile_to_lds(Ks, pk, lane);
    tile_to_lds(Vs, pv, lane);
    tile_to_lds(Ds, pd, lane);
    rowinfo[lane].x = plse;
    rowinfo[lane].z = __int_as_float(g.region(wy, wx, lane));
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float d = dot8(pd[it], po[it]);
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      if ((lane & 3) == 0) rowinfo[(lane >> 2) + 16 * it].y = d;
    }
    __syncthreads();
    {                                                         // next window (the last one re-reads itself: no branch)
      const int wn = w + nchunk < nwin ? w + nchunk : w;
      win_pos(wn, g.nw, nb, nwy, nwx);
      chunk_rows(nrow, g, nb, nwy, nwx, lane);
      tile_load(pq, qkv, ld, col, nrow);
      tile_load(pk, qkv, ld, g.C + col, nrow);
      tile_load(pv, qkv, ld, 2 * g.C + col, nrow);
      tile_load(pd, dout, g.C, col, nrow);
      tile_load(po, out, g.C, col, nrow);
      plse = lse_g[(int64_t)g.token_row(nb, nwy, nwx, lane) * g.heads + head];
    }
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
          accb[qi][kj][e] += dsv;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          dVt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ds, 32 * qi + 16 * ks, lane, true), acc_frag(s, ks), dVt, 0, 0, 0);
          dKt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qs, 32 * qi + 16 * ks, lane, true), acc_frag(p, ks), dKt, 0, 0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) Xs[mma32_row(e, h) * LDK + r] = (__bf16)p[e];          // dS tile [query][key]
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          dq[qi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Xs, r, ks, h), tr_frag(Ks, 32 * kj + 16 * ks, lane, false), dq[qi], 0, 0, 0);
      }
      // dK, dV of keys 32kj .. 32kj+31: [e][key] accumulators -> LDS [key][e] -> 16-byte chunks
      __syncthreads();
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        *reinterpret_cast<uint2*>(Xs + r * LDK + 8 * g4 + 4 * h) =
            make_uint2(pack_bf2(dKt[4 * g4] * kScale, dKt[4 * g4 + 1] * kScale), pack_bf2(dKt[4 * g4 + 2] * kScale, dKt[4 * g4 + 3] * kScale));
        *reinterpret_cast<uint2*>(Xs + (32 + r) * LDK + 8 * g4 + 4 * h) =
            make_uint2(pack_bf2(dVt[4 * g4], dVt[4 * g4 + 1]), pack_bf2(dVt[4 * g4 + 2], dVt[4 * g4 + 3]));
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int rl = (lane >> 2) + 16 * t;
        __bf16* dst = dqkv + row[2 * kj + t] * ld + col;
        stream_store(dst + g.C, *reinterpret_cast<const uint4*>(Xs + rl * LDK + 8 * (lane & 3)));
        stream_store(dst + 2 * g.C, *reinterpret_cast<const uint4*>(Xs + (32 + rl) * LDK + 8 * (lane & 3)));
      }
    }
    // dQ: [query][e] accumulators (lane = e) -> LDS -> 16-byte chunks
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int e = 0; e < 16; ++e) Xs[(32 * qi + mma32_row(e, h)) * LDK + r] = (__bf16)(dq[qi][e] * kScale);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it)
      stream_store(dqkv + row[it] * ld + col, *reinterpret_cast<const uint4*>(Xs + ((lane >> 2) + 16 * it) * LDK + 8 * (lane & 3)));
  }
  // relative-position-bias gradient: registers -> LDS bins -> one global atomic per bin.  (LDS float atomics cost
  // ~0.3 us per instruction here, so the bins are filled with plain read-add-write rounds instead: for one
  // accumulator slot the 32 lanes of a half-wave hit 32 distinct bins, and the 4 slots x 2 half-waves of a round each
  // own a private copy of the bins, so no two lanes of a round touch the same word.)
  __syncthreads();
  float* scr = reinterpret_cast<float*>(tiles);               // 8 copies x 232 floats
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
    // ~1000 blocks adding into the same 225 x heads words serialise in the L2 atomic units (15-20 us): with a scratch
    // buffer every block stores its bins and bias_grad_reduce_kernel sums them (deterministically)
    if (part) part[((int64_t)chunk * g.heads + head) * 232 + t] = sum;
    else atomicAdd(dtable + t * g.heads + head, sum);
  }
}

// dtable[t][head] += sum over chunks of part[chunk][head][t]; block (head, segment) sums `per` chunks with independent
// loads in flight and adds its partial with one atomic per bin (<= 32 adders per word)
__global__ __launch_bounds__(256) void bias_grad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dtable,
                                                               int heads, int nchunk, int per) {
  const int head = blockIdx.x, t = threadIdx.x;
  if (t >= 225) return;
  const int c0 = blockIdx.y * per, c1 = min(nchunk, c0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int64_t stride = (int64_t)heads * 232;
  const float* p = part + ((int64_t)c0 * heads + head) * 232 + t;
  int c = c0;
  for (; c + 4 <= c1; c += 4, p += 4 * stride) {
    s0 += p[0];
    s1 += p[stride];
    s2 += p[2 * stride];
    s3 += p[3 * stride];
  }
  for (; c < c1; ++c, p += stride) s0 += p[0];
  atomicAdd(dtable + t * heads + head, (s0 + s1) + (s2 + s3));
}

}  // namespace

int wm_attn_fast_fwd(const __bf16* qkv, const float* table, __bf16* out, float* lse, int B, int hw, int C, int shift,
                     hipStream_t st) {
  Geom g{hw, shift, hw / 8, C, C / 32};
  const int nwin = B * g.nw * g.nw;
  int blocks, nchunk;
  grid_for(nwin, g.heads, 8, blocks, nchunk);
  hipLaunchKernelGGL(attn_fwd_fast_kernel, dim3(blocks), dim3(64), 0, st, qkv, table, out, lse, g, nwin, nchunk);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

int wm_attn_fast_bwd(const __bf16* qkv, const float* table, const __bf16* out, const float* lse, const __bf16* dout,
                     __bf16* dqkv, float* dtable, float* scratch, WmReduceJob* defer, int B, int hw, int C, int shift,
                     hipStream_t st) {
  Geom g{hw, shift, hw / 8, C, C / 32};
  const int nwin = B * g.nw * g.nw;
  int blocks, nchunk;
  grid_for(nwin, g.heads, 4, blocks, nchunk);
  hipLaunchKernelGGL(attn_bwd_fast_kernel, dim3(blocks), dim3(64), 0, st, qkv, table, out, lse, dout, dqkv, dtable, scratch, g,
                     nwin, nchunk);
  WM_CHECK_LAUNCH();
  if (defer) {                                                // one entry: dtable[t][head] out of part[chunk][head][232]
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

// dtable[t][head] += the nchunk per-block bins of a backward launch that was not deferred (also used by attn_split.hip)
int wm_attn_bias_reduce(const float* scratch, float* dtable, int heads, int nchunk, hipStream_t st) {
  int segs = nchunk / 16 < 1 ? 1 : (nchunk / 16 > 32 ? 32 : nchunk / 16);
  const int per = (nchunk + segs - 1) / segs;
  segs = (nchunk + per - 1) / per;
  hipLaunchKernelGGL(bias_grad_reduce_kernel, dim3(heads, segs), dim3(256), 0, st, scratch, dtable, heads, nchunk, per);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
