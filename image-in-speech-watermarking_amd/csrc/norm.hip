// LayerNorm over the channel dim of token-major activations [rows, C], fused with the window
// "modulator" add of the decoder blocks (reference model.py:982,1017 and :996-997).
// A row is handled by L = min(64, C/4) lanes holding float4 slices (coalesced 16-B accesses);
// statistics are reduced with in-wave xor shuffles, no LDS.
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

constexpr float LN_EPS = 1e-5f;

__device__ __forceinline__ int window_pos(int64_t row, int hw, int shift) {
  const int t = (int)(row % ((int64_t)hw * hw));
  const int yy = t / hw, xx = t - yy * hw;
  const int i = (yy - shift + hw) % hw, j = (xx - shift + hw) % hw;
  return (i & 7) * 8 + (j & 7);
}

template <int L>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = L / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int C, typename YT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ mod,
                                                     YT* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int64_t rows, int hw, int shift) {
  constexpr int L = (C / 4 < 64) ? C / 4 : 64;
  constexpr int V = C / (4 * L);
  constexpr int RPW = 64 / L;                       // rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % L, rg = lane / L;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * RPW + rg;
  if (row >= rows) return;                          // whole lane groups exit together
  // every global operand is requested up front (gamma / beta / modulator too): a load issued after the statistics
  // would cost a second full memory round trip per row
  const float* mrow = mod ? mod + (int64_t)window_pos(row, hw, shift) * C : nullptr;
  float4 v[V], gmv[V], btv[V], mmv[V];
#pragma unroll
  for (int q = 0; q < V; ++q) v[q] = *reinterpret_cast<const float4*>(x + row * C + (q * L + sub) * 4);
#pragma unroll
  for (int q = 0; q < V; ++q) {
    gmv[q] = *reinterpret_cast<const float4*>(gamma + (q * L + sub) * 4);
    btv[q] = *reinterpret_cast<const float4*>(beta + (q * L + sub) * 4);
  }
  if (mrow) {
#pragma unroll
    for (int q = 0; q < V; ++q) mmv[q] = *reinterpret_cast<const float4*>(mrow + (q * L + sub) * 4);
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < V; ++q) s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
  const float mu = group_sum<L>(s) * (1.0f / C);
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < V; ++q) {
    const float a = v[q].x - mu, b = v[q].y - mu, c = v[q].z - mu, d = v[q].w - mu;
    ss += (a * a + b * b) + (c * c + d * d);
  }
  const float rs = 1.0f / sqrtf(group_sum<L>(ss) * (1.0f / C) + LN_EPS);
#pragma unroll
  for (int q = 0; q < V; ++q) {
    const int c = (q * L + sub) * 4;
    const float4 gm = gmv[q], bt = btv[q];
    float4 o;
    o.x = (v[q].x - mu) * rs * gm.x + bt.x;
    o.y = (v[q].y - mu) * rs * gm.y + bt.y;
    o.z = (v[q].z - mu) * rs * gm.z + bt.z;
    o.w = (v[q].w - mu) * rs * gm.w + bt.w;
    if (mrow) { o.x += mmv[q].x; o.y += mmv[q].y; o.z += mmv[q].z; o.w += mmv[q].w; }
    IO<YT>::st4(y + row * C + c, o);
  }
  if (sub == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma.
// dgamma / dbeta: per-lane partial sums over the rows this wave visits, then LDS + atomics.
constexpr int LNB_WAVES = 8;      // 512-thread blocks: 4x fewer same-address atomics per launch than 2048 x 256 threads
template <int C, typename DT>
__global__ __launch_bounds__(512) void ln_bwd_kernel(const DT* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     float* __restrict__ dx, __bf16* __restrict__ dxh,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ part, int64_t rows,
                                                     const float* __restrict__ hscale, int rows_per_scale) {
  constexpr int L = (C / 4 < 64) ? C / 4 : 64;
  constexpr int V = C / (4 * L);
  constexpr int RPW = 64 / L;
  __shared__ float red[2][LNB_WAVES][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % L, rg = lane / L;
  float4 ag[V], ab[V], gm[V];
#pragma unroll
  for (int q = 0; q < V; ++q) {
    ag[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    gm[q] = *reinterpret_cast<const float4*>(gamma + (q * L + sub) * 4);
  }
  const int64_t stride = (int64_t)gridDim.x * LNB_WAVES * RPW;
  for (int64_t row = ((int64_t)blockIdx.x * LNB_WAVES + wave) * RPW + rg; row < rows; row += stride) {
    const float mu = mean[row], rs = rstd[row];
    // the bf16 copy may carry the per-sample DropPath factor of the GEMMs that read it (dX and dW both want dy * s)
    const float hs = hscale ? hscale[row / rows_per_scale] : 1.f;
    float4 xh[V], g[V], dr[V];
    float s1 = 0.f, s2 = 0.f;
    if (dres) {                                               // requested with the other operands, used after the reductions
#pragma unroll
      for (int q = 0; q < V; ++q) dr[q] = *reinterpret_cast<const float4*>(dres + row * C + (q * L + sub) * 4);
    }
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const int c = (q * L + sub) * 4;
      const float4 xv = *reinterpret_cast<const float4*>(x + row * C + c);
      const float4 dv = IO<DT>::ld4(dy + row * C + c);
      xh[q] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
      g[q] = make_float4(dv.x * gm[q].x, dv.y * gm[q].y, dv.z * gm[q].z, dv.w * gm[q].w);
      s1 += (g[q].x + g[q].y) + (g[q].z + g[q].w);
      s2 += (g[q].x * xh[q].x + g[q].y * xh[q].y) + (g[q].z * xh[q].z + g[q].w * xh[q].w);
      ag[q].x += dv.x * xh[q].x; ag[q].y += dv.y * xh[q].y; ag[q].z += dv.z * xh[q].z; ag[q].w += dv.w * xh[q].w;
      ab[q].x += dv.x; ab[q].y += dv.y; ab[q].z += dv.z; ab[q].w += dv.w;
    }
    s1 = group_sum<L>(s1) * (1.0f / C);
    s2 = group_sum<L>(s2) * (1.0f / C);
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const int c = (q * L + sub) * 4;
      float4 o;
      o.x = rs * (g[q].x - s1 - xh[q].x * s2);
      o.y = rs * (g[q].y - s1 - xh[q].y * s2);
      o.z = rs * (g[q].z - s1 - xh[q].z * s2);
      o.w = rs * (g[q].w - s1 - xh[q].w * s2);
      if (dres) { o.x += dr[q].x; o.y += dr[q].y; o.z += dr[q].z; o.w += dr[q].w; }
      stream_store(dx + row * C + c, o);
      // the next GEMMs (dX, dW) round this gradient to bf16 while staging it anyway: a bf16 copy written here halves
      // what they read and lets them run as bf16 x bf16 instances (identical numbers)
      if (dxh) IO<__bf16>::st4(dxh + row * C + c, make_float4(o.x * hs, o.y * hs, o.z * hs, o.w * hs));
    }
  }
  // fold the RPW row groups of a wave (lanes sub, sub+L, ...) then the 4 waves
#pragma unroll
  for (int q = 0; q < V; ++q) {
#pragma unroll
    for (int o = L; o < 64; o <<= 1) {
      ag[q].x += __shfl_xor(ag[q].x, o); ag[q].y += __shfl_xor(ag[q].y, o);
      ag[q].z += __shfl_xor(ag[q].z, o); ag[q].w += __shfl_xor(ag[q].w, o);
      ab[q].x += __shfl_xor(ab[q].x, o); ab[q].y += __shfl_xor(ab[q].y, o);
      ab[q].z += __shfl_xor(ab[q].z, o); ab[q].w += __shfl_xor(ab[q].w, o);
    }
    if (rg == 0) {
      const int c = (q * L + sub) * 4;
      *reinterpret_cast<float4*>(&red[0][wave][c]) = ag[q];
      *reinterpret_cast<float4*>(&red[1][wave][c]) = ab[q];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 512) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < LNB_WAVES; ++w) { sg += red[0][w][c]; sb += red[1][w][c]; }
    if (part) {                                               // 512 workgroups adding into the same 2C words serialise in
      part[(int64_t)blockIdx.x * 2 * C + c] = sg;              // the L2 atomic units: store, let ln_param_reduce_kernel sum
      part[(int64_t)blockIdx.x * 2 * C + C + c] = sb;
    } else {
      atomicAdd(dgamma + c, sg);
      atomicAdd(dbeta + c, sb);
    }
  }
}

// dgamma[c] / dbeta[c] += sum over the P workgroups of part[p][0 / 1][c]; block (x: 256 outputs, y: segment of `per`)
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int P, int C, int per) {
  const int i = blockIdx.x * 256 + threadIdx.x, n = 2 * C;
  if (i >= n) return;
  const int p0 = blockIdx.y * per, p1 = min(P, p0 + per);
  const float* p = part + (int64_t)p0 * n + i;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int q = p0;
  for (; q + 4 <= p1; q += 4, p += 4 * n) {
    s0 += p[0];
    s1 += p[n];
    s2 += p[2 * n];
    s3 += p[3 * n];
  }
  for (; q < p1; ++q, p += n) s0 += p[0];
  atomicAdd(i < C ? dgamma + i : dbeta + (i - C), (s0 + s1) + (s2 + s3));
}

// dmod[pos, c] += sum over windows of dy[token(window, pos), c].  A thread owns V channels (bf16: 8 = one 16-byte load) and
// walks its windows four at a time with the four loads issued before the first add (one 8-byte load per iteration and two
// divisions per load kept the kernel at 2.4 TB/s).
template <typename DT, int V>
__global__ __launch_bounds__(256) void ln_dmod_kernel(const DT* __restrict__ dy, float* __restrict__ dmod,
                                                      int B, int hw, int C, int shift, int wchunk) {
  __shared__ float red[256 * V];
  const int pos = blockIdx.x, py = pos >> 3, px = pos & 7;
  const int cvn = C / V, nsub = 256 / cvn;
  const int cv = threadIdx.x % cvn, wsub = threadIdx.x / cvn;
  const int nw = hw / 8, nwin = B * nw * nw;
  const int w0 = blockIdx.y * wchunk, w1 = min(nwin, w0 + wchunk);
  float acc[V];
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = 0.f;
  auto addr = [&](int w) {
    const int b = w / (nw * nw), wr = w - b * nw * nw, wy = wr / nw, wx = wr - wy * nw;
    int yy = wy * 8 + py + shift, xx = wx * 8 + px + shift;
    yy -= yy >= hw ? hw : 0;
    xx -= xx >= hw ? hw : 0;
    return dy + ((int64_t)b * hw * hw + yy * hw + xx) * C + cv * V;
  };
  auto add = [&](const DT* p) {
    const float4 v0 = IO<DT>::ld4(p);
    acc[0] += v0.x; acc[1] += v0.y; acc[2] += v0.z; acc[3] += v0.w;
    if constexpr (V == 8) {
      const float4 v1 = IO<DT>::ld4(p + 4);
      acc[4] += v1.x; acc[5] += v1.y; acc[6] += v1.z; acc[7] += v1.w;
    }
  };
  int w = w0 + wsub;
  for (; w + 3 * nsub < w1; w += 4 * nsub) {
    const DT *p0 = addr(w), *p1 = addr(w + nsub), *p2 = addr(w + 2 * nsub), *p3 = addr(w + 3 * nsub);
    if constexpr (V == 8) {                                   // four 16-byte loads in flight
      const uint4 r0 = *reinterpret_cast<const uint4*>(p0), r1 = *reinterpret_cast<const uint4*>(p1);
      const uint4 r2 = *reinterpret_cast<const uint4*>(p2), r3 = *reinterpret_cast<const uint4*>(p3);
      const uint4 rr[4] = {r0, r1, r2, r3};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[0] += bf_lo(rr[q].x); acc[1] += bf_hi(rr[q].x); acc[2] += bf_lo(rr[q].y); acc[3] += bf_hi(rr[q].y);
        acc[4] += bf_lo(rr[q].z); acc[5] += bf_hi(rr[q].z); acc[6] += bf_lo(rr[q].w); acc[7] += bf_hi(rr[q].w);
      }
    } else {
      add(p0); add(p1); add(p2); add(p3);
    }
  }
  for (; w < w1; w += nsub) add(addr(w));
#pragma unroll
  for (int i = 0; i < V; ++i) red[threadIdx.x * V + i] = acc[i];
  __syncthreads();
  if (wsub == 0) {
    for (int s = 1; s < nsub; ++s)
#pragma unroll
      for (int i = 0; i < V; ++i) acc[i] += red[(s * cvn + cv) * V + i];
    float* d = dmod + (int64_t)pos * C + cv * V;
#pragma unroll
    for (int i = 0; i < V; ++i) atomicAdd(d + i, acc[i]);
  }
}

template <int C>
int ln_fwd_launch(const float* x, const float* g, const float* b, const float* mod, void* y, int y_bf16, float* mean,
                  float* rstd, int64_t rows, int hw, int shift, hipStream_t st) {
  constexpr int L = (C / 4 < 64) ? C / 4 : 64;
  constexpr int RPB = 4 * (64 / L);
  const int64_t blocks = (rows + RPB - 1) / RPB;
  if (y_bf16)
    hipLaunchKernelGGL((ln_fwd_kernel<C, __bf16>), dim3((unsigned)blocks), dim3(256), 0, st, x, g, b, mod,
                       static_cast<__bf16*>(y), mean, rstd, rows, hw, shift);
  else
    hipLaunchKernelGGL((ln_fwd_kernel<C, float>), dim3((unsigned)blocks), dim3(256), 0, st, x, g, b, mod,
                       static_cast<float*>(y), mean, rstd, rows, hw, shift);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
template <int C>
int ln_bwd_launch(const void* dy, int dy_bf16, const float* x, const float* g, const float* mean, const float* rstd,
                  const float* dres, float* dx, __bf16* dxh, float* dg, float* db, float* scratch, WmReduceJob* defer, int64_t rows,
                  const float* hscale, int rows_per_scale, hipStream_t st) {
  constexpr int L = (C / 4 < 64) ? C / 4 : 64;
  constexpr int RPB = LNB_WAVES * (64 / L);
  int64_t blocks = (rows + RPB - 1) / RPB;
  if (blocks > 512) blocks = 512;
  if (dy_bf16)
    hipLaunchKernelGGL((ln_bwd_kernel<C, __bf16>), dim3((unsigned)blocks), dim3(512), 0, st,
                       static_cast<const __bf16*>(dy), x, g, mean, rstd, dres, dx, dxh, dg, db, scratch, rows, hscale, rows_per_scale);
  else
    hipLaunchKernelGGL((ln_bwd_kernel<C, float>), dim3((unsigned)blocks), dim3(512), 0, st,
                       static_cast<const float*>(dy), x, g, mean, rstd, dres, dx, dxh, dg, db, scratch, rows, hscale, rows_per_scale);
  WM_CHECK_LAUNCH();
  if (defer) {                                                // two entries: dgamma, dbeta
    WmReduceJob jb{};
    defer[0] = jb;
    defer[1] = jb;
    if (scratch) {
      jb.src = scratch; jb.pstride = 2 * C; jb.src_ld = C; jb.count = (int)blocks; jb.rows = 1; jb.cols = C;
      jb.dst = dg; jb.dst_rs = C; jb.dst_cs = 1;
      defer[0] = jb;
      jb.src = scratch + C; jb.dst = db;
      defer[1] = jb;
      return WM_OK;
    }
  }
  if (scratch) {
    const int P = (int)blocks;
    int segs = P / 8 < 1 ? 1 : (P / 8 > 32 ? 32 : P / 8);
    const int per = (P + segs - 1) / segs;
    segs = (P + per - 1) / per;
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * C + 255) / 256, segs), dim3(256), 0, st, scratch, dg, db, P, C, per);
    WM_CHECK_LAUNCH();
  }
  return WM_OK;
}

}  // namespace

extern "C" int wm_layernorm_fwd(const float* x, const float* gamma, const float* beta, const float* modulator,
                                void* y, int32_t y_bf16, float* mean, float* rstd, int64_t rows, int32_t C,
                                int32_t hw, int32_t shift, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0) return WM_EINVAL;
  if (modulator && (hw <= 0 || (hw & 7) || rows % ((int64_t)hw * hw))) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (C) {
    case 32: return ln_fwd_launch<32>(x, gamma, beta, modulator, y, y_bf16, mean, rstd, rows, hw, shift, st);
    case 64: return ln_fwd_launch<64>(x, gamma, beta, modulator, y, y_bf16, mean, rstd, rows, hw, shift, st);
    case 128: return ln_fwd_launch<128>(x, gamma, beta, modulator, y, y_bf16, mean, rstd, rows, hw, shift, st);
    case 256: return ln_fwd_launch<256>(x, gamma, beta, modulator, y, y_bf16, mean, rstd, rows, hw, shift, st);
    case 512: return ln_fwd_launch<512>(x, gamma, beta, modulator, y, y_bf16, mean, rstd, rows, hw, shift, st);
    default: return WM_EINVAL;
  }
}

extern "C" int64_t wm_layernorm_bwd_scratch_floats(int32_t C) { return 512LL * 2 * C; }

extern "C" int wm_layernorm_bwd(const void* dy, int32_t dy_bf16, const float* x, const float* gamma, const float* mean,
                                const float* rstd, const float* dres, float* dx, void* dx_bf16, float* dgamma,
                                float* dbeta, float* dmod, float* scratch, WmReduceJob* defer, int64_t rows, int32_t C,
                                int32_t hw, int32_t shift, const float* copy_scale, int32_t rows_per_scale, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows <= 0) return WM_EINVAL;
  if (copy_scale && (!dx_bf16 || rows_per_scale <= 0)) return WM_EINVAL;
  if (dmod && (hw <= 0 || (hw & 7) || rows % ((int64_t)hw * hw) || C < 64 || C > 1024)) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc;
  switch (C) {
    case 32: rc = ln_bwd_launch<32>(dy, dy_bf16, x, gamma, mean, rstd, dres, dx, static_cast<__bf16*>(dx_bf16), dgamma, dbeta, scratch, defer, rows, copy_scale, rows_per_scale, st); break;
    case 64: rc = ln_bwd_launch<64>(dy, dy_bf16, x, gamma, mean, rstd, dres, dx, static_cast<__bf16*>(dx_bf16), dgamma, dbeta, scratch, defer, rows, copy_scale, rows_per_scale, st); break;
    case 128: rc = ln_bwd_launch<128>(dy, dy_bf16, x, gamma, mean, rstd, dres, dx, static_cast<__bf16*>(dx_bf16), dgamma, dbeta, scratch, defer, rows, copy_scale, rows_per_scale, st); break;
    case 256: rc = ln_bwd_launch<256>(dy, dy_bf16, x, gamma, mean, rstd, dres, dx, static_cast<__bf16*>(dx_bf16), dgamma, dbeta, scratch, defer, rows, copy_scale, rows_per_scale, st); break;
    case 512: rc = ln_bwd_launch<512>(dy, dy_bf16, x, gamma, mean, rstd, dres, dx, static_cast<__bf16*>(dx_bf16), dgamma, dbeta, scratch, defer, rows, copy_scale, rows_per_scale, st); break;
    default: return WM_EINVAL;
  }
  if (rc) return rc;
  if (dmod) {
    const int B = (int)(rows / ((int64_t)hw * hw));
    const int nwin = B * (hw / 8) * (hw / 8);
    int chunks = (nwin + 63) / 64;
    if (chunks > 32) chunks = 32;
    const int wchunk = (nwin + chunks - 1) / chunks;
    if (dy_bf16 && !(C & 7))
      hipLaunchKernelGGL((ln_dmod_kernel<__bf16, 8>), dim3(64, chunks), dim3(256), 0, st, static_cast<const __bf16*>(dy), dmod,
                         B, hw, C, shift, wchunk);
    else if (dy_bf16)
      hipLaunchKernelGGL((ln_dmod_kernel<__bf16, 4>), dim3(64, chunks), dim3(256), 0, st, static_cast<const __bf16*>(dy), dmod,
                         B, hw, C, shift, wchunk);
    else
      hipLaunchKernelGGL((ln_dmod_kernel<float, 4>), dim3(64, chunks), dim3(256), 0, st, static_cast<const float*>(dy), dmod,
                         B, hw, C, shift, wchunk);
    WM_CHECK_LAUNCH();
  }
  return WM_OK;
}
