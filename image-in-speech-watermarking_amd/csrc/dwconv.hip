// LeFF depthwise 3x3 convolution on token-major activations [B, hw, hw, Ch] (reference
// model.py:703-709: rearrange -> Conv2d(groups=Ch, k=3, p=1) -> GELU -> rearrange).  The token-major
// layout makes the channel dim contiguous, so no b(hw)c <-> bchw transposes are materialised.
// Input is linear1's PRE-activation; GELU is applied once per staged element (tile + halo in LDS).
// Output is the pre-activation of the second GELU (applied by the consumer GEMM while staging).
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

constexpr int TP = 8;              // tile is TP x TP pixels
constexpr int HP = TP + 2;         // with halo
constexpr int TC = 64;             // channels per workgroup

template <typename ST>
__device__ __forceinline__ void load_halo_tile(float* tile, const ST* __restrict__ src, int b, int ty, int tx,
                                               int hw, int Ch, int cbase, bool gelu, int tid) {
  for (int idx = tid; idx < HP * HP * (TC / 4); idx += 256) {
    const int pix = idx / (TC / 4), c4 = idx % (TC / 4);
    const int ly = pix / HP, lx = pix % HP;
    const int gy = ty * TP + ly - 1, gx = tx * TP + lx - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy >= 0 && gy < hw && gx >= 0 && gx < hw) {
      v = IO<ST>::ld4(src + (((int64_t)b * hw + gy) * hw + gx) * Ch + cbase + 4 * c4);
      if (gelu) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
    }
    *reinterpret_cast<float4*>(tile + pix * TC + 4 * c4) = v;
  }
}

template <typename ST>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const ST* __restrict__ h1, const float* __restrict__ w,
                                                         const float* __restrict__ bias, ST* __restrict__ h2,
                                                         ST* __restrict__ a2, ST* __restrict__ g2, int B, int hw,
                                                         int Ch) {
  __shared__ __attribute__((aligned(16))) float tile[HP * HP * TC];
  const int tid = threadIdx.x, nt = hw / TP;
  const int t = blockIdx.x, b = t / (nt * nt), tr = t - b * nt * nt, ty = tr / nt, tx = tr - ty * nt;
  const int cbase = blockIdx.y * TC;
  load_halo_tile(tile, h1, b, ty, tx, hw, Ch, cbase, true, tid);
  const int c4 = tid & 15, pg = tid >> 4;
  float wr[4][9], br[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    br[cc] = bias[cbase + 4 * c4 + cc];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[cc][k] = w[(cbase + 4 * c4 + cc) * 9 + k];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int pix = pg * 4 + q, py = pix >> 3, px = pix & 7;
    float o[4] = {br[0], br[1], br[2], br[3]};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float4 a = *reinterpret_cast<const float4*>(tile + ((py + dy) * HP + px + dx) * TC + 4 * c4);
        o[0] += wr[0][dy * 3 + dx] * a.x; o[1] += wr[1][dy * 3 + dx] * a.y;
        o[2] += wr[2][dy * 3 + dx] * a.z; o[3] += wr[3][dy * 3 + dx] * a.w;
      }
    const int gy = ty * TP + py, gx = tx * TP + px;
    const int64_t off = (((int64_t)b * hw + gy) * hw + gx) * Ch + cbase + 4 * c4;
    if (h2) IO<ST>::st4(h2 + off, make_float4(o[0], o[1], o[2], o[3]));
    // second GELU and its derivative are materialised once here, from one shared exp(): evaluating them inside the
    // consumer GEMMs (A-operand staging of linear2 / dW2, epilogue of d_h2) repeated them per column tile and made
    // those launches VALU-bound
    if (a2 || g2) {
      float av[4], gv[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const float e = __expf(-0.5f * o[cc] * o[cc]);
        const float t = phi_tail_f(o[cc], e);
        const float cdf = o[cc] >= 0.f ? 1.0f - t : t;
        av[cc] = o[cc] * cdf;
        gv[cc] = cdf + o[cc] * 0.39894228040143268f * e;
      }
      if (a2) IO<ST>::st4(a2 + off, make_float4(av[0], av[1], av[2], av[3]));
      if (g2) IO<ST>::st4(g2 + off, make_float4(gv[0], gv[1], gv[2], gv[3]));
    }
  }
}

// dh1 = gelu'(h1) * sum_k w[k] * dh2[p - off_k];  dw[c][k] += sum_q gelu(h1)[q] * dh2[q - off_k];  db += sum dh2.
// Only dh2 needs a halo: the weight gradient is written as a sum over the CENTRE pixels q of a1[q] * dh2[q - off]
// (zero padding of the dh2 halo outside the image makes it equal to the usual sum over p of dh2[p] * a1[p + off]),
// so every tap needs ONE LDS read that feeds both the transposed convolution and the weight gradient, and
// gelu / gelu' of h1 are evaluated once per element from one shared exp().
template <typename ST>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const ST* __restrict__ h1, const ST* __restrict__ dh2,
                                                         const float* __restrict__ w, ST* __restrict__ dh1,
                                                         float* __restrict__ dw, float* __restrict__ db, int B,
                                                         int hw, int Ch) {
  __shared__ __attribute__((aligned(16))) float tg[HP * HP * TC];      // dh2 with halo
  __shared__ float wacc[TC * 10];
  const int tid = threadIdx.x, nt = hw / TP;
  const int b = blockIdx.x / nt, ty = blockIdx.x % nt;
  const int cbase = blockIdx.y * TC;
  const int c4 = tid & 15, pg = tid >> 4;
  float wr[4][9], aw[4][9], ab[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    ab[cc] = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { wr[cc][k] = w[(cbase + 4 * c4 + cc) * 9 + k]; aw[cc][k] = 0.f; }
  }
  for (int i = tid; i < TC * 10; i += 256) wacc[i] = 0.f;
  for (int tx = 0; tx < nt; ++tx) {
    __syncthreads();
    load_halo_tile(tg, dh2, b, ty, tx, hw, Ch, cbase, false, tid);
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const int pix = pg * 4 + q, py = pix >> 3, px = pix & 7;
      const int gy = ty * TP + py, gx = tx * TP + px;
      const int64_t off = (((int64_t)b * hw + gy) * hw + gx) * Ch + cbase + 4 * c4;
      const float4 hv = IO<ST>::ld4(h1 + off);
      float a1[4], gp[4];
      {
        const float xs[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const float e = __expf(-0.5f * xs[cc] * xs[cc]);
          const float t = phi_tail_f(xs[cc], e);
          const float cdf = xs[cc] >= 0.f ? 1.0f - t : t;
          a1[cc] = xs[cc] * cdf;
          gp[cc] = cdf + xs[cc] * 0.39894228040143268f * e;
        }
      }
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int k = dy * 3 + dx;
          const float4 g = *reinterpret_cast<const float4*>(tg + ((py + 2 - dy) * HP + px + 2 - dx) * TC + 4 * c4);
          o[0] += wr[0][k] * g.x; o[1] += wr[1][k] * g.y; o[2] += wr[2][k] * g.z; o[3] += wr[3][k] * g.w;
          aw[0][k] += a1[0] * g.x; aw[1][k] += a1[1] * g.y; aw[2][k] += a1[2] * g.z; aw[3][k] += a1[3] * g.w;
        }
      const float4 gc = *reinterpret_cast<const float4*>(tg + ((py + 1) * HP + px + 1) * TC + 4 * c4);
      ab[0] += gc.x; ab[1] += gc.y; ab[2] += gc.z; ab[3] += gc.w;
      IO<ST>::st4(dh1 + off, make_float4(o[0] * gp[0], o[1] * gp[1], o[2] * gp[2], o[3] * gp[3]));
    }
  }
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
    for (int k = 0; k < 9; ++k) atomicAdd(&wacc[(4 * c4 + cc) * 10 + k], aw[cc][k]);
    atomicAdd(&wacc[(4 * c4 + cc) * 10 + 9], ab[cc]);
  }
  __syncthreads();
  for (int i = tid; i < TC * 10; i += 256) {
    const int c = i / 10, k = i % 10;
    if (k < 9) atomicAdd(dw + (cbase + c) * 9 + k, wacc[i]);
    else atomicAdd(db + cbase + c, wacc[i]);
  }
}

bool ok(int B, int hw, int Ch) { return B > 0 && hw >= 8 && (hw & 7) == 0 && Ch >= 64 && (Ch & 63) == 0; }

}  // namespace

extern "C" int wm_dwconv_fwd(const void* h1, const float* w, const float* b, void* h2, void* a2, void* g2, int32_t io_bf16,
                             int32_t B, int32_t hw, int32_t Ch, void* stream) {
  if (!h1 || !w || !b || (!h2 && !a2 && !g2) || !ok(B, hw, Ch)) return WM_EINVAL;
  const int nt = hw / TP;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (io_bf16)
    hipLaunchKernelGGL(dwconv_fwd_kernel<__bf16>, dim3(B * nt * nt, Ch / TC), dim3(256), 0, st,
                       static_cast<const __bf16*>(h1), w, b, static_cast<__bf16*>(h2), static_cast<__bf16*>(a2), static_cast<__bf16*>(g2), B, hw, Ch);
  else
    hipLaunchKernelGGL(dwconv_fwd_kernel<float>, dim3(B * nt * nt, Ch / TC), dim3(256), 0, st,
                       static_cast<const float*>(h1), w, b, static_cast<float*>(h2), static_cast<float*>(a2), static_cast<float*>(g2), B, hw, Ch);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_dwconv_bwd(const void* h1, const void* dh2, const float* w, void* dh1, float* dw, float* db,
                             int32_t io_bf16, int32_t B, int32_t hw, int32_t Ch, void* stream) {
  if (!h1 || !dh2 || !w || !dh1 || !dw || !db || !ok(B, hw, Ch)) return WM_EINVAL;
  const int nt = hw / TP;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (io_bf16)
    hipLaunchKernelGGL(dwconv_bwd_kernel<__bf16>, dim3(B * nt, Ch / TC), dim3(256), 0, st, static_cast<const __bf16*>(h1),
                       static_cast<const __bf16*>(dh2), w, static_cast<__bf16*>(dh1), dw, db, B, hw, Ch);
  else
    hipLaunchKernelGGL(dwconv_bwd_kernel<float>, dim3(B * nt, Ch / TC), dim3(256), 0, st, static_cast<const float*>(h1),
                       static_cast<const float*>(dh2), w, static_cast<float*>(dh1), dw, db, B, hw, Ch);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
