// Downsample (Conv2d 4 x 4, stride 2, padding 1, C -> 2C: reference uformerWM/model.py:759-775) WITHOUT an im2col image.
//
// Pad the image by one pixel, cut it into 2 x 2 pixel blocks of 4 C values ((sy, sx) sub-pixel major, channel minor) and put
// block (Y, X) at position (Y + 1, X + 1) of a grid of Wg x Wg rows per image, Wg = n + 2, n = hw / 2.  The convolution is then a
// 2 x 2, stride-1 VALID correlation over blocks,
//     y[Y][X] = sum_{by, bx in {0, 1}} W'[by][bx] . xg[(Y + 1 + by, X + 1 + bx)],      tap (ky, kx) = (2 by + sy, 2 bx + sx),
// and with the output gradient on the SAME grid (dy[Y][X] at (Y + 1, X + 1), zero border) every operand row of the three
// products lies at a constant row offset {0, 1, Wg, Wg + 1} from its partner's (WmGemm::a_grid_n / a_jump_k / b_seg_cols):
//     forward  y[m]   = [xg[f(m)], xg[f(m) + 1], xg[f(m) + Wg], xg[f(m) + Wg + 1]] . w_fwd^T       K = 16 C, one pointer jump
//     dX       dxg[g] = [dyg[g - Wg - 1], dyg[g - Wg], dyg[g - 1], dyg[g]] . w_dx^T                K = 8 C,  one pointer jump
//     dW       dw'    = dyg^T . xg[. + delta]  over the grid rows as the token axis (border rows of dyg are zero)
// The kernels here are the layout passes: x -> xg, dy -> dyg, dxg -> dx, and the 16-tap permutations of the weight.
// Buffers: xg  [B Wg^2 + Wg + 2 rows][4 C] bf16 (the tail rows are zero: the weight gradient reads up to Wg + 1 rows past a token)
//          dyg [Wg + 1 + B Wg^2 rows][2 C] bf16 (the FRONT rows are zero: dX reads up to Wg + 1 rows before a grid row)
//          dxg [B Wg^2 rows][4 C] bf16
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

__device__ __forceinline__ uint2 pack4f(const float4 v) { return make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)); }

__global__ __launch_bounds__(256) void x_to_grid_kernel(const float* __restrict__ x, __bf16* __restrict__ xg, int B, int hw, int C,
                                                        int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cq = C >> 2, n = hw >> 1, Wg = n + 2, G = Wg * Wg;
  const int c4 = (int)(idx % cq), s = (int)((idx / cq) & 3);
  const int64_t g = idx / C;                                       // (4 sub-pixels x C / 4 threads per grid row)
  uint2 v = make_uint2(0u, 0u);
  if (g < (int64_t)B * G) {
    const int b = (int)(g / G), t = (int)(g - (int64_t)b * G), gy = t / Wg, gx = t - gy * Wg;
    const int r = 2 * gy - 3 + (s >> 1), q = 2 * gx - 3 + (s & 1);
    if (r >= 0 && r < hw && q >= 0 && q < hw)
      v = pack4f(*reinterpret_cast<const float4*>(x + (((int64_t)b * hw + r) * hw + q) * C + 4 * c4));
  }
  *reinterpret_cast<uint2*>(xg + g * 4 * C + s * C + 4 * c4) = v;
}

template <typename DT>
__global__ __launch_bounds__(256) void dy_to_grid_kernel(const DT* __restrict__ dy, __bf16* __restrict__ dyg, int B, int hw, int C2,
                                                         int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cq = C2 >> 2, n = hw >> 1, Wg = n + 2, G = Wg * Wg;
  const int c4 = (int)(idx % cq);
  const int64_t row = idx / cq, g = row - (Wg + 1);                // (Wg + 1 zero rows in front)
  uint2 v = make_uint2(0u, 0u);
  if (g >= 0) {
    const int b = (int)(g / G), t = (int)(g - (int64_t)b * G), gy = t / Wg, gx = t - gy * Wg;
    if (gy >= 1 && gy <= n && gx >= 1 && gx <= n)
      v = pack4f(IO<DT>::ld4(dy + (((int64_t)b * n + gy - 1) * n + gx - 1) * C2 + 4 * c4));
  }
  *reinterpret_cast<uint2*>(dyg + row * C2 + 4 * c4) = v;
}

__global__ __launch_bounds__(256) void grid_to_dx_kernel(const __bf16* __restrict__ dxg, float* __restrict__ dx, int B, int hw, int C,
                                                         int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cq = C >> 2, n = hw >> 1, Wg = n + 2;
  const int c4 = (int)(idx % cq);
  const int64_t pix = idx / cq;
  const int q = (int)(pix % hw), r = (int)((pix / hw) % hw), b = (int)(pix / ((int64_t)hw * hw));
  const int rp = r + 1, qp = q + 1;                                // padded coordinates
  const int64_t g = (int64_t)b * Wg * Wg + ((rp >> 1) + 1) * Wg + (qp >> 1) + 1;
  const uint2 u = *reinterpret_cast<const uint2*>(dxg + g * 4 * C + ((rp & 1) * 2 + (qp & 1)) * C + 4 * c4);
  *reinterpret_cast<float4*>(dx + pix * C + 4 * c4) = make_float4(bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y));
}

// weight [2C][C][4][4] (fp32 parameter)  ->  w_fwd [2C][16 C]:  [o][((by 2 + bx) 4 + sy 2 + sx) C + c]
//                                            w_dx  [4C][8 C]:   [(sy 2 + sx) C + c][((1 - by) 2 + (1 - bx)) 2C + o]
// and back: grad[2C][C][4][4] += dw' (in w_fwd order)
template <int MODE>
__global__ __launch_bounds__(256) void conv_weight_kernel(const float* __restrict__ w, __bf16* __restrict__ w_fwd, __bf16* __restrict__ w_dx,
                                                          const float* __restrict__ dwp, float* __restrict__ grad, int C) {
  const int idx = blockIdx.x * 256 + threadIdx.x;                  // (o, t', c)
  if (idx >= 2 * C * 16 * C) return;
  const int c = idx % C, tp = (idx / C) & 15, o = idx / (16 * C);
  const int by = tp >> 3, bx = (tp >> 2) & 1, sy = (tp >> 1) & 1, sx = tp & 1;
  const int src = (o * C + c) * 16 + (2 * by + sy) * 4 + 2 * bx + sx;
  if constexpr (MODE == 0) {
    const __bf16 v = (__bf16)w[src];
    w_fwd[idx] = v;
    w_dx[(int64_t)((sy * 2 + sx) * C + c) * (8 * C) + ((1 - by) * 2 + (1 - bx)) * 2 * C + o] = v;
  } else {
    grad[src] += dwp[idx];
  }
}

inline int blocks_for(int64_t n) { return (int)((n + 255) / 256); }

}  // namespace

extern "C" int64_t wm_conv4x4s2_xg_rows(int32_t B, int32_t hw) { const int64_t Wg = hw / 2 + 2; return B * Wg * Wg + Wg + 2; }
extern "C" int64_t wm_conv4x4s2_dyg_rows(int32_t B, int32_t hw) { const int64_t Wg = hw / 2 + 2; return Wg + 1 + B * Wg * Wg; }
extern "C" int64_t wm_conv4x4s2_grid_rows(int32_t B, int32_t hw) { const int64_t Wg = hw / 2 + 2; return B * Wg * Wg; }

extern "C" int wm_conv4x4s2_x_to_grid(const float* x, void* xg, int32_t B, int32_t hw, int32_t C, void* stream) {
  if (!x || !xg || B <= 0 || hw < 2 || (hw & 1) || C < 4 || (C & 3)) return WM_EINVAL;
  const int64_t total = wm_conv4x4s2_xg_rows(B, hw) * C;
  hipLaunchKernelGGL(x_to_grid_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, static_cast<__bf16*>(xg),
                     B, hw, C, total);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_conv4x4s2_dy_to_grid(const void* dy, int32_t dy_bf16, void* dyg, int32_t B, int32_t hw, int32_t C, void* stream) {
  if (!dy || !dyg || B <= 0 || hw < 2 || (hw & 1) || C < 4 || (C & 3)) return WM_EINVAL;
  const int64_t total = wm_conv4x4s2_dyg_rows(B, hw) * (2 * C / 4);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dy_bf16)
    hipLaunchKernelGGL((dy_to_grid_kernel<__bf16>), dim3(blocks_for(total)), dim3(256), 0, st, static_cast<const __bf16*>(dy),
                       static_cast<__bf16*>(dyg), B, hw, 2 * C, total);
  else
    hipLaunchKernelGGL((dy_to_grid_kernel<float>), dim3(blocks_for(total)), dim3(256), 0, st, static_cast<const float*>(dy),
                       static_cast<__bf16*>(dyg), B, hw, 2 * C, total);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_conv4x4s2_grid_to_dx(const void* dxg, float* dx, int32_t B, int32_t hw, int32_t C, void* stream) {
  if (!dxg || !dx || B <= 0 || hw < 2 || (hw & 1) || C < 4 || (C & 3)) return WM_EINVAL;
  const int64_t total = (int64_t)B * hw * hw * (C / 4);
  hipLaunchKernelGGL(grid_to_dx_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const __bf16*>(dxg), dx, B, hw, C, total);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_conv4x4s2_weights(const float* w, void* w_fwd, void* w_dx, int32_t C, void* stream) {
  if (!w || !w_fwd || !w_dx || C < 4) return WM_EINVAL;
  hipLaunchKernelGGL((conv_weight_kernel<0>), dim3(blocks_for(32 * C * C)), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                     static_cast<__bf16*>(w_fwd), static_cast<__bf16*>(w_dx), nullptr, nullptr, C);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_conv4x4s2_dweight(const float* dw_fwd_order, float* grad, int32_t C, void* stream) {
  if (!dw_fwd_order || !grad || C < 4) return WM_EINVAL;
  hipLaunchKernelGGL((conv_weight_kernel<1>), dim3(blocks_for(32 * C * C)), dim3(256), 0, static_cast<hipStream_t>(stream), nullptr, nullptr,
                     nullptr, dw_fwd_order, grad, C);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
