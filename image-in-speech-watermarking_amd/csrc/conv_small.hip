// Direct convolutions for the thin layers at the model's edges, where channel counts are 1..64 and
// an MFMA GEMM would be all padding: InputProj (model.py:814), OutputProj (:846), stft_layer (:2305),
// ConvAutoencoder (:1700-1709), EncoderTransformerWM.conv2 (:1566) and the 2x2/16x8 max-pools.
// One thread per output (forward / dX) or one workgroup per weight element (dW): these layers are
// < 0.1 % of the step's FLOPs and bandwidth.
#include "common.h"
#include "../../include/wm_kernels.h"

// full-resolution 2 <-> 32/64 channel layers (conv_edge.hip); -1 = not such a layer
int wm_conv_edge_fwd(const WmConv& c, const float* x, const float* w, const float* b, float* y, hipStream_t st);
int wm_conv_edge_dx(const WmConv& c, const float* w, const float* y, const float* dy, float* dx, int accumulate,
                    hipStream_t st);
int wm_conv_edge_dw(const WmConv& c, const float* x, const float* y, const float* dy, float* dw, float* db, hipStream_t st);

namespace {

__device__ __forceinline__ float act_f(float v, int act) {
  switch (act) {
    case 1: return v > 0.f ? v : 0.f;
    case 2: return v > 0.f ? v : 0.01f * v;
    case 3: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}
// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
  switch (act) {
    case 1: return y > 0.f ? 1.f : 0.f;
    case 2: return y > 0.f ? 1.f : 0.01f;
    case 3: return y * (1.0f - y);
    default: return 1.f;
  }
}
__device__ __forceinline__ int64_t in_off(const WmConv& c, int b, int ci, int y, int x) {
  return c.in_nhwc ? (((int64_t)b * c.H + y) * c.W + x) * c.Cin + ci : (((int64_t)b * c.Cin + ci) * c.H + y) * c.W + x;
}
__device__ __forceinline__ int64_t out_off(const WmConv& c, int b, int co, int y, int x) {
  return c.out_nhwc ? (((int64_t)b * c.OH + y) * c.OW + x) * c.Cout + co
                    : (((int64_t)b * c.Cout + co) * c.OH + y) * c.OW + x;
}
// IT = int when the tensor has fewer than 2^31 elements: 64-bit div / mod costs ~100 instructions each, four per output
template <typename IT>
__device__ __forceinline__ void out_decode(const WmConv& c, IT idx, int& b, int& co, int& y, int& x) {
  if (c.out_nhwc) {
    co = (int)(idx % c.Cout); idx /= c.Cout; x = (int)(idx % c.OW); idx /= c.OW; y = (int)(idx % c.OH); b = (int)(idx / c.OH);
  } else {
    x = (int)(idx % c.OW); idx /= c.OW; y = (int)(idx % c.OH); idx /= c.OH; co = (int)(idx % c.Cout); b = (int)(idx / c.Cout);
  }
}
template <typename IT>
__device__ __forceinline__ void in_decode(const WmConv& c, IT idx, int& b, int& ci, int& y, int& x) {
  if (c.in_nhwc) {
    ci = (int)(idx % c.Cin); idx /= c.Cin; x = (int)(idx % c.W); idx /= c.W; y = (int)(idx % c.H); b = (int)(idx / c.H);
  } else {
    x = (int)(idx % c.W); idx /= c.W; y = (int)(idx % c.H); idx /= c.H; ci = (int)(idx % c.Cin); b = (int)(idx / c.Cin);
  }
}

template <typename IT>
__global__ void conv_fwd_kernel(const WmConv c, const float* __restrict__ x, const float* __restrict__ w,
                                const float* __restrict__ bias, float* __restrict__ y, int64_t total_) {
  const IT total = (IT)total_;
  for (IT idx = (IT)blockIdx.x * (IT)blockDim.x + (IT)threadIdx.x; idx < total; idx += (IT)gridDim.x * (IT)blockDim.x) {
    int b, co, oy, ox;
    out_decode<IT>(c, idx, b, co, oy, ox);
    float acc = bias ? bias[co] : 0.f;
    if (!c.transposed && c.in_nhwc && (c.Cin & 3) == 0) {
      // channel-contiguous input (OutputProj): 16-B loads along Cin; the weight index is wave-uniform for NCHW output
      for (int ky = 0; ky < c.KH; ++ky) {
        const int iy = oy * c.SH - c.PH + ky;
        if (iy < 0 || iy >= c.H) continue;
        for (int kx = 0; kx < c.KW; ++kx) {
          const int ix = ox * c.SW - c.PW + kx;
          if (ix < 0 || ix >= c.W) continue;
          const float4* xp = reinterpret_cast<const float4*>(x + (((int64_t)b * c.H + iy) * c.W + ix) * c.Cin);
          const float* wp = w + ((int64_t)co * c.Cin * c.KH + ky) * c.KW + kx;
          const int ws = c.KH * c.KW;
          for (int c4 = 0; c4 < c.Cin / 4; ++c4) {
            const float4 v = xp[c4];
            acc += v.x * wp[(4 * c4 + 0) * ws] + v.y * wp[(4 * c4 + 1) * ws] + v.z * wp[(4 * c4 + 2) * ws] +
                   v.w * wp[(4 * c4 + 3) * ws];
          }
        }
      }
    } else if (!c.transposed) {
      for (int ci = 0; ci < c.Cin; ++ci)
        for (int ky = 0; ky < c.KH; ++ky) {
          const int iy = oy * c.SH - c.PH + ky;
          if (iy < 0 || iy >= c.H) continue;
          for (int kx = 0; kx < c.KW; ++kx) {
            const int ix = ox * c.SW - c.PW + kx;
            if (ix < 0 || ix >= c.W) continue;
            acc += x[in_off(c, b, ci, iy, ix)] * w[((co * c.Cin + ci) * c.KH + ky) * c.KW + kx];
          }
        }
    } else {
      const int iy = oy / c.SH, ky = oy % c.SH, ix = ox / c.SW, kx = ox % c.SW;
      for (int ci = 0; ci < c.Cin; ++ci)
        acc += x[in_off(c, b, ci, iy, ix)] * w[((ci * c.Cout + co) * c.KH + ky) * c.KW + kx];
    }
    y[idx] = act_f(acc, c.act);
  }
}

template <typename IT>
__global__ void conv_dx_kernel(const WmConv c, const float* __restrict__ w, const float* __restrict__ y,
                               const float* __restrict__ dy, float* __restrict__ dx, int accumulate, int64_t total_) {
  const IT total = (IT)total_;
  for (IT idx = (IT)blockIdx.x * (IT)blockDim.x + (IT)threadIdx.x; idx < total; idx += (IT)gridDim.x * (IT)blockDim.x) {
    int b, ci, iy, ix;
    in_decode<IT>(c, idx, b, ci, iy, ix);
    float acc = 0.f;
    if (!c.transposed && c.out_nhwc && (c.Cout & 3) == 0 && c.SH == 1 && c.SW == 1) {
      // channel-contiguous output gradient (InputProj): 16-B loads of dy and y along Cout
      for (int ky = 0; ky < c.KH; ++ky) {
        const int oy = iy + c.PH - ky;
        if (oy < 0 || oy >= c.OH) continue;
        for (int kx = 0; kx < c.KW; ++kx) {
          const int ox = ix + c.PW - kx;
          if (ox < 0 || ox >= c.OW) continue;
          const int64_t o = (((int64_t)b * c.OH + oy) * c.OW + ox) * c.Cout;
          const float4* gp = reinterpret_cast<const float4*>(dy + o);
          const float4* yp = reinterpret_cast<const float4*>(y + o);
          const float* wp = w + ((int64_t)ci * c.KH + ky) * c.KW + kx;
          const int ws = c.Cin * c.KH * c.KW;
          for (int c4 = 0; c4 < c.Cout / 4; ++c4) {
            const float4 gv = gp[c4], yv = yp[c4];
            acc += gv.x * act_grad_from_out(yv.x, c.act) * wp[(4 * c4 + 0) * ws] +
                   gv.y * act_grad_from_out(yv.y, c.act) * wp[(4 * c4 + 1) * ws] +
                   gv.z * act_grad_from_out(yv.z, c.act) * wp[(4 * c4 + 2) * ws] +
                   gv.w * act_grad_from_out(yv.w, c.act) * wp[(4 * c4 + 3) * ws];
          }
        }
      }
    } else if (!c.transposed) {
      for (int ky = 0; ky < c.KH; ++ky) {
        const int ty = iy + c.PH - ky;
        if (ty < 0 || ty % c.SH) continue;
        const int oy = ty / c.SH;
        if (oy >= c.OH) continue;
        for (int kx = 0; kx < c.KW; ++kx) {
          const int tx = ix + c.PW - kx;
          if (tx < 0 || tx % c.SW) continue;
          const int ox = tx / c.SW;
          if (ox >= c.OW) continue;
          for (int co = 0; co < c.Cout; ++co) {
            const int64_t o = out_off(c, b, co, oy, ox);
            acc += dy[o] * act_grad_from_out(y[o], c.act) * w[((co * c.Cin + ci) * c.KH + ky) * c.KW + kx];
          }
        }
      }
    } else {
      for (int co = 0; co < c.Cout; ++co)
        for (int ky = 0; ky < c.KH; ++ky)
          for (int kx = 0; kx < c.KW; ++kx) {
            const int64_t o = out_off(c, b, co, iy * c.SH + ky, ix * c.SW + kx);
            acc += dy[o] * act_grad_from_out(y[o], c.act) * w[((ci * c.Cout + co) * c.KH + ky) * c.KW + kx];
          }
    }
    if (accumulate) dx[idx] += acc; else dx[idx] = acc;
  }
}

// 3x3 / stride 1 / pad 1 layers in NCHW with few channels (stft_layer 2 -> 4 -> 2 at 128 x 128, ConvAutoencoder 1 -> 16 and
// 16 -> 4): a thread owns ONE pixel and all NO output channels (forward) or all input channels (input gradient), so the
// nine taps of an input plane are read once per pixel instead of once per (pixel, channel), the weights are wave-uniform
// scalar loads and no index is divided more than once per pixel.
template <int NO>
__global__ __launch_bounds__(256) void conv3x3_nchw_fwd_kernel(const WmConv c, const float* __restrict__ x,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ y) {
  const int plane = c.H * c.W, p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (p >= plane) return;
  const int oy = p / c.W, ox = p - oy * c.W;
  float acc[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) acc[o] = bias ? bias[o] : 0.f;
  const float* xb = x + (int64_t)b * c.Cin * plane;
  for (int ci = 0; ci < c.Cin; ++ci) {
    float t[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy + ky - 1, ix = ox + kx - 1;
        const bool in = (unsigned)iy < (unsigned)c.H && (unsigned)ix < (unsigned)c.W;
        const float v = xb[ci * plane + (in ? iy * c.W + ix : p)];
        t[ky * 3 + kx] = in ? v : 0.f;
      }
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[o] += t[k] * w[(o * c.Cin + ci) * 9 + k];
  }
  float* yb = y + (int64_t)b * NO * plane + p;
#pragma unroll
  for (int o = 0; o < NO; ++o) yb[o * plane] = act_f(acc[o], c.act);
}

// dx[b, ci, p] = sum over co, taps of (dy * act'(y))[b, co, p + 1 - k] * w[co][ci][k]; NI = Cin
template <int NI>
__global__ __launch_bounds__(256) void conv3x3_nchw_dx_kernel(const WmConv c, const float* __restrict__ w,
                                                              const float* __restrict__ y, const float* __restrict__ dy,
                                                              float* __restrict__ dx, int accumulate) {
  const int plane = c.H * c.W, p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (p >= plane) return;
  const int iy = p / c.W, ix = p - iy * c.W;
  float acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = 0.f;
  const int64_t ob = (int64_t)b * c.Cout * plane;
  for (int co = 0; co < c.Cout; ++co) {
    float t[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int oy = iy + 1 - ky, ox = ix + 1 - kx;
        const bool in = (unsigned)oy < (unsigned)c.H && (unsigned)ox < (unsigned)c.W;
        const int64_t o = ob + co * plane + (in ? oy * c.W + ox : p);
        const float g = dy[o] * act_grad_from_out(y[o], c.act);
        t[ky * 3 + kx] = in ? g : 0.f;
      }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[i] += t[k] * w[(co * NI + i) * 9 + k];
  }
  float* db_ = dx + (int64_t)b * NI * plane + p;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    if (accumulate) db_[i * plane] += acc[i]; else db_[i * plane] = acc[i];
  }
}

// Weight / bias gradient of the same few-channel 3x3 NCHW layers: a thread walks pixels and keeps ALL NO x NI x 9 products and
// the NO bias sums in registers (at most 160 + 16), the workgroup folds them with wave shuffles + LDS and issues one atomic per
// weight.  (The tiled kernel below re-stages every 8 x 8 tile through LDS and spends two LDS reads per multiply-add.)
template <int NO, int NI>
__global__ __launch_bounds__(256) void conv3x3_nchw_dw_kernel(const WmConv c, const float* __restrict__ x,
                                                              const float* __restrict__ y, const float* __restrict__ dy,
                                                              float* __restrict__ dw, float* __restrict__ db, int per) {
  constexpr int NA = NO * NI * 9 + NO;
  __shared__ float red[4 * NA];
  const int plane = c.H * c.W, tid = threadIdx.x;
  float acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = 0.f;
  const int64_t n = (int64_t)c.B * plane;
  const int64_t q0 = (int64_t)blockIdx.x * per, q1 = q0 + per < n ? q0 + per : n;
  for (int64_t q = q0 + tid; q < q1; q += 256) {
    const int b = (int)(q / plane), p = (int)(q - (int64_t)b * plane), py = p / c.W, px = p - py * c.W;
    float g[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const int64_t oo = ((int64_t)b * NO + o) * plane + p;
      g[o] = dy[oo] * act_grad_from_out(y[oo], c.act);
      acc[NO * NI * 9 + o] += g[o];
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float* xp = x + ((int64_t)b * NI + i) * plane;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = py + ky - 1, ix = px + kx - 1;
          const bool in = (unsigned)iy < (unsigned)c.H && (unsigned)ix < (unsigned)c.W;
          const float v = xp[in ? iy * c.W + ix : p];
          const float t = in ? v : 0.f;
#pragma unroll
          for (int o = 0; o < NO; ++o) acc[(o * NI + i) * 9 + ky * 3 + kx] += g[o] * t;
        }
    }
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const float sum = wave_sum(acc[i]);
    if ((tid & 63) == 0) red[(tid >> 6) * NA + i] = sum;
  }
  __syncthreads();
  for (int i = tid; i < NA; i += 256) {
    const float sum = (red[i] + red[NA + i]) + (red[2 * NA + i] + red[3 * NA + i]);
    if (i < NO * NI * 9) atomicAdd(dw + i, sum);
    else if (db) atomicAdd(db + (i - NO * NI * 9), sum);
  }
}

// workgroup (x: weight element, + Cout extra for the bias; y: slice of the batch): one atomic per workgroup.  The batch is
// sliced because one workgroup per weight walked B * OH * OW positions alone (217 us per launch at B = 192).
__global__ __launch_bounds__(256) void conv_dw_kernel(const WmConv c, const float* __restrict__ x,
                                                      const float* __restrict__ y, const float* __restrict__ dy,
                                                      float* __restrict__ dw, float* __restrict__ db, int bper) {
  __shared__ float red[4];
  const int nW = c.Cout * c.Cin * c.KH * c.KW;
  const int e = blockIdx.x;
  const int b0 = blockIdx.y * bper, b1 = min(c.B, b0 + bper);
  float acc = 0.f;
  if (e < nW) {
    int t = e;
    const int kx = t % c.KW; t /= c.KW;
    const int ky = t % c.KH; t /= c.KH;
    int co, ci;
    if (!c.transposed) { ci = t % c.Cin; co = t / c.Cin; } else { co = t % c.Cout; ci = t / c.Cout; }
    if (!c.transposed) {
      const int n = (b1 - b0) * c.OH * c.OW;
      for (int p = threadIdx.x; p < n; p += 256) {
        const int ox = p % c.OW, oy = (p / c.OW) % c.OH, b = b0 + p / (c.OW * c.OH);
        const int iy = oy * c.SH - c.PH + ky, ix = ox * c.SW - c.PW + kx;
        if (iy < 0 || iy >= c.H || ix < 0 || ix >= c.W) continue;
        const int64_t o = out_off(c, b, co, oy, ox);
        acc += dy[o] * act_grad_from_out(y[o], c.act) * x[in_off(c, b, ci, iy, ix)];
      }
    } else {
      const int n = (b1 - b0) * c.H * c.W;
      for (int p = threadIdx.x; p < n; p += 256) {
        const int ix = p % c.W, iy = (p / c.W) % c.H, b = b0 + p / (c.W * c.H);
        const int64_t o = out_off(c, b, co, iy * c.SH + ky, ix * c.SW + kx);
        acc += dy[o] * act_grad_from_out(y[o], c.act) * x[in_off(c, b, ci, iy, ix)];
      }
    }
  } else {
    const int co = e - nW;
    const int n = (b1 - b0) * c.OH * c.OW;
    for (int p = threadIdx.x; p < n; p += 256) {
      const int ox = p % c.OW, oy = (p / c.OW) % c.OH, b = b0 + p / (c.OW * c.OH);
      const int64_t o = out_off(c, b, co, oy, ox);
      acc += dy[o] * act_grad_from_out(y[o], c.act);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    if (e < nW) atomicAdd(dw + e, s); else if (db) atomicAdd(db + (e - nW), s);
  }
}


// Weight/bias gradient of a 3x3, stride 1, pad 1 convolution (InputProj, OutputProj, stft_layer, ConvAutoencoder):
// one workgroup walks 8x8-pixel tiles, staging the input tile + halo and the activation-corrected output gradient
// in LDS; every thread owns up to MAXW weight elements (and the biases) and accumulates over the tile's 64 pixels.
// One atomic per weight per workgroup instead of one workgroup per weight re-reading the whole tensor.
constexpr int DW_MAXW = 5;                 // 256 threads x 5 >= 1152 + 64 (largest: OutputProj 2x64x3x3 + bias)
__global__ __launch_bounds__(256) void conv3x3_dw_kernel(const WmConv c, const float* __restrict__ x,
                                                         const float* __restrict__ y, const float* __restrict__ dy,
                                                         float* __restrict__ dw, float* __restrict__ db,
                                                         int tiles_per_block) {
  extern __shared__ float sm[];
  float* xin = sm;                          // [100][Cin]
  float* dpre = sm + 100 * c.Cin;           // [64][Cout]
  const int tid = threadIdx.x;
  const int nW = c.Cout * c.Cin * 9, nAll = nW + (db ? c.Cout : 0);
  const int tw = c.W / 8, th = c.H / 8, ntiles = c.B * th * tw;
  int wco[DW_MAXW], wxi[DW_MAXW];           // per owned element: output channel, LDS offset of its input tap
  float acc[DW_MAXW];
#pragma unroll
  for (int q = 0; q < DW_MAXW; ++q) {
    const int e = tid + 256 * q;
    acc[q] = 0.f;
    wco[q] = -1; wxi[q] = 0;
    if (e < nW) {
      const int kx = e % 3, ky = (e / 3) % 3, ci = (e / 9) % c.Cin;
      wco[q] = e / (9 * c.Cin);
      wxi[q] = (ky * 10 + kx) * c.Cin + ci;
    } else if (e < nAll) {
      wco[q] = e - nW;
      wxi[q] = -1;                          // bias: sum of dpre only
    }
  }
  for (int t = blockIdx.x * tiles_per_block; t < min(ntiles, (int)(blockIdx.x + 1) * tiles_per_block); ++t) {
    const int b = t / (th * tw), tr = t - b * th * tw, ty = tr / tw, tx = tr - ty * tw;
    __syncthreads();
    for (int i = tid; i < 100 * c.Cin; i += 256) {
      int pix, ci;
      if (c.in_nhwc) { ci = i % c.Cin; pix = i / c.Cin; } else { pix = i % 100; ci = i / 100; }
      const int gy = ty * 8 + pix / 10 - 1, gx = tx * 8 + pix % 10 - 1;
      xin[pix * c.Cin + ci] = (gy >= 0 && gy < c.H && gx >= 0 && gx < c.W) ? x[in_off(c, b, ci, gy, gx)] : 0.f;
    }
    for (int i = tid; i < 64 * c.Cout; i += 256) {
      int p, co;
      if (c.out_nhwc) { co = i % c.Cout; p = i / c.Cout; } else { p = i % 64; co = i / 64; }
      const int64_t o = out_off(c, b, co, ty * 8 + (p >> 3), tx * 8 + (p & 7));
      dpre[p * c.Cout + co] = dy[o] * act_grad_from_out(y[o], c.act);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < DW_MAXW; ++q) {
      if (wco[q] < 0) continue;
      float a = 0.f;
      if (wxi[q] >= 0) {
        for (int p = 0; p < 64; ++p) a += dpre[p * c.Cout + wco[q]] * xin[((p >> 3) * 10 + (p & 7)) * c.Cin + wxi[q]];
      } else {
        for (int p = 0; p < 64; ++p) a += dpre[p * c.Cout + wco[q]];
      }
      acc[q] += a;
    }
  }
#pragma unroll
  for (int q = 0; q < DW_MAXW; ++q) {
    const int e = tid + 256 * q;
    if (wco[q] < 0) continue;
    if (e < nW) atomicAdd(dw + e, acc[q]); else atomicAdd(db + (e - nW), acc[q]);
  }
}

__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int kh, int kw,
                                   int64_t total) {
  const int oh = H / kh, ow = W / kw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % ow), oy = (int)((idx / ow) % oh);
    const int64_t pl = idx / ((int64_t)ow * oh);
    const float* p = x + (pl * H + oy * kh) * W + ox * kw;
    float m = -INFINITY;
    for (int a = 0; a < kh; ++a)
      for (int b = 0; b < kw; ++b) m = fmaxf(m, p[a * W + b]);
    y[idx] = m;
  }
}
// gradient goes to the FIRST maximal element of each (non-overlapping) window, as ATen does
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                   int H, int W, int kh, int kw, int accumulate, int64_t total) {
  const int oh = H / kh, ow = W / kw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % ow), oy = (int)((idx / ow) % oh);
    const int64_t pl = idx / ((int64_t)ow * oh);
    const int64_t base = (pl * H + oy * kh) * W + ox * kw;
    float m = -INFINITY;
    int am = 0;
    for (int a = 0; a < kh; ++a)
      for (int b = 0; b < kw; ++b) {
        const float v = x[base + a * W + b];
        if (v > m) { m = v; am = a * kw + b; }
      }
    const float g = dy[idx];
    for (int a = 0; a < kh; ++a)
      for (int b = 0; b < kw; ++b) {
        const float v = (a * kw + b == am) ? g : 0.f;
        float* d = dx + base + a * W + b;
        if (accumulate) *d += v; else *d = v;
      }
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}
bool conv_ok(const WmConv* c) {
  if (!c || c->B <= 0 || c->Cin <= 0 || c->Cout <= 0 || c->KH <= 0 || c->KW <= 0 || c->SH <= 0 || c->SW <= 0) return false;
  if (c->transposed) {
    return c->KH == c->SH && c->KW == c->SW && c->PH == 0 && c->PW == 0 && c->OH == c->H * c->SH && c->OW == c->W * c->SW;
  }
  return c->OH == (c->H + 2 * c->PH - c->KH) / c->SH + 1 && c->OW == (c->W + 2 * c->PW - c->KW) / c->SW + 1;
}

}  // namespace

extern "C" int wm_conv_fwd(const WmConv* c, const float* x, const float* w, const float* b, float* y, void* stream) {
  if (!conv_ok(c) || !x || !w || !y) return WM_EINVAL;
  const int edge = wm_conv_edge_fwd(*c, x, w, b, y, static_cast<hipStream_t>(stream));
  if (edge >= 0) return edge;
  const bool nchw3 = !c->transposed && !c->in_nhwc && !c->out_nhwc && c->KH == 3 && c->KW == 3 && c->SH == 1 && c->SW == 1 &&
                     c->PH == 1 && c->PW == 1 && c->B <= 65535 && c->Cin <= 64;
  if (nchw3 && (c->Cout == 1 || c->Cout == 2 || c->Cout == 4 || c->Cout == 16)) {
    const dim3 grid((c->H * c->W + 255) / 256, c->B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (c->Cout) {
      case 1: hipLaunchKernelGGL(conv3x3_nchw_fwd_kernel<1>, grid, dim3(256), 0, st, *c, x, w, b, y); break;
      case 2: hipLaunchKernelGGL(conv3x3_nchw_fwd_kernel<2>, grid, dim3(256), 0, st, *c, x, w, b, y); break;
      case 4: hipLaunchKernelGGL(conv3x3_nchw_fwd_kernel<4>, grid, dim3(256), 0, st, *c, x, w, b, y); break;
      default: hipLaunchKernelGGL(conv3x3_nchw_fwd_kernel<16>, grid, dim3(256), 0, st, *c, x, w, b, y); break;
    }
    WM_CHECK_LAUNCH();
    return WM_OK;
  }
  const int64_t total = (int64_t)c->B * c->Cout * c->OH * c->OW;
  if (total + 8192 * 256 < (int64_t)1 << 31)
    hipLaunchKernelGGL(conv_fwd_kernel<int>, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), *c, x, w, b, y, total);
  else
    hipLaunchKernelGGL(conv_fwd_kernel<int64_t>, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), *c, x, w, b, y, total);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int wm_conv_bwd(const WmConv* c, const float* x, const float* w, const float* y, const float* dy,
                           float* dx, int32_t dx_accumulate, float* dw, float* db, void* stream) {
  if (!conv_ok(c) || !x || !w || !y || !dy) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dx) {
    const int edge = wm_conv_edge_dx(*c, w, y, dy, dx, dx_accumulate, st);
    if (edge > 0) return edge;
    const bool nchw3 = !c->transposed && !c->in_nhwc && !c->out_nhwc && c->KH == 3 && c->KW == 3 && c->SH == 1 && c->SW == 1 &&
                       c->PH == 1 && c->PW == 1 && c->B <= 65535 && c->Cout <= 64;
    if (edge < 0 && nchw3 && (c->Cin == 1 || c->Cin == 2 || c->Cin == 4 || c->Cin == 16)) {
      const dim3 grid((c->H * c->W + 255) / 256, c->B);
      switch (c->Cin) {
        case 1: hipLaunchKernelGGL(conv3x3_nchw_dx_kernel<1>, grid, dim3(256), 0, st, *c, w, y, dy, dx, dx_accumulate); break;
        case 2: hipLaunchKernelGGL(conv3x3_nchw_dx_kernel<2>, grid, dim3(256), 0, st, *c, w, y, dy, dx, dx_accumulate); break;
        case 4: hipLaunchKernelGGL(conv3x3_nchw_dx_kernel<4>, grid, dim3(256), 0, st, *c, w, y, dy, dx, dx_accumulate); break;
        default: hipLaunchKernelGGL(conv3x3_nchw_dx_kernel<16>, grid, dim3(256), 0, st, *c, w, y, dy, dx, dx_accumulate); break;
      }
      WM_CHECK_LAUNCH();
    } else if (edge < 0) {
      const int64_t total = (int64_t)c->B * c->Cin * c->H * c->W;
      if (total + 8192 * 256 < (int64_t)1 << 31)
        hipLaunchKernelGGL(conv_dx_kernel<int>, dim3(grid_for(total)), dim3(256), 0, st, *c, w, y, dy, dx, dx_accumulate, total);
      else
        hipLaunchKernelGGL(conv_dx_kernel<int64_t>, dim3(grid_for(total)), dim3(256), 0, st, *c, w, y, dy, dx, dx_accumulate, total);
      WM_CHECK_LAUNCH();
    }
  }
  if (dw) {
    const int edge = wm_conv_edge_dw(*c, x, y, dy, dw, db, st);
    if (edge >= 0) return edge;
  }
  if (dw && !c->transposed && !c->in_nhwc && !c->out_nhwc && c->KH == 3 && c->KW == 3 && c->SH == 1 && c->SW == 1 && c->PH == 1 &&
      c->PW == 1) {
    const int64_t n = (int64_t)c->B * c->H * c->W;
    int64_t per = (n + 511) / 512;                            // up to 512 workgroups (each ends in NA contended atomics), >= 1024 pixels each
    if (per < 1024) per = 1024;
    per = (per + 255) / 256 * 256;
    const dim3 grid((unsigned)((n + per - 1) / per));
#define WM_DW3(NO, NI) \
    { hipLaunchKernelGGL((conv3x3_nchw_dw_kernel<NO, NI>), grid, dim3(256), 0, st, *c, x, y, dy, dw, db, (int)per); WM_CHECK_LAUNCH(); return WM_OK; }
    if (c->Cout == 4 && c->Cin == 2) WM_DW3(4, 2)
    if (c->Cout == 2 && c->Cin == 4) WM_DW3(2, 4)
    if (c->Cout == 16 && c->Cin == 1) WM_DW3(16, 1)
    if (c->Cout == 1 && c->Cin == 1) WM_DW3(1, 1)
#undef WM_DW3
  }
  const bool tiled = dw && !c->transposed && c->KH == 3 && c->KW == 3 && c->SH == 1 && c->SW == 1 && c->PH == 1 &&
                     c->PW == 1 && (c->H % 8) == 0 && (c->W % 8) == 0 &&
                     c->Cout * c->Cin * 9 + c->Cout <= 256 * DW_MAXW && (100 * c->Cin + 64 * c->Cout) * 4 <= 60000;
  if (tiled) {
    const int ntiles = c->B * (c->H / 8) * (c->W / 8);
    const int tpb = ntiles >= 4096 ? 8 : (ntiles >= 512 ? 2 : 1);
    const size_t smem = sizeof(float) * (100 * c->Cin + 64 * c->Cout);
    hipLaunchKernelGGL(conv3x3_dw_kernel, dim3((ntiles + tpb - 1) / tpb), dim3(256), smem, st, *c, x, y, dy, dw, db, tpb);
    WM_CHECK_LAUNCH();
  } else if (dw) {
    const int nW = c->Cout * c->Cin * c->KH * c->KW, nx = nW + (db ? c->Cout : 0);
    // batch slices: about 4096 positions per workgroup, at most ~16 k workgroups, a slice never larger than 2^31 / 256 positions
    const int64_t plane = (int64_t)(c->transposed ? c->H * c->W : c->OH * c->OW);
    int bper = (int)(4096 / plane < 1 ? 1 : 4096 / plane);
    while ((int64_t)nx * ((c->B + bper - 1) / bper) > 16384 && bper < c->B) bper *= 2;
    if (plane * bper > (1 << 23)) return WM_EINVAL;
    hipLaunchKernelGGL(conv_dw_kernel, dim3(nx, (c->B + bper - 1) / bper), dim3(256), 0, st, *c, x, y, dy, dw, db, bper);
    WM_CHECK_LAUNCH();
  }
  return WM_OK;
}

extern "C" int wm_maxpool_fwd(const float* x, float* y, int32_t planes, int32_t H, int32_t W, int32_t kh, int32_t kw,
                              void* stream) {
  if (!x || !y || planes <= 0 || kh <= 0 || kw <= 0 || H % kh || W % kw) return WM_EINVAL;
  const int64_t total = (int64_t)planes * (H / kh) * (W / kw);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y,
                     H, W, kh, kw, total);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_maxpool_bwd(const float* x, const float* dy, float* dx, int32_t planes, int32_t H, int32_t W,
                              int32_t kh, int32_t kw, int32_t accumulate, void* stream) {
  if (!x || !dy || !dx || planes <= 0 || kh <= 0 || kw <= 0 || H % kh || W % kw) return WM_EINVAL;
  const int64_t total = (int64_t)planes * (H / kh) * (W / kw);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, dy,
                     dx, H, W, kh, kw, accumulate, total);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
