// Pure data-movement kernels around the MFMA GEMMs: im2col / col2im for Downsample (conv 4x4 s2 p1,
// reference model.py:763), pixel (un)shuffle for Upsample (ConvTranspose 2x2 s2, model.py:789), the
// skip-connection channel concat (model.py:1225) and a generic strided 3-D gather used for the
// weight re-layouts.  All HBM-bound: float4 per lane, channel dim contiguous.
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

// IX: index type of the flat loops below (int when the element count fits: 64-bit div / mod costs ~100 instructions each and
// these kernels decode four to five of them per 16 bytes moved).  V: channels per thread (8 for bf16 columns: one 16-byte store).
template <typename OT, typename IX, int V>
__global__ void im2col_kernel(const float* __restrict__ x, OT* __restrict__ cols, int B, int hw, int C,
                              int64_t total_) {
  const IX total = (IX)total_;
  const int cvn = C / V, oh = hw / 2;
  // Every input pixel belongs to four patches.  Blocks are dealt round-robin to the 8 XCDs, so neighbouring patches (neighbouring
  // blocks) read the shared pixels through different L2s and HBM delivered x about twice (PMC 1.37 x the kernel's bytes):
  // XCD k takes the k-th eighth of each grid-stride window instead (the grid is a multiple of 8).
  const IX vb = (IX)(blockIdx.x & 7) * (IX)(gridDim.x >> 3) + (IX)(blockIdx.x >> 3);
  for (IX idx = vb * (IX)blockDim.x + (IX)threadIdx.x; idx < total; idx += (IX)gridDim.x * (IX)blockDim.x) {
    const int cv = (int)(idx % cvn);
    const int tap = (int)((idx / cvn) % 16);
    const IX m = idx / ((IX)cvn * 16);
    const int j = (int)(m % oh), i = (int)((m / oh) % oh), b = (int)(m / ((IX)oh * oh));
    const int yy = 2 * i - 1 + (tap >> 2), xx = 2 * j - 1 + (tap & 3);
    const bool in = yy >= 0 && yy < hw && xx >= 0 && xx < hw;
    const float* src = x + (((int64_t)b * hw + (in ? yy : 0)) * hw + (in ? xx : 0)) * C + V * cv;
    OT* dst = cols + ((int64_t)m * 16 + tap) * C + V * cv;
    float4 v0 = *reinterpret_cast<const float4*>(src);
    if (!in) v0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (V == 8) {
      float4 v1 = *reinterpret_cast<const float4*>(src + 4);
      if (!in) v1 = make_float4(0.f, 0.f, 0.f, 0.f);
      stream_store(dst, make_uint4(pack_bf2(v0.x, v0.y), pack_bf2(v0.z, v0.w), pack_bf2(v1.x, v1.y), pack_bf2(v1.z, v1.w)));
    } else {
      IO<OT>::st4(dst, v0);
    }
  }
}

template <typename IT, typename IX>
__global__ void col2im_kernel(const IT* __restrict__ dcols, float* __restrict__ dx, int B, int hw, int C,
                              int64_t total_) {
  const IX total = (IX)total_;
  const int c4n = C / 4, oh = hw / 2;
  for (IX idx = (IX)blockIdx.x * (IX)blockDim.x + (IX)threadIdx.x; idx < total; idx += (IX)gridDim.x * (IX)blockDim.x) {
    const int c4 = (int)(idx % c4n);
    const IX pix = idx / c4n;
    const int xx = (int)(pix % hw), yy = (int)((pix / hw) % hw), b = (int)(pix / ((IX)hw * hw));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int i = ((yy + 1) >> 1) - a, di = yy + 1 - 2 * i;
      if (i < 0 || i >= oh || di > 3) continue;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int j = ((xx + 1) >> 1) - c, dj = xx + 1 - 2 * j;
        if (j < 0 || j >= oh || dj > 3) continue;
        const int64_t m = ((int64_t)b * oh + i) * oh + j;
        const float4 v = IO<IT>::ld4(dcols + (m * 16 + di * 4 + dj) * C + 4 * c4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    stream_store(dx + (int64_t)pix * C + 4 * c4, acc);
  }
}

// y4[m, t*Co + o] (+ bias[o]) -> dst[b, 2i+di, 2j+dj, c0+o]
template <typename IT, typename IX>
__global__ void shuffle_kernel(const IT* __restrict__ y4, const float* __restrict__ bias, float* __restrict__ dst,
                               int B, int hw, int Co, int64_t ldd, int c0, int64_t total_) {
  const IX total = (IX)total_;
  const int o4n = Co / 4;
  for (IX idx = (IX)blockIdx.x * (IX)blockDim.x + (IX)threadIdx.x; idx < total; idx += (IX)gridDim.x * (IX)blockDim.x) {
    const int o4 = (int)(idx % o4n);
    const int t = (int)((idx / o4n) % 4);
    const IX m = idx / ((IX)o4n * 4);
    const int j = (int)(m % hw), i = (int)((m / hw) % hw), b = (int)(m / ((IX)hw * hw));
    float4 v = IO<IT>::ld4(y4 + ((int64_t)m * 4 + t) * Co + 4 * o4);
    if (bias) {
      const float4 bb = *reinterpret_cast<const float4*>(bias + 4 * o4);
      v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
    }
    const int64_t row = ((int64_t)b * 2 * hw + 2 * i + (t >> 1)) * 2 * hw + 2 * j + (t & 1);
    stream_store(dst + row * ldd + c0 + 4 * o4, v);
  }
}

template <typename OT, typename IX>
__global__ void unshuffle_kernel(const float* __restrict__ dsrc, OT* __restrict__ dy4, int B, int hw, int Co,
                                 int64_t lds, int c0, int64_t total_) {
  const IX total = (IX)total_;
  const int o4n = Co / 4;
  for (IX idx = (IX)blockIdx.x * (IX)blockDim.x + (IX)threadIdx.x; idx < total; idx += (IX)gridDim.x * (IX)blockDim.x) {
    const int o4 = (int)(idx % o4n);
    const int t = (int)((idx / o4n) % 4);
    const IX m = idx / ((IX)o4n * 4);
    const int j = (int)(m % hw), i = (int)((m / hw) % hw), b = (int)(m / ((IX)hw * hw));
    const int64_t row = ((int64_t)b * 2 * hw + 2 * i + (t >> 1)) * 2 * hw + 2 * j + (t & 1);
    IO<OT>::st4(dy4 + ((int64_t)m * 4 + t) * Co + 4 * o4, *reinterpret_cast<const float4*>(dsrc + row * lds + c0 + 4 * o4));
  }
}

__global__ void copy_cols_kernel(const float* __restrict__ src, int64_t lds, int s0, float* __restrict__ dst,
                                 int64_t ldd, int c0, int64_t rows, int C, int acc, int64_t total) {
  const int c4n = C / 4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    const int64_t r = idx / c4n;
    float4 v = *reinterpret_cast<const float4*>(src + r * lds + s0 + 4 * c4);
    float* d = dst + r * ldd + c0 + 4 * c4;
    if (acc) {
      const float4 o = *reinterpret_cast<const float4*>(d);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    stream_store(d, v);
  }
}

template <typename OT, typename IX>                          // IX = int when every offset fits (weight re-layouts: always)
__global__ void strided3_kernel(const float* __restrict__ in, OT* __restrict__ out, int n0, int n1, int n2,
                                int64_t si0_, int64_t si1_, int64_t si2_, int64_t so0_, int64_t so1_, int64_t so2_,
                                int atomic) {
  const IX si0 = (IX)si0_, si1 = (IX)si1_, si2 = (IX)si2_, so0 = (IX)so0_, so1 = (IX)so1_, so2 = (IX)so2_;
  const IX total = (IX)n0 * n1 * n2;
  for (IX idx = (IX)blockIdx.x * (IX)blockDim.x + (IX)threadIdx.x; idx < total; idx += (IX)gridDim.x * (IX)blockDim.x) {
    const int i2 = (int)(idx % n2), i1 = (int)((idx / n2) % n1), i0 = (int)(idx / ((IX)n1 * n2));
    const float v = in[i0 * si0 + i1 * si1 + i2 * si2];
    OT* d = out + (i0 * so0 + i1 * so1 + i2 * so2);
    if constexpr (sizeof(OT) == 4) { if (atomic) atomicAdd(d, v); else *d = v; } else { *d = (OT)v; }
  }
}

__global__ void axpbyz_kernel(float* __restrict__ out, float a, const float* __restrict__ x, float b,
                              const float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

__global__ void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    IO<__bf16>::st4(dst + 4 * i, *reinterpret_cast<const float4*>(src + 4 * i));
}

// Transposed bf16 copies of many [rows, cols] fp32 matrices of one arena in ONE launch (the dX GEMMs of the big-tile kernel
// read their weights K-contiguous: gemm_big.hip; the fused attention-half backward reads Wqkv^T: attn_fused.hip).  job =
// {element offset of the matrix in both arenas, rows, cols, first tile}; a workgroup moves one 64 x 64 tile through LDS
// (256-byte row reads, 128-byte row writes); rows, cols: multiples of 4, edge tiles are partial.
__global__ __launch_bounds__(256) void transpose_cast_jobs_kernel(const float* __restrict__ src, __bf16* __restrict__ dst,
                                                                  const int4* __restrict__ jobs, int njobs) {
  __shared__ float tile[64][65];
  int lo = 0, hi = njobs - 1;                                    // last job whose first tile is <= this block
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].w <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int4 jb = jobs[lo];
  const int rows = jb.y, cols = jb.z, tc = (cols + 63) / 64, t = blockIdx.x - jb.w, r0 = (t / tc) * 64, c0 = (t % tc) * 64;
  const float* s = src + jb.x + (int64_t)r0 * cols + c0;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;        // 16 threads x float4 per row, 16 rows per sweep
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + ty + 16 * q < rows && c0 + 4 * tx < cols) v = *reinterpret_cast<const float4*>(s + (int64_t)(ty + 16 * q) * cols + 4 * tx);
    tile[ty + 16 * q][4 * tx] = v.x; tile[ty + 16 * q][4 * tx + 1] = v.y; tile[ty + 16 * q][4 * tx + 2] = v.z; tile[ty + 16 * q][4 * tx + 3] = v.w;
  }
  __syncthreads();
  __bf16* d = dst + jb.x + (int64_t)c0 * rows + r0;              // dst[c][r]
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = ty + 16 * q;
    if (c0 + c >= cols || r0 + 4 * tx >= rows) continue;
    const float4 v = make_float4(tile[4 * tx][c], tile[4 * tx + 1][c], tile[4 * tx + 2][c], tile[4 * tx + 3][c]);
    IO<__bf16>::st4(d + (int64_t)c * rows + 4 * tx, v);
  }
}

// Range-style audio_scale of the reference's data preparation (audio_test.py:33-55 normalize_batch, :331-338, :559-571):
// min / max over a whole tile tensor, then an affine map into [lo, hi] (or back).  Two launches for the reduction (per-block
// partials, then one block), the statistics stay on the device.
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  __shared__ float sl[256], sh[256];
  sl[threadIdx.x] = lo; sh[threadIdx.x] = hi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sl[threadIdx.x] = fminf(sl[threadIdx.x], sl[threadIdx.x + o]); sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = sl[0]; part[2 * blockIdx.x + 1] = sh[0]; }
}
__global__ __launch_bounds__(256) void minmax_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  float lo = INFINITY, hi = -INFINITY;
  for (int i = threadIdx.x; i < nb; i += 256) { lo = fminf(lo, part[2 * i]); hi = fmaxf(hi, part[2 * i + 1]); }
  __shared__ float sl[256], sh[256];
  sl[threadIdx.x] = lo; sh[threadIdx.x] = hi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sl[threadIdx.x] = fminf(sl[threadIdx.x], sl[threadIdx.x + o]); sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = sl[0]; out[1] = sh[0]; }
}
// forward: out = (x - min) / (max - min) * (hi - lo) + lo      inverse: out = (x - lo) / (hi - lo) * (max - min) + min
__global__ void range_map_kernel(const float* __restrict__ x, float* __restrict__ out, const float* __restrict__ mm, float lo, float hi,
                                 int inverse, int64_t n) {
  const float dmin = mm[0], dmax = mm[1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    out[i] = inverse ? (v - lo) / (hi - lo) * (dmax - dmin) + dmin : (v - dmin) / (dmax - dmin) * (hi - lo) + lo;
  }
}

// Payload feature [B,4,64] tiled (1,16,8) into the first 512 channels of the bottleneck concat
// [B,64,1024] (reference model.py:2389,2411) and the matching gradient fold.
__global__ void msg_expand_fwd_kernel(const float* __restrict__ feat, float* __restrict__ dst, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % 512), r = (int)((idx / 512) % 64);
    const int64_t b = idx / (512 * 64);
    dst[(b * 64 + r) * 1024 + c] = feat[b * 256 + (r & 3) * 64 + (c & 63)];
  }
}
__global__ void msg_expand_bwd_kernel(const float* __restrict__ dcat, float* __restrict__ dfeat, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx % 64), a = (int)((idx / 64) % 4);
    const int64_t b = idx / 256;
    float acc = 0.f;
    for (int rr = 0; rr < 16; ++rr)
      for (int cc = 0; cc < 8; ++cc) acc += dcat[(b * 64 + rr * 4 + a) * 1024 + cc * 64 + e];
    dfeat[idx] += acc;
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// dst[i] = sum over r of float(src[r][i]): the data-parallel gradient exchange in bf16 (train.GradSync(exchange="bf16")) gathers every
// rank's bf16-rounded bucket and accumulates them HERE in fp32, in rank order on every rank (replicas stay bit-identical)
__global__ __launch_bounds__(256) void sum_bf16_rows_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, int rows, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= n) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < rows; ++r) {
    const uint4 v = *reinterpret_cast<const uint4*>(src + (int64_t)r * n + i);
    acc[0] += bf_lo(v.x); acc[1] += bf_hi(v.x); acc[2] += bf_lo(v.y); acc[3] += bf_hi(v.y);
    acc[4] += bf_lo(v.z); acc[5] += bf_hi(v.z); acc[6] += bf_lo(v.w); acc[7] += bf_hi(v.w);
  }
  *reinterpret_cast<float4*>(dst + i) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *reinterpret_cast<float4*>(dst + i + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

}  // namespace

extern "C" int wm_im2col_4x4s2(const float* x, void* cols, int32_t cols_bf16, int32_t B, int32_t hw, int32_t C,
                               void* stream) {
  if (!x || !cols || B <= 0 || hw < 2 || (hw & 1) || C < 4 || (C & 3)) return WM_EINVAL;
  const bool v8 = cols_bf16 && !(C & 7);
  const int64_t total = (int64_t)B * (hw / 2) * (hw / 2) * 16 * (C / (v8 ? 8 : 4));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = (grid_for(total) + 7) & ~7u;          // a multiple of 8: see the XCD mapping in the kernel
  const bool small = total + (int64_t)grid * 256 < ((int64_t)1 << 31);
  if (v8 && small)
    hipLaunchKernelGGL((im2col_kernel<__bf16, int, 8>), dim3(grid), dim3(256), 0, st, x, static_cast<__bf16*>(cols), B, hw, C, total);
  else if (v8)
    hipLaunchKernelGGL((im2col_kernel<__bf16, int64_t, 8>), dim3(grid), dim3(256), 0, st, x, static_cast<__bf16*>(cols), B, hw, C, total);
  else if (cols_bf16)
    hipLaunchKernelGGL((im2col_kernel<__bf16, int64_t, 4>), dim3(grid), dim3(256), 0, st, x, static_cast<__bf16*>(cols), B, hw, C, total);
  else
    hipLaunchKernelGGL((im2col_kernel<float, int64_t, 4>), dim3(grid), dim3(256), 0, st, x, static_cast<float*>(cols), B, hw, C, total);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_col2im_4x4s2(const void* dcols, int32_t dcols_bf16, float* dx, int32_t B, int32_t hw, int32_t C,
                               void* stream) {
  if (!dcols || !dx || B <= 0 || hw < 2 || (hw & 1) || C < 4 || (C & 3)) return WM_EINVAL;
  const int64_t total = (int64_t)B * hw * hw * (C / 4);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool small = total + (int64_t)grid_for(total) * 256 < ((int64_t)1 << 31);
  if (dcols_bf16 && small)
    hipLaunchKernelGGL((col2im_kernel<__bf16, int>), dim3(grid_for(total)), dim3(256), 0, st, static_cast<const __bf16*>(dcols), dx,
                       B, hw, C, total);
  else if (dcols_bf16)
    hipLaunchKernelGGL((col2im_kernel<__bf16, int64_t>), dim3(grid_for(total)), dim3(256), 0, st, static_cast<const __bf16*>(dcols),
                       dx, B, hw, C, total);
  else
    hipLaunchKernelGGL((col2im_kernel<float, int64_t>), dim3(grid_for(total)), dim3(256), 0, st, static_cast<const float*>(dcols), dx,
                       B, hw, C, total);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_pixel_shuffle2(const void* y4, int32_t y4_bf16, const float* bias, float* dst, int32_t B, int32_t hw,
                                 int32_t Co, int64_t ldd, int32_t c0, void* stream) {
  if (!y4 || !dst || B <= 0 || hw <= 0 || Co < 4 || (Co & 3) || (ldd & 3) || (c0 & 3) || !al16(dst)) return WM_EINVAL;
  const int64_t total = (int64_t)B * hw * hw * 4 * (Co / 4);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool small = total + (int64_t)grid_for(total) * 256 < ((int64_t)1 << 31);
  if (y4_bf16 && small)
    hipLaunchKernelGGL((shuffle_kernel<__bf16, int>), dim3(grid_for(total)), dim3(256), 0, st, static_cast<const __bf16*>(y4), bias,
                       dst, B, hw, Co, ldd, c0, total);
  else if (y4_bf16)
    hipLaunchKernelGGL((shuffle_kernel<__bf16, int64_t>), dim3(grid_for(total)), dim3(256), 0, st, static_cast<const __bf16*>(y4),
                       bias, dst, B, hw, Co, ldd, c0, total);
  else
    hipLaunchKernelGGL((shuffle_kernel<float, int64_t>), dim3(grid_for(total)), dim3(256), 0, st, static_cast<const float*>(y4), bias,
                       dst, B, hw, Co, ldd, c0, total);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_pixel_unshuffle2(const float* dsrc, void* dy4, int32_t dy4_bf16, int32_t B, int32_t hw, int32_t Co,
                                   int64_t lds, int32_t c0, void* stream) {
  if (!dsrc || !dy4 || B <= 0 || hw <= 0 || Co < 4 || (Co & 3) || (lds & 3) || (c0 & 3) || !al16(dsrc))
    return WM_EINVAL;
  const int64_t total = (int64_t)B * hw * hw * 4 * (Co / 4);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool small = total + (int64_t)grid_for(total) * 256 < ((int64_t)1 << 31);
  if (dy4_bf16 && small)
    hipLaunchKernelGGL((unshuffle_kernel<__bf16, int>), dim3(grid_for(total)), dim3(256), 0, st, dsrc, static_cast<__bf16*>(dy4), B,
                       hw, Co, lds, c0, total);
  else if (dy4_bf16)
    hipLaunchKernelGGL((unshuffle_kernel<__bf16, int64_t>), dim3(grid_for(total)), dim3(256), 0, st, dsrc, static_cast<__bf16*>(dy4),
                       B, hw, Co, lds, c0, total);
  else
    hipLaunchKernelGGL((unshuffle_kernel<float, int64_t>), dim3(grid_for(total)), dim3(256), 0, st, dsrc, static_cast<float*>(dy4), B,
                       hw, Co, lds, c0, total);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_copy_cols(const float* src, int64_t lds, int32_t s0, float* dst, int64_t ldd, int32_t c0,
                            int64_t rows, int32_t C, int32_t accumulate, void* stream) {
  if (!src || !dst || rows <= 0 || C < 4 || (C & 3) || (lds & 3) || (ldd & 3) || (s0 & 3) || (c0 & 3) ||
      !al16(src) || !al16(dst))
    return WM_EINVAL;
  const int64_t total = rows * (C / 4);
  hipLaunchKernelGGL(copy_cols_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), src, lds,
                     s0, dst, ldd, c0, rows, C, accumulate, total);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_strided3(const float* in, void* out, int32_t out_bf16, int32_t n0, int32_t n1, int32_t n2, int64_t si0,
                           int64_t si1, int64_t si2, int64_t so0, int64_t so1, int64_t so2, int32_t atomic, void* stream) {
  if (!in || !out || n0 <= 0 || n1 <= 0 || n2 <= 0 || (out_bf16 && atomic)) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t total = (int64_t)n0 * n1 * n2, lim = (int64_t)1 << 30;
  auto span = [&](int64_t a, int64_t b, int64_t c) {
    return (n0 - 1) * (a < 0 ? -a : a) + (n1 - 1) * (b < 0 ? -b : b) + (n2 - 1) * (c < 0 ? -c : c);
  };
  const bool small = total < lim && span(si0, si1, si2) < lim && span(so0, so1, so2) < lim && si0 >= 0 && si1 >= 0 && si2 >= 0 &&
                     so0 >= 0 && so1 >= 0 && so2 >= 0;
  const dim3 grid(grid_for(total));
  if (out_bf16 && small)
    hipLaunchKernelGGL((strided3_kernel<__bf16, int>), grid, dim3(256), 0, st, in, static_cast<__bf16*>(out), n0, n1, n2, si0, si1,
                       si2, so0, so1, so2, 0);
  else if (out_bf16)
    hipLaunchKernelGGL((strided3_kernel<__bf16, int64_t>), grid, dim3(256), 0, st, in, static_cast<__bf16*>(out), n0, n1, n2, si0, si1,
                       si2, so0, so1, so2, 0);
  else if (small)
    hipLaunchKernelGGL((strided3_kernel<float, int>), grid, dim3(256), 0, st, in, static_cast<float*>(out), n0, n1, n2, si0, si1, si2,
                       so0, so1, so2, atomic);
  else
    hipLaunchKernelGGL((strided3_kernel<float, int64_t>), grid, dim3(256), 0, st, in, static_cast<float*>(out), n0, n1, n2, si0, si1,
                       si2, so0, so1, so2, atomic);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_axpbyz(float* out, float a, const float* x, float b, const float* y, int64_t n, void* stream) {
  if (!out || !x || n <= 0) return WM_EINVAL;
  hipLaunchKernelGGL(axpbyz_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), out, a, x, b, y,
                     n);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0 || (n & 3) || !al16(src)) return WM_EINVAL;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                     static_cast<__bf16*>(dst), n / 4);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_sum_bf16_rows(const void* src_bf16, float* dst, int32_t rows, int64_t n, void* stream) {
  if (!src_bf16 || !dst || rows <= 0 || n <= 0 || (n & 7)) return WM_EINVAL;
  hipLaunchKernelGGL(sum_bf16_rows_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const __bf16*>(src_bf16), dst, rows, n);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_transpose_cast_jobs(const float* arena, void* arena_t_bf16, const int32_t* jobs, int32_t njobs,
                                      int32_t total_tiles, void* stream) {
  if (!arena || !arena_t_bf16 || !jobs || njobs <= 0 || total_tiles <= 0) return WM_EINVAL;
  hipLaunchKernelGGL(transpose_cast_jobs_kernel, dim3(total_tiles), dim3(256), 0, static_cast<hipStream_t>(stream), arena,
                     static_cast<__bf16*>(arena_t_bf16), reinterpret_cast<const int4*>(jobs), njobs);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int64_t wm_minmax_scratch_floats(void) { return 2 * 1024; }
extern "C" int wm_minmax(const float* x, int64_t n, float* out_min_max, float* scratch, void* stream) {
  if (!x || n <= 0 || !out_min_max || !scratch) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int nb = (int)((n + 256 * 16 - 1) / (256 * 16));
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(nb), dim3(256), 0, st, x, n, scratch);
  WM_CHECK_LAUNCH();
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, st, scratch, nb, out_min_max);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_range_map(const float* x, float* out, const float* min_max, float lo, float hi, int32_t inverse, int64_t n,
                            void* stream) {
  if (!x || !out || !min_max || n <= 0 || hi == lo) return WM_EINVAL;
  hipLaunchKernelGGL(range_map_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, min_max, lo, hi,
                     inverse, n);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_msg_expand_fwd(const float* feat, float* dst, int32_t B, void* stream) {
  if (!feat || !dst || B <= 0) return WM_EINVAL;
  const int64_t total = (int64_t)B * 64 * 512;
  hipLaunchKernelGGL(msg_expand_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), feat,
                     dst, total);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
extern "C" int wm_msg_expand_bwd(const float* dcat, float* dfeat, int32_t B, void* stream) {
  if (!dcat || !dfeat || B <= 0) return WM_EINVAL;
  const int64_t total = (int64_t)B * 256;
  hipLaunchKernelGGL(msg_expand_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), dcat,
                     dfeat, total);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
