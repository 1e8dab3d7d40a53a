// 3x3 / stride 1 / pad 1 convolutions between a 2-plane NCHW tensor and a 32/64-channel NHWC tensor at full 128x128
// resolution: InputProj (model.py:814, 2 -> C, LeakyReLU) and OutputProj (model.py:846, 2C -> 2), forward and input
// gradient.  The generic one-thread-per-output kernels of conv_small.hip re-read every tap from global memory with
// scalar loads and took 0.25-0.9 ms per launch; these are < 1 GFLOP layers whose cost is the NHWC tensor's bytes.
//
//   fan-out  out[p][o] = act(bias[o] + sum_{s,ty,tx} in[s][p + (ty-1, tx-1)] * Wt[s][ty][tx][o])      2 planes -> NO channels
//            (InputProj forward; OutputProj dX with Wt = spatially flipped w)
//   fan-in   out[s][p] = bias[s] + sum_{ty,tx,c} in[p + (ty-1, tx-1)][c] * Wt[ty][tx][c][s]            NC channels -> 2 planes
//            (OutputProj forward; InputProj dX with in = dy * act'(y) and Wt = spatially flipped w)
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

__device__ __forceinline__ float act_apply(float v, int act) {
  return act == 1 ? fmaxf(v, 0.f) : (act == 2 ? (v > 0.f ? v : 0.01f * v) : (act == 3 ? 1.0f / (1.0f + __expf(-v)) : v));
}
__device__ __forceinline__ float act_grad_out(float y, int act) {
  return act == 1 ? (y > 0.f ? 1.f : 0.f) : (act == 2 ? (y > 0.f ? 1.f : 0.01f) : (act == 3 ? y * (1.f - y) : 1.f));
}

// ------------------------------------------------------------------------------------------------------------ fan-out
// 16x16-pixel tile per workgroup; thread = (channel quad q, pixel group): weights of its 4 channels stay in registers,
// the 18 inputs of a pixel are LDS broadcasts, every pixel's NO channels leave as one contiguous run of 16-byte stores.
template <int NO>
__global__ __launch_bounds__(256) void conv_fanout_kernel(const float* __restrict__ in, const float* __restrict__ gy,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, int H, int W, int flip, int act,
                                                          int gact, int accumulate) {
  constexpr int NQ = NO / 4, NPG = 256 / NQ, PPT = 256 / NPG;          // channel quads, pixel groups, pixels per thread
  __shared__ float tile[2][18][20];
  __shared__ __attribute__((aligned(16))) float wt[18][NO];
  const int tid = threadIdx.x, tw = W / 16;
  const int b = blockIdx.x / ((H / 16) * tw), tr = blockIdx.x % ((H / 16) * tw), ty0 = (tr / tw) * 16, tx0 = (tr % tw) * 16;
  // Wt[s][ty][tx][o]: forward w[o][s][ty][tx]; dX (flip) w[s][o][2-ty][2-tx]
  for (int i = tid; i < 18 * NO; i += 256) {
    const int o = i % NO, k = i / NO, s = k / 9, t = k % 9;
    wt[k][o] = flip ? w[(s * NO + o) * 9 + (8 - t)] : w[(o * 2 + s) * 9 + t];
  }
  for (int i = tid; i < 2 * 18 * 18; i += 256) {
    const int s = i / 324, r = (i % 324) / 18, c = i % 18;
    const int y = ty0 + r - 1, x = tx0 + c - 1;
    float v = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const int64_t o = (((int64_t)b * 2 + s) * H + y) * W + x;
      v = in[o];
      if (gy) v *= act_grad_out(gy[o], gact);
    }
    tile[s][r][c] = v;
  }
  __syncthreads();
  const int q = tid % NQ, pg = tid / NQ;
  float4 wr[18];
#pragma unroll
  for (int k = 0; k < 18; ++k) wr[k] = *reinterpret_cast<const float4*>(&wt[k][4 * q]);
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) b4 = *reinterpret_cast<const float4*>(bias + 4 * q);
#pragma unroll 2
  for (int i = 0; i < PPT; ++i) {
    const int p = pg + NPG * i, py = p >> 4, px = p & 15;
    float4 a = b4;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float v = tile[s][py + t / 3][px + t % 3];
        const float4 ww = wr[s * 9 + t];
        a.x += v * ww.x; a.y += v * ww.y; a.z += v * ww.z; a.w += v * ww.w;
      }
    a.x = act_apply(a.x, act); a.y = act_apply(a.y, act); a.z = act_apply(a.z, act); a.w = act_apply(a.w, act);
    float4* dst = reinterpret_cast<float4*>(out + (((int64_t)b * H + ty0 + py) * W + tx0 + px) * NO + 4 * q);
    if (accumulate) {
      const float4 o = *dst;
      a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
    }
    *dst = a;
  }
}

// ------------------------------------------------------------------------------------------------------------- fan-in
// 8x8-pixel tile (+ halo) x NC channels staged in LDS with all loads issued up front; 4 threads share a pixel (NC / 4
// channels each) and fold their two partial sums with lane exchanges.
template <int NC>
__global__ __launch_bounds__(256) void conv_fanin_kernel(const float* __restrict__ in, const float* __restrict__ gy,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int H, int W, int flip, int act,
                                                         int gact, int accumulate) {
  constexpr int CPP = NC / 4, NCH = 100 * CPP, NL = (NCH + 255) / 256;   // 16-byte chunks per pixel / tile / thread
  __shared__ __attribute__((aligned(16))) float tile[100 * NC];
  __shared__ __attribute__((aligned(16))) float wt[9 * NC * 2];
  const int tid = threadIdx.x, tw = W / 8;
  const int b = blockIdx.x / ((H / 8) * tw), tr = blockIdx.x % ((H / 8) * tw), ty0 = (tr / tw) * 8, tx0 = (tr % tw) * 8;
  float4 v[NL], g[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int idx = min(tid + 256 * k, NCH - 1), pix = idx / CPP, part = idx % CPP;
    const int y = min(max(ty0 + pix / 10 - 1, 0), H - 1), x = min(max(tx0 + pix % 10 - 1, 0), W - 1);
    const int64_t o = (((int64_t)b * H + y) * W + x) * NC + 4 * part;
    v[k] = *reinterpret_cast<const float4*>(in + o);
    if (gy) g[k] = *reinterpret_cast<const float4*>(gy + o);
  }
  // Wt[t][c][s]: forward w[s][c][t]; dX (flip) w[c][s][8 - t]
  for (int i = tid; i < 9 * NC * 2; i += 256) {
    const int s = i & 1, c = (i >> 1) % NC, t = i / (2 * NC);
    wt[i] = flip ? w[(c * 2 + s) * 9 + (8 - t)] : w[(s * NC + c) * 9 + t];
  }
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int idx = tid + 256 * k;
    if (idx < NCH) {
      const int pix = idx / CPP;
      const int y = ty0 + pix / 10 - 1, x = tx0 + pix % 10 - 1;
      float4 t = v[k];
      if (gy) {
        t.x *= act_grad_out(g[k].x, gact); t.y *= act_grad_out(g[k].y, gact);
        t.z *= act_grad_out(g[k].z, gact); t.w *= act_grad_out(g[k].w, gact);
      }
      if (y < 0 || y >= H || x < 0 || x >= W) t = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(tile + idx * 4) = t;
    }
  }
  __syncthreads();
  const int sub = tid & 3, p = tid >> 2, py = p >> 3, px = p & 7;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float* src = tile + ((py + t / 3) * 10 + px + t % 3) * NC;
    const float* ww = wt + t * NC * 2;
#pragma unroll
    for (int j = 0; j < NC / 16; ++j) {
      const int c = 4 * (sub + 4 * j);                        // interleaved quads: conflict-free LDS reads across the 4 lanes
      const float4 xv = *reinterpret_cast<const float4*>(src + c);
      const float4 w01 = *reinterpret_cast<const float4*>(ww + 2 * c), w23 = *reinterpret_cast<const float4*>(ww + 2 * c + 4);
      a0 += xv.x * w01.x + xv.y * w01.z + xv.z * w23.x + xv.w * w23.z;
      a1 += xv.x * w01.y + xv.y * w01.w + xv.z * w23.y + xv.w * w23.w;
    }
  }
  a0 += __shfl_xor(a0, 1); a1 += __shfl_xor(a1, 1);
  a0 += __shfl_xor(a0, 2); a1 += __shfl_xor(a1, 2);
  if (sub < 2) {
    float r = (sub == 0 ? a0 : a1) + (bias ? bias[sub] : 0.f);
    r = act_apply(r, act);
    float* dst = out + (((int64_t)b * 2 + sub) * H + ty0 + py) * W + tx0 + px;
    *dst = accumulate ? *dst + r : r;
  }
}

// ---------------------------------------------------------------------------------------------------- weight gradients
// dw[co][ci][t] = sum_pixels dypre[co][p] * x[ci][p + off_t], db[co] = sum dypre[co][p]  (dypre = dy * act'(y)).
// The generic kernel gave every thread 5 weight elements and two LDS reads per multiply-add; here a thread owns a
// (tap, 4-channel) slot so that one 16-byte LDS read feeds 4 or 8 multiply-adds, a workgroup walks `tpb` tiles with
// the sums in registers, and adds them with one atomic per weight at the end.

// "fan-in" layer (OutputProj): x NHWC [B,H,W,NC], dy / y NCHW [B,2,H,W];  thread = (tap, channel quad), 8 sums
template <int NC>
__global__ __launch_bounds__(256) void conv_fanin_dw_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ dy, float* __restrict__ dw,
                                                            float* __restrict__ db, int B, int H, int W, int act, int tpb) {
  constexpr int CPP = NC / 4, NCH = 100 * CPP, NL = (NCH + 255) / 256, NSLOT = 9 * CPP;
  static_assert(NSLOT + 2 <= 256, "one thread per (tap, channel quad) + the two bias sums");
  __shared__ __attribute__((aligned(16))) float tile[100 * NC];
  __shared__ float dpre[2][64];
  const int tid = threadIdx.x, tw = W / 8, ntiles = B * (H / 8) * tw;
  const int tap = tid / CPP, quad = tid % CPP;
  float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f}, bsum = 0.f;
  for (int t = blockIdx.x * tpb; t < min(ntiles, (int)(blockIdx.x + 1) * tpb); ++t) {
    const int b = t / ((H / 8) * tw), tr = t % ((H / 8) * tw), ty0 = (tr / tw) * 8, tx0 = (tr % tw) * 8;
    float4 v[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int idx = min(tid + 256 * k, NCH - 1), pix = idx / CPP, part = idx % CPP;
      const int yy = min(max(ty0 + pix / 10 - 1, 0), H - 1), xx = min(max(tx0 + pix % 10 - 1, 0), W - 1);
      v[k] = *reinterpret_cast<const float4*>(x + (((int64_t)b * H + yy) * W + xx) * NC + 4 * part);
    }
    float dp = 0.f;
    if (tid < 128) {
      const int co = tid >> 6, p = tid & 63;
      const int64_t o = (((int64_t)b * 2 + co) * H + ty0 + (p >> 3)) * W + tx0 + (p & 7);
      dp = dy[o] * act_grad_out(act ? y[o] : 1.f, act);
    }
    __syncthreads();                                          // previous tile fully consumed
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int idx = tid + 256 * k;
      if (idx < NCH) {
        const int pix = idx / CPP;
        const int yy = ty0 + pix / 10 - 1, xx = tx0 + pix % 10 - 1;
        *reinterpret_cast<float4*>(tile + idx * 4) = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? make_float4(0.f, 0.f, 0.f, 0.f) : v[k];
      }
    }
    if (tid < 128) dpre[tid >> 6][tid & 63] = dp;
    __syncthreads();
    if (tid < NSLOT) {
      const float* src = tile + ((tap / 3) * 10 + tap % 3) * NC + 4 * quad;
#pragma unroll 8
      for (int p = 0; p < 64; ++p) {
        const float4 xv = *reinterpret_cast<const float4*>(src + ((p >> 3) * 10 + (p & 7)) * NC);
        const float d0 = dpre[0][p], d1 = dpre[1][p];
        acc0[0] += d0 * xv.x; acc0[1] += d0 * xv.y; acc0[2] += d0 * xv.z; acc0[3] += d0 * xv.w;
        acc1[0] += d1 * xv.x; acc1[1] += d1 * xv.y; acc1[2] += d1 * xv.z; acc1[3] += d1 * xv.w;
      }
    } else if (tid < NSLOT + 2) {
      for (int p = 0; p < 64; ++p) bsum += dpre[tid - NSLOT][p];
    }
  }
  if (tid < NSLOT) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      atomicAdd(dw + ((int64_t)0 * NC + 4 * quad + c) * 9 + tap, acc0[c]);
      atomicAdd(dw + ((int64_t)1 * NC + 4 * quad + c) * 9 + tap, acc1[c]);
    }
  } else if (tid < NSLOT + 2 && db) {
    atomicAdd(db + (tid - NSLOT), bsum);
  }
}

// "fan-out" layer (InputProj): x NCHW [B,2,H,W], dy / y NHWC [B,H,W,NO];  thread = (input plane x tap, output quad), 4 sums
template <int NO>
__global__ __launch_bounds__(256) void conv_fanout_dw_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ dy, float* __restrict__ dw,
                                                             float* __restrict__ db, int B, int H, int W, int act, int tpb) {
  constexpr int NQ = NO / 4, NSLOT = 18 * NQ;                  // 8x8-pixel tiles: x tile [2][10][10], dypre [64][NO]
  static_assert(NSLOT + NQ <= 256, "one thread per (plane, tap, output quad) + the bias quads");
  __shared__ float xt[2][10][10];
  __shared__ __attribute__((aligned(16))) float dpre[64 * NO];
  const int tid = threadIdx.x, tw = W / 8, ntiles = B * (H / 8) * tw;
  const int slot = tid / NQ, quad = tid % NQ, s_pl = slot / 9, tap = slot % 9;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int NL = (64 * NQ + 255) / 256;
  for (int t = blockIdx.x * tpb; t < min(ntiles, (int)(blockIdx.x + 1) * tpb); ++t) {
    const int b = t / ((H / 8) * tw), tr = t % ((H / 8) * tw), ty0 = (tr / tw) * 8, tx0 = (tr % tw) * 8;
    float4 g[NL], yv[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int idx = min(tid + 256 * k, 64 * NQ - 1), p = idx / NQ, q4 = idx % NQ;
      const int64_t o = (((int64_t)b * H + ty0 + (p >> 3)) * W + tx0 + (p & 7)) * NO + 4 * q4;
      g[k] = *reinterpret_cast<const float4*>(dy + o);
      yv[k] = act ? *reinterpret_cast<const float4*>(y + o) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    float xv = 0.f;
    if (tid < 200) {
      const int s = tid / 100, r = (tid % 100) / 10, c = tid % 10;
      const int yy = ty0 + r - 1, xx = tx0 + c - 1;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) xv = x[(((int64_t)b * 2 + s) * H + yy) * W + xx];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int idx = tid + 256 * k;
      if (idx < 64 * NQ)
        *reinterpret_cast<float4*>(dpre + idx * 4) = make_float4(g[k].x * act_grad_out(yv[k].x, act), g[k].y * act_grad_out(yv[k].y, act),
                                                                 g[k].z * act_grad_out(yv[k].z, act), g[k].w * act_grad_out(yv[k].w, act));
    }
    if (tid < 200) xt[tid / 100][(tid % 100) / 10][tid % 10] = xv;
    __syncthreads();
    if (tid < NSLOT) {
#pragma unroll 8
      for (int p = 0; p < 64; ++p) {
        const float xs = xt[s_pl][(p >> 3) + tap / 3][(p & 7) + tap % 3];
        const float4 d = *reinterpret_cast<const float4*>(dpre + p * NO + 4 * quad);
        acc[0] += d.x * xs; acc[1] += d.y * xs; acc[2] += d.z * xs; acc[3] += d.w * xs;
      }
    } else if (tid < NSLOT + NQ) {
      const int q4 = tid - NSLOT;
      for (int p = 0; p < 64; ++p) {
        const float4 d = *reinterpret_cast<const float4*>(dpre + p * NO + 4 * q4);
        acc[0] += d.x; acc[1] += d.y; acc[2] += d.z; acc[3] += d.w;
      }
    }
  }
  if (tid < NSLOT) {
#pragma unroll
    for (int c = 0; c < 4; ++c) atomicAdd(dw + ((int64_t)(4 * quad + c) * 2 + s_pl) * 9 + tap, acc[c]);
  } else if (tid < NSLOT + NQ && db) {
#pragma unroll
    for (int c = 0; c < 4; ++c) atomicAdd(db + 4 * (tid - NSLOT) + c, acc[c]);
  }
}

}  // namespace

// Returns -1 when the layer is not one of the edge shapes (caller uses the generic kernels).
int wm_conv_edge_fwd(const WmConv& c, const float* x, const float* w, const float* b, float* y, hipStream_t st) {
  if (c.transposed || c.KH != 3 || c.KW != 3 || c.SH != 1 || c.SW != 1 || c.PH != 1 || c.PW != 1) return -1;
  if (!c.in_nhwc && c.out_nhwc && c.Cin == 2 && (c.Cout == 32 || c.Cout == 64) && !(c.H & 15) && !(c.W & 15)) {
    const dim3 grid(c.B * (c.H / 16) * (c.W / 16));
    if (c.Cout == 32) hipLaunchKernelGGL(conv_fanout_kernel<32>, grid, dim3(256), 0, st, x, nullptr, w, b, y, c.H, c.W, 0, c.act, 0, 0);
    else hipLaunchKernelGGL(conv_fanout_kernel<64>, grid, dim3(256), 0, st, x, nullptr, w, b, y, c.H, c.W, 0, c.act, 0, 0);
    WM_CHECK_LAUNCH();
    return WM_OK;
  }
  if (c.in_nhwc && !c.out_nhwc && c.Cout == 2 && (c.Cin == 32 || c.Cin == 64) && !(c.H & 7) && !(c.W & 7)) {
    const dim3 grid(c.B * (c.H / 8) * (c.W / 8));
    if (c.Cin == 32) hipLaunchKernelGGL(conv_fanin_kernel<32>, grid, dim3(256), 0, st, x, nullptr, w, b, y, c.H, c.W, 0, c.act, 0, 0);
    else hipLaunchKernelGGL(conv_fanin_kernel<64>, grid, dim3(256), 0, st, x, nullptr, w, b, y, c.H, c.W, 0, c.act, 0, 0);
    WM_CHECK_LAUNCH();
    return WM_OK;
  }
  return -1;
}

int wm_conv_edge_dx(const WmConv& c, const float* w, const float* y, const float* dy, float* dx, int accumulate,
                    hipStream_t st) {
  if (c.transposed || c.KH != 3 || c.KW != 3 || c.SH != 1 || c.SW != 1 || c.PH != 1 || c.PW != 1) return -1;
  const float* gy = c.act ? y : nullptr;
  if (c.in_nhwc && !c.out_nhwc && c.Cout == 2 && (c.Cin == 32 || c.Cin == 64) && !(c.H & 15) && !(c.W & 15)) {
    // OutputProj-shaped layer: dy is the 2-plane NCHW tensor, dx the NHWC one -> fan-out with flipped weights
    const dim3 grid(c.B * (c.H / 16) * (c.W / 16));
    if (c.Cin == 32) hipLaunchKernelGGL(conv_fanout_kernel<32>, grid, dim3(256), 0, st, dy, gy, w, nullptr, dx, c.H, c.W, 1, 0, c.act, accumulate);
    else hipLaunchKernelGGL(conv_fanout_kernel<64>, grid, dim3(256), 0, st, dy, gy, w, nullptr, dx, c.H, c.W, 1, 0, c.act, accumulate);
    WM_CHECK_LAUNCH();
    return WM_OK;
  }
  if (!c.in_nhwc && c.out_nhwc && c.Cin == 2 && (c.Cout == 32 || c.Cout == 64) && !(c.H & 7) && !(c.W & 7)) {
    // InputProj-shaped layer: dy (and y) are NHWC, dx is the 2-plane NCHW tensor -> fan-in with flipped weights
    const dim3 grid(c.B * (c.H / 8) * (c.W / 8));
    if (c.Cout == 32) hipLaunchKernelGGL(conv_fanin_kernel<32>, grid, dim3(256), 0, st, dy, gy, w, nullptr, dx, c.H, c.W, 1, 0, c.act, accumulate);
    else hipLaunchKernelGGL(conv_fanin_kernel<64>, grid, dim3(256), 0, st, dy, gy, w, nullptr, dx, c.H, c.W, 1, 0, c.act, accumulate);
    WM_CHECK_LAUNCH();
    return WM_OK;
  }
  return -1;
}

int wm_conv_edge_dw(const WmConv& c, const float* x, const float* y, const float* dy, float* dw, float* db, hipStream_t st) {
  if (c.transposed || c.KH != 3 || c.KW != 3 || c.SH != 1 || c.SW != 1 || c.PH != 1 || c.PW != 1 || (c.H & 7) || (c.W & 7)) return -1;
  const int ntiles = c.B * (c.H / 8) * (c.W / 8);
  int tpb = (ntiles + 1023) / 1024;                           // ~1024 workgroups: their atomics per weight stay cheap
  if (tpb < 1) tpb = 1;
  const dim3 grid((ntiles + tpb - 1) / tpb);
  if (c.in_nhwc && !c.out_nhwc && c.Cout == 2 && (c.Cin == 32 || c.Cin == 64)) {
    if (c.Cin == 32) hipLaunchKernelGGL(conv_fanin_dw_kernel<32>, grid, dim3(256), 0, st, x, y, dy, dw, db, c.B, c.H, c.W, c.act, tpb);
    else hipLaunchKernelGGL(conv_fanin_dw_kernel<64>, grid, dim3(256), 0, st, x, y, dy, dw, db, c.B, c.H, c.W, c.act, tpb);
    WM_CHECK_LAUNCH();
    return WM_OK;
  }
  if (!c.in_nhwc && c.out_nhwc && c.Cin == 2 && c.Cout == 32) {
    hipLaunchKernelGGL(conv_fanout_dw_kernel<32>, grid, dim3(256), 0, st, x, y, dy, dw, db, c.B, c.H, c.W, c.act, tpb);
    WM_CHECK_LAUNCH();
    return WM_OK;
  }
  return -1;
}
