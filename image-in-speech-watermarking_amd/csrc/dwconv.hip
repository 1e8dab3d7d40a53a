// LeFF depthwise 3x3 convolution on token-major activations [B, hw, hw, Ch] (reference
// model.py:703-709: rearrange -> Conv2d(groups=Ch, k=3, p=1) -> GELU -> rearrange).  The token-major
// layout makes the channel dim contiguous, so no b(hw)c <-> bchw transposes are materialised.
// Input is linear1's PRE-activation; GELU is applied once per staged element (tile + halo in LDS).
//
// Work split (single-tile kernels; the forward of strips of two tiles and more is dwconv_fwd_strip_kernel below): a
// 256-thread workgroup owns an 8x8-pixel tile x 64 channels; a thread owns TWO channels of one tile row
// (8 pixels), so the 3x3 window slides along x in registers (30 LDS reads per 8 outputs instead of 72) and every
// multiply-add is a packed 2 x fp32 instruction.  Global traffic is 16-byte chunks, all issued before the first use
// (the compiler does not hoist loads across LDS stores by itself), and the backward kernel walks a strip of tiles
// with the next tile's loads in flight while the current one is computed.
#include "common.h"
#include "gelu_poly.h"
#include "../../include/wm_kernels.h"

namespace {

constexpr int TP = 8;              // tile is TP x TP pixels
constexpr int HP = TP + 2;         // with halo
constexpr int TC = 64;             // channels per workgroup
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <typename ST> struct Chunk;                          // one 16-byte global chunk
template <> struct Chunk<__bf16> {
  static constexpr int NE = 8;
  static __device__ __forceinline__ void unpack(u32x4 v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf_lo(v[i]); f[2 * i + 1] = bf_hi(v[i]); }
  }
};
template <> struct Chunk<float> {
  static constexpr int NE = 4;
  static __device__ __forceinline__ void unpack(u32x4 v, float (&f)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
  }
};
template <typename ST> struct Halo {
  static constexpr int NE = Chunk<ST>::NE, CPP = TC / NE, NCH = HP * HP * CPP, NL = (NCH + 255) / 256;
};

// two adjacent channels of one pixel
template <typename ST> struct Pair;
template <> struct Pair<__bf16> {
  typedef uint32_t raw;
  static __device__ __forceinline__ void put(__bf16* p, f32x2 v) { stream_store(p, pack_bf2(v.x, v.y)); }
  static __device__ __forceinline__ f32x2 up(raw u) { f32x2 v = {bf_lo(u), bf_hi(u)}; return v; }
  static __device__ __forceinline__ raw down(f32x2 v) { return pack_bf2(v.x, v.y); }
};
template <> struct Pair<float> {
  typedef float2 raw;
  static __device__ __forceinline__ void put(float* p, f32x2 v) { stream_store(p, make_float2(v.x, v.y)); }
  static __device__ __forceinline__ f32x2 up(raw u) { f32x2 v = {u.x, u.y}; return v; }
  static __device__ __forceinline__ raw down(f32x2 v) { return make_float2(v.x, v.y); }
};

// (tile + halo) x 64 channels of `src` into registers; out-of-image pixels read a clamped address and become zero
template <typename ST>
__device__ __forceinline__ void halo_load(u32x4 (&c)[Halo<ST>::NL], const ST* __restrict__ src, int b, int ty, int tx,
                                          int hw, int Ch, int cbase, int tid) {
  typedef Halo<ST> H;
#pragma unroll
  for (int k = 0; k < H::NL; ++k) {
    int idx = tid + 256 * k;
    idx = idx < H::NCH ? idx : H::NCH - 1;
    const int pix = idx / H::CPP, part = idx - pix * H::CPP;
    const int ly = pix / HP, lx = pix - ly * HP;
    const int gy = ty * TP + ly - 1, gx = tx * TP + lx - 1;
    const bool inb = (unsigned)gy < (unsigned)hw && (unsigned)gx < (unsigned)hw;
    const int cy = min(max(gy, 0), hw - 1), cx = min(max(gx, 0), hw - 1);
    u32x4 v = *reinterpret_cast<const u32x4*>(src + (((int64_t)b * hw + cy) * hw + cx) * Ch + cbase + part * H::NE);
    if (!inb) v = (u32x4){0u, 0u, 0u, 0u};
    c[k] = v;
  }
}
// raw copy of the staged halo into an LDS image of the storage type (no conversion: the backward keeps dh2 as it lies)
template <typename ST>
__device__ __forceinline__ void halo_to_lds_raw(ST* tile, const u32x4 (&c)[Halo<ST>::NL], int tid) {
  typedef Halo<ST> H;
#pragma unroll
  for (int k = 0; k < H::NL; ++k) {
    const int idx = tid + 256 * k;
    if (idx < H::NCH) *reinterpret_cast<u32x4*>(tile + idx * H::NE) = c[k];
  }
}
template <typename ST, bool GELU>
__device__ __forceinline__ void halo_to_lds(float* tile, const u32x4 (&c)[Halo<ST>::NL], int tid) {
  typedef Halo<ST> H;
#pragma unroll
  for (int k = 0; k < H::NL; ++k) {
    const int idx = tid + 256 * k;
    if (idx < H::NCH) {
      float f[H::NE];
      Chunk<ST>::unpack(c[k], f);
      if (GELU) {
#pragma unroll
        for (int i = 0; i < H::NE; i += 2) {
          f32x2 a, g;
          gelu_pair<ST>((f32x2){f[i], f[i + 1]}, a, g);
          f[i] = a.x;
          f[i + 1] = a.y;
        }
      }
#pragma unroll
      for (int i = 0; i < H::NE; i += 4)
        *reinterpret_cast<float4*>(tile + idx * H::NE + i) = make_float4(f[i], f[i + 1], f[i + 2], f[i + 3]);
    }
  }
}

// wl[k][c] = w[cbase + c][k] (k < 9), wl[9][c] = bias[cbase + c] (or 0)
__device__ __forceinline__ void stage_weights(float* wl, const float* __restrict__ w, const float* __restrict__ bias,
                                              int cbase, int tid) {
  for (int i = tid; i < 10 * TC; i += 256) {
    const int k = i / TC, c = i - k * TC;
    wl[i] = k < 9 ? w[(cbase + c) * 9 + k] : (bias ? bias[cbase + c] : 0.f);
  }
}

template <typename ST>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const ST* __restrict__ h1, const float* __restrict__ w,
                                                         const float* __restrict__ bias, ST* __restrict__ h2,
                                                         ST* __restrict__ a2, ST* __restrict__ g2, int B, int hw,
                                                         int Ch) {
  __shared__ __attribute__((aligned(16))) float tile[HP * HP * TC];
  __shared__ __attribute__((aligned(16))) float wl[10 * TC];
  const int tid = threadIdx.x, nt = hw / TP;
  const int t = blockIdx.x, b = t / (nt * nt), tr = t - b * nt * nt, ty = tr / nt, tx = tr - ty * nt;
  const int cbase = blockIdx.y * TC;
  u32x4 hreg[Halo<ST>::NL];
  halo_load<ST>(hreg, h1, b, ty, tx, hw, Ch, cbase, tid);
  stage_weights(wl, w, bias, cbase, tid);
  halo_to_lds<ST, true>(tile, hreg, tid);
  __syncthreads();
  const int c2 = tid & 31, py = tid >> 5;
  f32x2 wr[9], br;
#pragma unroll
  for (int k = 0; k < 9; ++k) wr[k] = *reinterpret_cast<const f32x2*>(wl + k * TC + 2 * c2);
  br = *reinterpret_cast<const f32x2*>(wl + 9 * TC + 2 * c2);
  f32x2 col[3][3];                                            // col[x % 3][dy]
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) col[x][dy] = *reinterpret_cast<const f32x2*>(tile + ((py + dy) * HP + x) * TC + 2 * c2);
  const int64_t row0 = (((int64_t)b * hw + ty * TP + py) * hw + tx * TP) * Ch + cbase + 2 * c2;
#pragma unroll
  for (int px = 0; px < TP; ++px) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
      col[(px + 2) % 3][dy] = *reinterpret_cast<const f32x2*>(tile + ((py + dy) * HP + px + 2) * TC + 2 * c2);
    f32x2 o = br;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) o += wr[dy * 3 + dx] * col[(px + dx) % 3][dy];
    const int64_t off = row0 + (int64_t)px * Ch;
    if (h2) Pair<ST>::put(h2 + off, o);
    // second GELU and its derivative are materialised once here, from one shared exp(): evaluating them inside the
    // consumer GEMMs (A-operand staging of linear2 / dW2, epilogue of d_h2) repeated them per column tile and made
    // those launches VALU-bound
    if (a2 || g2) {
      f32x2 av, gv;
      gelu_pair<ST>(o, av, gv);
      if (a2) Pair<ST>::put(a2 + off, av);
      if (g2) Pair<ST>::put(g2 + off, gv);
    }
  }
}

// Strip-walking forward: one workgroup takes the row of tiles (b, ty, *) of 64 channels.  The halo image in LDS is a ring
// of 16 pixel columns (image column gx lives in slot (gx + 1) & 15): moving to the next tile only adds its 8 new columns,
// so the first GELU runs on 10 x 8 instead of 10 x 10 pixels per tile (2.25 instead of 2.56 evaluations per output with
// the second GELU), and the next tile's columns are in flight while the current tile is computed.
constexpr int RING = 16;
template <typename ST, int NCOL> struct Cols {                  // NCOL pixel columns of the halo rows: HP rows x NCOL columns x 64 channels
  static constexpr int NE = Chunk<ST>::NE, CPP = TC / NE, NCH = HP * NCOL * CPP, NL = (NCH + 255) / 256;
};
// columns [gx0, gx0 + NCOL) of rows ty * TP - 1 .. ty * TP + TP into registers (out-of-image pixels become zero)
template <typename ST, int NCOL, int NLD>
__device__ __forceinline__ void cols_load(u32x4 (&c)[NLD], const ST* __restrict__ src, int b, int ty, int gx0,
                                          int hw, int Ch, int cbase, int tid) {
  typedef Cols<ST, NCOL> H;
#pragma unroll
  for (int k = 0; k < H::NL; ++k) {
    int idx = tid + 256 * k;
    idx = idx < H::NCH ? idx : H::NCH - 1;
    const int pix = idx / H::CPP, part = idx - pix * H::CPP;
    const int ly = pix / NCOL, lx = pix - ly * NCOL;
    const int gy = ty * TP + ly - 1, gx = gx0 + lx;
    const bool inb = (unsigned)gy < (unsigned)hw && (unsigned)gx < (unsigned)hw;
    const int cy = min(max(gy, 0), hw - 1), cx = min(max(gx, 0), hw - 1);
    u32x4 v = *reinterpret_cast<const u32x4*>(src + (((int64_t)b * hw + cy) * hw + cx) * Ch + cbase + part * H::NE);
    if (!inb) v = (u32x4){0u, 0u, 0u, 0u};
    c[k] = v;
  }
}
// gelu of the staged columns into their ring slots (fp32 in LDS: a bf16 ring would let five instead of three workgroups
// share a CU but was measured at 3 % for one more rounding of gelu(h1); the kernel is bound by its VALU instruction count)
template <typename ST, int NCOL, int NLD>
__device__ __forceinline__ void cols_to_ring(float* ring, const u32x4 (&c)[NLD], int gx0, int tid) {
  typedef Cols<ST, NCOL> H;
  static_assert(NLD == H::NL, "register image of the columns");
#pragma unroll
  for (int k = 0; k < H::NL; ++k) {
    const int idx = tid + 256 * k;
    if (idx < H::NCH) {
      const int pix = idx / H::CPP, part = idx - pix * H::CPP;
      const int ly = pix / NCOL, lx = pix - ly * NCOL;
      float f[H::NE];
      Chunk<ST>::unpack(c[k], f);
#pragma unroll
      for (int i = 0; i < H::NE; i += 2) {
        f32x2 a, g;
        gelu_pair<ST>((f32x2){f[i], f[i + 1]}, a, g);
        f[i] = a.x;
        f[i + 1] = a.y;
      }
      float* dst = ring + ((ly * RING + ((gx0 + lx + 1) & (RING - 1))) * TC) + part * H::NE;
#pragma unroll
      for (int i = 0; i < H::NE; i += 4) *reinterpret_cast<float4*>(dst + i) = make_float4(f[i], f[i + 1], f[i + 2], f[i + 3]);
    }
  }
}

// TRAIN: the outputs of the train step (a2 and g2, no h2) are known at compile time: the per-pixel null checks of the
// general form cost four scalar branches per pixel and keep the two GELU polynomials in separate blocks
template <typename ST, bool TRAIN>
__global__ __launch_bounds__(256) void dwconv_fwd_strip_kernel(const ST* __restrict__ h1, const float* __restrict__ w,
                                                               const float* __restrict__ bias, ST* __restrict__ h2,
                                                               ST* __restrict__ a2, ST* __restrict__ g2, int B, int hw,
                                                               int Ch) {
  __shared__ __attribute__((aligned(16))) float ring[HP * RING * TC];
  __shared__ __attribute__((aligned(16))) float wl[10 * TC];
  const int tid = threadIdx.x, nt = hw / TP;
  const int b = blockIdx.y / nt, ty = blockIdx.y % nt;     // channel chunks of one strip are neighbours in dispatch order
  const int cbase = blockIdx.x * TC;
  const int c2 = tid & 31, py = tid >> 5;
  u32x4 creg[Cols<ST, TP>::NL];
  {
    u32x4 c0[Cols<ST, 2>::NL];
    cols_load<ST, 2>(c0, h1, b, ty, -1, hw, Ch, cbase, tid);  // the zero column left of the image and column 0
    cols_load<ST, TP>(creg, h1, b, ty, 1, hw, Ch, cbase, tid);   // tile 0 adds columns 1 .. 8
    stage_weights(wl, w, bias, cbase, tid);
    cols_to_ring<ST, 2>(ring, c0, -1, tid);                   // gelu(0) = 0 for the padding
  }
  __syncthreads();
  f32x2 wr[9], br;
#pragma unroll
  for (int k = 0; k < 9; ++k) wr[k] = *reinterpret_cast<const f32x2*>(wl + k * TC + 2 * c2);
  br = *reinterpret_cast<const f32x2*>(wl + 9 * TC + 2 * c2);
  const int64_t strip0 = (((int64_t)b * hw + ty * TP + py) * hw) * Ch + cbase + 2 * c2;
  for (int tx = 0; tx < nt; ++tx) {
    __syncthreads();                                          // the previous tile is done with the slots about to be reused
    cols_to_ring<ST, TP>(ring, creg, tx * TP + 1, tid);
    __syncthreads();
    if (tx + 1 < nt) cols_load<ST, TP>(creg, h1, b, ty, (tx + 1) * TP + 1, hw, Ch, cbase, tid);
    const int s0 = tx * TP;                                   // ring slot of halo column lx: (s0 + lx) & 15
    const int64_t tile0 = strip0 + (int64_t)tx * TP * Ch;
    auto at = [&](int dy, int x) {
      return *reinterpret_cast<const f32x2*>(ring + ((py + dy) * RING + ((s0 + x) & (RING - 1))) * TC + 2 * c2);
    };
    f32x2 col[3][3];                                          // col[x % 3][dy]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) col[x][dy] = at(dy, x);
#pragma unroll
    for (int px = 0; px < TP; ++px) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) col[(px + 2) % 3][dy] = at(dy, px + 2);
      f32x2 o = br;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) o += wr[dy * 3 + dx] * col[(px + dx) % 3][dy];
      const int64_t off = tile0 + px * Ch;                  // 32-bit multiply-add on top of the tile's 64-bit offset
      if (TRAIN) {
        f32x2 av, gv;
        gelu_pair<ST>(o, av, gv);
        Pair<ST>::put(a2 + off, av);
        Pair<ST>::put(g2 + off, gv);
      } else {
        if (h2) Pair<ST>::put(h2 + off, o);
        if (a2 || g2) {
          f32x2 av, gv;
          gelu_pair<ST>(o, av, gv);
          if (a2) Pair<ST>::put(a2 + off, av);
          if (g2) Pair<ST>::put(g2 + off, gv);
        }
      }
    }
  }
}

// dh1 = gelu'(h1) * sum_k w[k] * dh2[p - off_k];  dw[c][k] += sum_q gelu(h1)[q] * dh2[q - off_k];  db += sum dh2.
// Only dh2 needs a halo: the weight gradient is written as a sum over the CENTRE pixels q of a1[q] * dh2[q - off]
// (zero padding of the dh2 halo outside the image makes it equal to the usual sum over p of dh2[p] * a1[p + off]),
// so every tap feeds both the transposed convolution and the weight gradient, and gelu / gelu' of h1 are evaluated
// once per element from one shared exp().
// One workgroup walks the strip of tiles (b, ty, *) for 64 channels.  The weight / bias gradients stay in registers
// over the strip; at the end they are reduced over the 8 tile rows and either stored to `part` (one slot per
// workgroup, summed by dwconv_grad_reduce_kernel) or added with global atomics when there is no scratch buffer.
template <typename ST>
__global__ __launch_bounds__(256, sizeof(ST) == 2 ? 4 : 1) void dwconv_bwd_kernel(const ST* __restrict__ h1, const ST* __restrict__ dh2,
                                                         const float* __restrict__ w, ST* __restrict__ dh1,
                                                         float* __restrict__ dw, float* __restrict__ db,
                                                         float* __restrict__ part, int B, int hw, int Ch) {
  // dh2 with halo, in the storage type (bf16: 12.5 KB); the same bytes take the 4 x 64 x 10 partial sums at the end
  __shared__ __attribute__((aligned(16))) float tgf[(HP * HP * TC * sizeof(ST) > 4 * TC * 10 * 4 ? HP * HP * TC * sizeof(ST) : 4 * TC * 10 * 4) / 4];
  __shared__ __attribute__((aligned(16))) float wl[10 * TC];
  ST* const tg = reinterpret_cast<ST*>(tgf);
  typedef typename Pair<ST>::raw raw;
  const int tid = threadIdx.x, nt = hw / TP;
  const int b = blockIdx.y / nt, ty = blockIdx.y % nt;     // channel chunks of one strip are neighbours in dispatch order
  const int cbase = blockIdx.x * TC;
  const int c2 = tid & 31, py = tid >> 5;
  u32x4 hreg[Halo<ST>::NL];
  raw hv[TP];                                                 // h1 of the current tile; each slot is refilled for the next tile once it is consumed
  halo_load<ST>(hreg, dh2, b, ty, 0, hw, Ch, cbase, tid);
  const int64_t strip0 = (((int64_t)b * hw + ty * TP + py) * hw) * Ch + cbase + 2 * c2;
#pragma unroll
  for (int px = 0; px < TP; ++px) hv[px] = *reinterpret_cast<const raw*>(h1 + strip0 + (int64_t)px * Ch);
  stage_weights(wl, w, nullptr, cbase, tid);
  __syncthreads();
  f32x2 wr[9], aw[9], ab = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    wr[k] = *reinterpret_cast<const f32x2*>(wl + k * TC + 2 * c2);
    aw[k] = (f32x2){0.f, 0.f};
  }
  for (int tx = 0; tx < nt; ++tx) {
    __syncthreads();
    halo_to_lds_raw<ST>(tg, hreg, tid);
    __syncthreads();
    const int txn = tx + 1 < nt ? tx + 1 : tx;               // next tile's loads fly during this tile's arithmetic
    halo_load<ST>(hreg, dh2, b, ty, txn, hw, Ch, cbase, tid);
    const int64_t tile0 = strip0 + (int64_t)tx * TP * Ch, next0 = strip0 + (int64_t)txn * TP * Ch;
    f32x2 col[3][3];                                          // col[x % 3][row]: dh2 at halo row py + row, halo column x
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) col[x][rr] = Pair<ST>::up(*reinterpret_cast<const raw*>(tg + ((py + rr) * HP + x) * TC + 2 * c2));
#pragma unroll
    for (int px = 0; px < TP; ++px) {
#pragma unroll
      for (int rr = 0; rr < 3; ++rr)
        col[(px + 2) % 3][rr] = Pair<ST>::up(*reinterpret_cast<const raw*>(tg + ((py + rr) * HP + px + 2) * TC + 2 * c2));
      f32x2 a1, gp;
      gelu_pair<ST>(Pair<ST>::up(hv[px]), a1, gp);
      hv[px] = *reinterpret_cast<const raw*>(h1 + next0 + px * Ch);
      f32x2 o = {0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const f32x2 g = col[(px + 2 - dx) % 3][2 - dy];    // dh2 at (py + 1 - (dy - 1), px + 1 - (dx - 1))
          o += wr[dy * 3 + dx] * g;
          aw[dy * 3 + dx] += a1 * g;
        }
      ab += col[(px + 1) % 3][1];
      Pair<ST>::put(dh1 + tile0 + px * Ch, o * gp);
    }
  }
  // reduce the 20 per-thread sums over the 8 tile rows: the two rows of a wave by a lane exchange, the 4 waves via LDS
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    aw[k].x += __shfl_xor(aw[k].x, 32);
    aw[k].y += __shfl_xor(aw[k].y, 32);
  }
  ab.x += __shfl_xor(ab.x, 32);
  ab.y += __shfl_xor(ab.y, 32);
  __syncthreads();
  float* red = tgf;                                           // [4 waves][64 channels][10]
  if ((tid & 63) < 32) {
    float* dst = red + ((tid >> 6) * TC + 2 * c2) * 10;
#pragma unroll
    for (int k = 0; k < 9; ++k) { dst[k] = aw[k].x; dst[10 + k] = aw[k].y; }
    dst[9] = ab.x;
    dst[19] = ab.y;
  }
  __syncthreads();
  for (int i = tid; i < TC * 10; i += 256) {
    const float s = red[i] + red[TC * 10 + i] + red[2 * TC * 10 + i] + red[3 * TC * 10 + i];
    const int c = i / 10, k = i - c * 10;
    if (part) part[((int64_t)blockIdx.y * Ch + cbase) * 10 + i] = s;
    else if (k < 9) atomicAdd(dw + (cbase + c) * 9 + k, s);
    else atomicAdd(db + cbase + c, s);
  }
}

// dw[c][k] / db[c] += sum over the P strips of part[p][c][k]; block (x: 256 outputs, y: segment of `per` strips)
__global__ __launch_bounds__(256) void dwconv_grad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                 float* __restrict__ db, int P, int Ch, int per) {
  const int i = blockIdx.x * 256 + threadIdx.x, n = Ch * 10;
  if (i >= n) return;
  const int p0 = blockIdx.y * per, p1 = min(P, p0 + per);
  const float* p = part + (int64_t)p0 * n + i;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int q = p0;
  for (; q + 4 <= p1; q += 4, p += 4 * (int64_t)n) {
    s0 += p[0];
    s1 += p[n];
    s2 += p[2 * (int64_t)n];
    s3 += p[3 * (int64_t)n];
  }
  for (; q < p1; ++q, p += n) s0 += p[0];
  const float s = (s0 + s1) + (s2 + s3);
  const int c = i / 10, k = i - c * 10;
  if (k < 9) atomicAdd(dw + c * 9 + k, s);
  else atomicAdd(db + c, s);
}

bool ok(int B, int hw, int Ch) { return B > 0 && hw >= 8 && (hw & 7) == 0 && Ch >= 64 && (Ch & 63) == 0; }

}  // namespace

extern "C" int wm_dwconv_fwd(const void* h1, const float* w, const float* b, void* h2, void* a2, void* g2, int32_t io_bf16,
                             int32_t B, int32_t hw, int32_t Ch, void* stream) {
  if (!h1 || !w || !b || (!h2 && !a2 && !g2) || !ok(B, hw, Ch)) return WM_EINVAL;
  const int nt = hw / TP;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nt >= 2) {                                              // strips of two tiles and more: the ring saves up to a fifth of the first GELUs
    const bool train = !h2 && a2 && g2;
    if (io_bf16 && train)
      hipLaunchKernelGGL((dwconv_fwd_strip_kernel<__bf16, true>), dim3(Ch / TC, B * nt), dim3(256), 0, st,
                         static_cast<const __bf16*>(h1), w, b, static_cast<__bf16*>(h2), static_cast<__bf16*>(a2), static_cast<__bf16*>(g2), B, hw, Ch);
    else if (io_bf16)
      hipLaunchKernelGGL((dwconv_fwd_strip_kernel<__bf16, false>), dim3(Ch / TC, B * nt), dim3(256), 0, st,
                         static_cast<const __bf16*>(h1), w, b, static_cast<__bf16*>(h2), static_cast<__bf16*>(a2), static_cast<__bf16*>(g2), B, hw, Ch);
    else
      hipLaunchKernelGGL((dwconv_fwd_strip_kernel<float, false>), dim3(Ch / TC, B * nt), dim3(256), 0, st,
                         static_cast<const float*>(h1), w, b, static_cast<float*>(h2), static_cast<float*>(a2), static_cast<float*>(g2), B, hw, Ch);
    WM_CHECK_LAUNCH();
    return WM_OK;
  }
  if (io_bf16)
    hipLaunchKernelGGL(dwconv_fwd_kernel<__bf16>, dim3(B * nt * nt, Ch / TC), dim3(256), 0, st,
                       static_cast<const __bf16*>(h1), w, b, static_cast<__bf16*>(h2), static_cast<__bf16*>(a2), static_cast<__bf16*>(g2), B, hw, Ch);
  else
    hipLaunchKernelGGL(dwconv_fwd_kernel<float>, dim3(B * nt * nt, Ch / TC), dim3(256), 0, st,
                       static_cast<const float*>(h1), w, b, static_cast<float*>(h2), static_cast<float*>(a2), static_cast<float*>(g2), B, hw, Ch);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

extern "C" int64_t wm_dwconv_bwd_scratch_floats(int32_t B, int32_t hw, int32_t Ch) {
  return (int64_t)B * (hw / TP) * Ch * 10;
}

extern "C" int wm_dwconv_bwd(const void* h1, const void* dh2, const float* w, void* dh1, float* dw, float* db,
                             float* scratch, WmReduceJob* defer, int32_t io_bf16, int32_t B, int32_t hw, int32_t Ch,
                             void* stream) {
  if (!h1 || !dh2 || !w || !dh1 || !dw || !db || !ok(B, hw, Ch)) return WM_EINVAL;
  const int nt = hw / TP;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (io_bf16)
    hipLaunchKernelGGL(dwconv_bwd_kernel<__bf16>, dim3(Ch / TC, B * nt), dim3(256), 0, st, static_cast<const __bf16*>(h1),
                       static_cast<const __bf16*>(dh2), w, static_cast<__bf16*>(dh1), dw, db, scratch, B, hw, Ch);
  else
    hipLaunchKernelGGL(dwconv_bwd_kernel<float>, dim3(Ch / TC, B * nt), dim3(256), 0, st, static_cast<const float*>(h1),
                       static_cast<const float*>(dh2), w, static_cast<float*>(dh1), dw, db, scratch, B, hw, Ch);
  WM_CHECK_LAUNCH();
  if (defer) {                                                // two entries: dw [Ch][9], db [Ch] out of part[p][Ch][10]
    WmReduceJob jb{};
    defer[0] = jb;
    defer[1] = jb;
    if (scratch) {
      jb.src = scratch; jb.pstride = (int64_t)Ch * 10; jb.src_ld = 10; jb.count = B * nt; jb.rows = Ch; jb.cols = 9;
      jb.dst = dw; jb.dst_rs = 9; jb.dst_cs = 1;
      defer[0] = jb;
      jb.src = scratch + 9; jb.cols = 1; jb.dst = db; jb.dst_rs = 1;
      defer[1] = jb;
      return WM_OK;
    }
  }
  if (scratch) {
    const int P = B * nt;
    int segs = P / 8 < 1 ? 1 : (P / 8 > 32 ? 32 : P / 8);
    const int per = (P + segs - 1) / segs;
    segs = (P + per - 1) / per;
    hipLaunchKernelGGL(dwconv_grad_reduce_kernel, dim3((Ch * 10 + 255) / 256, segs), dim3(256), 0, st, scratch, dw, db, P, Ch, per);
    WM_CHECK_LAUNCH();
  }
  return WM_OK;
}
