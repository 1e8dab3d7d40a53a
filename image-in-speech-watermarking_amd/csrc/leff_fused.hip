// LeFF as two fused kernels for the blocks with C <= 128 (bf16 mode): reference LeFF.forward, model.py:695-714
//   x -> linear1 -> GELU -> rearrange -> depthwise 3x3 -> GELU -> rearrange -> linear2      (+ DropPath, + residual, :1016-1017)
// and its autograd backward.  These blocks carry 59 % of the step's token-channels and sit a factor 5-10 below the machine's
// flop / byte ridge (C / 2 flop per byte), so what they cost is HBM round trips between operators: the un-fused chain moves
// 50 C bytes per token forward (fc1 10 C, depthwise conv 24 C, fc2 16 C) and 52 C in the two input-gradient products and the
// depthwise backward.  Here a workgroup keeps an 8 x 8-pixel tile on chip across the three operators:
//
//   forward   xn2 halo tile (10 x 10 x C, bf16) -> LDS; per 64-channel slice of the hidden width: fc1 on the matrix cores (hidden
//             on the accumulator rows, pixel on the lane) -> h1 (saved, centre pixels) -> GELU in place -> depthwise 3 x 3 with the
//             window sliding in registers -> h2 (saved) -> GELU -> A operand of fc2, whose [64 x C] accumulator stays in registers
//             over the slices; epilogue (acc + b2) * DropPath + x1.   2 C (x 1.56 halo, mostly L2) + 4 C + 4 C + 16 C saved = 27 C.
//   backward  dy halo tile -> LDS; per slice: dh2 = (dy W2) * gelu'(h2) * DropPath on the halo (the saved h2 slice is staged in
//             the buffer the product lands in), transposed depthwise conv * gelu'(h1) -> dh1 (written for the fc1 weight
//             gradient) -> A operand of dxn2 += dh1 W1; depthwise weight / bias gradients summed in LDS over the strip of tiles
//             a workgroup walks.  Weights are staged as they lie and read with ds_read_b64_tr_b16 where the product wants
//             them transposed.  ~30 C instead of 52 C.
// fc1 / fc2 weight gradients stay on the TN GEMM (K = tokens): fc2's reads the saved h2 with GELU applied while staging.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "gelu_poly.h"
#include "../../include/wm_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int TP = 8;               // tile is TP x TP pixels
constexpr int HP = TP + 2;          // with halo
constexpr int NPIX = HP * HP;       // 100 halo pixels (4 MFMA row tiles of 32, the last one mostly padding)
constexpr int TCH = 64;             // hidden channels per slice
constexpr int HS = TCH + 8;         // LDS row stride (bf16) of the 64-wide images: 144 B (conflict-free 16-byte row reads)

template <int C> struct Cfg {
  static constexpr int XS = C + 8;                                  // row stride of the C-wide images
  static constexpr int CPR = C / 8;                                 // 16-byte chunks per C-wide row
  static constexpr int NCH = 4 * C / TCH;                           // hidden slices
  static constexpr int WS_EL = TCH * XS > C * HS ? TCH * XS : C * HS;   // W1 slice [64][C] or W2 slice [C][64]
  static constexpr int NT = 2 * (C / 32);                           // 32 x 32 output tiles of a [64 x C] product
  static constexpr int NTW = (NT + 3) / 4;                          // ... per wave
  static constexpr int NX = NPIX * CPR, NXL = (NX + 255) / 256;     // chunks of the halo tile, per thread
  static constexpr int FWD_SMEM = (NPIX * XS + WS_EL + NPIX * HS + 64 * HS) * 2 + (10 * TCH + TCH) * 4;
  static constexpr int BWD_SMEM = (NPIX * XS + WS_EL + NPIX * HS + 64 * HS) * 2 + (9 * TCH + 4 * 10 * TCH + NCH * 10 * TCH) * 4;
};

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the vector-memory counter
// (s_waitcnt vmcnt(0)): every barrier of a slice would then wait until the h1 / h2 / dh1 rows stored in the phase before it
// have been acknowledged by the L2 (1 - 2 us under load, five times per slice).  No phase reads global data another thread of
// the workgroup wrote, so only LDS visibility matters; loads are waited for by the compiler where their registers are used.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int virtual_block(int id, int n) { return (n & 7) ? id : (id & 7) * (n >> 3) + (id >> 3); }

// as-it-lies [k rows][columns] image -> 32x32x16 operand of the rows-are-k kind: lane l gets X[row0 + 8h + j][l & 31], j = 0..7
template <int LD>
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int row0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, h = g >> 1;
  const __bf16* a = tile + (row0 + 8 * h + q) * LD + 16 * (g & 1) + 4 * p;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * LD));
  bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return f;
}
template <int LD>
__device__ __forceinline__ bf16x8 row_frag(const __bf16* tile, int row, int ks, int h) {
  return *reinterpret_cast<const bf16x8*>(tile + row * LD + 16 * ks + 8 * h);
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) { return make_uint2(pack_bf2(a, b), pack_bf2(c, d)); }
__device__ __forceinline__ f32x2 up2(uint32_t u) { f32x2 v = {bf_lo(u), bf_hi(u)}; return v; }

// eight channels of one pixel as bf16 (one 16-byte chunk of the LDS image) from a bf16 or an fp32 tensor
template <typename T> struct Load8;
template <> struct Load8<__bf16> {
  static __device__ __forceinline__ u32x4 ld(const __bf16* p) { return *reinterpret_cast<const u32x4*>(p); }
};
template <> struct Load8<float> {
  static __device__ __forceinline__ u32x4 ld(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    u32x4 v = {pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w)};
    return v;
  }
};

// halo pixel index 0..99 of tile (ty, tx) -> image coordinates
struct Pix {
  int ly, lx, gy, gx;
  bool inb, centre;
  __device__ __forceinline__ Pix(int pix, int ty, int tx, int hw) {
    ly = pix / HP;
    lx = pix - ly * HP;
    gy = ty * TP + ly - 1;
    gx = tx * TP + lx - 1;
    inb = (unsigned)gy < (unsigned)hw && (unsigned)gx < (unsigned)hw;
    centre = (unsigned)(ly - 1) < (unsigned)TP && (unsigned)(lx - 1) < (unsigned)TP;
  }
};

// the [10 x 10] halo tile of a [tokens, C] tensor: loads (clamped addresses) and LDS stores (zero outside the image) apart,
// so that the loads are in flight while something else is done
template <int C, typename T>
__device__ __forceinline__ void halo_load(u32x4 (&xr)[Cfg<C>::NXL], const T* __restrict__ src, int b, int ty, int tx, int hw, int tid) {
  typedef Cfg<C> K;
#pragma unroll
  for (int k = 0; k < K::NXL; ++k) {
    int idx = tid + 256 * k;
    idx = idx < K::NX ? idx : K::NX - 1;
    const int pix = idx / K::CPR, part = idx - pix * K::CPR;
    const Pix p(pix, ty, tx, hw);
    const int cy = min(max(p.gy, 0), hw - 1), cx = min(max(p.gx, 0), hw - 1);
    xr[k] = Load8<T>::ld(src + (((int64_t)b * hw + cy) * hw + cx) * C + part * 8);
  }
}
template <int C>
__device__ __forceinline__ void halo_store(__bf16* Xs, const u32x4 (&xr)[Cfg<C>::NXL], int ty, int tx, int hw, int tid) {
  typedef Cfg<C> K;
#pragma unroll
  for (int k = 0; k < K::NXL; ++k) {
    const int idx = tid + 256 * k;
    if (idx < K::NX) {
      const int pix = idx / K::CPR, part = idx - pix * K::CPR;
      const Pix p(pix, ty, tx, hw);
      u32x4 v = xr[k];
      if (!p.inb) v = (u32x4){0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(Xs + pix * K::XS + part * 8) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ forward
// Persistent workgroups: each walks a contiguous range of tiles.  Nothing the next phase needs is requested at the moment it is
// needed: the next hidden slice's W1 rows / depthwise taps / fc1 bias, the next tile's halo image and this tile's residual rows are
// loaded into registers one phase (or one slice) ahead, so that a phase only waits for data requested a few microseconds earlier
// (the first version loaded and stored in place: ~6 us per slice and workgroup, four fifths of it memory round trips).
template <int C> struct FwdPre {                                       // next slice's operands, in flight
  u32x4 w1[TCH * Cfg<C>::CPR / 256];
  float wv[3], bv;
};
template <int C>
__device__ __forceinline__ void fwd_prefetch(FwdPre<C>& p, const __bf16* __restrict__ w1, const float* __restrict__ b1,
                                             const float* __restrict__ wd, const float* __restrict__ bd, int hb, int tid) {
  typedef Cfg<C> K;
#pragma unroll
  for (int k = 0; k < TCH * K::CPR / 256; ++k) {
    const int idx = tid + 256 * k, row = idx / K::CPR, part = idx - row * K::CPR;
    p.w1[k] = *reinterpret_cast<const u32x4*>(w1 + (int64_t)(hb + row) * C + part * 8);
  }
  // depthwise weights of the slice: 576 consecutive floats [channel][tap] (coalesced), biases behind them
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i = tid + 256 * k;
    p.wv[k] = i < 9 * TCH ? wd[hb * 9 + i] : bd[hb + (i < 10 * TCH ? i - 9 * TCH : 0)];
  }
  p.bv = b1[hb + (tid & 63)];
}
template <int C>
__device__ __forceinline__ void fwd_stage(const FwdPre<C>& p, __bf16* Ws, float* wl, float* b1s, int tid) {
  typedef Cfg<C> K;
#pragma unroll
  for (int k = 0; k < TCH * K::CPR / 256; ++k) {
    const int idx = tid + 256 * k, row = idx / K::CPR, part = idx - row * K::CPR;
    *reinterpret_cast<u32x4*>(Ws + row * K::XS + part * 8) = p.w1[k];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (tid + 256 * k < 10 * TCH) wl[tid + 256 * k] = p.wv[k];        // [channel][9] as it lies, then the 64 biases
  if (tid < TCH) b1s[tid] = p.bv;
}

template <int C, int SAVE, bool NTS>
__global__ __launch_bounds__(256, C == 32 ? 4 : C == 64 ? 3 : 2) void leff_fwd_kernel(const __bf16* __restrict__ xn2, const float* __restrict__ x1,
                                                       const __bf16* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ wd, const float* __restrict__ bd,
                                                       const __bf16* __restrict__ w2, const float* __restrict__ b2,
                                                       const float* __restrict__ s2, float* __restrict__ x2,
                                                       __bf16* __restrict__ h1, __bf16* __restrict__ h2, int B, int hw,
                                                       int ntiles, int per) {
  typedef Cfg<C> K;
  extern __shared__ __attribute__((aligned(16))) char smem[];        // one array (> 64 KB at C = 128: dynamic allocation)
  __bf16* const Xs = reinterpret_cast<__bf16*>(smem);                 // LayerNorm output, tile + halo (operand of fc1)
  __bf16* const Ws = Xs + NPIX * K::XS;                               // W1 slice, then W2 slice
  __bf16* const A1s = Ws + K::WS_EL;                                  // h1 slice, then gelu(h1) (zero outside the image)
  __bf16* const A2s = A1s + NPIX * HS;                                // gelu(h2): A operand of fc2
  float* const wl = reinterpret_cast<float*>(A2s + 64 * HS);          // depthwise weights [channel][tap], then the 64 biases
  float* const b1s = wl + 10 * TCH;
  const int tid0 = threadIdx.x;
  const int nt = hw / TP, Ch = 4 * C;
  constexpr bool EARLY = C <= 64;                                     // when the next tile's image / this tile's residual are requested
  const int t0 = blockIdx.x * per, t1 = min(ntiles, t0 + per);       // consecutive tiles: neighbours share halo columns in L2
  if (t0 >= t1) return;
  auto tile_pos = [&](int t, int& b, int& ty, int& tx) {
    b = t / (nt * nt);
    const int tr_ = t - b * nt * nt;
    ty = tr_ / nt;
    tx = tr_ - ty * nt;
  };

  u32x4 xr[K::NXL];
  FwdPre<C> pre;
  int b, ty, tx;
  tile_pos(t0, b, ty, tx);
  if constexpr (EARLY) halo_load<C, __bf16>(xr, xn2, b, ty, tx, hw, tid0);
  fwd_prefetch<C>(pre, w1, b1, wd, bd, 0, tid0);
  for (int t = t0; t < t1; ++t) {
    // The per-thread indices are re-derived from an OPAQUE copy of the thread id in every tile: hipcc otherwise hoists every
    // loop-invariant address component of the tile body (halo decode, epilogue row offsets, chunk indices) out of the
    // persistent loop and keeps ~80 registers of them alive through all phases (184 -> 121 registers at C = 32: four
    // workgroups per CU instead of two)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int c2 = tid & 31, py = tid >> 5;                           // depthwise stage: two channels of one tile row
    tile_pos(t, b, ty, tx);
    if constexpr (!EARLY) halo_load<C, __bf16>(xr, xn2, b, ty, tx, hw, tid);
    halo_store<C>(Xs, xr, ty, tx, hw, tid);                           // (every reader of the previous tile's image is behind a barrier)
    f32x16 acc2[K::NTW];
#pragma unroll
    for (int i = 0; i < K::NTW; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
    const int64_t row0 = (((int64_t)b * hw + ty * TP + py) * hw + tx * TP) * Ch + 2 * c2;
    float res[K::NTW][16];                                            // x1 of this tile in accumulator layout, requested in the last slice

    // one hidden slice; LAST (compile-time) = the tile's last slice, which also requests the next tile's image and this tile's
    // residual rows: as a run-time condition inside one loop those registers would be live through every slice
    auto slice = [&](const int kc, auto last_c) {
      constexpr bool LAST = decltype(last_c)::value;
      const int hb = kc * TCH;
      fwd_stage<C>(pre, Ws, wl, b1s, tid);
      wg_barrier();
      // W2 slice [C][64] into registers: lands while fc1 runs
      u32x4 w2r[C / 32];
#pragma unroll
      for (int k = 0; k < C / 32; ++k) {
        const int idx = tid + 256 * k, n = idx >> 3, part = idx & 7;
        w2r[k] = *reinterpret_cast<const u32x4*>(w2 + (int64_t)n * Ch + hb + part * 8);
      }
      {                                                               // fc1: h1[hidden][pixel] = W1 slice . xn2^T
        const int pix = 32 * wave + r;
        const int prow = pix < NPIX ? pix : NPIX - 1;
        f32x16 a[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int e = 0; e < 16; ++e) a[ct][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < C / 16; ++ks) {
          const bf16x8 bfrag = row_frag<K::XS>(Xs, prow, ks, h);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            a[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<K::XS>(Ws, 32 * ct + r, ks, h), bfrag, a[ct], 0, 0, 0);
        }
        if (pix < NPIX) {
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int hid = 32 * ct + 8 * g + 4 * h;
              const float4 bb = *reinterpret_cast<const float4*>(b1s + hid);
              *reinterpret_cast<uint2*>(A1s + pix * HS + hid) =
                  pack4(a[ct][4 * g] + bb.x, a[ct][4 * g + 1] + bb.y, a[ct][4 * g + 2] + bb.z, a[ct][4 * g + 3] + bb.w);
            }
        }
      }
      wg_barrier();
      // the next slice's operands (slice 0 again behind the last one: the next tile's), the next tile's image and this tile's
      // residual rows are requested here and used a slice / an epilogue later
      fwd_prefetch<C>(pre, w1, b1, wd, bd, kc + 1 < K::NCH ? hb + TCH : 0, tid);
      auto tile_prefetch = [&]() {
        if constexpr (EARLY) {                                         // (C = 128: 28 registers too many; the image is loaded at the tile's top)
          int nb, nty, ntx;
          tile_pos(t + 1 < t1 ? t + 1 : t, nb, nty, ntx);
          halo_load<C, __bf16>(xr, xn2, nb, nty, ntx, hw, tid);
        }
#pragma unroll
        for (int i = 0; i < K::NTW; ++i) {
          const int tile = wave + 4 * i;
          const int rt = tile & 1, ct = (tile < K::NT ? tile : 0) >> 1;
          const float* xrow = x1 + (((int64_t)b * hw + ty * TP + 4 * rt + (h >> 0) * 0) * hw + tx * TP) * C + 32 * ct + r;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int tok = mma32_row(e, h);                          // token 32 rt + tok: tile row 4 rt + (tok >> 3), column tok & 7
            res[i][e] = xrow[((tok >> 3) * hw + (tok & 7)) * C];
          }
        }
      };
      if constexpr (EARLY && LAST) tile_prefetch();
      {                                                               // h1 out (centre pixels), GELU in place; W2 slice -> LDS
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int idx = tid + 256 * k;
          if (idx < NPIX * 8) {
            const int pix = idx >> 3, part = idx & 7;
            const Pix p(pix, ty, tx, hw);
            __bf16* cell = A1s + pix * HS + part * 8;
            const u32x4 v = *reinterpret_cast<const u32x4*>(cell);
            if ((SAVE & 1) && p.centre) {
              __bf16* dst = h1 + (((int64_t)b * hw + p.gy) * hw + p.gx) * Ch + hb + part * 8;
              if (NTS) stream_store(dst, make_uint4(v.x, v.y, v.z, v.w));
              else *reinterpret_cast<u32x4*>(dst) = v;
            }
            u32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              f32x2 av, gv;
              gelu_pair<__bf16>(up2(v[i]), av, gv);
              o[i] = p.inb ? pack_bf2(av.x, av.y) : 0u;               // the convolution pads gelu(h1) with zeros
            }
            *reinterpret_cast<u32x4*>(cell) = o;
          }
        }
#pragma unroll
        for (int k = 0; k < C / 32; ++k) {
          const int idx = tid + 256 * k, n = idx >> 3, part = idx & 7;
          *reinterpret_cast<u32x4*>(Ws + n * HS + part * 8) = w2r[k];
        }
      }
      wg_barrier();
      {                                                               // depthwise 3 x 3, second GELU
        f32x2 wr[9], br;
#pragma unroll
        for (int k = 0; k < 9; ++k) wr[k] = (f32x2){wl[(2 * c2) * 9 + k], wl[(2 * c2 + 1) * 9 + k]};
        br = *reinterpret_cast<const f32x2*>(wl + 9 * TCH + 2 * c2);
        auto at = [&](int dy, int x) { return up2(*reinterpret_cast<const uint32_t*>(A1s + ((py + dy) * HP + x) * HS + 2 * c2)); };
        f32x2 col[3][3];                                              // col[x % 3][dy]
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
          if (SAVE & 2) {
            if (NTS) stream_store(h2 + row0 + (int64_t)px * Ch + hb, pack_bf2(o.x, o.y));
            else *reinterpret_cast<uint32_t*>(h2 + row0 + (int64_t)px * Ch + hb) = pack_bf2(o.x, o.y);
          }
          f32x2 av, gv;
          gelu_pair<__bf16>(o, av, gv);
          *reinterpret_cast<uint32_t*>(A2s + (py * TP + px) * HS + 2 * c2) = pack_bf2(av.x, av.y);
        }
      }
      wg_barrier();
      (void)tile_prefetch;
#pragma unroll
      for (int i = 0; i < K::NTW; ++i) {                              // fc2: x2[token][channel] += gelu(h2) . W2 slice^T
        const int tile = wave + 4 * i;
        if (tile < K::NT) {
          const int rt = tile & 1, ct = tile >> 1;
#pragma unroll
          for (int ks = 0; ks < TCH / 16; ++ks)
            acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HS>(A2s, 32 * rt + r, ks, h),
                                                              row_frag<HS>(Ws, 32 * ct + r, ks, h), acc2[i], 0, 0, 0);
        }
      }
      wg_barrier();
    };
    for (int kc = 0; kc < K::NCH - 1; ++kc) slice(kc, std::false_type{});
    slice(K::NCH - 1, std::true_type{});
    // (acc + b2) * DropPath + x1: a lane owns one channel, the 16 accumulator registers are 16 tokens
    const float sc = s2 ? s2[b] : 1.0f;
#pragma unroll
    for (int i = 0; i < K::NTW; ++i) {
      const int tile = wave + 4 * i;
      if (tile < K::NT) {
        const int rt = tile & 1, ct = tile >> 1, ch = 32 * ct + r;
        const float bias = b2[ch];
        float* orow = x2 + (((int64_t)b * hw + ty * TP + 4 * rt) * hw + tx * TP) * C + ch;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int tok = mma32_row(e, h);
          const int64_t o_ = (int64_t)((tok >> 3) * hw + (tok & 7)) * C;
          float rv;
          if constexpr (EARLY) rv = res[i][e];
          else rv = x1[(orow - x2) + o_];                             // C = 128: no registers for an early request (8 slices per tile amortise it)
          __builtin_nontemporal_store((acc2[i][e] + bias) * sc + rv, orow + o_);
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------------------- backward
// One workgroup walks the strip of tiles (b, ty, *); dw / db of the depthwise convolution are summed in LDS over the strip.
// As in the forward kernel, a slice's operands (W2 rows, the saved h2 halo slice, h1 of the thread's pixels, the depthwise taps)
// are requested one slice ahead, the W1 rows one phase ahead and the next tile's gradient image during the tile's last slice.
template <int C> struct BwdPre {
  u32x4 w2[C / 32], h2[4];
  uint32_t hv[TP];
  float wv[3];
};
template <int C>
__device__ __forceinline__ void bwd_prefetch(BwdPre<C>& p, const __bf16* __restrict__ h1, const __bf16* __restrict__ h2,
                                             const float* __restrict__ wd, const __bf16* __restrict__ w2, int b, int ty, int tx,
                                             int hw, int hb, int64_t row0, int tid) {
  const int Ch = 4 * C;
#pragma unroll
  for (int k = 0; k < C / 32; ++k) {
    const int idx = tid + 256 * k, n = idx >> 3, part_ = idx & 7;
    p.w2[k] = *reinterpret_cast<const u32x4*>(w2 + (int64_t)n * Ch + hb + part_ * 8);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int idx = tid + 256 * k;
    idx = idx < NPIX * 8 ? idx : NPIX * 8 - 1;
    const int pix = idx >> 3, part_ = idx & 7;
    const Pix q(pix, ty, tx, hw);
    const int cy = min(max(q.gy, 0), hw - 1), cx = min(max(q.gx, 0), hw - 1);
    p.h2[k] = *reinterpret_cast<const u32x4*>(h2 + (((int64_t)b * hw + cy) * hw + cx) * Ch + hb + part_ * 8);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i = tid + 256 * k;
    p.wv[k] = wd[hb * 9 + (i < 9 * TCH ? i : 0)];                     // [channel][tap] as it lies: 576 consecutive floats
  }
#pragma unroll
  for (int px = 0; px < TP; ++px) p.hv[px] = *reinterpret_cast<const uint32_t*>(h1 + row0 + (int64_t)px * Ch + hb);
}

template <int C, typename DT>
__global__ __launch_bounds__(256, 2) void leff_bwd_kernel(const DT* __restrict__ dy, const float* __restrict__ s2,
                                                          const __bf16* __restrict__ h1, const __bf16* __restrict__ h2,
                                                          const __bf16* __restrict__ w1, const float* __restrict__ wd,
                                                          const __bf16* __restrict__ w2, __bf16* __restrict__ dh1,
                                                          __bf16* __restrict__ dxn2, float* __restrict__ dwd,
                                                          float* __restrict__ dbd, float* __restrict__ part, int B, int hw) {
  typedef Cfg<C> K;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* const DYs = reinterpret_cast<__bf16*>(smem);                // incoming gradient, tile + halo; then the dxn2 tile
  __bf16* const Ws = DYs + NPIX * K::XS;                              // W2 slice [C][64] as it lies, then W1 slice [64][C]
  __bf16* const DH2s = Ws + K::WS_EL;                                 // h2 slice, then dh2 = (dy W2) gelu'(h2)
  __bf16* const DH1s = DH2s + NPIX * HS;                              // dh1: operand of dxn2 += dh1 W1
  float* const wl = reinterpret_cast<float*>(DH1s + 64 * HS);         // depthwise taps [channel][9]
  float* const RED = wl + 9 * TCH;                                    // per-wave sums of the slice's dw / db
  float* const ACC = RED + 4 * 10 * TCH;                              // [slice][tap | bias][channel] over the strip
  const int tid0 = threadIdx.x;
  const int nt = hw / TP;
  const int strip = virtual_block(blockIdx.x, gridDim.x);
  const int b = strip / nt, ty = strip - b * nt;
  const int Ch = 4 * C;
  constexpr bool EARLY = C <= 64;                                     // the next tile's gradient image is requested a slice ahead
  const float sc = s2 ? s2[b] : 1.0f;
  for (int i = tid0; i < K::NCH * 10 * TCH; i += 256) ACC[i] = 0.f;

  u32x4 xr[K::NXL];
  BwdPre<C> pre;
  if constexpr (EARLY) halo_load<C, DT>(xr, dy, b, ty, 0, hw, tid0);
  bwd_prefetch<C>(pre, h1, h2, wd, w2, b, ty, 0, hw, 0, (((int64_t)b * hw + ty * TP + (tid0 >> 5)) * hw) * Ch + 2 * (tid0 & 31), tid0);
  for (int tx = 0; tx < nt; ++tx) {
    // per-thread indices from an OPAQUE copy of the thread id in every tile (see the forward kernel: hipcc otherwise keeps every
    // hoisted address component of the tile body alive across the whole strip)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int c2 = tid & 31, py = tid >> 5;
    const int64_t strip0 = (((int64_t)b * hw + ty * TP + py) * hw) * Ch + 2 * c2;
    if constexpr (!EARLY) halo_load<C, DT>(xr, dy, b, ty, tx, hw, tid);
    halo_store<C>(DYs, xr, ty, tx, hw, tid);
    f32x16 acc2[K::NTW];
#pragma unroll
    for (int i = 0; i < K::NTW; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
    const int64_t row0 = strip0 + (int64_t)tx * TP * Ch;
    const int txn = tx + 1 < nt ? tx + 1 : tx;

    auto slice = [&](const int kc, auto last_c) {
      constexpr bool LAST = decltype(last_c)::value;
      const int hb = kc * TCH;
      uint32_t hv[TP];                                                // h1 of this thread's 8 pixels x 2 channels
#pragma unroll
      for (int px = 0; px < TP; ++px) hv[px] = pre.hv[px];
#pragma unroll
      for (int k = 0; k < C / 32; ++k) {                              // W2 slice as it lies, h2 halo slice, depthwise taps
        const int idx = tid + 256 * k, n = idx >> 3, part_ = idx & 7;
        *reinterpret_cast<u32x4*>(Ws + n * HS + part_ * 8) = pre.w2[k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int idx = tid + 256 * k;
        if (idx < NPIX * 8) *reinterpret_cast<u32x4*>(DH2s + (idx >> 3) * HS + (idx & 7) * 8) = pre.h2[k];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (tid + 256 * k < 9 * TCH) wl[tid + 256 * k] = pre.wv[k];
      wg_barrier();
      // W1 slice [64][C] into registers: lands while the first product runs
      constexpr int NL1 = TCH * K::CPR / 256;
      u32x4 w1r[NL1];
#pragma unroll
      for (int k = 0; k < NL1; ++k) {
        const int idx = tid + 256 * k, row = idx / K::CPR, part_ = idx - row * K::CPR;
        w1r[k] = *reinterpret_cast<const u32x4*>(w1 + (int64_t)(hb + row) * C + part_ * 8);
      }
      {                                                               // dh2[hidden][pixel] = (W2 slice)^T dy^T, times gelu'(h2)
        const int pix = 32 * wave + r;
        const int prow = pix < NPIX ? pix : NPIX - 1;
        f32x16 a[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int e = 0; e < 16; ++e) a[ct][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < C / 16; ++ks) {
          const bf16x8 bfrag = row_frag<K::XS>(DYs, prow, ks, h);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            a[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HS>(Ws + 32 * ct, 16 * ks, lane), bfrag, a[ct], 0, 0, 0);
        }
        if (pix < NPIX) {
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              uint2* cell = reinterpret_cast<uint2*>(DH2s + pix * HS + 32 * ct + 8 * g + 4 * h);
              const uint2 hraw = *cell;
              f32x2 a01, g01, a23, g23;
              gelu_pair<__bf16>(up2(hraw.x), a01, g01);
              gelu_pair<__bf16>(up2(hraw.y), a23, g23);
              *cell = pack4(a[ct][4 * g] * sc * g01.x, a[ct][4 * g + 1] * sc * g01.y, a[ct][4 * g + 2] * sc * g23.x,
                            a[ct][4 * g + 3] * sc * g23.y);
            }
        }
      }
      wg_barrier();
#pragma unroll
      for (int k = 0; k < NL1; ++k) {                                 // W1 slice as it lies (the first product is done with Ws)
        const int idx = tid + 256 * k, row = idx / K::CPR, part_ = idx - row * K::CPR;
        *reinterpret_cast<u32x4*>(Ws + row * K::XS + part_ * 8) = w1r[k];
      }
      // the next slice's operands (behind the last slice: slice 0 of the next tile) and, in the last slice, the next tile's image
      if constexpr (LAST) {
        bwd_prefetch<C>(pre, h1, h2, wd, w2, b, ty, txn, hw, 0, strip0 + (int64_t)txn * TP * Ch, tid);
        if constexpr (EARLY) halo_load<C, DT>(xr, dy, b, ty, txn, hw, tid);
      } else {
        bwd_prefetch<C>(pre, h1, h2, wd, w2, b, ty, tx, hw, hb + TCH, row0, tid);
      }
      {                                                               // transposed depthwise conv, gelu'(h1); dw / db sums
        f32x2 wr[9], aw[9], ab = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          wr[k] = (f32x2){wl[(2 * c2) * 9 + k], wl[(2 * c2 + 1) * 9 + k]};
          aw[k] = (f32x2){0.f, 0.f};
        }
        auto at = [&](int rr, int x) { return up2(*reinterpret_cast<const uint32_t*>(DH2s + ((py + rr) * HP + x) * HS + 2 * c2)); };
        f32x2 col[3][3];                                              // col[x % 3][row]
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int rr = 0; rr < 3; ++rr) col[x][rr] = at(rr, x);
#pragma unroll
        for (int px = 0; px < TP; ++px) {
#pragma unroll
          for (int rr = 0; rr < 3; ++rr) col[(px + 2) % 3][rr] = at(rr, px + 2);
          f32x2 a1, gp;
          gelu_pair<__bf16>(up2(hv[px]), a1, gp);
          f32x2 o = {0.f, 0.f};
#pragma unroll
          for (int dy_ = 0; dy_ < 3; ++dy_)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const f32x2 g = col[(px + 2 - dx) % 3][2 - dy_];         // dh2 at (py + 1 - (dy - 1), px + 1 - (dx - 1))
              o += wr[dy_ * 3 + dx] * g;
              aw[dy_ * 3 + dx] += a1 * g;
            }
          ab += col[(px + 1) % 3][1];
          const f32x2 d = o * gp;
          const uint32_t packed = pack_bf2(d.x, d.y);
          stream_store(dh1 + row0 + (int64_t)px * Ch + hb, packed);
          *reinterpret_cast<uint32_t*>(DH1s + (py * TP + px) * HS + 2 * c2) = packed;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          aw[k].x += __shfl_xor(aw[k].x, 32);
          aw[k].y += __shfl_xor(aw[k].y, 32);
        }
        ab.x += __shfl_xor(ab.x, 32);
        ab.y += __shfl_xor(ab.y, 32);
        if (lane < 32) {
          float* dst = RED + wave * 10 * TCH + 2 * c2;
#pragma unroll
          for (int k = 0; k < 9; ++k) *reinterpret_cast<float2*>(dst + k * TCH) = make_float2(aw[k].x, aw[k].y);
          *reinterpret_cast<float2*>(dst + 9 * TCH) = make_float2(ab.x, ab.y);
        }
      }
      wg_barrier();
#pragma unroll
      for (int i = 0; i < K::NTW; ++i) {                              // dxn2[channel][token] += (W1 slice)^T dh1^T
        const int tile = wave + 4 * i;
        if (tile < K::NT) {
          const int rt = tile & 1, ct = tile >> 1;
#pragma unroll
          for (int ks = 0; ks < TCH / 16; ++ks)
            acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<K::XS>(Ws + 32 * ct, 16 * ks, lane),
                                                              row_frag<HS>(DH1s, 32 * rt + r, ks, h), acc2[i], 0, 0, 0);
        }
      }
      for (int i = tid; i < 10 * TCH; i += 256)
        ACC[kc * 10 * TCH + i] += (RED[i] + RED[10 * TCH + i]) + (RED[2 * 10 * TCH + i] + RED[3 * 10 * TCH + i]);
      wg_barrier();
    };
    for (int kc = 0; kc < K::NCH - 1; ++kc) slice(kc, std::false_type{});
    slice(K::NCH - 1, std::true_type{});
    // dxn2 tile: [channel on the registers][token on the lane] -> LDS [token][C] -> 16-byte rows
#pragma unroll
    for (int i = 0; i < K::NTW; ++i) {
      const int tile = wave + 4 * i;
      if (tile < K::NT) {
        const int rt = tile & 1, ct = tile >> 1;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<uint2*>(DYs + (32 * rt + r) * K::XS + 32 * ct + 8 * g + 4 * h) =
              pack4(acc2[i][4 * g], acc2[i][4 * g + 1], acc2[i][4 * g + 2], acc2[i][4 * g + 3]);
      }
    }
    wg_barrier();
#pragma unroll
    for (int k = 0; k < C / 32; ++k) {
      const int idx = tid + 256 * k, tok = idx / K::CPR, part_ = idx - tok * K::CPR;
      const int64_t grow = ((int64_t)b * hw + ty * TP + (tok >> 3)) * hw + tx * TP + (tok & 7);
      const u32x4 v = *reinterpret_cast<const u32x4*>(DYs + tok * K::XS + part_ * 8);
      stream_store(dxn2 + grow * C + part_ * 8, make_uint4(v.x, v.y, v.z, v.w));
    }
    wg_barrier();
  }
  // the strip's dw [Ch][9] / db [Ch] as part[strip][Ch][10] (summed by the deferred reduction), or atomics without scratch;
  // ACC holds [slice][tap | bias][channel], the slices' taps lie [channel][9] in wl but were summed tap-major
  for (int i = tid0; i < Ch * 10; i += 256) {
    const int ch = i / 10, k = i - ch * 10;
    const float v = ACC[(ch >> 6) * 10 * TCH + k * TCH + (ch & 63)];
    if (part) part[(int64_t)strip * Ch * 10 + i] = v;
    else if (k < 9) atomicAdd(dwd + ch * 9 + k, v);
    else atomicAdd(dbd + ch, v);
  }
}

bool ok(int B, int hw, int C) { return B > 0 && hw >= 8 && (hw & 7) == 0 && (C == 32 || C == 64 || C == 128); }

// resident workgroups per CU (LDS and registers permitting): the persistent grid of the forward kernel
template <int C> constexpr int fwd_wgs_per_cu() { return C == 32 ? 4 : C == 64 ? 3 : 2; }

template <int C, int SAVE, bool NTS>
int launch_fwd_v(const void* xn2, const float* x1, const void* w1, const float* b1, const float* wd, const float* bd, const void* w2,
                 const float* b2, const float* s2, float* x2, void* h1, void* h2, int B, int hw, hipStream_t st) {
  const int tiles = B * (hw / TP) * (hw / TP);
  constexpr int SM = Cfg<C>::FWD_SMEM;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(leff_fwd_kernel<C, SAVE, NTS>), hipFuncAttributeMaxDynamicSharedMemorySize, SM);
    attr = true;
  }
  // a contiguous range of tiles per workgroup; ranges of at most 8 tiles so that small launches still cover the chip
  const int slots = 256 * fwd_wgs_per_cu<C>();
  int per = (tiles + slots - 1) / slots;
  per = per < 1 ? 1 : (per > 8 ? 8 : per);
  const int grid = (tiles + per - 1) / per;
  hipLaunchKernelGGL((leff_fwd_kernel<C, SAVE, NTS>), dim3(grid), dim3(256), SM, st, static_cast<const __bf16*>(xn2), x1,
                     static_cast<const __bf16*>(w1), b1, wd, bd, static_cast<const __bf16*>(w2), b2, s2, x2,
                     static_cast<__bf16*>(h1), static_cast<__bf16*>(h2), B, hw, tiles, per);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
template <int C>
int launch_fwd(const void* xn2, const float* x1, const void* w1, const float* b1, const float* wd, const float* bd, const void* w2,
               const float* b2, const float* s2, float* x2, void* h1, void* h2, int B, int hw, hipStream_t st) {
  static const bool plain = getenv("WM_LEFF_PLAIN_STORES") != nullptr;       // (experiment switch)
  const int save = (h1 ? 1 : 0) | (h2 ? 2 : 0);
#define WM_LF(S, N) return launch_fwd_v<C, S, N>(xn2, x1, w1, b1, wd, bd, w2, b2, s2, x2, h1, h2, B, hw, st)
  if (plain) { if (save == 3) WM_LF(3, false); if (save == 2) WM_LF(2, false); if (save == 1) WM_LF(1, false); }
  if (save == 3) WM_LF(3, true);
  if (save == 2) WM_LF(2, true);
  if (save == 1) WM_LF(1, true);
  WM_LF(0, true);
#undef WM_LF
}

template <int C>
int launch_bwd(const void* dy, int dy_bf16, const float* s2, const void* h1, const void* h2, const void* w1, const float* wd,
               const void* w2, void* dh1, void* dxn2, float* dwd, float* dbd, float* scratch, int B, int hw, hipStream_t st) {
  const int strips = B * (hw / TP);
  constexpr int SM = Cfg<C>::BWD_SMEM;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(leff_bwd_kernel<C, __bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, SM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(leff_bwd_kernel<C, float>), hipFuncAttributeMaxDynamicSharedMemorySize, SM);
    attr = true;
  }
  if (dy_bf16)
    hipLaunchKernelGGL((leff_bwd_kernel<C, __bf16>), dim3(strips), dim3(256), SM, st, static_cast<const __bf16*>(dy), s2,
                       static_cast<const __bf16*>(h1), static_cast<const __bf16*>(h2), static_cast<const __bf16*>(w1), wd,
                       static_cast<const __bf16*>(w2), static_cast<__bf16*>(dh1), static_cast<__bf16*>(dxn2), dwd, dbd, scratch, B, hw);
  else
    hipLaunchKernelGGL((leff_bwd_kernel<C, float>), dim3(strips), dim3(256), SM, st, static_cast<const float*>(dy), s2,
                       static_cast<const __bf16*>(h1), static_cast<const __bf16*>(h2), static_cast<const __bf16*>(w1), wd,
                       static_cast<const __bf16*>(w2), static_cast<__bf16*>(dh1), static_cast<__bf16*>(dxn2), dwd, dbd, scratch, B, hw);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

}  // namespace

extern "C" int32_t wm_leff_fusable(int32_t hw, int32_t C) { return ok(1, hw, C) ? 1 : 0; }

extern "C" int wm_leff_fwd(const void* xn2, const float* x1, const void* w1, const float* b1, const float* wd, const float* bd,
                           const void* w2, const float* b2, const float* rowscale, float* x2, void* h1, void* h2, int32_t B,
                           int32_t hw, int32_t C, void* stream) {
  if (!xn2 || !x1 || !w1 || !b1 || !wd || !bd || !w2 || !b2 || !x2 || !ok(B, hw, C)) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (C == 32) return launch_fwd<32>(xn2, x1, w1, b1, wd, bd, w2, b2, rowscale, x2, h1, h2, B, hw, st);
  if (C == 64) return launch_fwd<64>(xn2, x1, w1, b1, wd, bd, w2, b2, rowscale, x2, h1, h2, B, hw, st);
  return launch_fwd<128>(xn2, x1, w1, b1, wd, bd, w2, b2, rowscale, x2, h1, h2, B, hw, st);
}

extern "C" int64_t wm_leff_bwd_scratch_floats(int32_t B, int32_t hw, int32_t C) { return (int64_t)B * (hw / TP) * 4 * C * 10; }

extern "C" int wm_leff_bwd(const void* dy, int32_t dy_bf16, const float* rowscale, const void* h1, const void* h2, const void* w1,
                           const float* wd, const void* w2, void* dh1, void* dxn2, float* dwd, float* dbd, float* scratch,
                           WmReduceJob* defer, int32_t B, int32_t hw, int32_t C, void* stream) {
  if (!dy || !h1 || !h2 || !w1 || !wd || !w2 || !dh1 || !dxn2 || !dwd || !dbd || !ok(B, hw, C)) return WM_EINVAL;
  if (defer && !scratch) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc;
  if (C == 32) rc = launch_bwd<32>(dy, dy_bf16, rowscale, h1, h2, w1, wd, w2, dh1, dxn2, dwd, dbd, scratch, B, hw, st);
  else if (C == 64) rc = launch_bwd<64>(dy, dy_bf16, rowscale, h1, h2, w1, wd, w2, dh1, dxn2, dwd, dbd, scratch, B, hw, st);
  else rc = launch_bwd<128>(dy, dy_bf16, rowscale, h1, h2, w1, wd, w2, dh1, dxn2, dwd, dbd, scratch, B, hw, st);
  if (rc != WM_OK || !scratch) return rc;
  const int Ch = 4 * C, P = B * (hw / TP);
  WmReduceJob jw{}, jb{};
  jw.src = scratch; jw.pstride = (int64_t)Ch * 10; jw.src_ld = 10; jw.count = P; jw.rows = Ch; jw.cols = 9;
  jw.dst = dwd; jw.dst_rs = 9; jw.dst_cs = 1;
  jb = jw;
  jb.src = scratch + 9; jb.cols = 1; jb.dst = dbd; jb.dst_rs = 1;
  if (defer) {
    defer[0] = jw;
    defer[1] = jb;
    return WM_OK;
  }
  WmReduceJob both[2] = {jw, jb};
  return wm_reduce_jobs(both, 2, stream);
}
