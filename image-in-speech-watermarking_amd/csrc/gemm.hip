// MFMA GEMM with fused operand transforms and epilogue (see include/wm_kernels.h: WmGemm).
//
// C[M,N] = epi( A[M,K] * B[N,K]^T ).  Either operand may be K-contiguous or row-contiguous in HBM;
// both are staged global -> registers -> (convert / GELU / scale / transpose) -> LDS as K-contiguous
// rows, so the MFMA fragment reads are identical in all four layout combinations:
//   forward  Y  = X  W^T        A kc, B kc
//   dX          = dY W          A kc, B row-contiguous
//   dW (+db)    = dY^T [X | 1]  A row-contiguous, B row-contiguous (+ virtual ones row)
// One workgroup = 256 threads = 4 waves; each wave owns TM x TN tiles of 32x32 (MFMA 32x32xK, f32
// accumulators).  LDS is double-buffered: the next K-tile's global loads are issued before the current
// tile's MFMAs and written to the other buffer after them, one barrier per K-tile.  The tile shape is
// picked per launch so that the grid covers the 256 CUs (small-M stages use 64x64 tiles).
#include <cstdlib>
#include "common.h"
#include "tn_stage.h"
#include "../../include/wm_kernels.h"

namespace {

template <int R, int BK> struct TileRegs {
  static constexpr int KC_N = R * BK / 4 / 256 > 0 ? R * BK / 4 / 256 : 1;                  // float4 per thread, K-contiguous
  static constexpr int TPK = R / 4;                                                        // threads per 4-wide k block
  static constexpr int KBP = 256 / TPK;                                                    // k blocks per pass
  static constexpr int PASSES = (BK / 4 + KBP - 1) / KBP;
  static constexpr int N = KC_N > 4 * PASSES ? KC_N : 4 * PASSES;
  float4 v[N];
};

template <int R, int BK>
__device__ __forceinline__ void load_tile(TileRegs<R, BK>& t, const float* __restrict__ p, bool kc, int64_t ld,
                                          int row0, int rmax, int k0, int kend, bool ones, int tid) {
  using TR = TileRegs<R, BK>;
  if (kc) {
    constexpr int TPR = BK / 4;                         // threads per row
    constexpr int RPP = 256 / TPR;                      // rows per pass
    const int r = tid / TPR, kq = tid % TPR;
    const int k = k0 + 4 * kq;
#pragma unroll
    for (int q = 0; q < TR::KC_N; ++q) {
      const int rl = r + RPP * q;
      const int row = row0 + rl;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rl < R && k < kend) {
        if (row < rmax) v = *reinterpret_cast<const float4*>(p + (int64_t)row * ld + k);
        else if (ones && row == rmax) v = make_float4(1.f, 1.f, 1.f, 1.f);
      }
      t.v[q] = v;
    }
  } else {
    const int kb = tid / TR::TPK, r4 = tid % TR::TPK;
    const int row = row0 + 4 * r4;
#pragma unroll
    for (int q = 0; q < TR::PASSES; ++q) {
      const int kbb = kb + q * TR::KBP;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = k0 + 4 * kbb + i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kbb < BK / 4 && k < kend) {
          const float* src = p + (int64_t)k * ld + row;
          if (row + 3 < rmax) {
            v = *reinterpret_cast<const float4*>(src);
          } else {
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int rr = row + j;
              e[j] = rr < rmax ? src[j] : ((ones && rr == rmax) ? 1.f : 0.f);
            }
            v = make_float4(e[0], e[1], e[2], e[3]);
          }
        }
        t.v[q * 4 + i] = v;
      }
    }
  }
}

__device__ __forceinline__ float apply_op(float x, int op) { return op == 1 ? gelu_f(x) : x; }

template <int R, int BK, typename LT>
__device__ __forceinline__ void store_tile(const TileRegs<R, BK>& t, LT* __restrict__ s, bool kc, int op,
                                           const float* __restrict__ kscale, int kps, int row0, int rmax,
                                           int k0, int tid) {
  using TR = TileRegs<R, BK>;
  constexpr int LD = BK + LdsT<LT>::PAD;
  if (kc) {
    constexpr int TPR = BK / 4;
    constexpr int RPP = 256 / TPR;
    const int r = tid / TPR, kq = tid % TPR;
    float sc = 1.f;
    if (kscale) sc = kscale[(k0 + 4 * kq) / kps];
#pragma unroll
    for (int q = 0; q < TR::KC_N; ++q) {
      const int rl = r + RPP * q;
      if (rl < R) {
        float4 v = t.v[q];
        if (op && row0 + rl < rmax) {
          v.x = apply_op(v.x, op); v.y = apply_op(v.y, op); v.z = apply_op(v.z, op); v.w = apply_op(v.w, op);
        }
        LdsT<LT>::store4(s + rl * LD + 4 * kq, v.x * sc, v.y * sc, v.z * sc, v.w * sc);
      }
    }
  } else {
    const int kb = tid / TR::TPK, r4 = tid % TR::TPK;
#pragma unroll
    for (int q = 0; q < TR::PASSES; ++q) {
      const int kbb = kb + q * TR::KBP;
      if (kbb < BK / 4) {
        float sc = 1.f;
        if (kscale) sc = kscale[(k0 + 4 * kbb) / kps];
        const float4 a = t.v[q * 4 + 0], b = t.v[q * 4 + 1], c = t.v[q * 4 + 2], d = t.v[q * 4 + 3];
        const float m[4][4] = {{a.x, b.x, c.x, d.x}, {a.y, b.y, c.y, d.y}, {a.z, b.z, c.z, d.z}, {a.w, b.w, c.w, d.w}};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rl = 4 * r4 + j;
          float e0 = m[j][0], e1 = m[j][1], e2 = m[j][2], e3 = m[j][3];
          if (op && row0 + rl < rmax) {
            e0 = apply_op(e0, op); e1 = apply_op(e1, op); e2 = apply_op(e2, op); e3 = apply_op(e3, op);
          }
          LdsT<LT>::store4(s + rl * LD + 4 * kbb, e0 * sc, e1 * sc, e2 * sc, e3 * sc);
        }
      }
    }
  }
}

// ---- bf16 operands in HBM (wide activations / weight copies in bf16 mode): 16-byte loads of 8 elements ----------
template <int R, int BK> struct TileRegsH {
  static constexpr int TPR = BK / 8;                       // threads per row (K-contiguous)
  static constexpr int RPP = 256 / TPR;
  static constexpr int KC_N = (R + RPP - 1) / RPP;
  // row-contiguous source (transposing): one unit = RW rows x 8 k; RW chosen so that ~256 units exist per tile
  static constexpr int RW = R >= 256 ? 8 : (R >= 128 ? 4 : 2);
  static constexpr int NU = (R / RW) * (BK / 8);
  static constexpr int UPT = (NU + 255) / 256;             // units per thread
  static constexpr int WPL = RW / 2;                       // 32-bit words per k-line of a unit
  static constexpr int NW = UPT * 8 * WPL;                 // words held for the transposing path
  static constexpr int N = (4 * KC_N > NW ? 4 * KC_N : NW);
  uint32_t w[N];
};

template <int R, int BK>
__device__ __forceinline__ void load_tile_h(TileRegsH<R, BK>& t, const __bf16* __restrict__ p, bool kc, int64_t ld,
                                            int row0, int rmax, int k0, int kend, bool ones, int tid) {
  using TR = TileRegsH<R, BK>;
  if (kc) {
    const int r = tid / TR::TPR, kq = tid % TR::TPR;
    const int k = k0 + 8 * kq;
#pragma unroll
    for (int q = 0; q < TR::KC_N; ++q) {
      const int rl = r + TR::RPP * q, row = row0 + rl;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (rl < R && k < kend) {
        if (row < rmax) v = *reinterpret_cast<const uint4*>(p + (int64_t)row * ld + k);
        else if (ones && row == rmax) v = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
      }
      t.w[4 * q + 0] = v.x; t.w[4 * q + 1] = v.y; t.w[4 * q + 2] = v.z; t.w[4 * q + 3] = v.w;
    }
  } else {
    constexpr int RW = TR::RW, WPL = TR::WPL;
#pragma unroll
    for (int uq = 0; uq < TR::UPT; ++uq) {
      const int u = tid + 256 * uq;
      const int kb = u / (R / RW), rg = u % (R / RW);
      const int row = row0 + RW * rg;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = k0 + 8 * kb + i;
        uint32_t wv[WPL];
#pragma unroll
        for (int x = 0; x < WPL; ++x) wv[x] = 0;
        if (u < TR::NU && k < kend) {
          const __bf16* src = p + (int64_t)k * ld + row;
          if (row + RW - 1 < rmax) {
            if constexpr (WPL == 1) wv[0] = *reinterpret_cast<const uint32_t*>(src);
            else if constexpr (WPL == 2) { const uint2 v = *reinterpret_cast<const uint2*>(src); wv[0] = v.x; wv[1] = v.y; }
            else { const uint4 v = *reinterpret_cast<const uint4*>(src); wv[0] = v.x; wv[1] = v.y; wv[2] = v.z; wv[3] = v.w; }
          } else {
#pragma unroll
            for (int j = 0; j < RW; ++j) {
              const int rr = row + j;
              const uint32_t e = rr < rmax ? reinterpret_cast<const uint16_t*>(src)[j] : ((ones && rr == rmax) ? 0x3F80u : 0u);
              wv[j >> 1] |= e << (16 * (j & 1));
            }
          }
        }
#pragma unroll
        for (int x = 0; x < WPL; ++x) t.w[(uq * 8 + i) * WPL + x] = wv[x];
      }
    }
  }
}

template <int R, int BK>
__device__ __forceinline__ void store_tile_h(const TileRegsH<R, BK>& t, __bf16* __restrict__ s, bool kc, int op,
                                             const float* __restrict__ kscale, int kps, int row0, int rmax, int k0,
                                             int tid) {
  using TR = TileRegsH<R, BK>;
  constexpr int LD = BK + LdsT<__bf16>::PAD;
  const bool scaled = kscale != nullptr;
  if (kc) {
    const int r = tid / TR::TPR, kq = tid % TR::TPR;
    float sc = 1.f;
    if (scaled) sc = kscale[(k0 + 8 * kq) / kps];
#pragma unroll
    for (int q = 0; q < TR::KC_N; ++q) {
      const int rl = r + TR::RPP * q;
      if (rl < R) {
        uint32_t w[4] = {t.w[4 * q], t.w[4 * q + 1], t.w[4 * q + 2], t.w[4 * q + 3]};
        const bool real = row0 + rl < rmax;
        if ((op && real) || scaled) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float a = bf_lo(w[j]), b = bf_hi(w[j]);
            if (op && real) { a = apply_op(a, op); b = apply_op(b, op); }
            w[j] = pack_bf2(a * sc, b * sc);
          }
        }
        *reinterpret_cast<uint4*>(s + rl * LD + 8 * kq) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
  } else {
    constexpr int RW = TR::RW, WPL = TR::WPL;
#pragma unroll
    for (int uq = 0; uq < TR::UPT; ++uq) {
      const int u = tid + 256 * uq;
      if (u < TR::NU) {
        const int kb = u / (R / RW), rg = u % (R / RW);
        float sc = 1.f;
        if (scaled) sc = kscale[(k0 + 8 * kb) / kps];
#pragma unroll
        for (int j = 0; j < RW; ++j) {                      // output row j of the unit: k = 0..7 from word (j>>1) of each k-line
          const int rl = RW * rg + j;
          const bool real = row0 + rl < rmax;
          uint32_t o[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const uint32_t a = t.w[(uq * 8 + 2 * kk) * WPL + (j >> 1)], b = t.w[(uq * 8 + 2 * kk + 1) * WPL + (j >> 1)];
            const uint32_t lo = (j & 1) ? (a >> 16) : (a & 0xffffu);
            const uint32_t hi = (j & 1) ? (b & 0xffff0000u) : (b << 16);
            if ((op && real) || scaled) {
              float fa = __uint_as_float(lo << 16), fb = __uint_as_float(hi);
              if (op && real) { fa = apply_op(fa, op); fb = apply_op(fb, op); }
              o[kk] = pack_bf2(fa * sc, fb * sc);
            } else {
              o[kk] = lo | hi;
            }
          }
          *reinterpret_cast<uint4*>(s + rl * LD + 8 * kb) = make_uint4(o[0], o[1], o[2], o[3]);
        }
      }
    }
  }
}

// operand stager: picks the float or bf16 global->LDS path from the source type
template <typename ST, int R, int BK, typename LT> struct Stager {
  TileRegs<R, BK> regs;
  __device__ __forceinline__ void load(const void* p, bool kc, int64_t ld, int row0, int rmax, int k0, int kend, bool ones, int tid) {
    load_tile<R, BK>(regs, static_cast<const float*>(p), kc, ld, row0, rmax, k0, kend, ones, tid);
  }
  __device__ __forceinline__ void store(LT* s, bool kc, int op, const float* ks, int kps, int row0, int rmax, int k0, int tid) const {
    store_tile<R, BK, LT>(regs, s, kc, op, ks, kps, row0, rmax, k0, tid);
  }
};
template <int R, int BK> struct Stager<__bf16, R, BK, __bf16> {
  TileRegsH<R, BK> regs;
  __device__ __forceinline__ void load(const void* p, bool kc, int64_t ld, int row0, int rmax, int k0, int kend, bool ones, int tid) {
    load_tile_h<R, BK>(regs, static_cast<const __bf16*>(p), kc, ld, row0, rmax, k0, kend, ones, tid);
  }
  __device__ __forceinline__ void store(__bf16* s, bool kc, int op, const float* ks, int kps, int row0, int rmax, int k0, int tid) const {
    store_tile_h<R, BK>(regs, s, kc, op, ks, kps, row0, rmax, k0, tid);
  }
};

template <typename BT, int R, int BK, typename LT, bool TR> struct BStager { using type = Stager<BT, R, BK, LT>; };
template <typename BT, int R, int BK> struct BStager<BT, R, BK, __bf16, true> { using type = TnStage<BT, R, BK>; };

// TM x TN register-blocked MFMA over one K-tile held in LDS
template <typename LT, int TM, int TN, int BK> struct TileMma;
template <int TM, int TN, int BK> struct TileMma<__bf16, TM, TN, BK> {
  static __device__ __forceinline__ void run(f32x16 (&acc)[TM][TN], const __bf16* a_rows, const __bf16* b_rows, int h) {
    constexpr int LD = BK + LdsT<__bf16>::PAD;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(a_rows + i * 32 * LD + 16 * ks + 8 * h);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(b_rows + j * 32 * LD + 16 * ks + 8 * h);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
};
template <int TM, int TN, int BK> struct TileMma<float, TM, TN, BK> {
  static __device__ __forceinline__ void run(f32x16 (&acc)[TM][TN], const float* a_rows, const float* b_rows, int h) {
    constexpr int LD = BK + LdsT<float>::PAD;
#pragma unroll
    for (int ks = 0; ks < BK / 8; ++ks) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(a_rows + i * 32 * LD + 8 * ks + 4 * h);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(b_rows + j * 32 * LD + 8 * ks + 4 * h);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
  }
};

// B_TR: the B operand is row-contiguous bf16 (dX: the [out,in] weight read along `in`): staged as it lies ([k][n]) and
// read through transposing LDS reads instead of being transposed with VALU shuffles.
template <typename LT, typename AT, typename BT, bool B_TR, int WM, int WN, int TM, int TN, int BK>
__global__ __launch_bounds__(256) void gemm_kernel(const WmGemm g, const int klen, const bool vec_ok, const int tiles_n,
                                                   const int tiles_m, const int ksplit) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int LD = BK + LdsT<LT>::PAD;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int LDBT = BN + 32;                              // row stride of the as-it-lies B image
  constexpr int BSZ = B_TR ? BK * LDBT : BN * LD;            // elements per B buffer
  LT* As = reinterpret_cast<LT*>(smem_raw);                  // [2][BM*LD]
  LT* Bs = As + 2 * BM * LD;                                 // [2][BSZ]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware tile order (speed only, never correctness): workgroups are dealt round-robin over the 8 XCDs, each
  // with a private L2.  Workgroups that re-read the same activation slab are given ids that are equal mod 8:
  //   plain GEMM  - all column tiles of one row tile (they share the A row slab),
  //   split-K dW  - all output tiles of one K range (they share both K slabs).
  int bx, by, bz;
  {
    const int L = blockIdx.x, j = L & 7, sq = L >> 3;
    if (ksplit > 1) {
      const int tiles = tiles_n * tiles_m, t = sq % tiles;
      bz = (sq / tiles) * 8 + j;
      bx = t % tiles_n; by = t / tiles_n;
      if (bz >= ksplit) return;
    } else {
      bz = 0;
      bx = sq % tiles_n;
      by = (sq / tiles_n) * 8 + j;
      if (by >= tiles_m) return;
    }
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * klen;
  const int kend = min(g.K, kbeg + klen);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  Stager<AT, BM, BK, LT> ra;
  typename BStager<BT, BN, BK, LT, B_TR>::type rb;
  ra.load(g.A, g.a_kc, g.lda, m0, g.M, kbeg, kend, false, tid);
  ra.store(As, g.a_kc, g.a_op, g.kscale, g.k_per_scale, m0, g.M, kbeg, tid);
  if constexpr (B_TR) {
    rb.load(g.B, g.ldb, n0, g.N, kbeg, kend, tid);
    rb.store(reinterpret_cast<__bf16*>(Bs), g.b_op, nullptr, 1, kbeg, tid);
  } else {
    rb.load(g.B, g.b_kc, g.ldb, n0, g.N, kbeg, kend, g.b_ones != 0, tid);
    rb.store(Bs, g.b_kc, g.b_op, nullptr, 1, n0, g.N, kbeg, tid);
  }
  __syncthreads();

  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {
      ra.load(g.A, g.a_kc, g.lda, m0, g.M, k0 + BK, kend, false, tid);
      if constexpr (B_TR) rb.load(g.B, g.ldb, n0, g.N, k0 + BK, kend, tid);
      else rb.load(g.B, g.b_kc, g.ldb, n0, g.N, k0 + BK, kend, g.b_ones != 0, tid);
    }
    if constexpr (B_TR) {
      const __bf16* Ab = reinterpret_cast<const __bf16*>(As) + buf * BM * LD + ((wm * TM) * 32 + r) * LD;
      const __bf16* Bb = reinterpret_cast<const __bf16*>(Bs) + buf * BSZ;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * LD + 16 * ks + 8 * h);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = tr_frag(Bb, LDBT, (wn * TN + j) * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
      TileMma<LT, TM, TN, BK>::run(acc, As + buf * BM * LD + ((wm * TM) * 32 + r) * LD,
                                   Bs + buf * BSZ + ((wn * TN) * 32 + r) * LD, h);
    }
    if (more) {
      ra.store(As + (buf ^ 1) * BM * LD, g.a_kc, g.a_op, g.kscale, g.k_per_scale, m0, g.M, k0 + BK, tid);
      if constexpr (B_TR) rb.store(reinterpret_cast<__bf16*>(Bs) + (buf ^ 1) * BSZ, g.b_op, nullptr, 1, k0 + BK, tid);
      else rb.store(Bs + (buf ^ 1) * BSZ, g.b_kc, g.b_op, nullptr, 1, n0, g.N, k0 + BK, tid);
    }
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue: accumulators -> LDS (fp32 tile) -> coalesced float4 rows -------------------------
  // The MFMA C layout gives each lane one column of 16 scattered rows (4-byte accesses, 128-B runs); going
  // through LDS turns every HBM access of the epilogue (bias / residual / gelu' operand / output) into full
  // 16-B-per-lane row segments.
  constexpr int LDC = BN + 4;
  float* Cs = reinterpret_cast<float*>(smem_raw);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        Cs[((wm * TM + i) * 32 + mma32_row(e, h)) * LDC + (wn * TN + j) * 32 + r] = acc[i][j][e];
  __syncthreads();
  constexpr int TPRW = BN / 4, RPP = 256 / TPRW;
  const int c4 = tid % TPRW;
  const int n = n0 + 4 * c4;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool full = vec_ok && n + 3 < g.N;
  if (full && g.bias) bias4 = *reinterpret_cast<const float4*>(g.bias + n);
  for (int rr = tid / TPRW; rr < BM; rr += RPP) {
    const int m = m0 + rr;
    if (m >= g.M) break;
    float4 v = *reinterpret_cast<const float4*>(Cs + rr * LDC + 4 * c4);
    if (full) {
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      if (g.rowscale) {
        const float sc = g.rowscale[m / g.rows_per_scale];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
      }
      if (g.gaux) {
        float4 a;
        if (g.gaux_bf16) {
          const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(g.gaux) + (int64_t)m * g.ldg + n);
          a = make_float4(bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y));
        } else {
          a = *reinterpret_cast<const float4*>(static_cast<const float*>(g.gaux) + (int64_t)m * g.ldg + n);
        }
        if (g.gaux_raw) { v.x *= a.x; v.y *= a.y; v.z *= a.z; v.w *= a.w; }
        else { v.x *= gelu_grad_f(a.x); v.y *= gelu_grad_f(a.y); v.z *= gelu_grad_f(a.z); v.w *= gelu_grad_f(a.w); }
      }
      if (g.res) {
        const float4 a = *reinterpret_cast<const float4*>(g.res + (int64_t)m * g.ldr + n);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      if (g.c_bf16) {
        *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(g.C) + (int64_t)m * g.ldc + n) =
            make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
      } else {
        float* dst = static_cast<float*>(g.C) + (int64_t)m * g.ldc + n;
        if (g.atomic_out) {
          atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
        } else {
          *reinterpret_cast<float4*>(dst) = v;
        }
      }
    } else {
      const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nn = n + q;
        float val = e4[q];
        if (nn < g.N) {
          if (g.bias) val += g.bias[nn];
          if (g.rowscale) val *= g.rowscale[m / g.rows_per_scale];
          if (g.gaux) {
            const float av = g.gaux_bf16 ? (float)reinterpret_cast<const __bf16*>(g.gaux)[(int64_t)m * g.ldg + nn]
                                         : static_cast<const float*>(g.gaux)[(int64_t)m * g.ldg + nn];
            val *= g.gaux_raw ? av : gelu_grad_f(av);
          }
          if (g.res) val += g.res[(int64_t)m * g.ldr + nn];
          if (g.c_bf16) {
            reinterpret_cast<__bf16*>(g.C)[(int64_t)m * g.ldc + nn] = (__bf16)val;
          } else {
            float* dst = static_cast<float*>(g.C) + (int64_t)m * g.ldc + nn;
            if (g.atomic_out) atomicAdd(dst, val); else *dst = val;
          }
        } else if (g.b_ones && nn == g.N) {
          atomicAdd(g.C2 + m, val);
        }
      }
    }
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct Plan { int tiles_m, tiles_n, ksplit, klen; };

template <int BM, int BN, int BK>
Plan make_plan(const WmGemm& g, int ksplit) {
  const int ncols = g.N + (g.b_ones ? 1 : 0);
  Plan p;
  p.tiles_n = (ncols + BN - 1) / BN;
  p.tiles_m = (g.M + BM - 1) / BM;
  if (ksplit <= 0) {
    ksplit = 1;
    if (g.atomic_out) {                       // weight-gradient GEMMs: tiny M x N, K = all tokens -> split K over the chip
      const int tiles = p.tiles_n * p.tiles_m;
      const int want = (1024 + tiles - 1) / tiles;
      const int maxs = (g.K + 511) / 512;                      // at least 512 of K per workgroup (bounds atomic traffic)
      ksplit = want < 1 ? 1 : (want > maxs ? maxs : want);
      if (ksplit < 1) ksplit = 1;
    }
  }
  p.klen = (g.K + ksplit - 1) / ksplit;
  p.klen = ((p.klen + BK - 1) / BK) * BK;
  p.ksplit = (g.K + p.klen - 1) / p.klen;
  return p;
}

template <typename LT, typename AT, typename BT, bool B_TR, int WM, int WN, int TM, int TN, int BK>
int launch(const WmGemm& g, int ksplit, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int LD = BK + LdsT<LT>::PAD;
  constexpr size_t bsz = B_TR ? BK * (BN + 32) : BN * LD;
  constexpr size_t stage = sizeof(LT) * 2 * (BM * LD + bsz), ctile = sizeof(float) * BM * (BN + 4);
  constexpr size_t smem = stage > ctile ? stage : ctile;
  const Plan p = make_plan<BM, BN, BK>(g, ksplit);
  static bool attr = false;
  if (!attr) {
    if (smem > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<LT, AT, BT, B_TR, WM, WN, TM, TN, BK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  const int groups = p.ksplit > 1 ? (p.ksplit + 7) / 8 * p.tiles_n * p.tiles_m : (p.tiles_m + 7) / 8 * p.tiles_n;
  dim3 grid(8 * groups);
  const bool vec_ok = al16(g.C) && !(g.ldc & 7) && (!g.bias || al16(g.bias)) && (!g.res || (al16(g.res) && !(g.ldr & 3))) &&
                      (!g.gaux || (al16(g.gaux) && !(g.ldg & 7)));
  hipLaunchKernelGGL((gemm_kernel<LT, AT, BT, B_TR, WM, WN, TM, TN, BK>), grid, dim3(256), smem, st, g, p.klen, vec_ok,
                     p.tiles_n, p.tiles_m, p.ksplit);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

// Tile choice: the largest tile whose grid still gives every CU at least ~1.5 workgroups; the transformer's
// coarse stages (M = B*256 or B*64 tokens) otherwise leave most of the 256 CUs idle behind 128x128 tiles.
template <typename LT, typename AT, typename BT, bool B_TR, int BK>
int dispatch(const WmGemm& g, int ksplit, hipStream_t st) {
  const int ncols = g.N + (g.b_ones ? 1 : 0);
  if (ncols <= 32) return launch<LT, AT, BT, B_TR, 4, 1, 2, 1, BK>(g, ksplit, st);           // 256 x 32
  const int64_t t128 = (int64_t)((g.M + 127) / 128) * ((ncols + 127) / 128);
  const int64_t t12864 = (int64_t)((g.M + 127) / 128) * ((ncols + 63) / 64);
  const int64_t want = 384;
  // weight-gradient GEMMs get ~4x more workgroups from split-K; more splits than that cost more in fp32 atomics
  // (1.3 TB/s chip-wide) than the larger tile saves in operand re-reads (measured: 128x128 + deep split-K was slower)
  const int ks = g.atomic_out ? 4 : 1;
  if (ncols > 64 && t128 * ks >= want) return launch<LT, AT, BT, B_TR, 2, 2, 2, 2, BK>(g, ksplit, st);       // 128 x 128
  if (t12864 * ks >= want || g.M >= 32768) return launch<LT, AT, BT, B_TR, 2, 2, 2, 1, BK>(g, ksplit, st);   // 128 x 64
  return launch<LT, AT, BT, B_TR, 2, 2, 1, 1, BK>(g, ksplit, st);                            // 64 x 64
}

}  // namespace

int wm_gemm_fast_bf16(const WmGemm& g, hipStream_t st);   // gemm_fast.hip: branch-free bf16 kernels, -1 = not applicable
int wm_gemm_fast_split(const WmGemm& g, hipStream_t st);  // ... their split-precision instances (mode 2), -1 = not applicable
int wm_gemm_big_bf16(const WmGemm& g, hipStream_t st);    // gemm_big.hip: 256 x 256 tiles for the compute-side shapes, -1 = not applicable
int wm_gemm_big_tn_bf16(const WmGemm& g, hipStream_t st); // ... its weight-gradient form

thread_local int wm_gemm_family_tls = WM_FAM_GENERIC;
extern "C" int32_t wm_last_gemm_family(void) { return wm_gemm_family_tls; }
extern "C" int wm_abi_version(void) { return 7; }
// split-K slices of one weight-gradient launch: the 128 x 128 kernel writes up to ~384 tiles, the 256 x 256 kernel one tile
// per CU (256 x 65536 floats = 67 MB)
extern "C" int64_t wm_gemm_scratch_floats(void) { return 256LL * 256 * 256 + 1024 * 1024; }

// The LayerNorm of the output rows can ride in the epilogue when a tile holds whole rows: N = 32 / 64 / 128 on the 128 x 128
// family (any row count that is a multiple of 256), N = 256 on the 256 x 256 tiles (which need at least 256 of them)
extern "C" int32_t wm_gemm_ln_fusable(int32_t M, int32_t N, int32_t K, int32_t mode, int32_t a_bf16) {
  if (mode != 1 || !a_bf16 || M <= 0 || (M % 256) || K < 8 || (K % 8)) return 0;
  if (N == 32 || N == 64 || N == 128) return 1;
  if (N == 256) {
    const char* off = getenv("WM_GEMM_BIG_OFF");
    return (K % 64) == 0 && K >= 128 && (M / 256) >= 256 && !(off && off[0] == '1');
  }
  return 0;
}

extern "C" int wm_gemm(const WmGemm* gp, void* stream) {
  if (!gp) return WM_EINVAL;
  const WmGemm& g = *gp;
  if (g.defer) *g.defer = WmReduceJob{};                       // count 0: nothing deferred unless the split-K path fills it in
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || !g.A || !g.B || !g.C) return WM_EINVAL;
  const int amask = g.a_bf16 ? 7 : 3, bmask = g.b_bf16 ? 7 : 3;
  if (!al16(g.A) || !al16(g.B) || (g.lda & amask) || (g.ldb & bmask)) return WM_EINVAL;
  if (g.a_kc && (g.K & amask)) return WM_EINVAL;
  if (g.b_kc && (g.K & bmask)) return WM_EINVAL;
  if (g.ksplit > 1 && !g.atomic_out) return WM_EINVAL;
  if (g.atomic_out && g.c_bf16) return WM_EINVAL;
  if ((g.mode == 0 || g.mode == 2) && (g.a_bf16 || g.b_bf16 || g.c_bf16)) return WM_EINVAL;   // parity modes keep fp32 storage end to end
  if (g.b_ones && !g.C2) return WM_EINVAL;
  if (g.kscale && (g.k_per_scale <= 0 || (g.k_per_scale & 3))) return WM_EINVAL;
  if (g.rowscale && g.rows_per_scale <= 0) return WM_EINVAL;
  if ((g.a_grid_n || g.a_jump_k || g.b_seg_cols) && (g.mode != 1 || g.ln_out)) return WM_EINVAL;   // grid addressing: bf16 mode, plain epilogues
  hipStream_t st = static_cast<hipStream_t>(stream);
  wm_gemm_family_tls = WM_FAM_GENERIC;
  if (g.ln_out) {                                              // fused LayerNorm: only the kernels that hold whole rows
    if (g.mode != 1 || !g.ln_gamma || !g.ln_beta || !g.ln_mean || !g.ln_rstd || !wm_gemm_ln_fusable(g.M, g.N, g.K, g.mode, g.a_bf16))
      return WM_EINVAL;
    int rc = g.N == 256 ? wm_gemm_big_bf16(g, st) : wm_gemm_fast_bf16(g, st);
    return rc >= 0 ? rc : WM_EINVAL;
  }
  if (g.mode == 2) {                                           // split precision; shapes the fast kernels decline run exactly
    const int rc = wm_gemm_fast_split(g, st);
    if (rc >= 0) return rc;
    return dispatch<float, float, float, false, 32>(g, g.ksplit, st);
  }
  if (g.mode == 0) return dispatch<float, float, float, false, 32>(g, g.ksplit, st);
  if (g.mode == 1) {
    int rc = wm_gemm_big_bf16(g, st);
    if (rc >= 0) return rc;
    rc = wm_gemm_big_tn_bf16(g, st);
    if (rc >= 0) return rc;
    rc = wm_gemm_fast_bf16(g, st);
    if (rc >= 0) return rc;
    if (g.a_grid_n || g.a_jump_k || g.b_seg_cols) return WM_EINVAL;    // grid addressing exists in the fast kernels only
  }
  if (g.mode == 1) {
    if (g.a_bf16 && g.b_bf16) return dispatch<__bf16, __bf16, __bf16, false, 64>(g, g.ksplit, st);
    if (g.a_bf16) return dispatch<__bf16, __bf16, float, false, 64>(g, g.ksplit, st);
    if (g.b_bf16) return dispatch<__bf16, float, __bf16, false, 64>(g, g.ksplit, st);
    return dispatch<__bf16, float, float, false, 64>(g, g.ksplit, st);
  }
  return WM_EINVAL;
}
