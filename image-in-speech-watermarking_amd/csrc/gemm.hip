// MFMA GEMM with fused operand transforms and epilogue (see include/wm_kernels.h: WmGemm).
//
// C[M,N] = epi( A[M,K] * B[N,K]^T ).  Either operand may be K-contiguous or row-contiguous in HBM;
// both are staged global -> registers -> (convert / GELU / scale / transpose) -> LDS as K-contiguous
// rows, so the MFMA fragment reads are identical in all four layout combinations:
//   forward  Y  = X  W^T        A kc, B kc
//   dX          = dY W          A kc, B row-contiguous
//   dW (+db)    = dY^T [X | 1]  A row-contiguous, B row-contiguous (+ virtual ones row)
// One workgroup = 256 threads = 4 waves; each wave owns TM x TN tiles of 32x32 (MFMA 32x32xK,
// f32 accumulators).  Next K-tile's global loads are issued before the current tile's MFMAs.
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

constexpr int BK = 32;

template <int R> struct TileRegs {
  static constexpr int N = (R >= 128) ? R / 32 : 4;
  float4 v[N];
};

template <int R>
__device__ __forceinline__ void load_tile(TileRegs<R>& t, const float* __restrict__ p, bool kc, int64_t ld,
                                          int row0, int rmax, int k0, int kend, bool ones, int tid) {
  if (kc) {
    const int r = tid >> 3, kq = tid & 7;
    const int k = k0 + 4 * kq;
#pragma unroll
    for (int q = 0; q < R / 32; ++q) {
      const int row = row0 + r + 32 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < kend) {
        if (row < rmax) v = *reinterpret_cast<const float4*>(p + (int64_t)row * ld + k);
        else if (ones && row == rmax) v = make_float4(1.f, 1.f, 1.f, 1.f);
      }
      t.v[q] = v;
    }
  } else {
    constexpr int TPK = R / 4;                                   // threads per 4-wide k block
    constexpr int KBP = 256 / TPK;                               // k blocks per pass
    constexpr int PASSES = (8 + KBP - 1) / KBP;
    const int kb = tid / TPK, r4 = tid % TPK;
    const int row = row0 + 4 * r4;
#pragma unroll
    for (int q = 0; q < PASSES; ++q) {
      const int kbb = kb + q * KBP;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = k0 + 4 * kbb + i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kbb < 8 && k < kend) {
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

template <int R, typename LT>
__device__ __forceinline__ void store_tile(const TileRegs<R>& t, LT* __restrict__ s, bool kc, int op,
                                           const float* __restrict__ kscale, int kps, int row0, int rmax,
                                           int k0, int tid) {
  constexpr int LD = BK + LdsT<LT>::PAD;
  if (kc) {
    const int r = tid >> 3, kq = tid & 7;
    float sc = 1.f;
    if (kscale) sc = kscale[(k0 + 4 * kq) / kps];
#pragma unroll
    for (int q = 0; q < R / 32; ++q) {
      const int rl = r + 32 * q;
      float4 v = t.v[q];
      if (op && row0 + rl < rmax) {
        v.x = apply_op(v.x, op); v.y = apply_op(v.y, op); v.z = apply_op(v.z, op); v.w = apply_op(v.w, op);
      }
      LdsT<LT>::store4(s + rl * LD + 4 * kq, v.x * sc, v.y * sc, v.z * sc, v.w * sc);
    }
  } else {
    constexpr int TPK = R / 4;
    constexpr int KBP = 256 / TPK;
    constexpr int PASSES = (8 + KBP - 1) / KBP;
    const int kb = tid / TPK, r4 = tid % TPK;
#pragma unroll
    for (int q = 0; q < PASSES; ++q) {
      const int kbb = kb + q * KBP;
      if (kbb < 8) {
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

template <typename LT, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_kernel(const WmGemm g, const int klen) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int LD = BK + LdsT<LT>::PAD;
  __shared__ __attribute__((aligned(16))) LT As[BM * LD];
  __shared__ __attribute__((aligned(16))) LT Bs[BN * LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * klen;
  const int kend = min(g.K, kbeg + klen);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  TileRegs<BM> ra;
  TileRegs<BN> rb;
  load_tile<BM>(ra, g.A, g.a_kc, g.lda, m0, g.M, kbeg, kend, false, tid);
  load_tile<BN>(rb, g.B, g.b_kc, g.ldb, n0, g.N, kbeg, kend, g.b_ones != 0, tid);

  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    store_tile<BM, LT>(ra, As, g.a_kc, g.a_op, g.kscale, g.k_per_scale, m0, g.M, k0, tid);
    store_tile<BN, LT>(rb, Bs, g.b_kc, g.b_op, nullptr, 1, n0, g.N, k0, tid);
    __syncthreads();
    if (k0 + BK < kend) {
      load_tile<BM>(ra, g.A, g.a_kc, g.lda, m0, g.M, k0 + BK, kend, false, tid);
      load_tile<BN>(rb, g.B, g.b_kc, g.ldb, n0, g.N, k0 + BK, kend, g.b_ones != 0, tid);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        mma32<LT>(acc[i][j], As + ((wm * TM + i) * 32 + r) * LD, Bs + ((wn * TN + j) * 32 + r) * LD, BK, h);
    __syncthreads();
  }

  // ---- epilogue --------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (wm * TM + i) * 32 + mma32_row(e, h);
        if (m >= g.M) continue;
        float v = acc[i][j][e];
        if (n < g.N) {
          if (g.bias) v += g.bias[n];
          if (g.rowscale) v *= g.rowscale[m / g.rows_per_scale];
          if (g.gaux) v *= gelu_grad_f(g.gaux[(int64_t)m * g.ldg + n]);
          if (g.res) v += g.res[(int64_t)m * g.ldr + n];
          float* dst = g.C + (int64_t)m * g.ldc + n;
          if (g.atomic_out) atomicAdd(dst, v); else *dst = v;
        } else if (g.b_ones && n == g.N) {
          atomicAdd(g.C2 + m, v);
        }
      }
    }
  }
}

template <typename LT, int WM, int WN, int TM, int TN>
int launch(const WmGemm& g, int ksplit, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int ncols = g.N + (g.b_ones ? 1 : 0);
  const int tiles_n = (ncols + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  if (ksplit <= 0) {
    ksplit = 1;
    if (g.atomic_out) {
      const int tiles = tiles_n * tiles_m;
      const int want = (1024 + tiles - 1) / tiles;
      const int maxs = (g.K + 255) / 256;
      ksplit = want < 1 ? 1 : (want > maxs ? maxs : want);
      if (ksplit < 1) ksplit = 1;
    }
  }
  int klen = (g.K + ksplit - 1) / ksplit;
  klen = ((klen + BK - 1) / BK) * BK;
  ksplit = (g.K + klen - 1) / klen;
  dim3 grid(tiles_n, tiles_m, ksplit);
  hipLaunchKernelGGL((gemm_kernel<LT, WM, WN, TM, TN>), grid, dim3(256), 0, st, g, klen);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

template <typename LT>
int dispatch(const WmGemm& g, int ksplit, hipStream_t st) {
  const int ncols = g.N + (g.b_ones ? 1 : 0);
  if (ncols <= 32) return launch<LT, 4, 1, 2, 1>(g, ksplit, st);
  if (ncols <= 64) return launch<LT, 2, 2, 2, 1>(g, ksplit, st);
  return launch<LT, 2, 2, 2, 2>(g, ksplit, st);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int wm_abi_version(void) { return 1; }

extern "C" int wm_gemm(const WmGemm* gp, void* stream) {
  if (!gp) return WM_EINVAL;
  const WmGemm& g = *gp;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || !g.A || !g.B || !g.C) return WM_EINVAL;
  if (!al16(g.A) || !al16(g.B) || (g.lda & 3) || (g.ldb & 3)) return WM_EINVAL;
  if (g.a_kc && (g.K & 3)) return WM_EINVAL;
  if (g.b_kc && (g.K & 3)) return WM_EINVAL;
  if (g.ksplit > 1 && !g.atomic_out) return WM_EINVAL;
  if (g.b_ones && !g.C2) return WM_EINVAL;
  if (g.kscale && (g.k_per_scale <= 0 || (g.k_per_scale & 3))) return WM_EINVAL;
  if (g.rowscale && g.rows_per_scale <= 0) return WM_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (g.mode == 0) return dispatch<float>(g, g.ksplit, st);
  if (g.mode == 1) return dispatch<__bf16>(g, g.ksplit, st);
  return WM_EINVAL;
}
