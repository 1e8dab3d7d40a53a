// Weight-gradient GEMM (bf16 MFMA mode):  C[M,N] += sum_k opA(A[k,m]) * kscale[k] * opB(B[k,n]),  C2[m] += sum_k A'[k,m]
// i.e. dW = dY^T X and db = colsum(dY) with K = all tokens of the batch.  Both operands are row-contiguous along the
// NON-reduction dim, which is the wrong way round for MFMA fragments (a lane needs 8 consecutive k).  Instead of
// transposing in registers (the generic kernel's path: ~2000 VALU instructions per K-tile, issue-bound at ~30 TF/s)
// the tiles are copied to LDS exactly as they lie in HBM ([k][m], 16-byte moves, no ALU) and the fragments are
// fetched with gfx950's transposing LDS read ds_read_b64_tr_b16 (4 k x 16 columns per 16-lane group, delivered
// column-major), two reads per 8-element fragment.
//   * fp32 sources (residual-stream gradients) are converted while staging; bf16 sources are moved untouched.
//   * K is split over the chip (split-K, fp32 atomics into the per-step-zeroed gradient arena); workgroups of one
//     K range get ids equal mod 8 so they share an XCD's L2.
//   * the bias gradient is the column sum of the staged A tile, done by the bx == 0 workgroups only.
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

constexpr int TBK = 32;                                    // k rows per tile
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int ld, int col0, int kstep, int lane) {
  // MFMA 32x32x16 operand for 32 columns starting at col0, k = 16*kstep .. +15 (lane l: column l&31, k = 8*(l>>5) + j)
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const __bf16* a = tile + (16 * kstep + 8 * (g >> 1) + q) * ld + col0 + 16 * (g & 1) + 4 * p;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * ld));
  bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return f;
}

// stage a [TBK][R] tile (row-contiguous source) into LDS [TBK][R + 32] bf16
template <typename ST, int R> struct TnStage;
template <int R> struct TnStage<__bf16, R> {
  static constexpr int CPR = R / 8, NCH = TBK * CPR, PT = (NCH + 255) / 256;     // 16-byte chunks
  uint4 v[PT];
  __device__ __forceinline__ void load(const void* p, int64_t ld, int c0, int cmax, int k0, int kend, int tid) {
    const __bf16* src = static_cast<const __bf16*>(p);
#pragma unroll
    for (int q = 0; q < PT; ++q) {
      const int ch = tid + 256 * q, kr = ch / CPR, c = c0 + 8 * (ch % CPR);
      uint4 x = make_uint4(0, 0, 0, 0);
      if (ch < NCH && k0 + kr < kend) {
        const __bf16* s = src + (int64_t)(k0 + kr) * ld + c;
        if (c + 7 < cmax) {
          x = *reinterpret_cast<const uint4*>(s);
        } else {
          uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (c + j < cmax) w[j >> 1] |= (uint32_t)reinterpret_cast<const uint16_t*>(s)[j] << (16 * (j & 1));
          x = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
      v[q] = x;
    }
  }
  __device__ __forceinline__ void store(__bf16* t, int op, const float* kscale, int kps, int k0, int tid) const {
    constexpr int LD = R + 32;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
      const int ch = tid + 256 * q, kr = ch / CPR, cl = 8 * (ch % CPR);
      if (ch < NCH) {
        uint4 x = v[q];
        if (op || kscale) {
          const float sc = kscale ? kscale[(k0 + kr) / kps] : 1.f;
          uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float a = bf_lo(w[j]), b = bf_hi(w[j]);
            if (op) { a = gelu_f(a); b = gelu_f(b); }
            w[j] = pack_bf2(a * sc, b * sc);
          }
          x = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *reinterpret_cast<uint4*>(t + kr * LD + cl) = x;
      }
    }
  }
};
template <int R> struct TnStage<float, R> {
  static constexpr int CPR = R / 4, NCH = TBK * CPR, PT = (NCH + 255) / 256;     // float4 chunks
  float4 v[PT];
  __device__ __forceinline__ void load(const void* p, int64_t ld, int c0, int cmax, int k0, int kend, int tid) {
    const float* src = static_cast<const float*>(p);
#pragma unroll
    for (int q = 0; q < PT; ++q) {
      const int ch = tid + 256 * q, kr = ch / CPR, c = c0 + 4 * (ch % CPR);
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ch < NCH && k0 + kr < kend) {
        const float* s = src + (int64_t)(k0 + kr) * ld + c;
        if (c + 3 < cmax) {
          x = *reinterpret_cast<const float4*>(s);
        } else {
          float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c + j < cmax) e[j] = s[j];
          x = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
      v[q] = x;
    }
  }
  __device__ __forceinline__ void store(__bf16* t, int op, const float* kscale, int kps, int k0, int tid) const {
    constexpr int LD = R + 32;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
      const int ch = tid + 256 * q, kr = ch / CPR, cl = 4 * (ch % CPR);
      if (ch < NCH) {
        float4 x = v[q];
        if (op) { x.x = gelu_f(x.x); x.y = gelu_f(x.y); x.z = gelu_f(x.z); x.w = gelu_f(x.w); }
        const float sc = kscale ? kscale[(k0 + kr) / kps] : 1.f;
        *reinterpret_cast<uint2*>(t + kr * LD + cl) = make_uint2(pack_bf2(x.x * sc, x.y * sc), pack_bf2(x.z * sc, x.w * sc));
      }
    }
  }
};

template <typename AT, typename BT, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const WmGemm g, const int klen, const int tiles_n, const int tiles_m,
                                                      const int ksplit) {
  constexpr int BM = 64 * TM, BN = 64 * TN, LDA = BM + 32, LDB = BN + 32;
  __shared__ __attribute__((aligned(16))) __bf16 As[2][TBK * LDA];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2][TBK * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  int bx, by, bz;
  {
    const int L = blockIdx.x, j = L & 7, sq = L >> 3, tiles = tiles_n * tiles_m, t = sq % tiles;
    bz = (sq / tiles) * 8 + j;
    bx = t % tiles_n; by = t / tiles_n;
    if (bz >= ksplit) return;                              // whole workgroup leaves together
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * klen, kend = min(g.K, kbeg + klen);
  const bool do_bias = g.b_ones && bx == 0;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float bsum = 0.f;

  TnStage<AT, BM> ra;
  TnStage<BT, BN> rb;
  ra.load(g.A, g.lda, m0, g.M, kbeg, kend, tid);
  rb.load(g.B, g.ldb, n0, g.N, kbeg, kend, tid);
  ra.store(As[0], 0, g.kscale, g.k_per_scale, kbeg, tid);
  rb.store(Bs[0], g.b_op, nullptr, 1, kbeg, tid);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += TBK) {
    const bool more = k0 + TBK < kend;
    if (more) {
      ra.load(g.A, g.lda, m0, g.M, k0 + TBK, kend, tid);
      rb.load(g.B, g.ldb, n0, g.N, k0 + TBK, kend, tid);
    }
#pragma unroll
    for (int ks = 0; ks < TBK / 16; ++ks) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = tr_frag(As[buf], LDA, (wm * TM + i) * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = tr_frag(Bs[buf], LDB, (wn * TN + j) * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (do_bias && tid < BM) {
#pragma unroll 8
      for (int kr = 0; kr < TBK; ++kr) bsum += (float)As[buf][kr * LDA + tid];
    }
    if (more) {
      ra.store(As[buf ^ 1], 0, g.kscale, g.k_per_scale, k0 + TBK, tid);
      rb.store(Bs[buf ^ 1], g.b_op, nullptr, 1, k0 + TBK, tid);
    }
    __syncthreads();
    buf ^= 1;
  }
  float* C = static_cast<float*>(g.C);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (wm * TM + i) * 32 + mma32_row(e, h);
        if (m < g.M && n < g.N) atomicAdd(C + (int64_t)m * g.ldc + n, acc[i][j][e]);
      }
    }
  if (do_bias && tid < BM && m0 + tid < g.M) atomicAdd(g.C2 + m0 + tid, bsum);
}

template <typename AT, typename BT, int TM, int TN>
int launch_tn(const WmGemm& g, hipStream_t st) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN, tiles = tiles_m * tiles_n;
  int ksplit = g.ksplit > 0 ? g.ksplit : (768 + tiles - 1) / tiles;
  const int maxs = (g.K + 255) / 256;
  if (ksplit > maxs) ksplit = maxs;
  if (ksplit < 1) ksplit = 1;
  int klen = (g.K + ksplit - 1) / ksplit;
  klen = (klen + TBK - 1) / TBK * TBK;
  ksplit = (g.K + klen - 1) / klen;
  const int groups = (ksplit + 7) / 8 * tiles;
  hipLaunchKernelGGL((gemm_tn_kernel<AT, BT, TM, TN>), dim3(8 * groups), dim3(256), 0, st, g, klen, tiles_n, tiles_m, ksplit);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

template <typename AT, typename BT>
int dispatch_tn(const WmGemm& g, hipStream_t st) {
  if (g.M > 64 && g.N > 64) return launch_tn<AT, BT, 2, 2>(g, st);
  if (g.M > 64) return launch_tn<AT, BT, 2, 1>(g, st);
  return launch_tn<AT, BT, 1, 1>(g, st);
}

}  // namespace

// called from wm_gemm for mode 1, a_kc == 0, b_kc == 0, atomic_out (the weight-gradient form)
int wm_gemm_tn_bf16(const WmGemm& g, hipStream_t st) {
  if (g.a_bf16 && g.b_bf16) return dispatch_tn<__bf16, __bf16>(g, st);
  if (g.a_bf16) return dispatch_tn<__bf16, float>(g, st);
  if (g.b_bf16) return dispatch_tn<float, __bf16>(g, st);
  return dispatch_tn<float, float>(g, st);
}
