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
#include "tn_stage.h"
#include "../../include/wm_kernels.h"

namespace {

constexpr int TBK = 32;                                    // k rows per tile

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

  TnStage<AT, BM, TBK> ra;
  TnStage<BT, BN, TBK> rb;
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
