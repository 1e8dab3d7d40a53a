// 256 x 256 tile bf16 GEMMs, "ping-pong" schedule: the schedule of gemm_big.hip kept both waves of a SIMD in lockstep (same
// barriers, same phase), so the matrix pipe idled while BOTH issued their direct loads, waited at the barrier and waited for
// the first LDS fragments of a k tile (measured 35-40 % of the dense bf16 rate).  Here the eight waves form two groups that
// share the SIMDs pairwise (waves w and w + 4) and run HALF A SLOT APART on 32-deep stages:
//
//   slot k (between barriers b_k and b_k+1)
//     group 0 (waves 0-3, tile rows 0-127)   : request stage k + 2  ->  read fragments of stage k  ->  16 MFMAs
//     group 1 (waves 4-7, tile rows 128-255) : 16 MFMAs of stage k - 1 (first fragments read before b_k)  ->  request stage k + 2
//                                              ->  read the first fragments of stage k
//   so one group's loads / LDS latency / barrier wait sit beside the other group's MFMAs, each of which can keep the pipe
//   busy alone.  Four 32 KB LDS stages hold [k - 1, k, k + 1, k + 2]: stage j's buffer is re-requested after b_(j-2) (both
//   groups have left stage j - 4) and must have landed before b_j, i.e. two slots of flight time; every wave waits for its own
//   pieces with a counted vmcnt (the younger stage stays in flight) before it arrives at the barrier.  One barrier per slot.
//
//   gemm_nt_pp_kernel  C[M,N] = epi(A[M,K] B[N,K]^T)      both operands K-contiguous ([256 rows][64 bytes] stage images, chunk
//                      c of row r at chunk c ^ ((r >> 2) & 3): conflict-free ds_read_b128 fragments)
//   gemm_tn_pp_kernel  part[z] = A[K,M]^T B[K,N]           both operands row-contiguous ([32 tokens][512 bytes] slabs, chunk c of
//                      token row k at c ^ ((k & 3) << 2), fragments by ds_read_b64_tr_b16), + column sums of A (bias gradient)
//
// LDS fragment reads are inline asm with hand-counted lgkmcnt waits: hipcc (ROCm 7.2) otherwise drains every direct load in
// flight (s_waitcnt vmcnt(0)) in front of LDS reads it cannot tell apart from the buffers being filled.
#include <cstdlib>
#include "common.h"
#include "../../include/wm_kernels.h"
#include "gemm_big_epi.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const void glb_cvoid;

constexpr int SK = 32;                              // k depth of a stage
constexpr int OPND_B = 16 * 1024;                   // one operand's stage image
constexpr int STAGE_B = 2 * OPND_B;                 // A | B
constexpr int NSTAGE = 4;
constexpr int SMEM_B = NSTAGE * STAGE_B;            // 128 KB (the epilogue's 64 x 260 fp32 image aliases it)

#define WM_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0f70 | (n))        /* vmcnt(n), n <= 15 */
#define WM_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(lds_char*)p; }

template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_tr(uint32_t addr, int) {          // two transposing 8-byte reads, rows k and k + 4
  bf16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 4 * 512));
  bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return f;
}

#define WM_PP_MMA(S)                                                                                   \
  if (!(WM_PP_DBG & 2)) {                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[S][i], b[S][j], acc[i][j], 0, 0, 0);     \
    __builtin_amdgcn_s_setprio(0);                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  }

// ------------------------------------------------------------------------------------------------------------------
// DBG: ablation switches of scripts/gemm_ablate.py (1 no operand loads, 2 no MFMAs, 4 no epilogue) as a TEMPLATE parameter: the
// product instance is DBG = 0 and carries none of them; the other instances exist only in a -DWM_GEMM_ABLATE build
#define WM_PP_DBG DBG
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(const WmGemm g, const int tiles_n, const int tiles_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;
  int bx, by;
  {
    const int L = blockIdx.x, j = L & 7, sq = L >> 3;     // XCD-aware order: blocks that share an A row panel have equal ids mod 8
    bx = sq % tiles_n; by = (sq / tiles_n) * 8 + j;
    if (by >= tiles_m) return;
  }
  const int m0 = by * 256, n0 = bx * 256, S = g.K / SK;

  // direct loads: a wave-instruction fills 16 rows x 64 bytes; pass q, wave w -> rows [128 q + 16 w, + 16) of the stage image
  const int lrow = 16 * wave + (lane >> 2);
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);              // ((128 q + lrow) >> 2) & 3 == (lane >> 4) & 3
  const __bf16* pa[2];
  const __bf16* pb[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    pa[q] = static_cast<const __bf16*>(g.A) + (int64_t)min(m0 + 128 * q + lrow, g.M - 1) * g.lda + 8 * chunk;   // rows past M: never stored
    pb[q] = static_cast<const __bf16*>(g.B) + (int64_t)(n0 + 128 * q + lrow) * g.ldb + 8 * chunk;
  }
  auto request = [&](int stage) {
    if constexpr (DBG & 1) return;                               // (ablation: no operand loads)
    char* t = smem + (stage & 3) * STAGE_B + 16 * wave * 64;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      __builtin_amdgcn_global_load_lds((glb_cvoid*)pa[q], (lds_void*)(t + 128 * q * 64), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_cvoid*)pb[q], (lds_void*)(t + OPND_B + 128 * q * 64), 16, 0, 0);
      pa[q] += SK; pb[q] += SK;
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment addresses inside a stage: row (.. + r) * 64 + (((2 ks + h) ^ ((r >> 2) & 3)) << 4)
  const int hs = (h ^ ((r >> 2) & 3)) << 4;
  const uint32_t lds0 = lds_addr(smem);
  const uint32_t fa0 = lds0 + (wm * 128 + r) * 64 + hs, fa1 = lds0 + (wm * 128 + r) * 64 + (hs ^ 32);
  const uint32_t fb0 = lds0 + OPND_B + (wn * 64 + r) * 64 + hs, fb1 = lds0 + OPND_B + (wn * 64 + r) * 64 + (hs ^ 32);
  bf16x8 a[2][4], b[2][2];
#define WM_NT_FRAGS(S_, AA, BB)                                            \
  {                                                                        \
    a[S_][0] = lds_read128<0>(AA); a[S_][1] = lds_read128<2048>(AA);       \
    a[S_][2] = lds_read128<4096>(AA); a[S_][3] = lds_read128<6144>(AA);    \
    b[S_][0] = lds_read128<0>(BB); b[S_][1] = lds_read128<2048>(BB);       \
  }

  request(0);
  if (S > 1) request(1);
  if (S > 1) WM_WAIT_VM(4); else WM_WAIT_VM(0);                   // stage 0 has landed (this wave's pieces)

  if (wm == 0) {
    for (int k = 0; k < S; ++k) {
      __builtin_amdgcn_s_barrier();                              // b_k: stage k complete, stage k - 2 free
      if (k + 2 < S) request(k + 2);
      const uint32_t so = (k & 3) * STAGE_B;
      WM_NT_FRAGS(0, fa0 + so, fb0 + so)
      WM_NT_FRAGS(1, fa1 + so, fb1 + so)
      WM_WAIT_LGKM(6);
      WM_PP_MMA(0)
      WM_WAIT_LGKM(0);
      WM_PP_MMA(1)
      if (k + 1 < S) { if (k + 2 < S) WM_WAIT_VM(4); else WM_WAIT_VM(0); }    // stage k + 1 landed
    }
    __builtin_amdgcn_s_barrier();                                // b_S: the other group's last slot begins
  } else {
    __builtin_amdgcn_s_barrier();                                // b_0
    WM_NT_FRAGS(0, fa0, fb0)                                     // first fragments of stage 0
    if (S > 2) request(2);                                       // (slot 0 of this group holds nothing else)
    if (S > 1) { if (S > 2) WM_WAIT_VM(4); else WM_WAIT_VM(0); }
    for (int k = 1; k <= S; ++k) {
      __builtin_amdgcn_s_barrier();                              // b_k
      const uint32_t so = ((k - 1) & 3) * STAGE_B;
      WM_NT_FRAGS(1, fa1 + so, fb1 + so)
      WM_WAIT_LGKM(6);
      WM_PP_MMA(0)
      WM_WAIT_LGKM(0);
      WM_PP_MMA(1)
      if (k + 2 < S) request(k + 2);
      if (k < S) {
        const uint32_t sn = (k & 3) * STAGE_B;
        WM_NT_FRAGS(0, fa0 + sn, fb0 + sn)                       // stage k landed before b_k: read ahead for the next slot
      }
      if (k + 1 < S) { if (k + 2 < S) WM_WAIT_VM(4); else WM_WAIT_VM(0); }
    }
  }
#undef WM_NT_FRAGS
  __builtin_amdgcn_s_barrier();                                  // b_(S+1): nobody reads an operand stage any more
  if constexpr (DBG & 4) {                                       // (ablation: no epilogue; keep the accumulators alive)
    if (acc[0][0][0] == 12345.678f) static_cast<float*>(g.C)[0] = acc[3][1][15] + acc[1][0][3];
    return;
  }
  bigtile::nt_epilogue<EPI>(acc, g, m0, n0, smem, tid, wm, wn, r, h);
}

#undef WM_PP_DBG
#define WM_PP_DBG 0
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ float frag_sum(const bf16x8& f, float acc) {      // acc + the eight elements (v_dot2c_f32_bf16 with ones)
  const bf16x2_t one = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const bf16x2_t pr = {f[2 * u], f[2 * u + 1]};
    acc = __builtin_amdgcn_fdot2_f32_bf16(pr, one, acc, false);
  }
  return acc;
}
// byte offset of a lane's first transposing read inside a [32 tokens][512 bytes] slab for the 32 columns from col0 (k step 0)
__device__ __forceinline__ int slab_off(int col0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int row = 8 * (g >> 1) + q;                            // (row + 16 ks [+ 4]) & 3 == q
  const int c = (col0 >> 3) + 2 * (g & 1) + (p >> 1);
  return row * 512 + ((c ^ (q << 2)) << 4) + 8 * (p & 1);
}

__global__ __launch_bounds__(512, 2) void gemm_tn_pp_kernel(const WmGemm g, const int klen, const int tiles_n, const int tiles_m,
                                                           const int ksplit, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;
  int bx, by, bz;
  {
    const int L = blockIdx.x, j = L & 7, sq = L >> 3, tiles = tiles_n * tiles_m, t = sq % tiles;
    bz = (sq / tiles) * 8 + j; bx = t % tiles_n; by = t / tiles_n;    // the slices of one k range share an XCD's L2
    if (bz >= ksplit) return;
  }
  const int m0 = by * 256, n0 = bx * 256;
  const int kbeg = bz * klen, kend = min(g.K, kbeg + klen), S = (kend - kbeg) / SK;     // host: K % 32 == 0, klen % 32 == 0

  // direct loads: a wave-instruction fills 2 token rows x 512 bytes; pass q, wave w -> rows 16 q + 2 w, + 1
  const int lrow = 2 * wave + (lane >> 5);
  const int chunk = (lane & 31) ^ ((lrow & 3) << 2);             // (16 q + lrow) & 3 == lrow & 3
  const __bf16* pa[2];
  const __bf16* pb[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    pa[q] = static_cast<const __bf16*>(g.A) + (int64_t)(kbeg + 16 * q + lrow) * g.lda + m0 + 8 * chunk;
    pb[q] = static_cast<const __bf16*>(g.B) + (int64_t)(kbeg + 16 * q + lrow) * g.ldb + n0 + 8 * chunk;
    if (g.b_seg_cols > 0) {                                      // grid addressing (WmGemm::b_seg_cols): this tile's column block lies rows further down
      const int seg = n0 / g.b_seg_cols;
      pb[q] += (int64_t)g.b_seg_rows[seg] * g.ldb - (int64_t)seg * g.b_seg_cols;
    }
  }
  const int64_t sa = (int64_t)SK * g.lda, sb = (int64_t)SK * g.ldb;
  auto request = [&](int stage) {
    char* t = smem + (stage & 3) * STAGE_B + 2 * wave * 512;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      __builtin_amdgcn_global_load_lds((glb_cvoid*)pa[q], (lds_void*)(t + 16 * q * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_cvoid*)pb[q], (lds_void*)(t + OPND_B + 16 * q * 512), 16, 0, 0);
      pa[q] += sa; pb[q] += sb;
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // bias gradient = column sums of A over the token range: fragment i == wn of wave (wm, wn) covers every 32-column block of the
  // slab exactly once over the eight waves; only the bx == 0 workgroups of a k range add theirs
  float bsum = 0.f;
  const bool do_bias = g.b_ones && bx == 0;

  const uint32_t lds0 = lds_addr(smem);
  uint32_t fa[4], fb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = lds0 + slab_off(wm * 128 + i * 32, lane);
#pragma unroll
  for (int j = 0; j < 2; ++j) fb[j] = lds0 + OPND_B + slab_off(wn * 64 + j * 32, lane);
  bf16x8 a[2][4], b[2][2];
#define WM_TN_FRAGS(S_, KS, SO)                                                                  \
  {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) a[S_][i] = lds_read_tr<KS * 16 * 512>(fa[i] + SO, 0);  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) b[S_][j] = lds_read_tr<KS * 16 * 512>(fb[j] + SO, 0);  \
  }
#define WM_TN_BIAS(S_)                                                                           \
  if (do_bias) {                                                                                 \
    const bf16x8 fs = wn == 0 ? a[S_][0] : wn == 1 ? a[S_][1] : wn == 2 ? a[S_][2] : a[S_][3];   \
    bsum = frag_sum(fs, bsum);                                                                   \
  }

  if (S > 0) request(0);
  if (S > 1) request(1);
  if (S > 1) WM_WAIT_VM(4); else WM_WAIT_VM(0);

  if (wm == 0) {
    for (int k = 0; k < S; ++k) {
      __builtin_amdgcn_s_barrier();
      if (k + 2 < S) request(k + 2);
      const uint32_t so = (k & 3) * STAGE_B;
      WM_TN_FRAGS(0, 0, so)
      WM_TN_FRAGS(1, 1, so)
      WM_WAIT_LGKM(12);                                          // 12 transposing reads per k step
      WM_PP_MMA(0)
      WM_TN_BIAS(0)
      WM_WAIT_LGKM(0);
      WM_PP_MMA(1)
      WM_TN_BIAS(1)
      if (k + 1 < S) { if (k + 2 < S) WM_WAIT_VM(4); else WM_WAIT_VM(0); }
    }
    __builtin_amdgcn_s_barrier();
  } else {
    __builtin_amdgcn_s_barrier();
    if (S > 0) WM_TN_FRAGS(0, 0, 0)
    if (S > 2) request(2);
    if (S > 1) { if (S > 2) WM_WAIT_VM(4); else WM_WAIT_VM(0); }
    for (int k = 1; k <= S; ++k) {
      __builtin_amdgcn_s_barrier();
      const uint32_t so = ((k - 1) & 3) * STAGE_B;
      WM_TN_FRAGS(1, 1, so)
      WM_WAIT_LGKM(12);
      WM_PP_MMA(0)
      WM_TN_BIAS(0)
      WM_WAIT_LGKM(0);
      WM_PP_MMA(1)
      WM_TN_BIAS(1)
      if (k + 2 < S) request(k + 2);
      if (k < S) {
        const uint32_t sn = (k & 3) * STAGE_B;
        WM_TN_FRAGS(0, 0, sn)
      }
      if (k + 1 < S) { if (k + 2 < S) WM_WAIT_VM(4); else WM_WAIT_VM(0); }
    }
  }
#undef WM_TN_FRAGS
#undef WM_TN_BIAS
  __builtin_amdgcn_s_barrier();
  bigtile::tn_store(acc, g, part, bz, m0, n0, wm, wn, r, h);
  if (do_bias) {
    bsum += __shfl_xor(bsum, 32);                                 // the two k halves of the fragment
    if (h == 0) atomicAdd(g.C2 + m0 + wm * 128 + wn * 32 + r, bsum);
  }
}

template <int EPI, int DBG = 0>
int launch_nt_pp(const WmGemm& g, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_pp_kernel<EPI, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
    attr = true;
  }
  const int tiles_n = g.N / 256, tiles_m = (g.M + 255) / 256;
  wm_gemm_family_tls = WM_FAM_NT_PP;
  hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, DBG>), dim3(8 * ((tiles_m + 7) / 8) * tiles_n), dim3(512), SMEM_B, st, g, tiles_n, tiles_m);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

}  // namespace

// launchers for gemm_big.hip's dispatch (which has checked shapes and alignment)
int wm_gemm_pp_nt_launch(const WmGemm& g, int epi, hipStream_t st) {
#ifdef WM_GEMM_ABLATE                                             // bench-only build (make ABLATE=1): WM_GEMM_DBG picks an ablated instance of EPI 0
  if (const char* d = getenv("WM_GEMM_DBG"); d && epi == 0) {
    switch (atoi(d)) {
      case 1: return launch_nt_pp<0, 1>(g, st);
      case 2: return launch_nt_pp<0, 2>(g, st);
      case 3: return launch_nt_pp<0, 3>(g, st);
      case 4: return launch_nt_pp<0, 4>(g, st);
      case 5: return launch_nt_pp<0, 5>(g, st);
      case 6: return launch_nt_pp<0, 6>(g, st);
      default: break;
    }
  }
#endif
  return epi == 0 ? launch_nt_pp<0>(g, st) : epi == 1 ? launch_nt_pp<1>(g, st) : epi == 2 ? launch_nt_pp<2>(g, st) : launch_nt_pp<4>(g, st);
}
int wm_gemm_pp_tn_launch(const WmGemm& g, int klen, int tiles_n, int tiles_m, int ksplit, float* part, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
    attr = true;
  }
  wm_gemm_family_tls = WM_FAM_TN_PP;
  hipLaunchKernelGGL(gemm_tn_pp_kernel, dim3(8 * ((ksplit + 7) / 8) * tiles_n * tiles_m), dim3(512), SMEM_B, st, g, klen, tiles_n, tiles_m,
                     ksplit, part);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
