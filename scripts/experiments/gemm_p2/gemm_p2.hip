// 128 x 256 tile bf16 NT GEMM, TWO workgroups per CU.  The 256 x 256 tile kernels (gemm_big.hip, gemm_pp.hip) hold a CU alone
// (128 KB of LDS, 512 threads), so nothing runs beside a workgroup's prologue and its epilogue: the ablation
// (profiles/r03_gemm_bigtile_ablation.txt) prices the exposed epilogue at 40 of 163 us on a K = 512 product, and the K = 256 products
// of the step ran at 2.2 x their roofline.  Here a workgroup is four waves (one per SIMD), each with the same 128 x 64 register tile
// as there (acc[4][2], 128 registers), three 24 KB LDS stages of 32-deep operand slabs (A 128 rows | B 256 rows, 64 bytes per row,
// chunk c of row r at chunk c ^ ((r >> 2) & 3)): 72 KB and <= 256 registers, so two workgroups share a CU and one's loads, barrier
// waits and epilogue sit beside the other's MFMAs — the interleave the ping-pong schedule builds inside one workgroup, between two.
// Price: L2 -> LDS operand traffic per flop is 1.5 x the 256 x 256 tile's (384 operand rows per 128 x 256 outputs instead of 512 per
// 256 x 256).
//
//   slot k (between barriers b_k and b_k+1):  request stage k + 2 (buffer of stage k - 1, free since b_k)  ->  read the fragments
//   of stage k  ->  16 MFMAs  ->  counted vmcnt: this wave's pieces of stage k + 1 have landed (stage k + 2 stays in flight)
#include <cstdlib>
#include "common.h"
#include "../../include/wm_kernels.h"
#include "gemm_big_epi.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const void glb_cvoid;

constexpr int SK = 32;                              // k depth of a stage
constexpr int A_B = 128 * 64;                       // A slab: 8 KB
constexpr int B_B = 256 * 64;                       // B slab: 16 KB
constexpr int STAGE_B = A_B + B_B;
constexpr int NSTAGE = 3;
constexpr int SMEM_B = NSTAGE * STAGE_B;            // 72 KB (the epilogue's 32 x 260 fp32 image aliases it)

#define WM_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0f70 | (n))        /* vmcnt(n), n <= 15 */
#define WM_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(lds_char*)p; }

template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

#define WM_P2_MMA(S)                                                                                   \
  {                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                    \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[S][i], b[S][j], acc[i][j], 0, 0, 0);     \
    __builtin_amdgcn_s_setprio(0);                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_p2_kernel(const WmGemm g, const int tiles_n, const int tiles_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave, r = lane & 31, h = lane >> 5;
  int bx, by;
  {
    const int L = blockIdx.x, j = L & 7, sq = L >> 3;     // XCD-aware order: blocks that share an A row panel have equal ids mod 8
    bx = sq % tiles_n; by = (sq / tiles_n) * 8 + j;
    if (by >= tiles_m) return;
  }
  const int m0 = by * 128, n0 = bx * 256, S = g.K / SK;

  // direct loads: a wave-instruction fills 16 rows x 64 bytes; pass q, wave w -> rows [64 q + 16 w, + 16) of a slab
  const int lrow = 16 * wave + (lane >> 2);
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);              // ((64 q + lrow) >> 2) & 3 == (lane >> 4) & 3
  const __bf16* pa[2];
  const __bf16* pb[4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
    pa[q] = static_cast<const __bf16*>(g.A) + (int64_t)min(m0 + 64 * q + lrow, g.M - 1) * g.lda + 8 * chunk;   // rows past M: never stored
#pragma unroll
  for (int q = 0; q < 4; ++q) pb[q] = static_cast<const __bf16*>(g.B) + (int64_t)(n0 + 64 * q + lrow) * g.ldb + 8 * chunk;
  auto request = [&](uint32_t so) {
    char* t = smem + so + 16 * wave * 64;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      __builtin_amdgcn_global_load_lds((glb_cvoid*)pa[q], (lds_void*)(t + 64 * q * 64), 16, 0, 0);
      pa[q] += SK;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_global_load_lds((glb_cvoid*)pb[q], (lds_void*)(t + A_B + 64 * q * 64), 16, 0, 0);
      pb[q] += SK;
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
  const uint32_t fa0 = lds0 + r * 64 + hs, fa1 = lds0 + r * 64 + (hs ^ 32);
  const uint32_t fb0 = lds0 + A_B + (wn * 64 + r) * 64 + hs, fb1 = lds0 + A_B + (wn * 64 + r) * 64 + (hs ^ 32);
  bf16x8 a[2][4], b[2][2];
#define WM_P2_FRAGS(S_, AA, BB)                                            \
  {                                                                        \
    a[S_][0] = lds_read128<0>(AA); a[S_][1] = lds_read128<2048>(AA);       \
    a[S_][2] = lds_read128<4096>(AA); a[S_][3] = lds_read128<6144>(AA);    \
    b[S_][0] = lds_read128<0>(BB); b[S_][1] = lds_read128<2048>(BB);       \
  }

  request(0);
  if (S > 1) request(STAGE_B);
  if (S > 1) WM_WAIT_VM(6); else WM_WAIT_VM(0);                   // stage 0 has landed (this wave's six pieces)
  uint32_t so = 0, sreq = 2 * STAGE_B;                            // buffer of stage k / of stage k + 2
  for (int k = 0; k < S; ++k) {
    __builtin_amdgcn_s_barrier();                                // b_k: stage k complete, the buffer of stage k - 1 free
    if (k + 2 < S) request(sreq);
    WM_P2_FRAGS(0, fa0 + so, fb0 + so)
    WM_P2_FRAGS(1, fa1 + so, fb1 + so)
    WM_WAIT_LGKM(6);
    WM_P2_MMA(0)
    WM_WAIT_LGKM(0);
    WM_P2_MMA(1)
    if (k + 1 < S) { if (k + 2 < S) WM_WAIT_VM(6); else WM_WAIT_VM(0); }    // stage k + 1 landed
    so = so == 2 * STAGE_B ? 0 : so + STAGE_B;
    sreq = sreq == 2 * STAGE_B ? 0 : sreq + STAGE_B;
  }
#undef WM_P2_FRAGS
  __builtin_amdgcn_s_barrier();                                  // nobody reads an operand stage any more
  bigtile::nt_epilogue128<EPI>(acc, g, m0, n0, smem, tid, wn, r, h);
}

template <int EPI>
int launch_nt_p2(const WmGemm& g, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_p2_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
    attr = true;
  }
  const int tiles_n = g.N / 256, tiles_m = (g.M + 127) / 128;
  wm_gemm_family_tls = WM_FAM_NT_PP;
  hipLaunchKernelGGL((gemm_nt_p2_kernel<EPI>), dim3(8 * ((tiles_m + 7) / 8) * tiles_n), dim3(256), SMEM_B, st, g, tiles_n, tiles_m);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

}  // namespace

// launcher for gemm_big.hip's dispatch (which has checked shapes and alignment)
int wm_gemm_p2_nt_launch(const WmGemm& g, int epi, hipStream_t st) {
  return epi == 0 ? launch_nt_p2<0>(g, st) : epi == 1 ? launch_nt_p2<1>(g, st) : epi == 2 ? launch_nt_p2<2>(g, st) : launch_nt_p2<4>(g, st);
}
