// 256 x 256 x 64 bf16 GEMM tiles for the compute-side shapes of the train step (C >= 256: K, N in 256 .. 2048, tens to
// hundreds of thousands of rows), where the 128 x 128 kernels of gemm_fast.hip are bound by the L2 -> LDS operand traffic
// (64 FLOP per operand byte per tile; this tile: 128) and by their per-thread staging instructions.
//
//   gemm_nt_big_kernel   C[M,N] = epi(A[M,K] * B[N,K]^T)   both operands K-contiguous bf16 (forward: B = weights [N,K];
//                        dX: B = the TRANSPOSED weight copy [K_in, N_out], so one kernel serves both directions)
//
// 512 threads = 8 waves (2 x 4), a wave owns 128 x 64 of the tile (4 x 2 accumulators of v_mfma_f32_32x32x16_bf16 = 128
// registers).  Operand tiles go global -> LDS directly (global_load_lds_dwordx4, no staging registers, no LDS store
// instructions): two 64 KB buffers (A | B tiles of one k step), tile t + 2 is requested as soon as every wave has finished
// reading tile t, a counted vmcnt leaves the younger tile's requests in flight across the barriers.  The LDS image of a tile
// is [256 rows][128 bytes]; the direct load writes lane-linear, so the bank swizzle (16-byte chunk c of row r lives at chunk
// c ^ ((r >> 1) & 7)) is applied to the per-lane SOURCE address and again on the fragment reads: the 16 lanes of a
// ds_read_b128 group then hit 16 different 16-byte slots of the 256-byte bank row.
#include <cstdlib>
#include "common.h"
#include "../../include/wm_kernels.h"
#include "gemm_big_epi.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_cvoid;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int ROWB = BK * 2;                       // bytes per LDS row
constexpr int TILE_B = BM * ROWB;                  // 32 KB per operand tile
constexpr int BUF_B = 2 * TILE_B;                  // A tile | B tile
constexpr int SMEM_B = 2 * BUF_B;                  // 128 KB (the epilogue's 64 x 260 fp32 image aliases it)

#define WM_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0f70 | (n))      /* vmcnt(n), n <= 15; lgkmcnt / expcnt untouched */

// one operand tile = 4 direct loads per thread: pass q covers rows [64 q, 64 q + 64), wave w rows 8 w .. 8 w + 7 of them
__device__ __forceinline__ void request_tile(const __bf16* const (&src)[4], char* tile, int wave_u) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    __builtin_amdgcn_global_load_lds((glb_cvoid*)src[q], (lds_void*)(tile + (64 * q + 8 * wave_u) * ROWB), 16, 0, 0);
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_big_kernel(const WmGemm g, const int tiles_n, const int tiles_m) {
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
  const int m0 = by * BM, n0 = bx * BN, K = g.K, nt = K / BK;

  // ---- source pointers of the direct loads (chunk swizzle on the source side) ----
  const int lr = 8 * wave + (lane >> 3);                        // row inside a 64-row pass
  const int chunk = (lane & 7) ^ ((lr >> 1) & 7);               // (64 q >> 1) & 7 == 0: the same for the four passes
  const __bf16* pa[4];
  const __bf16* pb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ra = min(m0 + 64 * q + lr, g.M - 1);              // rows past M re-read the last row (never stored)
    pa[q] = static_cast<const __bf16*>(g.A) + (int64_t)ra * g.lda + 8 * chunk;
    pb[q] = static_cast<const __bf16*>(g.B) + (int64_t)(n0 + 64 * q + lr) * g.ldb + 8 * chunk;
  }
  auto request = [&](int buf) {
    char* t = smem + buf * BUF_B;
    request_tile(pa, t, wave);
    request_tile(pb, t + TILE_B, wave);
#pragma unroll
    for (int q = 0; q < 4; ++q) { pa[q] += BK; pb[q] += BK; }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  request(0);
  if (nt > 1) request(1);

  // fragment addresses: row (.. + r), chunk (2 ks + h) ^ ((r >> 1) & 7) = (2 ks) ^ hs
  const int hs = (h ^ ((r >> 1) & 7)) << 4;
  const int a_off = (wm * 128 + r) * ROWB, b_off = TILE_B + (wn * 64 + r) * ROWB;

  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) WM_WAIT_VMCNT(8); else WM_WAIT_VMCNT(0);     // this wave's pieces of tile t have landed
    __builtin_amdgcn_s_barrier();                               // ... and everybody else's
    const char* base = smem + (t & 1) * BUF_B;
    bf16x8 a[2][4], b[2][2];
    auto frags = [&](int ks, bf16x8 (&fa)[4], bf16x8 (&fb)[2]) {
      const int x = (32 * ks) ^ hs;
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(base + a_off + i * 32 * ROWB + x);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(base + b_off + j * 32 * ROWB + x);
    };
    frags(0, a[0], b[0]);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      if (ks + 1 < BK / 16) frags(ks + 1, a[(ks + 1) & 1], b[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][i], b[ks & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();                               // every wave is done reading this buffer
    if (t + 2 < nt) request(t & 1);
  }

  bigtile::nt_epilogue<EPI>(acc, g, m0, n0, smem, tid, wm, wn, r, h);
}

// ------------------------------------------------------------------------------------------------------------------
//   gemm_tn_big_kernel   part[z][M,N] = A[K range z, M]^T * B[K range z, N]   (weight gradients: A = dY [tokens, N_out],
//                        B = X [tokens, K_in], both row-contiguous bf16; the reduction runs over the tokens)
//                        + column sums of A over the range (bias gradient) by atomics into C2
// Same pipeline; an operand tile is the [64 tokens][256 columns] slab as it lies in memory (512-byte rows, two rows per
// wave-instruction of the direct load), fragments are read with the transposing ds_read_b64_tr_b16.  Rows are 0 mod 256
// bytes apart, so the four token rows a 32-lane half reads would share 16 banks: chunk c of row k is kept at chunk
// c ^ ((k & 3) << 2), which puts them on four different quarters of the bank row.
constexpr int TROWB = 512, TTILE_B = 64 * TROWB;             // 32 KB per operand tile, as above

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

__device__ __forceinline__ void request_slab(const __bf16* const (&src)[4], char* tile, int wave_u) {
#pragma unroll
  for (int q = 0; q < 4; ++q)                                  // pass q: token rows [16 q, 16 q + 16), wave w rows 2 w, 2 w + 1
    __builtin_amdgcn_global_load_lds((glb_cvoid*)src[q], (lds_void*)(tile + (16 * q + 2 * wave_u) * TROWB), 16, 0, 0);
}

// MFMA operand (lane: column col0 + (lane & 31), k = 16 ks + 8 (lane >> 5) + j) out of the swizzled [k][256] slab, by two
// ds_read_b64_tr_b16.  The reads are issued as inline asm: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the
// transposing-read BUILTIN whenever a direct global -> LDS load is in flight (it cannot tell the buffers apart), which would
// drain the prefetched tile at every k step; asm reads are invisible to that pass, so their completion is waited for by hand
// (counted lgkmcnt, LDS returns in order) with a scheduling barrier behind the wait.
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(lds_char*)p; }
// byte offset of this lane's first read inside a slab, for the 32 columns starting at col0 (k step 0)
__device__ __forceinline__ int slab_frag_off(int col0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int row = 8 * (g >> 1) + q;                            // (row + 16 ks [+ 4]) & 3 == q
  const int c = (col0 >> 3) + 2 * (g & 1) + (p >> 1);          // 16-byte chunk of the row
  return row * TROWB + ((c ^ (q << 2)) << 4) + 8 * (p & 1);
}
template <int KS>
__device__ __forceinline__ bf16x8 slab_frag(uint32_t addr) {
  bf16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(KS * 16 * TROWB));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(KS * 16 * TROWB + 4 * TROWB));
  bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return f;
}
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

__global__ __launch_bounds__(512, 2) void gemm_tn_big_kernel(const WmGemm g, const int klen, const int tiles_n, const int tiles_m,
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
  const int kbeg = bz * klen, kend = min(g.K, kbeg + klen), nt = (kend - kbeg) / 64;    // host: K % 64 == 0, klen % 64 == 0

  const int lrow = 2 * wave + (lane >> 5);                      // token row inside a 16-row pass; (16 q + lrow) & 3 == lrow & 3
  const int chunk = (lane & 31) ^ ((lrow & 3) << 2);
  const __bf16* pa[4];
  const __bf16* pb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    pa[q] = static_cast<const __bf16*>(g.A) + (int64_t)(kbeg + 16 * q + lrow) * g.lda + m0 + 8 * chunk;
    pb[q] = static_cast<const __bf16*>(g.B) + (int64_t)(kbeg + 16 * q + lrow) * g.ldb + n0 + 8 * chunk;
  }
  const int64_t sa = 64 * g.lda, sb = 64 * g.ldb;
  auto request = [&](int buf) {
    char* t = smem + buf * BUF_B;
    request_slab(pa, t, wave);
    request_slab(pb, t + TTILE_B, wave);
#pragma unroll
    for (int q = 0; q < 4; ++q) { pa[q] += sa; pb[q] += sb; }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // bias gradient = column sums of A over the token range: the A fragments are already in registers, and the eight waves hold
  // every 32-column block of the slab exactly once as fragment i == wn of wave (wm, wn): four v_dot2c per k step and wave.
  // They do not depend on the N tile, so only the bx == 0 workgroups of a k range add theirs.
  float bsum = 0.f;
  const bool do_bias = g.b_ones && bx == 0;
  int aoff[4], boff[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = slab_frag_off(wm * 128 + i * 32, lane);
#pragma unroll
  for (int j = 0; j < 2; ++j) boff[j] = TTILE_B + slab_frag_off(wn * 64 + j * 32, lane);

  if (nt > 0) request(0);
  if (nt > 1) request(1);

  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) WM_WAIT_VMCNT(8); else WM_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    const uint32_t base = lds_addr(smem) + (t & 1) * BUF_B;
    bf16x8 a[2][4], b[2][2];
#define WM_TN_FRAGS(KS, S)                                                                     \
    {                                                                                          \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) a[S][i] = slab_frag<KS>(base + aoff[i]);   \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) b[S][j] = slab_frag<KS>(base + boff[j]);   \
    }
#define WM_TN_MMA(S)                                                                           \
    {                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      __builtin_amdgcn_s_setprio(1);                                                           \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                          \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[S][i], b[S][j], acc[i][j], 0, 0, 0); \
      __builtin_amdgcn_s_setprio(0);                                                           \
      if (do_bias) {                                                                           \
        const bf16x8 fs = wn == 0 ? a[S][0] : wn == 1 ? a[S][1] : wn == 2 ? a[S][2] : a[S][3]; \
        bsum = frag_sum(fs, bsum);                                                             \
      }                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                       \
    }
    // 12 reads per k step; "lgkmcnt(12)" = everything but the step just requested has returned
    WM_TN_FRAGS(0, 0)
    WM_TN_FRAGS(1, 1)
    asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
    WM_TN_MMA(0)
    WM_TN_FRAGS(2, 0)
    asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
    WM_TN_MMA(1)
    WM_TN_FRAGS(3, 1)
    asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
    WM_TN_MMA(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WM_TN_MMA(1)
#undef WM_TN_FRAGS
#undef WM_TN_MMA
    __builtin_amdgcn_s_barrier();
    if (t + 2 < nt) request(t & 1);
  }

  bigtile::tn_store(acc, g, part, bz, m0, n0, wm, wn, r, h);
  if (do_bias) {
    bsum += __shfl_xor(bsum, 32);                                 // the two k halves of the fragment
    if (h == 0) atomicAdd(g.C2 + m0 + wm * 128 + wn * 32 + r, bsum);
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int EPI>
int launch_nt_big(const WmGemm& g, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_big_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
    attr = true;
  }
  const int tiles_n = g.N / BN, tiles_m = (g.M + BM - 1) / BM;
  wm_gemm_family_tls = WM_FAM_NT_BIG;
  hipLaunchKernelGGL((gemm_nt_big_kernel<EPI>), dim3(8 * ((tiles_m + 7) / 8) * tiles_n), dim3(512), SMEM_B, st, g, tiles_n, tiles_m);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

}  // namespace

// gemm_pp.hip: the same tiles on the ping-pong schedule (the default); WM_GEMM_BIG_V=1 selects the kernels of this file
int wm_gemm_pp_nt_launch(const WmGemm& g, int epi, hipStream_t st);
int wm_gemm_pp_tn_launch(const WmGemm& g, int klen, int tiles_n, int tiles_m, int ksplit, float* part, hipStream_t st);
static bool use_pp() {
  const char* v = getenv("WM_GEMM_BIG_V");
  return !(v && v[0] == '1');
}

// Returns -1 when the launch does not fit (the caller goes on to the 128 x 128 kernels).
int wm_gemm_big_bf16(const WmGemm& g, hipStream_t st) {
  const char* off = getenv("WM_GEMM_BIG_OFF");                    // A/B switch for the microbenchmarks (read per call: host side, ~50 ns)
  if (off && off[0] == '1') return -1;
  if (g.mode != 1 || !g.a_kc || !g.b_kc || !g.a_bf16 || !g.b_bf16 || g.a_op || g.b_op || g.atomic_out || g.b_ones || g.kscale) return -1;
  if (g.a_grid_n || g.a_jump_k || g.b_seg_cols) return -1;       // grid addressing: the 128 x 128-family kernels
  if ((g.N % BN) || (g.K % BK) || g.K < 2 * BK || (g.M % 32) || (g.lda % 8) || (g.ldb % 8)) return -1;
  const int64_t tiles = (int64_t)((g.M + BM - 1) / BM) * (g.N / BN);
  if (tiles < 256) return -1;                                     // fewer tiles than CUs: the small tiles fill the chip better
  if (g.rowscale && (g.rows_per_scale % 32)) return -1;
  if (!al16(g.A) || !al16(g.B) || !al16(g.C) || (g.ldc % 8) || (g.bias && !al16(g.bias))) return -1;
  int epi = -1;
  if (g.ln_out) {                                                 // fused LayerNorm of whole output rows: one column tile
    if (g.N != BN || (g.M % BM) || g.c_bf16 || !g.res || g.gaux || !al16(g.res) || (g.ldr % 4) || !al16(g.ln_gamma) ||
        !al16(g.ln_beta) || !al16(g.ln_out) || (g.ln_mod && (!al16(g.ln_mod) || g.ln_hw <= 0)))
      return -1;
    epi = 4;
  } else if (g.c_bf16 && !g.gaux && !g.res) epi = 0;
  else if (!g.c_bf16 && g.res && !g.gaux && al16(g.res) && !(g.ldr % 4)) epi = 1;
  else if (g.c_bf16 && g.gaux && g.gaux_raw && g.gaux_bf16 && !g.res && !g.bias && al16(g.gaux) && !(g.ldg % 8)) epi = 2;
  if (epi < 0) return -1;
  if (use_pp()) return wm_gemm_pp_nt_launch(g, epi, st);
  return epi == 0 ? launch_nt_big<0>(g, st) : epi == 1 ? launch_nt_big<1>(g, st) : epi == 2 ? launch_nt_big<2>(g, st) : launch_nt_big<4>(g, st);
}

// Weight gradient dW[M,N] (+)= A[K,M]^T B[K,N] on 256 x 256 tiles, split over the token range so that about one workgroup per
// CU runs; the slices go to the scratch buffer and are summed by the caller's deferred reduction (or by splitk_reduce_big).
__global__ __launch_bounds__(256) void splitk_reduce_big_kernel(const float* __restrict__ part, float* __restrict__ C, int64_t ldc,
                                                                int M, int N, int ksplit) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x, mn = (int64_t)M * N;
  if (i4 * 4 >= mn) return;
  const float* p = part + i4 * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < ksplit; ++z, p += mn) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
  }
  const int m = (int)(i4 * 4 / N), n = (int)(i4 * 4 - (int64_t)m * N);
  float* c = C + m * ldc + n;
  c[0] += s.x; c[1] += s.y; c[2] += s.z; c[3] += s.w;
}

int wm_gemm_big_tn_bf16(const WmGemm& g, hipStream_t st) {
  const char* off = getenv("WM_GEMM_BIG_OFF");
  if (off && (off[0] == '1' || off[0] == '2')) return -1;          // 2: the weight-gradient kernel alone
  if (g.mode != 1 || g.a_kc || g.b_kc || !g.a_bf16 || !g.b_bf16 || g.a_op || g.b_op || !g.atomic_out || g.c_bf16 || g.kscale) return -1;
  if (g.a_grid_n || g.a_jump_k) return -1;
  if (g.b_seg_cols && ((g.b_seg_cols % 256) || g.N > 4 * g.b_seg_cols || !use_pp())) return -1;   // (a 256-wide tile inside one column block)
  if ((g.M % 256) || (g.N % 256) || (g.K % 64) || (g.lda % 8) || (g.ldb % 8) || !al16(g.A) || !al16(g.B)) return -1;
  const int tiles_m = g.M / 256, tiles_n = g.N / 256, tiles = tiles_m * tiles_n;
  // one or three output tiles (proj / qkv at C = 256) would mean 86 - 256 token ranges of a few k steps each and as many
  // 256 KB slices to sum: measured 91 against 37 us (proj) and 93 against 63 us (qkv) at 65 k tokens; from four tiles on the
  // big tile wins (fc1 / fc2 at C = 256: 183 against 255 us at 262 k tokens; qkv at C = 512: 207 against 306 us)
  if (tiles < 4 || g.K < 64 * 64 || !g.scratch) return -1;
  int ksplit = g.ksplit > 0 ? g.ksplit : (256 + tiles - 1) / tiles;
  const int64_t mn = (int64_t)g.M * g.N;
  if ((int64_t)ksplit * mn > g.scratch_floats) ksplit = (int)(g.scratch_floats / mn);
  if (ksplit > g.K / 512) ksplit = g.K / 512;                      // as in gemm_fast.hip launch_tn: no slice under 512 tokens
  if (ksplit < 1) return -1;
  int klen = (g.K + ksplit - 1) / ksplit;
  klen = (klen + 63) / 64 * 64;
  ksplit = (g.K + klen - 1) / klen;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
    attr = true;
  }
  float* part = ksplit > 1 ? g.scratch : nullptr;
  if (use_pp()) {
    const int rc = wm_gemm_pp_tn_launch(g, klen, tiles_n, tiles_m, ksplit, part, st);
    if (rc) return rc;
  } else {
    wm_gemm_family_tls = WM_FAM_TN_BIG;
    hipLaunchKernelGGL(gemm_tn_big_kernel, dim3(8 * ((ksplit + 7) / 8) * tiles), dim3(512), SMEM_B, st, g, klen, tiles_n, tiles_m, ksplit, part);
    WM_CHECK_LAUNCH();
  }
  if (g.defer) {
    WmReduceJob jb{};
    if (part) {
      jb.src = part; jb.pstride = mn; jb.src_ld = g.N; jb.count = ksplit; jb.rows = g.M; jb.cols = g.N;
      jb.dst = static_cast<float*>(g.C); jb.dst_rs = g.ldc; jb.dst_cs = 1;
    }
    *g.defer = jb;
    return WM_OK;
  }
  if (part) {
    hipLaunchKernelGGL(splitk_reduce_big_kernel, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, st, part, static_cast<float*>(g.C),
                       g.ldc, g.M, g.N, ksplit);
    WM_CHECK_LAUNCH();
  }
  return WM_OK;
}
