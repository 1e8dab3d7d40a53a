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
#include <utility>
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
// Epilogue of the NT kernel with every LDS access as inline asm (a C++ LDS access makes hipcc drain the direct loads in flight —
// here the NEXT tile's first two stages).  Image: [64 rows][256 fp32] = exactly stage buffers 2 | 3, 32-column halves of rows
// 4-7 of every 8 swapped (row rho, column c at rho * 256 + (c ^ ((rho & 4) << 3))): an accumulator register's 64 lanes (rows R /
// R + 4, 32 columns each) write 64 different banks, a staging row is read back as one 1 KB line.
//   EPI as in gemm_big_epi.h.  Four passes of 64 rows (tile row block i of every wave); two barriers per pass.
constexpr uint32_t IMG_B = 2 * STAGE_B;

template <int OFF>
__device__ __forceinline__ void lds_write32(uint32_t addr, float v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ float4 lds_read128f(uint32_t addr) {
  float4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int... E>
__device__ __forceinline__ void img_write(uint32_t addr, const f32x16& a, std::integer_sequence<int, E...>) {
  (lds_write32<((E & 3) + 8 * (E >> 2)) * 1024>(addr, a[E]), ...);
}
template <int... Q>
__device__ __forceinline__ void img_read(uint32_t addr, float4 (&v)[8], std::integer_sequence<int, Q...>) {
  ((v[Q] = lds_read128f<Q * 8192>(addr)), ...);
}

// The epilogue's vector loads (bias, residual / gate rows) are inline asm with hand-counted vmcnt waits as well: with loads, stores
// and LDS-DMA pending together hipcc treats the counter as unordered and waits vmcnt(0) for any load result — i.e. for the previous
// pass's stores (a store-acknowledge stall per pass) and for the stages requested ahead.  The hardware counts them in issue order, so
// the rows of pass i + 1 are requested BEFORE pass i's stores and waited for with those stores still in flight.  DropPath scales:
// scalar loads.  FULL: M % 256 == 0, no row of a tile is outside the matrix and every store is issued (the counts below rely on
// it); !FULL: loads from a clamped row, conditional stores, waits that assume no store was issued.  EPI 4 (LayerNorm, FULL only)
// has a lane-dependent store count and waits vmcnt(0).
__device__ __forceinline__ float4 gload128(const void* p) {
  float4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p));
  return v;
}
__device__ __forceinline__ uint2 gload64(const void* p) {
  uint2 v;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p));
  return v;
}
__device__ __forceinline__ float sload(const float* p) {         // uniform address; as asm: the compiler would use a VECTOR load here
  float v;                                                         // (stores may alias) and then wait vmcnt(0) for it
  asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(p));
  return v;
}
#define WM_WAIT_VMN(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int EPI, bool FULL>
__device__ __forceinline__ void nt_epilogue_pp(f32x16 (&acc)[4][2], const WmGemm& g, int m0, int n0, uint32_t lds0, int tid, int wm, int wn,
                                               int r, int h, const bool plain) {
  constexpr bool ROWS = EPI == 1 || EPI == 2 || EPI == 4;         // a second operand row per output row
  constexpr bool AHEAD = EPI != 4;
  const int c4 = tid & 63, rg = tid >> 6, n = n0 + 4 * c4;
  const uint32_t wbase = lds0 + IMG_B + (wm * 32 + 4 * h) * 1024 + (wn * 64 + r) * 4;
  const uint32_t w0 = wbase + (h << 7), w1 = wbase + ((h ^ 1) << 7);                 // accumulator column j at (j ^ h) * 32
  const uint32_t rbase = lds0 + IMG_B + rg * 1024 + ((16 * c4) ^ ((rg & 4) << 5));
  const int last = g.M - 1;
  auto grow = [&](int i, int q) { return m0 + (q >> 2) * 128 + 32 * i + ((rg + 8 * q) & 31); };   // image row rg + 8 q of pass i
  auto lrow = [&](int i, int q) { return (int64_t)(FULL ? grow(i, q) : min(grow(i, q), last)); };
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (EPI != 2) bias4 = gload128(g.bias ? static_cast<const void*>(g.bias + n) : g.A);   // (no bias: any valid 16 bytes, zeroed below)
  float sc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sc[i][0] = sc[i][1] = 1.f;
    if (g.rowscale) {
      const int i0 = __builtin_amdgcn_readfirstlane((FULL ? m0 + 32 * i : min(m0 + 32 * i, last)) / g.rows_per_scale);
      const int i1 = __builtin_amdgcn_readfirstlane((FULL ? m0 + 128 + 32 * i : min(m0 + 128 + 32 * i, last)) / g.rows_per_scale);
      sc[i][0] = sload(g.rowscale + i0);                         // (waited for with the lgkmcnt(0) in front of the first barrier)
      sc[i][1] = sload(g.rowscale + i1);
    }
  }
  uint2 ga[2][8];
  float4 re[2][8];
  auto fetch = [&](int i) {
    if constexpr (EPI == 2) {
#pragma unroll
      for (int q = 0; q < 8; ++q) ga[i & 1][q] = gload64(static_cast<const __bf16*>(g.gaux) + lrow(i, q) * g.ldg + n);
    }
    if constexpr (EPI == 1 || EPI == 4) {
#pragma unroll
      for (int q = 0; q < 8; ++q) re[i & 1][q] = gload128(g.res + lrow(i, q) * g.ldr + n);
    }
  };
  if constexpr (AHEAD) fetch(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i) __builtin_amdgcn_s_barrier();                         // every wave has read pass i - 1 (waited on below)
    img_write(w0, acc[i][0], std::make_integer_sequence<int, 16>{});
    img_write(w1, acc[i][1], std::make_integer_sequence<int, 16>{});
    if constexpr (AHEAD) { if (i + 1 < 4) fetch(i + 1); } else fetch(i);
    const bool ok0 = FULL || m0 + 32 * i < g.M, ok1 = FULL || m0 + 128 + 32 * i < g.M;   // M % 32 == 0 (host): a 32-row block is in or out
    WM_WAIT_LGKM(0);
    __builtin_amdgcn_s_barrier();                                // the image of pass i is complete
    float4 v[8];
    img_read(rbase, v, std::make_integer_sequence<int, 8>{});
    // rows of pass i (and, in pass 0, the bias) have landed; younger: pass i - 1's 8 stores, pass i + 1's 8 row loads
    if constexpr (!AHEAD) WM_WAIT_VMN(0);
    else if constexpr (!ROWS) { if (i == 0) WM_WAIT_VMN(0); }
    else if constexpr (FULL) { if (i == 0 || i == 3) WM_WAIT_VMN(8); else WM_WAIT_VMN(16); }
    else { if (i == 3) WM_WAIT_VMN(0); else WM_WAIT_VMN(8); }
    WM_WAIT_LGKM(0);
    __builtin_amdgcn_sched_barrier(0);                           // (nothing that reads the asm results moves above the waits)
    if (i == 0 && !g.bias) bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EPI == 4) {
      const float4 gm = *reinterpret_cast<const float4*>(g.ln_gamma + n), bt = *reinterpret_cast<const float4*>(g.ln_beta + n);
      float s1[8], s2[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float s = sc[i][q >> 2];
        const float4 rq = re[i & 1][q];
        float4 x = v[q];
        x.x = (x.x + bias4.x) * s + rq.x; x.y = (x.y + bias4.y) * s + rq.y;
        x.z = (x.z + bias4.z) * s + rq.z; x.w = (x.w + bias4.w) * s + rq.w;
        stream_store(static_cast<float*>(g.C) + (int64_t)grow(i, q) * g.ldc + n, x);
        v[q] = x;
        s1[q] = (x.x + x.y) + (x.z + x.w);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int q = 0; q < 8; ++q) s1[q] += __shfl_xor(s1[q], o);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float mu = s1[q] * (1.0f / 256.0f);
        s1[q] = mu;
        v[q].x -= mu; v[q].y -= mu; v[q].z -= mu; v[q].w -= mu;
        s2[q] = (v[q].x * v[q].x + v[q].y * v[q].y) + (v[q].z * v[q].z + v[q].w * v[q].w);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int q = 0; q < 8; ++q) s2[q] += __shfl_xor(s2[q], o);
      const int64_t img = (int64_t)g.ln_hw * g.ln_hw;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int64_t gr = grow(i, q);
        const float rs = 1.0f / sqrtf(s2[q] * (1.0f / 256.0f) + 1e-5f);
        float4 o = make_float4(v[q].x * rs * gm.x + bt.x, v[q].y * rs * gm.y + bt.y, v[q].z * rs * gm.z + bt.z, v[q].w * rs * gm.w + bt.w);
        if (g.ln_mod) {
          const int t = (int)(gr % img), yy = t / g.ln_hw, xx = t - yy * g.ln_hw;
          const int pi = (yy - g.ln_shift + g.ln_hw) % g.ln_hw, pj = (xx - g.ln_shift + g.ln_hw) % g.ln_hw;
          const float4 mm = *reinterpret_cast<const float4*>(g.ln_mod + (int64_t)((pi & 7) * 8 + (pj & 7)) * 256 + n);
          o.x += mm.x; o.y += mm.y; o.z += mm.z; o.w += mm.w;
        }
        stream_store(static_cast<__bf16*>(g.ln_out) + gr * 256 + n, make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w)));
        if (c4 == 0) { g.ln_mean[gr] = s1[q]; g.ln_rstd[gr] = rs; }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool ok = q < 4 ? ok0 : ok1;
        const float s = sc[i][q >> 2];
        float4 x = v[q];
        if constexpr (EPI == 1) {
          const float4 rq = re[i & 1][q];
          x.x = (x.x + bias4.x) * s + rq.x; x.y = (x.y + bias4.y) * s + rq.y;
          x.z = (x.z + bias4.z) * s + rq.z; x.w = (x.w + bias4.w) * s + rq.w;
          if (ok) { if (plain) *reinterpret_cast<float4*>(static_cast<float*>(g.C) + (int64_t)grow(i, q) * g.ldc + n) = x;
                    else stream_store(static_cast<float*>(g.C) + (int64_t)grow(i, q) * g.ldc + n, x); }
        } else {
          if constexpr (EPI == 0) { x.x = (x.x + bias4.x) * s; x.y = (x.y + bias4.y) * s; x.z = (x.z + bias4.z) * s; x.w = (x.w + bias4.w) * s; }
          else {
            const uint2 gq = ga[i & 1][q];
            x.x *= s * bf_lo(gq.x); x.y *= s * bf_hi(gq.x); x.z *= s * bf_lo(gq.y); x.w *= s * bf_hi(gq.y);
          }
          const uint2 pk = make_uint2(pack_bf2(x.x, x.y), pack_bf2(x.z, x.w));
          if (ok) { if (plain) *reinterpret_cast<uint2*>(static_cast<__bf16*>(g.C) + (int64_t)grow(i, q) * g.ldc + n) = pk;
                    else stream_store(static_cast<__bf16*>(g.C) + (int64_t)grow(i, q) * g.ldc + n, pk); }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// PERSISTENT tile loop: a workgroup walks the virtual tile ids L, L + gridDim.x, ... (gridDim.x a multiple of 8, so it keeps its
// XCD and its A row panels).  With K / 32 a multiple of 4 the stage ring runs THROUGH the tile boundary: slots S - 2 and S - 1 of
// a tile request stages 0 and 1 of the workgroup's next tile into buffers 0 | 1 (free since slot S - 3), so the next tile's
// prologue latency sits under the last two slots and the epilogue, whose image is buffers 2 | 3.  The host launches one
// workgroup per tile otherwise (K % 128 != 0 or WM_GEMM_NO_PERSIST=1): the loop then runs once and nothing is requested ahead.
// DBG: ablation switches of scripts/gemm_ablate.py (1 no operand loads, 2 no MFMAs, 4 no epilogue) as a TEMPLATE parameter: the
// product instance is DBG = 0 and carries none of them; the other instances exist only in a -DWM_GEMM_ABLATE build
#define WM_PP_DBG DBG
template <int EPI, bool FULL, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(const WmGemm g, const int tiles_n, const int tiles_m, const int plain) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (blockIdx.x & 8) for (int d = plain >> 1; d > 0; --d) __builtin_amdgcn_s_sleep(16);      // EXPERIMENT: half the CUs start half a tile late
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;
  const int total = 8 * ((tiles_m + 7) / 8) * tiles_n, G = gridDim.x;
  // XCD-aware order: virtual ids that share an A row panel are equal mod 8; by >= tiles_m: a hole of the last group of 8 panels
  auto row_panel = [&](int L) { return ((L >> 3) / tiles_n) * 8 + (L & 7); };
  auto next_id = [&](int L) {
    for (L += G; L < total && row_panel(L) >= tiles_m; L += G) {}
    return L;
  };
  int L = blockIdx.x;
  if (row_panel(L) >= tiles_m) L = next_id(L);
  if (L >= total) return;
  const int S = g.K / SK;

  // direct loads: a wave-instruction fills 16 rows x 64 bytes; pass q, wave w -> rows [128 q + 16 w, + 16) of the stage image
  const int lrow = 16 * wave + (lane >> 2);
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);              // ((128 q + lrow) >> 2) & 3 == (lane >> 4) & 3
  const __bf16* pa[2];
  const __bf16* pb[2];
  auto rebase = [&](int Lt) {
    const int m0 = row_panel(Lt) * 256, n0 = ((Lt >> 3) % tiles_n) * 256;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      pa[q] = static_cast<const __bf16*>(g.A) + (int64_t)min(m0 + 128 * q + lrow, g.M - 1) * g.lda + 8 * chunk;   // rows past M: never stored
      pb[q] = static_cast<const __bf16*>(g.B) + (int64_t)(n0 + 128 * q + lrow) * g.ldb + 8 * chunk;
    }
  };
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

  rebase(L);
  request(0);
  request(1);                                                    // host: K >= 128, S >= 4
  WM_WAIT_VM(4);                                                 // stage 0 has landed (this wave's pieces)

  for (;;) {
    const int Ln = next_id(L);
    const bool more = Ln < total;                                // (only with S % 4 == 0: the host launches one tile per workgroup otherwise)
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (wm == 0) {
      for (int k = 0; k < S; ++k) {
        __builtin_amdgcn_s_barrier();                            // b_k: stage k complete, stage k - 2 free
        if (k + 2 == S && more) rebase(Ln);
        const bool rq = k + 2 < S || more;                       // stages S, S + 1 = stages 0, 1 of the next tile
        if (rq) request(k + 2);
        const uint32_t so = (k & 3) * STAGE_B;
        WM_NT_FRAGS(0, fa0 + so, fb0 + so)
        WM_NT_FRAGS(1, fa1 + so, fb1 + so)
        WM_WAIT_LGKM(6);
        WM_PP_MMA(0)
        WM_WAIT_LGKM(0);
        WM_PP_MMA(1)
        if (k + 1 < S || more) { if (rq) WM_WAIT_VM(4); else WM_WAIT_VM(0); }    // stage k + 1 landed
      }
      __builtin_amdgcn_s_barrier();                              // b_S: the other group's last slot begins
    } else {
      __builtin_amdgcn_s_barrier();                              // b_0
      WM_NT_FRAGS(0, fa0, fb0)                                   // first fragments of stage 0
      request(2);                                                // (slot 0 of this group holds nothing else)
      WM_WAIT_VM(4);                                             // stage 1 (and a previous tile's stores) done
      for (int k = 1; k <= S; ++k) {
        __builtin_amdgcn_s_barrier();                            // b_k
        const uint32_t so = ((k - 1) & 3) * STAGE_B;
        WM_NT_FRAGS(1, fa1 + so, fb1 + so)
        WM_WAIT_LGKM(6);
        WM_PP_MMA(0)
        WM_WAIT_LGKM(0);
        WM_PP_MMA(1)
        if (k + 2 == S && more) rebase(Ln);
        const bool rq = k + 2 < S || (more && k < S);
        if (rq) request(k + 2);
        if (k < S) {
          const uint32_t sn = (k & 3) * STAGE_B;
          WM_NT_FRAGS(0, fa0 + sn, fb0 + sn)                     // stage k landed before b_k: read ahead for the next slot
        }
        if (k + 1 < S || (more && k < S)) { if (rq) WM_WAIT_VM(4); else WM_WAIT_VM(0); }
      }
    }
    __builtin_amdgcn_s_barrier();                                // b_(S+1): nobody reads stages S - 2, S - 1 (buffers 2 | 3) any more
    if constexpr (DBG & 4) {                                     // (ablation: no epilogue; keep the accumulators alive)
      if (acc[0][0][0] == 12345.678f) static_cast<float*>(g.C)[0] = acc[3][1][15] + acc[1][0][3];
    } else {
      nt_epilogue_pp<EPI, FULL>(acc, g, row_panel(L) * 256, ((L >> 3) % tiles_n) * 256, lds0, tid, wm, wn, r, h, (plain & 1) != 0);
    }
    if (!more) break;
    L = Ln;
  }
#undef WM_NT_FRAGS
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

template <int EPI, bool FULL, int DBG>
int launch_nt_pp_full(const WmGemm& g, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_pp_kernel<EPI, FULL, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
    attr = true;
  }
  const int tiles_n = g.N / 256, tiles_m = (g.M + 255) / 256, total = 8 * ((tiles_m + 7) / 8) * tiles_n;
  static const bool plain = [] { const char* v = getenv("WM_PP_PLAIN"); return v && v[0] == '1'; }();
  static const int delay = [] { const char* v = getenv("WM_PP_DELAY"); return v ? atoi(v) : 0; }();
  static const bool no_persist = [] { const char* v = getenv("WM_GEMM_NO_PERSIST"); return v && v[0] == '1'; }();
  // one workgroup per CU walks the tiles when the stage ring can run through a tile boundary (K / 32 a multiple of 4)
  const int grid = (g.K % (4 * SK) == 0 && !no_persist) ? min(total, 256) : total;
  wm_gemm_family_tls = WM_FAM_NT_PP;
  hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, FULL, DBG>), dim3(grid), dim3(512), SMEM_B, st, g, tiles_n, tiles_m, (plain ? 1 : 0) | (delay << 1));
  WM_CHECK_LAUNCH();
  return WM_OK;
}
template <int EPI, int DBG = 0>
int launch_nt_pp(const WmGemm& g, hipStream_t st) {
  if constexpr (EPI == 4 || DBG != 0) return launch_nt_pp_full<EPI, true, DBG>(g, st);     // (EPI 4: host has checked M % 256 == 0)
  else return g.M % 256 == 0 ? launch_nt_pp_full<EPI, true, DBG>(g, st) : launch_nt_pp_full<EPI, false, DBG>(g, st);
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
