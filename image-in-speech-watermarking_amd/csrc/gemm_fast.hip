// bf16-MFMA GEMMs of the train step with BRANCH-FREE operand staging (mode 1).
//
// The first generic kernel (gemm.hip) guards every global load with its own bounds branch and picks layouts at run
// time; its K loop compiled to ~1600 basic blocks in which loads are issued one at a time behind exec-mask
// branches, so each K tile paid several full memory latencies (measured 15k cycles per K tile against 512 cycles of
// MFMA).  Here every layout decision is a template parameter, addresses are clamped instead of guarded (out-of-range
// lanes load a valid address and the value is zeroed with a select), and row pointers are set up once before the
// loop, so a K tile is: N back-to-back 16-byte loads -> MFMAs of the previous tile -> 16-byte LDS stores -> barrier.
//
//   gemm_nt_kernel   C[M,N] = epi(A[M,K] * B^T),  A K-contiguous (fp32 or bf16),
//                    B K-contiguous ([N,K], forward) or row-contiguous ([K,N], dX; read with ds_read_b64_tr_b16)
//   gemm_tn_kernel   C[M,N] += A[K,M]^T B[K,N] (+ column sums of A): weight gradients, split-K, fp32 atomics
#include <cstdlib>
#include "common.h"
#include "gelu_poly.h"
#include "../../include/wm_kernels.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
#ifndef NT_SMALL_BK
#define NT_SMALL_BK 128
#endif
#ifndef NT_NBUF
#define NT_NBUF 1
#endif
#ifndef TN_NBUF
#define TN_NBUF 1
#endif
#ifndef TN_TARGET
#define TN_TARGET 512
#endif

// MFMA 32x32x16 operand out of an as-it-lies [k][columns] LDS image (lane l: column l&31, k = 8*(l>>5) + j)
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int ld, int col0, int kstep, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const __bf16* a = tile + (16 * kstep + 8 * (g >> 1) + q) * ld + col0 + 16 * (g & 1) + 4 * p;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * ld));
  bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return f;
}

template <typename ST> struct Chunk;                       // one 16-byte global load
template <> struct Chunk<__bf16> {
  static constexpr int E = 8;
  uint4 v;
  __device__ __forceinline__ void ld(const __bf16* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void zero_if(bool z) { if (z) v = make_uint4(0, 0, 0, 0); }
  template <bool GELU> __device__ __forceinline__ void st(__bf16* s, float sc, bool scaled) const {
    uint4 x = v;
    if (GELU || scaled) {
      uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 a = {bf_lo(w[j]), bf_hi(w[j])};
        if (GELU) {                                          // bf16 in, bf16 out: the packed polynomial form (gelu_poly.h) —
          f32x2 g;                                           // fc2's weight gradient reads the saved h2 of the fused LeFF this way
          gelu_pair<__bf16>(a, a, g);
        }
        w[j] = pack_bf2(a.x * sc, a.y * sc);
      }
      x = make_uint4(w[0], w[1], w[2], w[3]);
    }
    *reinterpret_cast<uint4*>(s) = x;
  }
};
template <> struct Chunk<float> {
  static constexpr int E = 4;
  float4 v;
  __device__ __forceinline__ void ld(const float* p) { v = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void zero_if(bool z) { if (z) v = make_float4(0.f, 0.f, 0.f, 0.f); }
  template <bool GELU> __device__ __forceinline__ void st(__bf16* s, float sc, bool) const {
    float4 x = v;
    if (GELU) { x.x = gelu_f(x.x); x.y = gelu_f(x.y); x.z = gelu_f(x.z); x.w = gelu_f(x.w); }
    *reinterpret_cast<uint2*>(s) = make_uint2(pack_bf2(x.x * sc, x.y * sc), pack_bf2(x.z * sc, x.w * sc));
  }
  // split precision (compute_mode "split"): x = hi + lo with hi = bf16(x), lo = bf16(x - hi); the two images lie lo_off apart
  __device__ __forceinline__ void st_split(__bf16* s, int lo_off, float sc) const {
    const float e[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
    bf16x4 hi, lo;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      hi[u] = (__bf16)e[u];
      lo[u] = (__bf16)(e[u] - (float)hi[u]);
    }
    *reinterpret_cast<bf16x4*>(s) = hi;
    *reinterpret_cast<bf16x4*>(s + lo_off) = lo;
  }
};

// Operand staging.  The k loop of both kernels spent more cycles issuing address arithmetic, bounds clamps and
// zero-selects (~240 VALU instructions per tile and thread) than MFMAs, so the per-tile work is reduced to
// "load through a pointer, bump the pointer": `load_full` / `store_full` touch no bounds at all and are used for every
// tile that lies completely inside K; only the last (partial) tile of a K range and workgroups on the M / N edge of the
// matrix take the `*_tail` / `edge` variants.  Nothing may READ a loaded register before the MFMAs of the current tile
// (not even to zero it): that would make the compiler wait for the data first, i.e. no overlap.

// [R rows][BK] tile of a K-contiguous operand -> LDS [R][BK + 8]
template <typename ST, int R, int BK, bool GELU, bool SPLIT = false> struct KcStage {
  static constexpr int E = Chunk<ST>::E, CPR = BK / E, RPP = 256 / CPR, NP = R / RPP, LD = BK + 8;
  static constexpr int LO = R * LD;                          // SPLIT: offset of the low-part image (elements)
  static_assert(R % RPP == 0, "tile rows must be a multiple of the rows covered per pass");
  const ST* ptr[NP];                                         // this thread's chunk of the NEXT tile to load
  Chunk<ST> c[NP];
  uint32_t rowok;
  int kq, soff;
  bool edge;                                                 // some row of this workgroup's tile is outside the matrix
  // grid_n > 0: row (b, Y, X) of the compact n x n enumeration lives at row b (n+2)^2 + (Y+1)(n+2) + X+1 of the operand (WmGemm::a_grid_n)
  __device__ __forceinline__ void init(const void* p, int64_t ld, int row0, int rmax, int k0, int tid, int grid_n = 0) {
    const int r0 = tid / CPR;
    kq = E * (tid % CPR); rowok = 0; soff = r0 * LD + kq;
    edge = row0 + R > rmax;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int row = row0 + r0 + RPP * q;
      if (row < rmax) rowok |= 1u << q;
      int64_t rr = min(row, rmax - 1);
      if (grid_n > 0) {
        const int nn = grid_n * grid_n, b = (int)rr / nn, t = (int)rr - b * nn, Y = t / grid_n, X = t - Y * grid_n;
        rr = (int64_t)b * (grid_n + 2) * (grid_n + 2) + (Y + 1) * (grid_n + 2) + X + 1;
      }
      ptr[q] = static_cast<const ST*>(p) + rr * ld + k0 + kq;
    }
  }
  __device__ __forceinline__ void jump(int64_t d) {
#pragma unroll
    for (int q = 0; q < NP; ++q) ptr[q] += d;
  }
  __device__ __forceinline__ void load_full() {
#pragma unroll
    for (int q = 0; q < NP; ++q) { c[q].ld(ptr[q]); ptr[q] += BK; }
  }
  __device__ __forceinline__ void load_tail(int k0, int kend) {       // K % 8 == 0: a chunk is inside or outside
    const int back = k0 + kq < kend ? 0 : k0 + kq;                   // outside: re-read column 0 (valid memory), zeroed at store
#pragma unroll
    for (int q = 0; q < NP; ++q) c[q].ld(ptr[q] - back);
  }
  __device__ __forceinline__ void store_full(__bf16* s) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      if (edge) c[q].zero_if(!((rowok >> q) & 1));
      if constexpr (SPLIT) c[q].st_split(s + soff + RPP * q * LD, LO, 1.f);
      else c[q].template st<GELU>(s + soff + RPP * q * LD, 1.f, false);
    }
  }
  __device__ __forceinline__ void store_tail(__bf16* s, int k0, int kend) {
    const bool kok = k0 + kq < kend;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      c[q].zero_if(!kok || !((rowok >> q) & 1));
      if constexpr (SPLIT) c[q].st_split(s + soff + RPP * q * LD, LO, 1.f);
      else c[q].template st<GELU>(s + soff + RPP * q * LD, 1.f, false);
    }
  }
};

// [KT k-rows][R columns] tile of a row-contiguous operand -> LDS [KT][R + 32] (as it lies)
template <typename ST, int R, int KT, bool GELU, bool SPLIT = false> struct TrStage {
  static constexpr int E = Chunk<ST>::E, CPR = R / E, NCH = KT * CPR, NP = NCH / 256, RPQ = 256 / CPR, LD = R + 32;
  static constexpr int LO = KT * LD;                         // SPLIT: offset of the low-part image (elements)
  static_assert(NCH % 256 == 0 && 256 % CPR == 0, "tile must be a multiple of 256 chunks");
  const ST* ptr[NP];                                         // this thread's chunk (row kr + RPQ q, fixed column) of the next tile
  Chunk<ST> c[NP];
  float sc;                                                  // per-tile scale on A (DropPath keep factor of the tile's sample)
  int64_t ld, step;
  int kr, soff;
  bool colok, edge;
  __device__ __forceinline__ void init(const void* p, int64_t ld_, int c0, int cmax, int k0, int tid) {
    ld = ld_; step = (int64_t)KT * ld_; sc = 1.f;
    kr = tid / CPR;
    const int col = c0 + E * (tid % CPR);
    colok = col < cmax;                                      // cmax is a multiple of E (checked on the host)
    edge = c0 + R > cmax;
    soff = kr * LD + E * (tid % CPR);
#pragma unroll
    for (int q = 0; q < NP; ++q) ptr[q] = static_cast<const ST*>(p) + (int64_t)(k0 + kr + RPQ * q) * ld_ + min(col, cmax - E);
  }
  __device__ __forceinline__ void load_full() {
#pragma unroll
    for (int q = 0; q < NP; ++q) { c[q].ld(ptr[q]); ptr[q] += step; }
  }
  __device__ __forceinline__ void load_tail(int k0, int kend) {       // rows at or beyond kend re-read row kend - 1
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int over = k0 + kr + RPQ * q - (kend - 1);
      c[q].ld(ptr[q] - (over > 0 ? over : 0) * ld);
    }
  }
  // kscale[k / kps] is constant over a tile when kps % KT == 0 and the K range starts on a tile boundary (checked by the
  // caller): one scalar load per tile instead of a division per chunk
  __device__ __forceinline__ void load_scale(const float* kscale, int kps, int k0) { sc = kscale[k0 / kps]; }
  __device__ __forceinline__ void store_full(__bf16* s, bool scaled) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      if (edge) c[q].zero_if(!colok);
      if constexpr (SPLIT) c[q].st_split(s + soff + RPQ * q * LD, LO, sc);
      else c[q].template st<GELU>(s + soff + RPQ * q * LD, sc, scaled);
    }
  }
  __device__ __forceinline__ void store_tail(__bf16* s, bool scaled, int k0, int kend) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      c[q].zero_if(k0 + kr + RPQ * q >= kend || !colok);
      if constexpr (SPLIT) c[q].st_split(s + soff + RPQ * q * LD, LO, sc);
      else c[q].template st<GELU>(s + soff + RPQ * q * LD, sc, scaled);
    }
  }
};

// XCD-aware tile order (speed only): workgroups that re-read one activation slab get ids equal mod 8 (same XCD L2)
__device__ __forceinline__ bool tile_of_block(int tiles_n, int tiles_m, int ksplit, int& bx, int& by, int& bz) {
  const int L = blockIdx.x, j = L & 7, sq = L >> 3;
  if (ksplit > 1) {
    const int tiles = tiles_n * tiles_m, t = sq % tiles;
    bz = (sq / tiles) * 8 + j; bx = t % tiles_n; by = t / tiles_n;
    return bz < ksplit;
  }
  bz = 0; bx = sq % tiles_n; by = (sq / tiles_n) * 8 + j;
  return by < tiles_m;
}

// Interior epilogue rows (whole pass inside M, 16-byte aligned operands) for the three operand combinations that carry
// almost every launch of the train step (EPI template parameter of the kernel).  A short-K launch spends as long here as
// in its k loop, so the per-row work is: one LDS read, the packed arithmetic, one store, pointer bumps - no bounds
// checks, no 64-bit multiplies, no branches.
//   EPI 0: C(bf16) = (acc + bias) * sc                       (qkv / fc1 forward, plain dX, proj dX with DropPath scale)
//   EPI 1: C(fp32) = (acc + bias) * sc + res                 (proj / fc2 forward with DropPath scale and residual)
//   EPI 2: C(bf16) = acc * sc * gaux(bf16 factor)            (fc2 dX times the stored gelu')
//   EPI 4: EPI 1 + the LayerNorm of the finished row (N == the tile width: the TPRW threads of a row hold all of it) written as
//          bf16 with its statistics: norm2 behind proj, the next block's norm1 (+ window modulator) behind fc2
__device__ __forceinline__ int ln_window_pos(int t, int hw, int shift) {      // t = token index inside its image
  const int yy = t / hw, xx = t - yy * hw;
  const int i = (yy - shift + hw) % hw, j = (xx - shift + hw) % hw;
  return (i & 7) * 8 + (j & 7);
}
template <int EPI, int NR, int RPP, int LDC, int TPRW = 32>
__device__ __forceinline__ void epi_interior(const float* cs, const WmGemm& g, int64_t row0, int n, float4 bias4, float sc) {
  const int64_t cstep = (int64_t)RPP * g.ldc;
  if constexpr (EPI == 4) {
    float4 re[NR], v[NR];
    {
      const float* rp = g.res + row0 * g.ldr + n;
      const int64_t rstep = (int64_t)RPP * g.ldr;
#pragma unroll
      for (int q = 0; q < NR; ++q, rp += rstep) re[q] = *reinterpret_cast<const float4*>(rp);
    }
    const float4 gm = *reinterpret_cast<const float4*>(g.ln_gamma + n), bt = *reinterpret_cast<const float4*>(g.ln_beta + n);
    float* cf = static_cast<float*>(g.C) + row0 * g.ldc + n;
    float s1[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q, cf += cstep) {
      float4 x = *reinterpret_cast<const float4*>(cs + q * RPP * LDC);
      x.x = (x.x + bias4.x) * sc + re[q].x; x.y = (x.y + bias4.y) * sc + re[q].y;
      x.z = (x.z + bias4.z) * sc + re[q].z; x.w = (x.w + bias4.w) * sc + re[q].w;
      stream_store(cf, x);
      v[q] = x;
      s1[q] = (x.x + x.y) + (x.z + x.w);
    }
#pragma unroll
    for (int o = TPRW / 2; o > 0; o >>= 1)
#pragma unroll
      for (int q = 0; q < NR; ++q) s1[q] += __shfl_xor(s1[q], o);
    constexpr float invn = 1.0f / (4 * TPRW);
    float s2[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const float mu = s1[q] * invn;
      s1[q] = mu;
      v[q].x -= mu; v[q].y -= mu; v[q].z -= mu; v[q].w -= mu;
      s2[q] = (v[q].x * v[q].x + v[q].y * v[q].y) + (v[q].z * v[q].z + v[q].w * v[q].w);
    }
#pragma unroll
    for (int o = TPRW / 2; o > 0; o >>= 1)
#pragma unroll
      for (int q = 0; q < NR; ++q) s2[q] += __shfl_xor(s2[q], o);
    __bf16* yo = static_cast<__bf16*>(g.ln_out) + row0 * g.N + n;
    const int tbase = g.ln_mod ? (int)(row0 % ((int64_t)g.ln_hw * g.ln_hw)) : 0;       // a row tile never crosses an image
#pragma unroll
    for (int q = 0; q < NR; ++q, yo += (int64_t)RPP * g.N) {
      const float rs = 1.0f / sqrtf(s2[q] * invn + 1e-5f);
      float4 o = make_float4(v[q].x * rs * gm.x + bt.x, v[q].y * rs * gm.y + bt.y, v[q].z * rs * gm.z + bt.z, v[q].w * rs * gm.w + bt.w);
      if (g.ln_mod) {
        const float4 mm = *reinterpret_cast<const float4*>(g.ln_mod + (int64_t)ln_window_pos(tbase + q * RPP, g.ln_hw, g.ln_shift) * g.N + n);
        o.x += mm.x; o.y += mm.y; o.z += mm.z; o.w += mm.w;
      }
      stream_store(yo, make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w)));
      if ((threadIdx.x % TPRW) == 0) { g.ln_mean[row0 + (int64_t)q * RPP] = s1[q]; g.ln_rstd[row0 + (int64_t)q * RPP] = rs; }
    }
    return;
  }
  uint2 ga[NR];
  float4 re[NR];
  if constexpr (EPI == 2) {
    const __bf16* gp = static_cast<const __bf16*>(g.gaux) + row0 * g.ldg + n;
    const int64_t gstep = (int64_t)RPP * g.ldg;
#pragma unroll
    for (int q = 0; q < NR; ++q, gp += gstep) ga[q] = *reinterpret_cast<const uint2*>(gp);
  }
  if constexpr (EPI == 1) {
    const float* rp = g.res + row0 * g.ldr + n;
    const int64_t rstep = (int64_t)RPP * g.ldr;
#pragma unroll
    for (int q = 0; q < NR; ++q, rp += rstep) re[q] = *reinterpret_cast<const float4*>(rp);
  }
  if constexpr (EPI == 1) {
    float* cf = static_cast<float*>(g.C) + row0 * g.ldc + n;
#pragma unroll
    for (int q = 0; q < NR; ++q, cf += cstep) {
      float4 v = *reinterpret_cast<const float4*>(cs + q * RPP * LDC);
      v.x = (v.x + bias4.x) * sc + re[q].x; v.y = (v.y + bias4.y) * sc + re[q].y;
      v.z = (v.z + bias4.z) * sc + re[q].z; v.w = (v.w + bias4.w) * sc + re[q].w;
      stream_store(cf, v);
    }
  } else {
    __bf16* ch = static_cast<__bf16*>(g.C) + row0 * g.ldc + n;
#pragma unroll
    for (int q = 0; q < NR; ++q, ch += cstep) {
      float4 v = *reinterpret_cast<const float4*>(cs + q * RPP * LDC);
      if constexpr (EPI == 0) { v.x = (v.x + bias4.x) * sc; v.y = (v.y + bias4.y) * sc; v.z = (v.z + bias4.z) * sc; v.w = (v.w + bias4.w) * sc; }
      else { v.x *= sc * bf_lo(ga[q].x); v.y *= sc * bf_hi(ga[q].x); v.z *= sc * bf_lo(ga[q].y); v.w *= sc * bf_hi(ga[q].y); }
      stream_store(ch, make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)));
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 x bf16 instances are held to 168 registers per lane (3 waves per SIMD = 3 workgroups per CU); they need 172 unconstrained
// SPLIT (compute_mode "split", fp32 operands): every operand tile is staged as a high and a low bf16 image (x = hi + lo) and a
// product is three MFMAs, hi*hi + hi*lo + lo*hi: 2^-17 relative per term at 3/16 of the fp32-MFMA time (the round-2 STFT trick)
// BKT: k depth of a staged tile.  64 everywhere but in the 64 x 64-tile instances, which run where a launch has about one
// workgroup per CU (the deep levels at small batches): nothing overlaps their load -> LDS -> MFMA chain there, an iteration lasts one
// memory round trip whatever it carries, so they carry 128.
template <typename AT, typename BT, bool B_TR, bool A_GELU, int WM, int WN, int TM, int TN, int EPI, bool SPLIT = false, int BKT = 64>
__global__ __launch_bounds__(256, (sizeof(AT) == 2 && EPI != 3 && EPI != 4 ? 3 : 2)) void gemm_nt_kernel(const WmGemm g, const bool vec_ok, const int tiles_n, const int tiles_m) {
  constexpr int BK = BKT, BM = WM * TM * 32, BN = WN * TN * 32, LDA = BK + 8;
  constexpr int LDB = B_TR ? BN + 32 : BK + 8, BSZ1 = B_TR ? BK * LDB : BN * LDB;
  constexpr int ASZ = (SPLIT ? 2 : 1) * BM * LDA, BSZ = (SPLIT ? 2 : 1) * BSZ1;
  constexpr int NBUF = NT_NBUF;                              // LDS operand buffers (1: more workgroups per CU, one more barrier per k tile)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);          // [NBUF][ASZ]
  __bf16* Bs = As + NBUF * ASZ;                              // [NBUF][BSZ]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
  int bx, by, bz;
  if (!tile_of_block(tiles_n, tiles_m, 1, bx, by, bz)) return;
  const int m0 = by * BM, n0 = bx * BN, K = g.K;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  KcStage<AT, BM, BK, A_GELU, SPLIT> sa;
  sa.init(g.A, g.lda, m0, g.M, 0, tid, g.a_grid_n);
  const int jump_tile = g.a_jump_k > 0 ? g.a_jump_k / BK : -1;   // the tile whose load is followed by the pointer jump (host: a_jump_k % 128 == 0)
  KcStage<BT, BN, BK, false, SPLIT> sbk;
  TrStage<BT, BN, BK, false, SPLIT> sbt;
  if constexpr (B_TR) sbt.init(g.B, g.ldb, n0, g.N, 0, tid); else sbk.init(g.B, g.ldb, n0, g.N, 0, tid);

  const int nfull = K / BK, ntiles = (K + BK - 1) / BK;     // tiles [0, nfull) lie completely inside K
  if (nfull > 0) {
    sa.load_full();
    if (jump_tile == 1) sa.jump(g.a_jump);
    if constexpr (B_TR) sbt.load_full(); else sbk.load_full();
    sa.store_full(As);
    if constexpr (B_TR) sbt.store_full(Bs, false); else sbk.store_full(Bs);
  } else {
    sa.load_tail(0, K);
    if constexpr (B_TR) sbt.load_tail(0, K); else sbk.load_tail(0, K);
    sa.store_tail(As, 0, K);
    if constexpr (B_TR) sbt.store_tail(Bs, false, 0, K); else sbk.store_tail(Bs, 0, K);
  }
  __syncthreads();
  int buf = 0;
  for (int t = 0; t < ntiles; ++t) {
    const int nx = t + 1;
    if (nx < nfull) {
      sa.load_full();
      if (nx + 1 == jump_tile) sa.jump(g.a_jump);             // (tiles [0, jump_tile) are loaded: the rest lie a_jump further on)
      if constexpr (B_TR) sbt.load_full(); else sbk.load_full();
    } else if (nx < ntiles) {
      sa.load_tail(nx * BK, K);
      if constexpr (B_TR) sbt.load_tail(nx * BK, K); else sbk.load_tail(nx * BK, K);
    }
    const __bf16* Ab = As + buf * ASZ + ((wm * TM) * 32 + r) * LDA;
    const __bf16* Bb = Bs + buf * BSZ;
    // operand fragments of k-step ks + 1 are read from LDS while the MFMAs of step ks run (at one or two waves per SIMD
    // nothing else hides the LDS latency: waiting for each step's reads in turn parked the wave ~40 % of the loop)
    bf16x8 a[2][TM], b[2][TN];
    auto frags = [&](int ks, bf16x8 (&fa)[TM], bf16x8 (&fb)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * LDA + 16 * ks + 8 * h);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (B_TR) fb[j] = tr_frag(Bb, LDB, (wn * TN + j) * 32, ks, lane);
        else fb[j] = *reinterpret_cast<const bf16x8*>(Bb + ((wn * TN + j) * 32 + r) * LDB + 16 * ks + 8 * h);
      }
    };
    if constexpr (SPLIT) {
      // low-part fragments straight before their use (the split instances run at two waves per SIMD either way)
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 ah[TM], bh[TN], al[TM], bl[TN];
        frags(ks, ah, bh);
#pragma unroll
        for (int i = 0; i < TM; ++i) al[i] = *reinterpret_cast<const bf16x8*>(Ab + BM * LDA + i * 32 * LDA + 16 * ks + 8 * h);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (B_TR) bl[j] = tr_frag(Bb + BSZ1, LDB, (wn * TN + j) * 32, ks, lane);
          else bl[j] = *reinterpret_cast<const bf16x8*>(Bb + BSZ1 + ((wn * TN + j) * 32 + r) * LDB + 16 * ks + 8 * h);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);     // small terms first
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
    } else {
    frags(0, a[0], b[0]);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      if (ks + 1 < BK / 16) frags(ks + 1, a[(ks + 1) & 1], b[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);                      // keep the reads ahead of the MFMAs (the scheduler sinks them back)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][i], b[ks & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    const int nb = NBUF == 2 ? buf ^ 1 : 0;
    if (NBUF == 1) __syncthreads();                           // every wave is done reading the single buffer
    if (nx < nfull) {
      sa.store_full(As + nb * ASZ);
      if constexpr (B_TR) sbt.store_full(Bs + nb * BSZ, false); else sbk.store_full(Bs + nb * BSZ);
    } else if (nx < ntiles) {
      sa.store_tail(As + nb * ASZ, nx * BK, K);
      if constexpr (B_TR) sbt.store_tail(Bs + nb * BSZ, false, nx * BK, K); else sbk.store_tail(Bs + nb * BSZ, nx * BK, K);
    }
    __syncthreads();
    buf = nb;
  }

  // epilogue: accumulators -> LDS (fp32) -> coalesced 16-byte rows, one wave-row (TM * 32 rows) of the tile per pass so that
  // the staging image is 1 / WM of the tile (the full fp32 tile was the largest LDS user and capped the workgroups per CU)
  constexpr int LDC = BN + 4, PR = TM * 32;                  // rows per pass
  float* Cs = reinterpret_cast<float*>(smem_raw);
  constexpr int TPRW = BN / 4, RPP = 256 / TPRW;
  const int c4 = tid % TPRW, n = n0 + 4 * c4;
  const bool full = vec_ok && n + 3 < g.N;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (full && g.bias) bias4 = *reinterpret_cast<const float4*>(g.bias + n);
  const bool sc_uniform = g.rowscale && (g.rows_per_scale % BM) == 0;       // one sample per row tile
  const float sc_tile = sc_uniform ? g.rowscale[m0 / g.rows_per_scale] : 1.f;
#pragma unroll 1
  for (int pass = 0; pass < WM; ++pass) {
    __syncthreads();
    if (wm == pass) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            Cs[(i * 32 + mma32_row(e, h)) * LDC + (wn * TN + j) * 32 + r] = acc[i][j][e];
    }
    __syncthreads();
    if (n >= g.N) continue;                                   // column-edge tile: this thread's four columns do not exist
    const int mp = m0 + pass * PR;                            // first row of this pass
    if constexpr (EPI != 3) {
      if (full && mp + PR <= g.M) {                           // (the host picked EPI from the operands that are present)
        epi_interior<EPI, PR / RPP, RPP, LDC, TPRW>(Cs + (tid / TPRW) * LDC + 4 * c4, g, mp + tid / TPRW, n, bias4, g.rowscale ? sc_tile : 1.f);
        continue;
      }
    }
    if (full) {
      // All epilogue operands of a group of rows (DropPath scale, gelu' factor, residual) are requested BEFORE the first
      // one is used: written as one loop the compiler waited for every load in turn, ~3 round trips per row, which
      // dominated the short-K launches.
      // (specialised instances run this general code only on edge tiles: a small group keeps its operand arrays from
      // setting the kernel's register count, which decides how many workgroups fit on a CU)
      constexpr int NR = PR / RPP, G = EPI != 3 ? (NR < 2 ? NR : 2) : (NR < 8 ? NR : 8);
      static_assert(NR >= 1 && NR % G == 0, "rows per thread");
      const int r0 = tid / TPRW;
#pragma unroll
      for (int g0 = 0; g0 < NR; g0 += G) {
        uint4 ga[G];
        float4 re[G];
        float sc[G];
        int64_t mrow[G];
#pragma unroll
        for (int q = 0; q < G; ++q) {
          mrow[q] = min(mp + r0 + (g0 + q) * RPP, g.M - 1);
          sc[q] = sc_tile;
        }
        // one straight-line loop per operand (a per-row if / else makes the compiler drain the loads row by row)
        if (g.rowscale && !sc_uniform) {
#pragma unroll
          for (int q = 0; q < G; ++q) sc[q] = g.rowscale[mrow[q] / g.rows_per_scale];
        }
        if (g.gaux && g.gaux_bf16) {
#pragma unroll
          for (int q = 0; q < G; ++q) {
            const uint2 t = *reinterpret_cast<const uint2*>(static_cast<const __bf16*>(g.gaux) + mrow[q] * g.ldg + n);
            ga[q] = make_uint4(t.x, t.y, 0u, 0u);
          }
        } else if (g.gaux) {
#pragma unroll
          for (int q = 0; q < G; ++q) ga[q] = *reinterpret_cast<const uint4*>(static_cast<const float*>(g.gaux) + mrow[q] * g.ldg + n);
        }
        if (g.res) {
#pragma unroll
          for (int q = 0; q < G; ++q) re[q] = *reinterpret_cast<const float4*>(g.res + mrow[q] * g.ldr + n);
        }
#pragma unroll
        for (int q = 0; q < G; ++q) {
          const int rr = r0 + (g0 + q) * RPP, m = mp + rr;
          if (m < g.M) {
            float4 v = *reinterpret_cast<const float4*>(Cs + rr * LDC + 4 * c4);
            v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
            if (g.rowscale) { v.x *= sc[q]; v.y *= sc[q]; v.z *= sc[q]; v.w *= sc[q]; }
            if (g.gaux) {
              float4 a;
              if (g.gaux_bf16) a = make_float4(bf_lo(ga[q].x), bf_hi(ga[q].x), bf_lo(ga[q].y), bf_hi(ga[q].y));
              else a = make_float4(__uint_as_float(ga[q].x), __uint_as_float(ga[q].y), __uint_as_float(ga[q].z), __uint_as_float(ga[q].w));
              if (g.gaux_raw) { v.x *= a.x; v.y *= a.y; v.z *= a.z; v.w *= a.w; }
              else { v.x *= gelu_grad_f(a.x); v.y *= gelu_grad_f(a.y); v.z *= gelu_grad_f(a.z); v.w *= gelu_grad_f(a.w); }
            }
            if (g.res) { v.x += re[q].x; v.y += re[q].y; v.z += re[q].z; v.w += re[q].w; }
            if (g.c_bf16) IO<__bf16>::st4(static_cast<__bf16*>(g.C) + (int64_t)m * g.ldc + n, v);
            else IO<float>::st4(static_cast<float*>(g.C) + (int64_t)m * g.ldc + n, v);
          }
        }
      }
    } else {
      for (int rr = tid / TPRW; rr < PR; rr += RPP) {         // column-edge tiles / unaligned operands: scalar path
        const int m = mp + rr;
        if (m >= g.M) break;
        const float4 v = *reinterpret_cast<const float4*>(Cs + rr * LDC + 4 * c4);
        const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int nn = n + q;
          if (nn >= g.N) continue;
          float val = e4[q];
          if (g.bias) val += g.bias[nn];
          if (g.rowscale) val *= g.rowscale[m / g.rows_per_scale];
          if (g.gaux) {
            const float av = g.gaux_bf16 ? (float)static_cast<const __bf16*>(g.gaux)[(int64_t)m * g.ldg + nn]
                                         : static_cast<const float*>(g.gaux)[(int64_t)m * g.ldg + nn];
            val *= g.gaux_raw ? av : gelu_grad_f(av);
          }
          if (g.res) val += g.res[(int64_t)m * g.ldr + nn];
          if (g.c_bf16) static_cast<__bf16*>(g.C)[(int64_t)m * g.ldc + nn] = (__bf16)val;
          else static_cast<float*>(g.C)[(int64_t)m * g.ldc + nn] = val;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
constexpr int TBK = 64;                                    // k rows per tile of the weight-gradient kernel (32 KB of loads in flight per workgroup)

template <typename AT, typename BT, bool B_GELU, int TM, int TN, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const WmGemm g, const int klen, const int tiles_n, const int tiles_m,
                                                      const int ksplit, float* __restrict__ part) {
  constexpr int BM = 64 * TM, BN = 64 * TN, LDA = BM + 32, LDB = BN + 32;
  constexpr int NBUF = TN_NBUF;                               // 1: half the LDS, twice the workgroups per CU, one more barrier per k tile
  __shared__ __attribute__((aligned(16))) __bf16 As[NBUF][(SPLIT ? 2 : 1) * TBK * LDA];     // SPLIT: high | low image
  __shared__ __attribute__((aligned(16))) __bf16 Bs[NBUF][(SPLIT ? 2 : 1) * TBK * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  int bx, by, bz;
  {
    const int L = blockIdx.x, j = L & 7, sq = L >> 3, tiles = tiles_n * tiles_m, t = sq % tiles;
    bz = (sq / tiles) * 8 + j; bx = t % tiles_n; by = t / tiles_n;
    if (bz >= ksplit) return;
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * klen, kend = min(g.K, kbeg + klen);
  // Bias gradient = column sums of A over this K range.  They do not depend on the N tile, so the k rows of every
  // staged A tile are dealt out over the tiles_n workgroups that hold it (and over the 256 / BM thread groups of each):
  // having the bx == 0 workgroups sum all 64 rows alone made them the stragglers of the launch (~1 us per k tile).
  const int bias_step = tiles_n * (256 / BM), bias_first = bx * (256 / BM) + tid / BM;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float bsum = 0.f;

  TrStage<AT, BM, TBK, false, SPLIT> sa;
  TrStage<BT, BN, TBK, B_GELU, SPLIT> sb;
  sa.init(g.A, g.lda, m0, g.M, kbeg, tid);
  const BT* Bp = static_cast<const BT*>(g.B);
  if (g.b_seg_cols > 0) {                                    // column block of this tile: rows further down, columns of the stored row
    const int seg = n0 / g.b_seg_cols;
    Bp += (int64_t)g.b_seg_rows[seg] * g.ldb - (int64_t)seg * g.b_seg_cols;
  }
  sb.init(Bp, g.ldb, n0, g.N, kbeg, tid);
  const bool scaled = g.kscale != nullptr;                   // host guarantees k_per_scale % TBK == 0 and klen % TBK == 0
  const int nfull = (kend - kbeg) / TBK, ntiles = (kend - kbeg + TBK - 1) / TBK;
  if (scaled) sa.load_scale(g.kscale, g.k_per_scale, kbeg);
  if (nfull > 0) {
    sa.load_full(); sb.load_full();
    sa.store_full(As[0], scaled); sb.store_full(Bs[0], false);
  } else {
    sa.load_tail(kbeg, kend); sb.load_tail(kbeg, kend);
    sa.store_tail(As[0], scaled, kbeg, kend); sb.store_tail(Bs[0], false, kbeg, kend);
  }
  __syncthreads();
  int buf = 0;
  for (int t = 0; t < ntiles; ++t) {
    const int nx = t + 1, kn = kbeg + nx * TBK;
    if (nx < ntiles && scaled) sa.load_scale(g.kscale, g.k_per_scale, kn);
    if (nx < nfull) { sa.load_full(); sb.load_full(); }
    else if (nx < ntiles) { sa.load_tail(kn, kend); sb.load_tail(kn, kend); }
    bf16x8 a[2][TM], b[2][TN];                              // fragments of step ks + 1 are read while step ks multiplies
    auto frags = [&](int ks, bf16x8 (&fa)[TM], bf16x8 (&fb)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = tr_frag(As[buf], LDA, (wm * TM + i) * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = tr_frag(Bs[buf], LDB, (wn * TN + j) * 32, ks, lane);
    };
    if constexpr (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < TBK / 16; ++ks) {
        bf16x8 ah[TM], bh[TN], al[TM], bl[TN];
        frags(ks, ah, bh);
#pragma unroll
        for (int i = 0; i < TM; ++i) al[i] = tr_frag(As[buf] + TBK * LDA, LDA, (wm * TM + i) * 32, ks, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) bl[j] = tr_frag(Bs[buf] + TBK * LDB, LDB, (wn * TN + j) * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
    } else {
    frags(0, a[0], b[0]);
#pragma unroll
    for (int ks = 0; ks < TBK / 16; ++ks) {
      if (ks + 1 < TBK / 16) frags(ks + 1, a[(ks + 1) & 1], b[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);                      // keep the reads ahead of the MFMAs (the scheduler sinks them back)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][i], b[ks & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    if (g.b_ones) {                                            // column sums of A (bias gradient), see bias_first above
#pragma unroll 4
      for (int kr = bias_first; kr < TBK; kr += bias_step) {
        bsum += (float)As[buf][kr * LDA + (tid & (BM - 1))];
        if constexpr (SPLIT) bsum += (float)As[buf][TBK * LDA + kr * LDA + (tid & (BM - 1))];
      }
    }
    const int nb = NBUF == 2 ? buf ^ 1 : 0;
    if (NBUF == 1) __syncthreads();                           // every wave is done reading the single buffer
    if (nx < nfull) { sa.store_full(As[nb], scaled); sb.store_full(Bs[nb], false); }
    else if (nx < ntiles) { sa.store_tail(As[nb], scaled, kn, kend); sb.store_tail(Bs[nb], false, kn, kend); }
    __syncthreads();
    buf = nb;
  }
  float* C = static_cast<float*>(g.C);
  const int how = part ? 0 : (ksplit == 1 ? 1 : 2);
  float* const dst = part ? part + (int64_t)bz * g.M * g.N : C;
  const int64_t ldd = part ? g.N : g.ldc;
  if (how == 0 && m0 + BM <= g.M && n0 + BN <= g.N) {
    // interior tile of a split-K slice (the common case): one running pointer per 32 x 32 accumulator, no bounds checks
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float* d = dst + (int64_t)(m0 + (wm * TM + i) * 32 + 4 * h) * ldd + n0 + (wn * TN + j) * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          *d = acc[i][j][e];
          d += ((e & 3) == 3) ? 5 * ldd : ldd;               // rows (e & 3) + 8 (e >> 2) + 4 h
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + r;
        float old[16];
        if (how == 1) {                                       // all 16 reads in flight before the first add
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = min(m0 + (wm * TM + i) * 32 + mma32_row(e, h), g.M - 1);
            old[e] = dst[m * ldd + min(n, g.N - 1)];
          }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + (wm * TM + i) * 32 + mma32_row(e, h);
          if (m < g.M && n < g.N) {
            float* d = dst + m * ldd + n;
            if (how == 0) *d = acc[i][j][e];
            else if (how == 1) *d = old[e] + acc[i][j][e];
            else atomicAdd(d, acc[i][j][e]);
          }
        }
      }
  }
  if (g.b_ones) {                                              // fold the thread groups through LDS: one atomic per column
    __syncthreads();
    float* red = reinterpret_cast<float*>(&As[0][0]);
    red[tid] = bsum;
    __syncthreads();
    if (tid < BM && m0 + tid < g.M) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 256 / BM; ++q) t += red[q * BM + tid];
      atomicAdd(g.C2 + m0 + tid, t);
    }
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename AT, typename BT, bool B_TR, bool A_GELU, int WM, int WN, int TM, int TN, int EPI, bool SPLIT = false, int BKT = 64>
int launch_nt_epi(const WmGemm& g, bool vec_ok, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = BKT;
  constexpr size_t bsz = B_TR ? BK * (BN + 32) : BN * (BK + 8);
  constexpr size_t stage = (SPLIT ? 2 : 1) * NT_NBUF * 2 * (BM * (BK + 8) + bsz), ctile = sizeof(float) * TM * 32 * (BN + 4);
  constexpr size_t smem = stage > ctile ? stage : ctile;
  static bool attr = false;
  if (!attr) {
    if (smem > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<AT, BT, B_TR, A_GELU, WM, WN, TM, TN, EPI, SPLIT, BKT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
  wm_gemm_family_tls = WM_FAM_NT;
  hipLaunchKernelGGL((gemm_nt_kernel<AT, BT, B_TR, A_GELU, WM, WN, TM, TN, EPI, SPLIT, BKT>), dim3(8 * ((tiles_m + 7) / 8) * tiles_n), dim3(256),
                     smem, st, g, vec_ok, tiles_n, tiles_m);
  WM_CHECK_LAUNCH();
  return WM_OK;
}

// split-precision launches (fp32 operands and outputs): the general epilogue only
template <bool B_TR>
int dispatch_nt_split(const WmGemm& g, hipStream_t st) {
  const bool vec_ok = al16(g.C) && !(g.ldc & 7) && (!g.bias || al16(g.bias)) && (!g.res || (al16(g.res) && !(g.ldr & 3))) &&
                      (!g.gaux || (al16(g.gaux) && !(g.ldg & 7)));
  const int64_t t128 = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128);
  const int64_t t12864 = (int64_t)((g.M + 127) / 128) * ((g.N + 63) / 64);
  if (g.N <= 32) return launch_nt_epi<float, float, B_TR, false, 4, 1, 2, 1, 3, true>(g, vec_ok, st);
  if (g.N > 64 && t128 >= 256) return launch_nt_epi<float, float, B_TR, false, 2, 2, 2, 2, 3, true>(g, vec_ok, st);
  if (t12864 >= 256 || g.M >= 32768) return launch_nt_epi<float, float, B_TR, false, 2, 2, 2, 1, 3, true>(g, vec_ok, st);
  return launch_nt_epi<float, float, B_TR, false, 2, 2, 1, 1, 3, true>(g, vec_ok, st);
}

// picks the epilogue specialisation from the operands that are present (3 = general code)
template <typename AT, typename BT, bool B_TR, bool A_GELU, int WM, int WN, int TM, int TN, int BKT = 64>
int launch_nt(const WmGemm& g, hipStream_t st) {
  constexpr int BM = WM * TM * 32;
  const bool vec_ok = al16(g.C) && !(g.ldc & 7) && (!g.bias || al16(g.bias)) && (!g.res || (al16(g.res) && !(g.ldr & 3))) &&
                      (!g.gaux || (al16(g.gaux) && !(g.ldg & 7)));
  const bool sc_ok = !g.rowscale || (g.rows_per_scale % BM) == 0;
  if (g.ln_out) {                                              // fused LayerNorm: whole rows in the tile, interior path only
    constexpr int BN = WN * TN * 32;
    if constexpr (BN <= 128 && !B_TR && !A_GELU && sizeof(AT) == 2) {
      if (!(vec_ok && sc_ok) || g.c_bf16 || !g.res || g.gaux || g.N != BN || (g.M % BM) || !al16(g.ln_gamma) || !al16(g.ln_beta) ||
          !al16(g.ln_out) || (g.ln_mod && (!al16(g.ln_mod) || g.ln_hw <= 0 || ((int64_t)g.ln_hw * g.ln_hw) % BM)))
        return WM_EINVAL;
      return launch_nt_epi<AT, BT, B_TR, A_GELU, WM, WN, TM, TN, 4, false, BKT>(g, vec_ok, st);
    } else {
      return WM_EINVAL;
    }
  }
  if (vec_ok && sc_ok) {
    if (g.c_bf16 && !g.gaux && !g.res) return launch_nt_epi<AT, BT, B_TR, A_GELU, WM, WN, TM, TN, 0, false, BKT>(g, vec_ok, st);
    if (!g.c_bf16 && g.res && !g.gaux) return launch_nt_epi<AT, BT, B_TR, A_GELU, WM, WN, TM, TN, 1, false, BKT>(g, vec_ok, st);
    if (g.c_bf16 && g.gaux && g.gaux_raw && g.gaux_bf16 && !g.res && !g.bias)
      return launch_nt_epi<AT, BT, B_TR, A_GELU, WM, WN, TM, TN, 2, false, BKT>(g, vec_ok, st);
  }
  return launch_nt_epi<AT, BT, B_TR, A_GELU, WM, WN, TM, TN, 3, false, BKT>(g, vec_ok, st);
}

template <typename AT, typename BT, bool B_TR, bool A_GELU>
int dispatch_nt(const WmGemm& g, hipStream_t st) {
  const int64_t t128 = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128);
  const int64_t t12864 = (int64_t)((g.M + 127) / 128) * ((g.N + 63) / 64);
  if (g.ln_out) {                                              // the tile must hold whole rows: width = N
    if (g.N == 32) return launch_nt<AT, BT, B_TR, A_GELU, 4, 1, 2, 1>(g, st);
    if (g.N == 64) return launch_nt<AT, BT, B_TR, A_GELU, 2, 2, 2, 1>(g, st);
    if (g.N == 128) return launch_nt<AT, BT, B_TR, A_GELU, 2, 2, 2, 2>(g, st);
    return WM_EINVAL;
  }
  if (g.N <= 32) return launch_nt<AT, BT, B_TR, A_GELU, 4, 1, 2, 1>(g, st);          // 256 x 32
  // N = 192 (qkv at C = 64) leaves the second 128-wide column tile half empty: three full 64-wide tiles instead
  // (measured at B = 96: 301.9 -> 218.1 us at 1.57 M rows, 77.4 -> 55.6 us at 393 k rows)
  if (g.N == 192 && t12864 >= 384) return launch_nt<AT, BT, B_TR, A_GELU, 2, 2, 2, 1>(g, st);
  if (g.N > 64 && t128 >= 384) return launch_nt<AT, BT, B_TR, A_GELU, 2, 2, 2, 2>(g, st);   // 128 x 128
  if (t12864 >= 384 || g.M >= 32768) return launch_nt<AT, BT, B_TR, A_GELU, 2, 2, 2, 1>(g, st);  // 128 x 64
  return launch_nt<AT, BT, B_TR, A_GELU, 2, 2, 1, 1, NT_SMALL_BK>(g, st);              // 64 x 64, 128 deep
}

// C[m][n] += sum over z in this block's segment of part[z][m][n]; thread = 4 consecutive n (N % 4 == 0)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, int64_t ldc,
                                                            int M, int N, int ksplit, int per) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x, mn = (int64_t)M * N;
  if (i4 * 4 >= mn) return;
  const int z0 = blockIdx.y * per, z1 = min(ksplit, z0 + per);
  const float* p = part + z0 * mn + i4 * 4;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  int z = z0;
  for (; z + 4 <= z1; z += 4, p += 4 * mn) {                  // four slices in flight
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + mn);
    const float4 c = *reinterpret_cast<const float4*>(p + 2 * mn), d = *reinterpret_cast<const float4*>(p + 3 * mn);
    s0.x += a.x + c.x; s0.y += a.y + c.y; s0.z += a.z + c.z; s0.w += a.w + c.w;
    s1.x += b.x + d.x; s1.y += b.y + d.y; s1.z += b.z + d.z; s1.w += b.w + d.w;
  }
  for (; z < z1; ++z, p += mn) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
  }
  const int m = (int)(i4 * 4 / N), n = (int)(i4 * 4 - (int64_t)m * N);
  float* c = C + m * ldc + n;
  if (gridDim.y == 1) {
    c[0] += s0.x + s1.x; c[1] += s0.y + s1.y; c[2] += s0.z + s1.z; c[3] += s0.w + s1.w;
  } else {
    atomicAdd(c + 0, s0.x + s1.x); atomicAdd(c + 1, s0.y + s1.y); atomicAdd(c + 2, s0.z + s1.z); atomicAdd(c + 3, s0.w + s1.w);
  }
}

template <typename AT, typename BT, bool B_GELU, int TM, int TN, bool SPLIT = false>
int launch_tn(const WmGemm& g, hipStream_t st) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN, tiles = tiles_m * tiles_n;
  const int64_t mn = (int64_t)g.M * g.N;
  const bool use_part = g.scratch && !(g.N & 3);
  static const int tn_target = [] { const char* v = getenv("WM_TN_TARGET"); const int t = v ? atoi(v) : 0; return t > 0 ? t : TN_TARGET; }();
  int ksplit = g.ksplit > 0 ? g.ksplit : ((use_part ? tn_target : 768) + tiles - 1) / tiles;
  // a slice shorter than 512 tokens writes and re-reads more partial sums than it reads operands (128 x 128 x 8 B against
  // 256 rows x 2 B per token): measured at B = 24 (deep levels: 1.5 k - 6 k tokens), 29.15 -> 28.41 ms per step
  const int mk = use_part ? 512 : 256;
  const int maxs = (g.K + mk - 1) / mk;
  if (ksplit > maxs) ksplit = maxs;
  if (use_part && (int64_t)ksplit * mn > g.scratch_floats) ksplit = (int)(g.scratch_floats / mn);
  if (ksplit < 1) ksplit = 1;
  int klen = (g.K + ksplit - 1) / ksplit;
  klen = (klen + TBK - 1) / TBK * TBK;
  ksplit = (g.K + klen - 1) / klen;
  float* part = (use_part && ksplit > 1) ? g.scratch : nullptr;
  wm_gemm_family_tls = WM_FAM_TN;
  hipLaunchKernelGGL((gemm_tn_kernel<AT, BT, B_GELU, TM, TN, SPLIT>), dim3(8 * ((ksplit + 7) / 8) * tiles), dim3(256), 0, st, g, klen,
                     tiles_n, tiles_m, ksplit, part);
  WM_CHECK_LAUNCH();
  if (g.defer) {                                              // the caller sums the slices later (wm_reduce_jobs)
    WmReduceJob jb{};
    if (part) {
      jb.src = part; jb.pstride = mn; jb.src_ld = g.N; jb.count = ksplit; jb.rows = g.M; jb.cols = g.N;
      jb.dst = static_cast<float*>(g.C); jb.dst_rs = g.ldc; jb.dst_cs = 1;
    }
    *g.defer = jb;
    return WM_OK;
  }
  if (part) {
    const int64_t blocks = (mn / 4 + 255) / 256;
    int segs = (int)(256 / blocks);
    if (segs > ksplit / 4) segs = ksplit / 4;
    if (segs < 1) segs = 1;
    const int per = (ksplit + segs - 1) / segs;
    segs = (ksplit + per - 1) / per;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks, segs), dim3(256), 0, st, part, static_cast<float*>(g.C), g.ldc,
                       g.M, g.N, ksplit, per);
    WM_CHECK_LAUNCH();
  }
  return WM_OK;
}

template <typename AT, typename BT, bool B_GELU>
int dispatch_tn(const WmGemm& g, hipStream_t st) {
  const int64_t t128 = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128);
  if (g.M > 64 && g.N > 64 && (t128 >= 8 || !g.scratch)) return launch_tn<AT, BT, B_GELU, 2, 2>(g, st);
  if (g.M > 64 && (g.N <= 64 || t128 >= 2)) return launch_tn<AT, BT, B_GELU, 2, 1>(g, st);
  return launch_tn<AT, BT, B_GELU, 1, 1>(g, st);
}

}  // namespace

// Returns -1 when the launch does not fit the fast kernels (caller falls back to the generic kernel of gemm.hip).
// compute_mode "split" (mode 2): fp32 storage everywhere, products in split bf16 on the fast kernels.  -1 = shape not covered
// (the caller then runs the exact fp32-MFMA kernel of gemm.hip, which costs time, not accuracy).
int wm_gemm_fast_split(const WmGemm& g, hipStream_t st) {
  if (g.defer) *g.defer = WmReduceJob{};
  if (g.mode != 2 || g.a_bf16 || g.b_bf16 || g.c_bf16 || g.gaux_bf16 || g.a_op || g.b_op || g.ln_out) return -1;
  if (!g.a_kc && !g.b_kc && g.atomic_out) {                                     // weight gradient
    if ((g.M % 4) || (g.N % 4) || (g.lda % 4) || (g.ldb % 4) || g.M < 4 || g.N < 4) return -1;
    if (g.kscale && (g.k_per_scale <= 0 || g.k_per_scale % TBK)) return -1;
    if (g.M > 64) return launch_tn<float, float, false, 2, 1, true>(g, st);
    return launch_tn<float, float, false, 1, 1, true>(g, st);
  }
  if (g.a_kc && !g.atomic_out && !g.b_ones && !g.kscale && (g.K % 8) == 0 && g.K >= 8 && (g.lda % 4) == 0) {
    if (g.b_kc) {
      if (g.ldb % 4) return -1;
      return dispatch_nt_split<false>(g, st);
    }
    if ((g.N % 8) || (g.ldb % 4) || g.N < 8) return -1;
    return dispatch_nt_split<true>(g, st);
  }
  return -1;
}

int wm_gemm_fast_bf16(const WmGemm& g, hipStream_t st) {
  if (g.defer) *g.defer = WmReduceJob{};                       // count 0 unless a split-K launch fills it in
  if (g.mode != 1) return -1;
  const int ea = g.a_bf16 ? 8 : 4, eb = g.b_bf16 ? 8 : 4;
  if (g.a_grid_n || g.a_jump_k) {                              // grid addressing of A: K-contiguous bf16 A, whole 64 / 128-deep tiles on both sides of the jump
    if (!g.a_kc || !g.a_bf16 || g.a_op || g.atomic_out || (g.K % 128) || (g.a_jump_k % 128) || g.a_jump_k >= g.K || (g.a_jump % 8)) return -1;
    if (g.a_grid_n && (g.M % (g.a_grid_n * g.a_grid_n))) return -1;
  }
  if (g.b_seg_cols) {                                          // ... of B: weight gradient only, tiles never straddle a column block
    if (g.a_kc || g.b_kc || !g.atomic_out || !g.b_bf16 || (g.b_seg_cols % 128) || g.N > 4 * g.b_seg_cols) return -1;
  }
  if (!g.a_kc && !g.b_kc && g.atomic_out && !g.a_op && !g.c_bf16) {          // weight gradient
    if ((g.M % ea) || (g.N % eb) || (g.lda % ea) || (g.ldb % eb) || g.M < ea || g.N < eb) return -1;
    if (g.kscale && (g.k_per_scale <= 0 || g.k_per_scale % TBK)) return -1;    // one scale per k tile (TrStage::load_scale)
    if (g.b_op) {
      if (g.a_bf16 && g.b_bf16) return dispatch_tn<__bf16, __bf16, true>(g, st);
      if (g.a_bf16) return dispatch_tn<__bf16, float, true>(g, st);
      if (g.b_bf16) return dispatch_tn<float, __bf16, true>(g, st);
      return dispatch_tn<float, float, true>(g, st);
    }
    if (g.a_bf16 && g.b_bf16) return dispatch_tn<__bf16, __bf16, false>(g, st);
    if (g.a_bf16) return dispatch_tn<__bf16, float, false>(g, st);
    if (g.b_bf16) return dispatch_tn<float, __bf16, false>(g, st);
    return dispatch_tn<float, float, false>(g, st);
  }
  if (g.a_kc && !g.atomic_out && !g.b_ones && !g.b_op && !g.kscale && g.b_bf16 && (g.K % 8) == 0 && g.K >= 8 &&
      (g.lda % ea) == 0) {
    if (g.b_kc) {                                                             // forward: B = [N, K] weight copy
      if (g.ldb % 8) return -1;
      if (g.a_op) return g.a_bf16 ? dispatch_nt<__bf16, __bf16, false, true>(g, st) : dispatch_nt<float, __bf16, false, true>(g, st);
      return g.a_bf16 ? dispatch_nt<__bf16, __bf16, false, false>(g, st) : dispatch_nt<float, __bf16, false, false>(g, st);
    }
    if (g.a_op || (g.N % 8) || (g.ldb % 8) || g.N < 8) return -1;             // dX: B = [K, N] weight copy
    return g.a_bf16 ? dispatch_nt<__bf16, __bf16, true, false>(g, st) : dispatch_nt<float, __bf16, true, false>(g, st);
  }
  return -1;
}
