// Window geometry, LDS tile layout and MFMA operand fetches shared by the window-attention kernels on the matrix cores
// (attn_fast.hip: bf16 storage; attn_split.hip: fp32 storage, split-precision products).
#pragma once
#include "common.h"
#include "../../include/wm_kernels.h"

namespace attnk {

constexpr int LDK = 40;                                       // LDS row stride (bf16): 80 B, 16-B aligned chunks
constexpr float kScale = 0.17677669529663687f;                // head_dim ** -0.5
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;    // one 16-byte chunk (native vector: stays in registers)

struct Geom {
  int hw, shift, nw, C, heads;
  __device__ __forceinline__ int wrap(int v) const { return v >= hw ? v - hw : v; }
  __device__ __forceinline__ int token_row(int b, int wy, int wx, int p) const {
    return (b * hw + wrap(wy * 8 + (p >> 3) + shift)) * hw + wrap(wx * 8 + (p & 7) + shift);
  }
  __device__ __forceinline__ int region1(int ii) const { return ii < hw - 8 ? 0 : (ii < hw - shift ? 1 : 2); }
  __device__ __forceinline__ int region(int wy, int wx, int p) const {          // 0 everywhere when there is no shift
    return shift ? 3 * region1(wy * 8 + (p >> 3)) + region1(wx * 8 + (p & 7)) : 0;
  }
};

// as-it-lies [k rows][32 columns] LDS image -> MFMA operand (lane l: column l & 31).
// perm == false: k slot (h, j) = row0 + 8h + j;  perm == true: row0 + 8(j>>2) + 4h + (j&3)  (accumulator row order)
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int row0, int lane, bool perm) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, h = g >> 1;
  const __bf16* a = tile + (row0 + (perm ? 4 * h : 8 * h) + q) * LDK + 16 * (g & 1) + 4 * p;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + (perm ? 8 : 4) * LDK));
  bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return f;
}
__device__ __forceinline__ bf16x8 row_frag(const __bf16* tile, int row, int ks, int h) {
  return *reinterpret_cast<const bf16x8*>(tile + row * LDK + 16 * ks + 8 * h);
}
// blocks are dealt round-robin to the 8 XCDs: renumber so that consecutive virtual ids (the heads of one window, which
// share 128-byte lines of q|k|v) land on the same XCD's L2
__device__ __forceinline__ int virtual_block(int id, int n) { return (n & 7) ? id : (id & 7) * (n >> 3) + (id >> 3); }

__device__ __forceinline__ void win_pos(int w, int nw, int& b, int& wy, int& wx) {
  b = w / (nw * nw);
  const int wr = w - b * nw * nw;
  wy = wr / nw;
  wx = wr - wy * nw;
}

// persistent grid: `waves_per_cu` resident single-wave workgroups on each of the 256 CUs, a whole number of chunks per head
inline void grid_for(int nwin, int heads, int waves_per_cu, int& blocks, int& nchunk) {
  nchunk = (256 * waves_per_cu) / heads;
  if (nchunk < 1) nchunk = 1;
  if (nchunk > nwin) nchunk = nwin;
  blocks = nchunk * heads;
}

}  // namespace attnk

int wm_attn_bias_reduce(const float* scratch, float* dtable, int heads, int nchunk, hipStream_t st);
