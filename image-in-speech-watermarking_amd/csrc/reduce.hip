// One launch for all partial-sum reductions of a group of kernels (wm_reduce_jobs).
//
// Parameter gradients that many workgroups contribute to are stored as per-workgroup partial sums in a scratch buffer
// (split-K slices of the weight-gradient GEMMs, LayerNorm gamma / beta, depthwise-conv weights, the relative-position
// bias table) and summed by a small kernel.  Each of those kernels costs 4-7 us of launch + dependent-load latency
// however little it adds up, and a LeWin block's backward produces eight of them: the producers can instead DEFER
// their reduction (they fill a WmReduceJob and leave the partials in their own scratch slice) and one launch of this
// kernel at the end of the block performs all of them.
#include "common.h"
#include "../../include/wm_kernels.h"

namespace {

constexpr int MAXJ = 16;
struct JobTable {
  WmReduceJob job[MAXJ];
  int first_block[MAXJ + 1];      // prefix sums of blocks per job
  int chunks[MAXJ];               // 256-output chunks of the job
  int per[MAXJ];                  // partials per segment
  int njobs;
};

// out(r, c) += sum_p src[p * pstride + r * src_ld + c]; block = (256 outputs of one job) x (one segment of its partials)
__global__ __launch_bounds__(256) void reduce_jobs_kernel(const JobTable t) {
  int j = 0;
  while (j + 1 < t.njobs && (int)blockIdx.x >= t.first_block[j + 1]) ++j;
  const WmReduceJob jb = t.job[j];
  const int rel = blockIdx.x - t.first_block[j], chunk = rel % t.chunks[j], seg = rel / t.chunks[j];
  const int64_t i = (int64_t)chunk * 256 + threadIdx.x, n = (int64_t)jb.rows * jb.cols;
  if (i >= n) return;
  const int r = (int)(i / jb.cols), c = (int)(i - (int64_t)r * jb.cols);
  const int p0 = seg * t.per[j], p1 = min(jb.count, p0 + t.per[j]);
  const float* p = jb.src + (int64_t)p0 * jb.pstride + (int64_t)r * jb.src_ld + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int q = p0;
  // a thread's partials are `pstride` apart: every load is its own memory round trip, so the launch lasts (partials per thread /
  // loads in flight) round trips.  Sixteen in flight: the 64-slice weight gradients of the wide layers took 16 trips with four.
  for (; q + 16 <= p1; q += 16, p += 16 * jb.pstride) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[u * jb.pstride];
#pragma unroll
    for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
  }
  for (; q + 4 <= p1; q += 4, p += 4 * jb.pstride) {          // four partials in flight
    s0 += p[0];
    s1 += p[jb.pstride];
    s2 += p[2 * jb.pstride];
    s3 += p[3 * jb.pstride];
  }
  for (; q < p1; ++q, p += jb.pstride) s0 += p[0];
  float* d = jb.dst + (int64_t)r * jb.dst_rs + (int64_t)c * jb.dst_cs;
  const float s = (s0 + s1) + (s2 + s3);
  if (t.per[j] >= jb.count) *d += s;                          // a single segment: this thread is the only writer
  else atomicAdd(d, s);
}

}  // namespace

extern "C" int wm_reduce_jobs(const WmReduceJob* jobs, int32_t njobs, void* stream) {
  if (!jobs || njobs < 0 || njobs > MAXJ) return WM_EINVAL;
  JobTable t;
  t.njobs = 0;
  int blocks = 0;
  for (int k = 0; k < njobs; ++k) {
    const WmReduceJob& jb = jobs[k];
    if (jb.count <= 0) continue;                              // producer had nothing to defer (e.g. a single K split)
    if (!jb.src || !jb.dst || jb.rows <= 0 || jb.cols <= 0) return WM_EINVAL;
    const int64_t n = (int64_t)jb.rows * jb.cols;
    const int chunks = (int)((n + 255) / 256);
    int segs = 256 / chunks;                                  // small outputs: spread their partials over more workgroups
    if (segs > jb.count / 4) segs = jb.count / 4;
    if (segs > 32) segs = 32;
    if (segs < 1) segs = 1;
    const int per = (jb.count + segs - 1) / segs;
    segs = (jb.count + per - 1) / per;
    const int j = t.njobs++;
    t.job[j] = jb;
    t.first_block[j] = blocks;
    t.chunks[j] = chunks;
    t.per[j] = per;
    blocks += chunks * segs;
  }
  if (t.njobs == 0) return WM_OK;
  t.first_block[t.njobs] = blocks;
  hipLaunchKernelGGL(reduce_jobs_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), t);
  WM_CHECK_LAUNCH();
  return WM_OK;
}
