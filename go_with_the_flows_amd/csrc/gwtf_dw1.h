// gwtf_dw1.h -- the two stages of the dW1 partial reduction as device functions, shared by the stand-alone kernels
// (gwtf_bwd.hip: gwtf_dw1_reduce) and by the train pipeline's merged tail kernels (gwtf_train.hip), which run them beside the
// sd0 fold / the gradient combine in ONE launch each.
#pragma once
#include <hip/hip_runtime.h>

namespace gwtf_dw1 {

constexpr int kStage = 64;    // stage 1 folds the partials into this many sums (fixed partition -> deterministic)

// PARTIAL RECORD  [2 branches][f columns (h feature i)][RP rows (dacc feature j)], RP = f rounded up to 4: the backward kernel's lane
// holds four CONSECUTIVE rows of one column of a 16 x 16 output tile (MFMA C layout), so a partial leaves the kernel as ONE 16-byte
// store per tile and lane (3 per branch at f = 37) where the row-major compact [f][f] record took a dword store per element (12, each
// with its own bound check): 8 % more bytes at f = 37 (RP = 40), a quarter of the store instructions.
__host__ __device__ inline int rows_padded(int f) { return (f + 3) / 4 * 4; }
__host__ __device__ inline int rec_floats(int f) { return 2 * f * rows_padded(f); }

// stage 1, block (bx = element tile of 256, by = chunk < kStage, component already applied to ws / mid):
// mid[by][e] = sum over the chunk's partials of ws[p][e], e over the whole partial record (coalesced)
__device__ __forceinline__ void fold_block(const float* __restrict__ ws, int n_partials, float* __restrict__ mid, int rec, int bx,
                                           int by, int tid) {
  const int e = bx * 256 + tid;
  if (e >= rec) return;
  const int per = (n_partials + kStage - 1) / kStage;
  const int p0 = by * per, p1 = min(n_partials, p0 + per);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  int p = p0;
  for (; p + 15 < p1; p += 16) {          // 16 loads in flight per thread (a chunk of the 64 x 2048 grid is exactly 16 partials)
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = __builtin_nontemporal_load(&ws[(size_t)(p + u) * rec + e]);
#pragma unroll
    for (int u = 0; u < 16; ++u) s[u & 3] += v[u];
  }
  for (; p + 3 < p1; p += 4)
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += ws[(size_t)(p + u) * rec + e];
  for (; p < p1; ++p) s[0] += ws[(size_t)p * rec + e];
  mid[(size_t)by * rec + e] = (s[0] + s[1]) + (s[2] + s[3]);
}

// stage 2, block bx = 64 consecutive elements of the RECORD (coalesced reads of the kStage sums; the padded rows are skipped) x 4 slices
// of the kStage sums, combined through LDS (`part`: 4 x 64 floats of the caller's LDS; all 256 threads of the block must call);
// grid: (rec_floats(f) + 63) / 64 blocks per component
__device__ __forceinline__ void reduce_block(const float* __restrict__ mid, float* __restrict__ out, int f, size_t branch_stride,
                                             int bx, int tid, float (*part)[64]) {
  const int e = tid & 63, sl = tid >> 6;
  const int t = bx * 64 + e, RP = rows_padded(f);
  const bool in = t < rec_floats(f);
  const int br = in ? t / (f * RP) : 0, i = in ? (t / RP) % f : 0, j = in ? t % RP : 0;      // element (branch, column i, row j)
  const bool on = in && j < f;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (on) {
    const float* src = mid + t;
    const size_t stride = (size_t)rec_floats(f);
#pragma unroll
    for (int c = sl * (kStage / 4); c < (sl + 1) * (kStage / 4); c += 4)
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] += src[(size_t)(c + u) * stride];
  }
  part[sl][e] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (sl == 0 && on) out[(size_t)br * branch_stride + (size_t)j * f + i] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
}

}  // namespace gwtf_dw1
