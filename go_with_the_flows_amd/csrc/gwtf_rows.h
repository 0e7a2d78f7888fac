// gwtf_rows.h -- building blocks of the per-SHAPE kernels that put the (<= 128) latent rows on the exact-fp32 MFMA's M axis with
// operands straight from L2 (gwtf_film_train.hip: FiLM heads; gwtf_prior.hip: global prior flow): clamped / vector operand loads,
// column totals over the rows in the accumulators' layout.
// v_mfma_f32_16x16x4_f32 operand map used throughout: A[i][k]: lane (i = lane & 15, q = lane >> 4) supplies k = 4 q + t of a 16-k step
// in MFMA t = 0..3 (both operands use the same order, which is all a contraction needs); D[4 q + r][lane & 15] in register r.
#pragma once
#include <hip/hip_runtime.h>

namespace gwtf_rows {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float swish(float h) { return h / (1.0f + expf(-h)); }

// four k values k .. k+3 of one operand row, clamped to the row's last element (the other operand is zero there)
__device__ __forceinline__ f32x4 load4(const float* __restrict__ rowp, int k, int K) {
  f32x4 v;
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = rowp[min(k + t, K - 1)];
  return v;
}
// the same where k is the contiguous index: ONE 16-byte load (global_load_dwordx4 needs dword alignment only -- the arena's records
// sit at odd float offsets) when the four values lie inside the row, the clamped scalar form at the row's end
typedef f32x4 __attribute__((aligned(4))) f32x4_u;
__device__ __forceinline__ f32x4 load4v(const float* __restrict__ rowp, int k, int K) {
  if (k + 3 < K) return *reinterpret_cast<const f32x4_u*>(rowp + k);
  return load4(rowp, k, K);
}
__device__ __forceinline__ f32x4 load4s(const float* __restrict__ colp, int k, int K, size_t stride) {   // k runs along a stride
  f32x4 v;
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = colp[(size_t)min(k + t, K - 1) * stride];
  return v;
}
__device__ __forceinline__ f32x4 zero_from(f32x4 v, int k, int K) {
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = k + t < K ? v[t] : 0.f;
  return v;
}

// totals of NT per-lane column partials over the 4 lane quarters and the 4 waves; every lane ends with the totals of its columns
template <int NT>
__device__ __forceinline__ void column_totals(float (&v)[NT], float (*red)[16 * NT], int wave, int c16, int q) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float s = v[nt];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (q == 0) red[wave][16 * nt + c16] = s;
  }
  __syncthreads();
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int j = 16 * nt + c16;
    v[nt] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
  }
  __syncthreads();
}

}  // namespace gwtf_rows
