// Diagnostic: (1) does v_mfma_f32_16x16x32_f16 honour f16 subnormal inputs? (2) its issue rate; (3) v_pk_fma_f32 rate;
// (4) v_cvt_pkrtz_f16_f32 subnormal output.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void denorm(float* out) {
  const int l = threadIdx.x;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0; b[j] = (_Float16)0; }
  // k index = 8*(l>>4)+j ; set A[i][k=0] = 1 for all rows i, B[k=0][col] = subnormal 2^-20
  if ((l >> 4) == 0) { a[0] = (_Float16)1.0f; b[0] = (_Float16)9.5367431640625e-07f; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (l == 0) out[0] = c[0];
  // pkrtz producing subnormal
  f16x2 r = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(3.0e-6f, 1.0f));
  if (l == 0) { out[1] = (float)r[0]; out[2] = (float)r[1]; }
}

template <int ROLE>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
  float x = threadIdx.x * 1e-3f;
  if (ROLE == 0) {
    f16x8 u, v; for (int j = 0; j < 8; ++j) { u[j] = (_Float16)(x + j); v[j] = (_Float16)(1 + j); }
    f32x4 a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, v, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, v, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, v, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(u, v, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (ROLE == 1) {
    f32x2 c0 = {x, x + 1}, c1 = {x + 2, x + 3}, c2 = {x + 4, x + 5}, c3 = {x + 6, x + 7};
    const f32x2 y = {1.0001f, 0.9999f}, z = {x, -x};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c0 = __builtin_elementwise_fma(c0, y, z); c1 = __builtin_elementwise_fma(c1, y, z);
        c2 = __builtin_elementwise_fma(c2, y, z); c3 = __builtin_elementwise_fma(c3, y, z);
      }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[0] + c3[1];
  } else {
    float c0 = x, c1 = x + 1, c2 = x + 2, c3 = x + 3, c4 = x + 4, c5 = x + 5, c6 = x + 6, c7 = x + 7; const float y = 1.0001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        c0 = __builtin_fmaf(c0, y, x); c1 = __builtin_fmaf(c1, y, x); c2 = __builtin_fmaf(c2, y, x); c3 = __builtin_fmaf(c3, y, x);
        c4 = __builtin_fmaf(c4, y, x); c5 = __builtin_fmaf(c5, y, x); c6 = __builtin_fmaf(c6, y, x); c7 = __builtin_fmaf(c7, y, x);
      }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  }
}
template <int R> float run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((rate<R>), dim3(256), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((rate<R>), dim3(256), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5 * 1000;
}
int main() {
  float* d; hipMalloc(&d, 256 * 256 * 4);
  hipLaunchKernelGGL(denorm, dim3(1), dim3(64), 0, 0, d);
  float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
  printf("mfma f16 with subnormal B=2^-20: got %.6e (expect 9.536743e-07; 0 means flushed)\n", h[0]);
  printf("cvt_pkrtz(3.0e-6, 1.0) = %.6e, %.3f (expect ~2.98e-06 subnormal, 1.0)\n", h[1], h[2]);
  const int it = 20000;
  float t0 = run<0>(d, it), t1 = run<1>(d, it), t2 = run<2>(d, it);
  printf("one wave/SIMD: 4 mfma_16x16x32_f16 per iter: %.1f us -> %.1f ns per MFMA\n", t0, t0 * 1000 / (it * 4.0));
  printf("one wave/SIMD: 16 v_pk_fma_f32 per iter:     %.1f us -> %.2f ns per instr\n", t1, t1 * 1000 / (it * 16.0));
  printf("one wave/SIMD: 16 v_fma_f32 per iter:        %.1f us -> %.2f ns per instr\n", t2, t2 * 1000 / (it * 16.0));
  return 0;
}
