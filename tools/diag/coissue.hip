// Diagnostic: do v_mfma_f32_16x16x4_f32 and f32 VALU overlap when issued by two waves of one SIMD?
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run role A, waves 4-7 role B.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int ROLE_A, int ROLE_B>  // 0 none, 1 mfma f32 16x16x4, 2 valu fma, 3 mfma f32 32x32x2, 4 mfma bf16 16x16x32
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? ROLE_A : ROLE_B;
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  if (role == 1) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (role == 3) {
    f32x16 a0 = {0}, a1 = {0};
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1];
  } else if (role == 4) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    bf16x8 u = {1, 2, 3, 4, 5, 6, 7, 8}, v = {8, 7, 6, 5, 4, 3, 2, 1};
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u, v, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u, v, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u, v, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u, v, a3, 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (role == 2) {
    float c0 = x, c1 = x + 1, c2 = x + 2, c3 = x + 3, c4 = x + 4, c5 = x + 5, c6 = x + 6, c7 = x + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // 32 independent-ish FMAs per iteration
        c0 = __builtin_fmaf(c0, y, x); c1 = __builtin_fmaf(c1, y, x); c2 = __builtin_fmaf(c2, y, x); c3 = __builtin_fmaf(c3, y, x);
        c4 = __builtin_fmaf(c4, y, x); c5 = __builtin_fmaf(c5, y, x); c6 = __builtin_fmaf(c6, y, x); c7 = __builtin_fmaf(c7, y, x);
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  }
}

template <int A, int B>
float run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5 * 1000;
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  const int it = 20000;
  printf("per launch, us (iters=%d): 4 MFMA16 or 2 MFMA32 or 32 VALU fma per iteration\n", it);
  printf("mfma16 alone        %8.1f\n", run<1, 0>(d, it));
  printf("valu alone          %8.1f\n", run<0, 2>(d, it));
  printf("mfma16 + valu       %8.1f\n", run<1, 2>(d, it));
  printf("mfma16 + mfma16     %8.1f\n", run<1, 1>(d, it));
  printf("valu + valu         %8.1f\n", run<2, 2>(d, it));
  printf("mfma32x32x2 alone   %8.1f\n", run<3, 0>(d, it));
  printf("mfma32x32x2 + valu  %8.1f\n", run<3, 2>(d, it));
  printf("bf16 mfma alone     %8.1f\n", run<4, 0>(d, it));
  printf("bf16 mfma + valu    %8.1f\n", run<4, 2>(d, it));
  return 0;
}
