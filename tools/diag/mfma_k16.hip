// Does the legacy v_mfma_f32_16x16x16_f16 cost half of v_mfma_f32_16x16x32_f16 on gfx950?  (It decides whether a short
// last k-step -- f = 33..40 has 1-2 valid k-slots per lane in its second k-step -- is worth issuing at K = 16.)
// build: hipcc -O3 --offload-arch=gfx950 mfma_k16.hip -o /tmp/mfma_k16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 a8, b8;
  f16x4 a4, b4;
  for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(threadIdx.x * 0.001f + e); b8[e] = (_Float16)(e * 0.5f); }
  for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
      if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
      if (MODE == 2) {   // K=16 with two VALU fillers per gap
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
        f[i] = fmaf(f[i], 1.0001f, 0.5f);
        f[(i + 4) & 7] = fmaf(f[(i + 4) & 7], 0.9999f, 0.25f);
      }
      if (MODE == 3) {   // K=32 with two VALU fillers per gap
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
        f[i] = fmaf(f[i], 1.0001f, 0.5f);
        f[(i + 4) & 7] = fmaf(f[(i + 4) & 7], 0.9999f, 0.25f);
      }
      if (MODE == 4) {   // K=16 with one VALU filler per gap
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
        f[i] = fmaf(f[i], 1.0001f, 0.5f);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out) {
  const int iters = 400000, grid = 256;   // one workgroup of 4 waves per CU: one wave per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 200000);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %8.3f ms  %6.2f ns per MFMA slot (at 2.4 GHz: %5.1f cycles)\n", name, ms, ms * 1e6 / (iters * 8.0), ms * 1e6 / (iters * 8.0) * 2.4);
}

int main() {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  run<0>("16x16x32_f16", out);
  run<1>("16x16x16_f16", out);
  run<3>("16x16x32_f16 + 2 valu", out);
  run<2>("16x16x16_f16 + 2 valu", out);
  run<4>("16x16x16_f16 + 1 valu", out);
  return 0;
}
