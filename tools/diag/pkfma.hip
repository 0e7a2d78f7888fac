// Issue cost of v_pk_fma_f32 against two v_fma_f32, alone and beside MFMAs (one wave per SIMD).
// build: hipcc -w -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form pkfma.hip -o pkfma.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: 16 v_fma  1: 8 v_pk_fma  2: 4 MFMA + 16 v_fma  3: 4 MFMA + 8 v_pk_fma  4: 4 MFMA  5: 16 v_max  6: 4 MFMA + 16 v_max
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 a8, b8;
  for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(threadIdx.x * 0.001f + e); b8[e] = (_Float16)(e * 0.5f); }
  float f[16];
  for (int i = 0; i < 16; ++i) f[i] = threadIdx.x + i;
  const float c0 = out[0], c1 = out[1];
  const f32x2 cc = {c0, c1}, dd = {c1, c0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 6) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
      if (MODE == 0 || MODE == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[4 * i + j]) : "v"(c0), "v"(c1));
      }
      if (MODE == 1 || MODE == 3) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x2 v = {f[4 * i + 2 * j], f[4 * i + 2 * j + 1]};
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(cc), "v"(dd));
          f[4 * i + 2 * j] = v[0]; f[4 * i + 2 * j + 1] = v[1];
        }
      }
      if (MODE == 5 || MODE == 6) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[4 * i + j]) : "v"(c0));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x + 2] = s;
}

template <int MODE>
void run(const char* name, float* out) {
  const int iters = 200000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %8.3f ms  %7.2f ns per iteration (at 2.4 GHz: %6.1f cycles)\n", name, ms, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
}

int main() {
  float* out; (void)hipMalloc(&out, (256 * 256 + 2) * 4); (void)hipMemset(out, 0, (256 * 256 + 2) * 4);
  run<0>("16 v_fma", out);
  run<1>("8 v_pk_fma", out);
  run<5>("16 v_max", out);
  run<4>("4 mfma", out);
  run<2>("4 mfma + 16 v_fma", out);
  run<3>("4 mfma + 8 v_pk_fma", out);
  run<6>("4 mfma + 16 v_max", out);
  return 0;
}
