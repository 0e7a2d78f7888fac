// quarter_reduce<4>: shuffle version against the lane-swap version (gfx950 v_permlane{32,16}_swap), 64 lanes, 4 inputs per lane
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ float qr_old(const float (&o)[4], int q) {
  const bool hi = q >= 2, odd = q & 1;
  const float s0 = hi ? o[0] : o[2], s1 = hi ? o[1] : o[3];
  const float t0 = (hi ? o[2] : o[0]) + __shfl_xor(s0, 32);
  const float t1 = (hi ? o[3] : o[1]) + __shfl_xor(s1, 32);
  return (odd ? t1 : t0) + __shfl_xor(odd ? t0 : t1, 16);
}
__device__ float qr_new(const float (&o)[4], int q) {
  const auto a = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, o[0]), __builtin_bit_cast(unsigned, o[2]), false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, o[1]), __builtin_bit_cast(unsigned, o[3]), false, false);
  const float s02 = __builtin_bit_cast(float, a[0]) + __builtin_bit_cast(float, a[1]);
  const float s13 = __builtin_bit_cast(float, b[0]) + __builtin_bit_cast(float, b[1]);
  const auto c = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s02), __builtin_bit_cast(unsigned, s13), false, false);
  return __builtin_bit_cast(float, c[0]) + __builtin_bit_cast(float, c[1]);
}
__device__ float qr_asm(const float (&o)[4], int q) {
  float a0 = o[0], a2 = o[2], a1 = o[1], a3 = o[3];
  asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(a2));
  asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a1), "+v"(a3));
  float s02 = a0 + a2, s13 = a1 + a3;
  asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(s02), "+v"(s13));
  return s02 + s13;
}
__global__ void k(const float* x, float* y) {
  const int lane = threadIdx.x, q = lane >> 4;
  float o[4];
  for (int i = 0; i < 4; ++i) o[i] = x[i * 64 + lane];
  y[lane] = qr_old(o, q);
  y[64 + lane] = qr_asm(o, q);
}
int main() {
  float h[256], r[128]; for (int i = 0; i < 256; ++i) h[i] = (float)((i * 37) % 101) * 0.25f;
  float *dx, *dy; hipMalloc(&dx, sizeof h); hipMalloc(&dy, sizeof r); hipMemcpy(dx, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dx, dy); hipMemcpy(r, dy, sizeof r, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) if (r[i] != r[64 + i]) { if (bad < 8) printf("lane %d old %g new %g\n", i, r[i], r[64 + i]); ++bad; }
  printf("mismatching lanes: %d\n", bad); return 0;
}
