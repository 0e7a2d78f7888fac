#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
  auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[128 + threadIdx.x] = s[0]; out[192 + threadIdx.x] = s[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
  for (int t = 0; t < 4; ++t) { printf("%s:", names[t]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[t * 64 + i]); printf("\n"); }
  return 0;
}
