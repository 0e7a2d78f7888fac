// Diagnostic: which hardware wave slots / SIMDs / CUs do the waves of co-resident workgroups get?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long* t) {
  __shared__ float big[18944];  // 75 KB like the stack kernel: 2 WG per CU
  big[threadIdx.x] = 0;
  if ((threadIdx.x & 63) == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    t[blockIdx.x * 4 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime();
  }
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
  if (big[threadIdx.x] != 0) out[0] = 1;
}
int main() {
  const int G = 512;
  unsigned* d; unsigned long long* t;
  hipMalloc(&d, G * 4 * 2 * sizeof(unsigned)); hipMalloc(&t, G * 4 * 8);
  hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, d, t);
  std::vector<unsigned> h(G * 8); std::vector<unsigned long long> ht(G * 4);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(ht.data(), t, ht.size() * 8, hipMemcpyDeviceToHost);
  for (int b = 0; b < G; b += 1) {
    if (b < 6 || (b >= 256 && b < 262) || b > 506) {
      printf("blk %3d:", b);
      for (int w = 0; w < 4; ++w) {
        unsigned hw = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1];
        printf(" [w%d slot=%u simd=%u cu=%u sh=%u se=%u xcc=%u t=%llu]", w, hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, x & 15, ht[b*4+w] % 100000000ull);
      }
      printf("\n");
    }
  }
  // histogram of slot ids and of co-residency: key = (xcc,se,sh,cu,simd) -> list of slots
  int hist[16] = {0};
  for (int i = 0; i < G * 4; ++i) hist[h[i * 2] & 15]++;
  printf("slot histogram:"); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("\n");
  return 0;
}
