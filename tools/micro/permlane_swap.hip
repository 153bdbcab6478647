#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned lane = threadIdx.x;
  unsigned a = 100 + lane, b = 200 + lane;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane] = r[0]; out[64 + lane] = r[1];
  auto r2 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + lane] = r2[0]; out[192 + lane] = r2[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[4] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1"};
  for (int j = 0; j < 4; ++j) { printf("%s:", nm[j]); for (int i = 0; i < 64; ++i) printf(" %u", h[64 * j + i]); printf("\n"); }
  return 0;
}
