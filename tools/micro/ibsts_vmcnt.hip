// does s_getreg_b32 hwreg(HW_REG_IB_STS) expose the wave's outstanding-VMEM count (VM_CNT [3:0] + VM_CNT_HI [23:22] on gfx9)?
// conv_wino.hip wants a wait whose threshold is a run-time value (s_waitcnt takes an immediate only).
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ int vm_cnt() {
  const int lo = __builtin_amdgcn_s_getreg(7 | (0 << 6) | (3 << 11));
  const int hi = __builtin_amdgcn_s_getreg(7 | (22 << 6) | (1 << 11));
  return lo | (hi << 4);
}
__global__ void k(const float4* in, float4* out, int* log, int stride) {
  float4 v[20];
  const int c0 = vm_cnt();
#pragma unroll
  for (int i = 0; i < 20; ++i) v[i] = in[(size_t)(threadIdx.x + 64 * i) * stride];
  const int c1 = vm_cnt();
  int spins = 0, c2;
  while ((c2 = vm_cnt()) > 12 && spins < 100000) ++spins;     // run-time threshold
  const int c3 = vm_cnt();
  asm volatile("s_waitcnt vmcnt(0)");
  const int c4 = vm_cnt();
  float4 s = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 20; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
  out[threadIdx.x] = s;
  const int c5 = vm_cnt();
  if (threadIdx.x == 0) { log[0] = c0; log[1] = c1; log[2] = c2; log[3] = spins; log[4] = c3; log[5] = c4; log[6] = c5; }
}
int main() {
  float4 *in, *out; int* log;
  hipMalloc(&in, 1 << 28); hipMalloc(&out, 4096); hipMalloc(&log, 64);
  hipMemset(in, 0, 1 << 28);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out, log, 1024);
  int h[8]; hipMemcpy(h, log, 32, hipMemcpyDeviceToHost);
  printf("before loads %d | after issuing 20 loads %d | first value <= 12: %d after %d spins | then %d | after vmcnt(0) %d | after a store %d\n",
         h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
  return 0;
}
