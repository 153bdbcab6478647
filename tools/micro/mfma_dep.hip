// micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 with accumulator dependency distance 1, 2, 3, 4
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int D>
__global__ void k(float* out, unsigned long long* cyc) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 2 + i); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const unsigned long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u) acc[u % D] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u % D], 0, 0, 0);
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[D] = t1 - t0;
}
int main() {
  float* o; unsigned long long* c;
  hipMalloc(&o, 1 << 20); hipMalloc(&c, 64);
  hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, o, c);
  hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, o, c);
  hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, o, c);
  hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, o, c);
  hipDeviceSynchronize();
  unsigned long long h[8]; hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
  for (int d = 1; d <= 4; ++d) printf("dependency distance %d: %.1f cycles per MFMA\n", d, (double)h[d] / (256.0 * 12));
  return 0;
}
