// micro-check: do ds_read_b64 / ds_read2_b64 return the right words from an LDS address that is only 4-byte aligned (gfx950, default
// SH_MEM_CONFIG alignment mode), and what do they cost against the aligned form?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MIS>
__global__ void k(float* out, unsigned long long* cyc) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  const unsigned base = (unsigned)(unsigned long long)((__attribute__((address_space(3))) void*)lds) + (unsigned)(2 * (threadIdx.x & 31) + MIS) * 4u + (threadIdx.x >> 5) * 4096u;
  float acc = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
    f4 v;
    asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + (unsigned)((it & 7) * 288)));
    acc += v.x + v.y + v.z + v.w;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  f4 w;
  asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(base));
  out[threadIdx.x * 5 + 0] = w.x; out[threadIdx.x * 5 + 1] = w.y; out[threadIdx.x * 5 + 2] = w.z; out[threadIdx.x * 5 + 3] = w.w; out[threadIdx.x * 5 + 4] = acc;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* o; unsigned long long* c; hipMalloc(&o, 256 * 5 * 4); hipMalloc(&c, 8);
  for (int mis = 0; mis < 2; ++mis) {
    if (mis) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, o, c); else hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, o, c);
    float h[256 * 5]; unsigned long long hc;
    if (hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed (misalignment %d)\n", mis); return 1; }
    hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) { const int e = 2 * (t & 31) + mis + (t >> 5) * 1024; for (int j = 0; j < 4; ++j) bad += h[t * 5 + j] != (float)(e + j); }
    printf("ds_read2_b64 at a %s address: %d wrong words of 1024, %.1f cycles per read (incl. wait)\n", mis ? "4-byte aligned (odd dword)" : "8-byte aligned", bad, (double)hc / 256);
  }
  return 0;
}
