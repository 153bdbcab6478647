// micro-benchmark: what the whole chip SUSTAINS (wall time, all 256 CUs, one wave per SIMD) on back-to-back bf16 MFMAs
//   * of the two shapes, 32x32x16 and 16x16x32 (same FLOPs per cycle on paper): does the 16-row shape -- which would let
//     a Cout = 48 layer run 3 x 16 rows instead of 2 x 32 -- sustain the same rate under the power limit?
//   * on random operands, on operands whose M rows 24..31 are zero (the padding of a Cout = 24 layer) and on all-zero
//     operands: what does a padded / zero MFMA cost in wall time?
// Prints cycles per MFMA (clock64), the effective clock (clock64 / wall) and TF/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline bf16x8 rnd8(unsigned& s, float zero_mask) {
  bf16x8 v;
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    v[i] = (__bf16)(zero_mask * ((float)(int)(s >> 8) * (1.f / 8388608.f) - 1.f));
  }
  return v;
}

// MODE 0: random A and B; 1: A rows 24..31 zero (lane & 31 >= 24); 2: everything zero
template <int SHAPE, int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  unsigned s = 1234567u + threadIdx.x * 7919u + blockIdx.x * 104729u;
  const int row = threadIdx.x & (SHAPE == 32 ? 31 : 15);
  const float za = MODE == 2 ? 0.f : ((MODE == 1 && SHAPE == 32 && row >= 24) ? 0.f : 1.f);
  const float zb = MODE == 2 ? 0.f : 1.f;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = rnd8(s, za); b[i] = rnd8(s, zb); }
  const unsigned long long t0 = clock64();
  float sum = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
        acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 3], b[(u >> 2) & 3], acc[u & 3], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
  } else {
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u)        // 2 x as many: a 16x16x32 MFMA is half the FLOPs
        acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u & 3], b[(u >> 2) & 3], acc[u & 3], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) sum += acc[j][r];
  }
  const unsigned long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int SHAPE, int MODE>
static void run(const char* what, float* o, unsigned long long* c) {
  const int iters = getenv("MFMA_ITERS") ? atoi(getenv("MFMA_ITERS")) : 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, MODE>), dim3(256), dim3(256), 0, 0, o, c, 2000);       // warm-up
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<SHAPE, MODE>), dim3(256), dim3(256), 0, 0, o, c, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h = 0;
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  const double n32 = (double)iters * 16;                      // 32x32x16-equivalents per wave
  const double flops = n32 * 2.0 * 32 * 32 * 16 * 1024;        // 1024 waves
  printf("%-34s %7.2f cycles per 32x32x16-equivalent, clock %.2f GHz, %7.1f TF/s (%.2f ms)\n", what, (double)h / n32,
         (double)h / (ms * 1e6), flops / (ms * 1e-3) / 1e12, ms);
}

int main() {
  float* o; unsigned long long* c;
  hipMalloc(&o, 1 << 20); hipMalloc(&c, 64);
  run<32, 0>("32x32x16 random", o, c);
  run<16, 0>("16x16x32 random", o, c);
  run<32, 1>("32x32x16 rows 24-31 of A zero", o, c);
  run<32, 2>("32x32x16 all zero", o, c);
  run<32, 0>("32x32x16 random (again)", o, c);
  return 0;
}
