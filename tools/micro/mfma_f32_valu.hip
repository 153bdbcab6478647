// micro-benchmark: does side work hide behind v_mfma_f32_32x32x2_f32 (64 matrix cycles) in a ONE-wave-per-SIMD kernel?
// 16 independent accumulators, N filler instructions after every MFMA; fillers: v_add_f32 (VALU), v_pk_add_f32, ds_read_b64,
// s_nop.  Also 2 waves per SIMD: wave A MFMA only, wave B fillers only.  Prints s_memtime cycles per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ACL "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15"
template <int P, int MODE> __device__ __forceinline__ void mf(float a, float b) {
  if (MODE == 1) asm volatile("v_mfma_f32_32x32x2_f32 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(16 * (P % 6)), "n"(16 * (P % 6) + 15), "v"(a), "v"(b) : "a95");
  else asm volatile("v_mfma_f32_32x32x2_f32 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(16 * P), "n"(16 * P + 15), "v"(a), "v"(b) : "a255");
}
template <int KIND, int N> __device__ __forceinline__ void fill(float (&x)[16], const float* lds) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i % 16]) : "v"(x[(i + 5) % 16]));
    if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&x[2 * (i % 8)])) : "v"(*reinterpret_cast<double*>(&x[2 * ((i + 3) % 8)])));
    if (KIND == 2) { float2 t = *reinterpret_cast<const float2*>(lds + ((threadIdx.x * 2 + i * 128) & 4094)); x[i % 16] += t.x; }
    if (KIND == 3) asm volatile("s_nop 0");
    if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i % 16]) : "v"(x[(i + 5) % 16]));
    if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i % 16]) : "v"(x[(i + 5) % 16]));
    if (KIND == 6) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i % 16]) : "v"(x[(i + 5) % 16]));
  }
}
template <int KIND, int N, int MODE>   // MODE 0: same wave does both; 1: waves 0-3 MFMA, waves 4-7 fillers; 2: fillers only
__global__ __launch_bounds__(MODE == 1 ? 512 : 256, 1) void k(float* out, unsigned long long* cyc, int slot) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
  const float a = x[3], b = x[7];
  const bool do_m = MODE == 0 || (MODE == 1 && threadIdx.x < 256);
  const bool do_f = MODE == 0 || MODE == 2 || (MODE == 1 && threadIdx.x >= 256);
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 200; ++it) {
#define ST(P) if (do_m) mf<P, MODE>(a, b); if (do_f) fill<KIND, N>(x, lds);
    ST(0) ST(1) ST(2) ST(3) ST(4) ST(5) ST(6) ST(7) ST(8) ST(9) ST(10) ST(11) ST(12) ST(13) ST(14) ST(15)
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x == 0) && blockIdx.x == 0) cyc[slot] = t1 - t0;
}
static float* o; static unsigned long long* c; static int slot = 0;
static const char* names[64];
template <int KIND, int N, int MODE> void run(const char* nm, int threads) {
  names[slot] = nm;
  hipLaunchKernelGGL((k<KIND, N, MODE>), dim3(256), dim3(threads), 0, 0, o, c, slot);
  ++slot;
}
int main() {
  hipMalloc(&o, 1 << 22); hipMalloc(&c, 64 * 8);
  run<0, 0, 0>("1 wave/SIMD: MFMA only", 256);
  run<0, 2, 0>("1 wave/SIMD: + 2 v_add_f32 per MFMA", 256);
  run<0, 4, 0>("1 wave/SIMD: + 4 v_add_f32", 256);
  run<0, 8, 0>("1 wave/SIMD: + 8 v_add_f32", 256);
  run<0, 12, 0>("1 wave/SIMD: + 12 v_add_f32", 256);
  run<0, 16, 0>("1 wave/SIMD: + 16 v_add_f32", 256);
  run<4, 8, 0>("1 wave/SIMD: + 8 v_fma_f32", 256);
  run<5, 8, 0>("1 wave/SIMD: + 8 v_cndmask", 256);
  run<6, 8, 0>("1 wave/SIMD: + 8 v_mov", 256);
  run<1, 4, 0>("1 wave/SIMD: + 4 v_pk_add_f32", 256);
  run<1, 8, 0>("1 wave/SIMD: + 8 v_pk_add_f32", 256);
  run<2, 2, 0>("1 wave/SIMD: + 2 ds_read_b64(+add)", 256);
  run<2, 4, 0>("1 wave/SIMD: + 4 ds_read_b64(+add)", 256);
  run<3, 8, 0>("1 wave/SIMD: + 8 s_nop", 256);
  run<3, 16, 0>("1 wave/SIMD: + 16 s_nop", 256);
  run<0, 8, 1>("2 waves/SIMD: A = MFMA, B = 8 v_add_f32 per slot (cycles of A)", 512);
  run<0, 16, 1>("2 waves/SIMD: A = MFMA, B = 16 v_add_f32 per slot", 512);
  run<1, 8, 1>("2 waves/SIMD: A = MFMA, B = 8 v_pk_add_f32 per slot", 512);
  run<0, 8, 2>("1 wave/SIMD: 8 v_add_f32 only (no MFMA)", 256);
  run<0, 16, 2>("1 wave/SIMD: 16 v_add_f32 only (no MFMA)", 256);
  hipDeviceSynchronize();
  unsigned long long h[64]; hipMemcpy(h, c, 64 * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < slot; ++i) printf("%-72s %7.1f cycles per slot\n", names[i], (double)h[i] / (200.0 * 16));
  return 0;
}
