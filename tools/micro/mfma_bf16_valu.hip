// micro-benchmark: how much VALU work hides behind v_mfma_f32_32x32x16_bf16 (8 passes = 32 matrix cycles) in a ONE-wave-per-SIMD
// kernel?  (The f32 MFMA shares the vector ALU: tools/micro/mfma_f32_valu.hip.)  16 independent accumulators, N filler
// instructions after every MFMA.  Fillers: v_add_f32, v_pk_add_f32, v_and_b32, v_perm_b32, and the exact 3-piece bf16 split of two
// fp32 values (9 instructions: and, and, pk_sub, and, and, pk_sub, perm x 3).  Prints s_memtime cycles per MFMA slot.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int P, int MODE> __device__ __forceinline__ void mf(s16x8 a, s16x8 b) {
  if (MODE == 1 || MODE == 3) asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(16 * (P % 6)), "n"(16 * (P % 6) + 15), "v"(a), "v"(b) : "a95");
  else asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(16 * P), "n"(16 * P + 15), "v"(a), "v"(b) : "a255");
}
template <int KIND, int N> __device__ __forceinline__ void fill(float (&x)[16], unsigned (&u)[6]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i % 16]) : "v"(x[(i + 5) % 16]));
    if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<f32x2*>(&x[2 * (i % 8)])) : "v"(*reinterpret_cast<f32x2*>(&x[2 * ((i + 3) % 8)])));
    if (KIND == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(x[i % 16]) : "v"(x[(i + 5) % 16]));
    if (KIND == 3) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x[i % 16]) : "v"(x[(i + 5) % 16]), "v"(x[(i + 7) % 16]), "s"(0x07060302u));
    if (KIND == 5) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x[i % 16]) : "v"(x[(i + 5) % 16]), "v"(0xbf80u));
    if (KIND == 6) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(x[i % 16]) : "v"(x[(i + 5) % 16]));
    if (KIND == 7) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i % 16]) : "v"(x[(i + 5) % 16]), "s"(0x07060302u));
    if (KIND == 8) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[i % 16]) : "v"(x[(i + 5) % 16]));
    if (KIND == 9) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x[i % 16]));
    if (KIND == 10) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i % 16]) : "v"(x[(i + 5) % 16]));
    if (KIND == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<f32x2*>(&x[2 * (i % 8)])) : "v"(*reinterpret_cast<f32x2*>(&x[2 * ((i + 3) % 8)])));
    if (KIND == 12) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i % 16]) : "v"(x[(i + 5) % 16]));
    if (KIND == 14) {   // dot2-based exact split of the pair x[2j], x[2j+1]: 7 instructions
      const int j = i % 8;
      float v0 = x[2 * j], v1 = x[2 * j + 1];
      asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[0]) : "v"(v1), "v"(v0), "s"(0x07060302u));
      asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v0) : "v"(u[0]), "v"(0x0000bf80u));
      asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v1) : "v"(u[0]), "v"(0xbf800000u));
      asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[1]) : "v"(v1), "v"(v0), "s"(0x07060302u));
      asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v0) : "v"(u[1]), "v"(0x0000bf80u));
      asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v1) : "v"(u[1]), "v"(0xbf800000u));
      asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[2]) : "v"(v1), "v"(v0), "s"(0x07060302u));
      x[2 * j] += __uint_as_float(u[2] & 1u) + __uint_as_float(u[1] & 1u);
    }
    if (KIND == 4) {   // one exact 3-piece split of the pair x[2j], x[2j+1]: N counts PAIRS here (9 instructions each)
      const int j = i % 8;
      f32x2 v = *reinterpret_cast<f32x2*>(&x[2 * j]), h, r, m, r2;
      asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(h.x) : "v"(v.x));
      asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(h.y) : "v"(v.y));
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(v), "v"(h));
      asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(m.x) : "v"(r.x));
      asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(m.y) : "v"(r.y));
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r2) : "v"(r), "v"(m));
      asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[0]) : "v"(v.y), "v"(v.x), "s"(0x07060302u));
      asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[1]) : "v"(r.y), "v"(r.x), "s"(0x07060302u));
      asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[2]) : "v"(r2.y), "v"(r2.x), "s"(0x07060302u));
      x[2 * j] += __uint_as_float(u[2] & 1u);      // keeps the chain alive without changing the instruction mix much (2 more VALU)
    }
  }
}
template <int KIND, int N, int MODE>   // MODE 0: MFMA + fillers; 2: fillers only; 1: 2 waves per SIMD, both do MFMA + fillers (128 AGPRs each)
__global__ __launch_bounds__((MODE == 1 || MODE == 3) ? 512 : 256, 1) void k(float* out, unsigned long long* cyc, int slot) {
  float x[16]; unsigned u[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 16; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
  s16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + i); b[i] = (short)(0x3f00 + threadIdx.x % 7); }
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE == 3) {
    // two waves per SIMD, waves 0-3 MFMAs only, waves 4-7 fillers only (the warp-specialised shape of conv_bf16x6.hip): ONE branch
    if (threadIdx.x < 256) {
#pragma unroll 1
      for (int it = 0; it < 200; ++it) {
#define SM(P) mf<P, MODE>(a, b);
        SM(0) SM(1) SM(2) SM(3) SM(4) SM(5) SM(6) SM(7) SM(8) SM(9) SM(10) SM(11) SM(12) SM(13) SM(14) SM(15)
      }
    } else {
#pragma unroll 1
      for (int it = 0; it < 6400 / (N > 0 ? N : 1); ++it) {      // (runs longer than the MFMA waves)
#define SF(P) fill<KIND, N>(x, u);
        SF(0) SF(1) SF(2) SF(3) SF(4) SF(5) SF(6) SF(7) SF(8) SF(9) SF(10) SF(11) SF(12) SF(13) SF(14) SF(15)
      }
    }
  } else
#pragma unroll 1
  for (int it = 0; it < 200; ++it) {
    const bool do_m = MODE != 2, do_f = true;
#define ST(P) if (do_m) mf<P, MODE>(a, b); if (do_f) fill<KIND, N>(x, u);
    ST(0) ST(1) ST(2) ST(3) ST(4) ST(5) ST(6) ST(7) ST(8) ST(9) ST(10) ST(11) ST(12) ST(13) ST(14) ST(15)
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + __uint_as_float(u[0] ^ u[1] ^ u[2]);
  if ((threadIdx.x == 0) && blockIdx.x == 0) cyc[slot] = t1 - t0;
}
static float* o; static unsigned long long* c; static int slot = 0;
static const char* names[64];
template <int KIND, int N, int MODE> void run(const char* nm) {
  names[slot] = nm;
  hipLaunchKernelGGL((k<KIND, N, MODE>), dim3(256), dim3((MODE == 1 || MODE == 3) ? 512 : 256), 0, 0, o, c, slot);
  ++slot;
}
__global__ void exact_k(const float* in, unsigned* bad, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v0 = in[2 * i], v1 = in[2 * i + 1]; const float o0 = v0, o1 = v1; unsigned h, m, l;
  asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(h) : "v"(v1), "v"(v0), "s"(0x07060302u));
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v0) : "v"(h), "v"(0x0000bf80u));
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v1) : "v"(h), "v"(0xbf800000u));
  asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(m) : "v"(v1), "v"(v0), "s"(0x07060302u));
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v0) : "v"(m), "v"(0x0000bf80u));
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v1) : "v"(m), "v"(0xbf800000u));
  asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(l) : "v"(v1), "v"(v0), "s"(0x07060302u));
  const float s0 = __uint_as_float(h << 16) + __uint_as_float(m << 16) + __uint_as_float(l << 16);
  const float s1 = __uint_as_float(h & 0xffff0000u) + __uint_as_float(m & 0xffff0000u) + __uint_as_float(l & 0xffff0000u);
  // reference truncation split
  const unsigned rh = __float_as_uint(o0) & 0xffff0000u; const float r = o0 - __uint_as_float(rh);
  const unsigned rm = __float_as_uint(r) & 0xffff0000u; const float r2 = r - __uint_as_float(rm);
  const bool same = (h << 16) == rh && (m << 16) == rm && (l << 16) == (__float_as_uint(r2) & 0xffff0000u);
  if (s0 != o0 || s1 != o1) atomicAdd(bad, 1u);
  if (!same) atomicAdd(bad + 1, 1u);
}
int main() {
  {
    const int n = 1 << 20; float* in; unsigned* bad; hipMalloc(&in, n * 8); hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    float* h = (float*)malloc(n * 8); unsigned sd = 12345u;
    for (int i = 0; i < 2 * n; ++i) { sd = sd * 1664525u + 1013904223u; unsigned b = sd; unsigned e = (b >> 23) & 0xff; if (e == 255) b ^= 0x00800000u;
      if ((i & 7) == 0) b = (b & 0x807fffffu) | ((1u + (sd >> 28)) << 23);      // some tiny values (remainders go denormal)
      if ((i & 1023) == 0) b = 0; memcpy(&h[i], &b, 4); }
    hipMemcpy(in, h, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(exact_k, dim3(n / 256), dim3(256), 0, 0, in, bad, n);
    unsigned hb[2]; hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    printf("dot2 split of %d random fp32 pairs: %u sums differ from the value, %u pairs differ from the and/sub split\n", n, hb[0], hb[1]);
  }
  hipMalloc(&o, 1 << 22); hipMalloc(&c, 64 * 8);
  run<0, 0, 0>("1 wave/SIMD: bf16 MFMA only");
  run<0, 6, 0>("+ 6 v_add_f32");
  run<1, 2, 0>("+ 2 v_pk_add_f32");
  run<0, 0, 3>("2 waves/SIMD: wave A MFMAs only, wave B idle (cycles of A per MFMA)");
  run<0, 4, 3>("  wave B: 4 v_add_f32 per slot");
  run<0, 8, 3>("  wave B: 8 v_add_f32 per slot");
  run<0, 16, 3>("  wave B: 16 v_add_f32 per slot");
  run<1, 2, 3>("  wave B: 2 v_pk_add_f32 per slot");
  run<1, 4, 3>("  wave B: 4 v_pk_add_f32 per slot");
  run<1, 8, 3>("  wave B: 8 v_pk_add_f32 per slot");
  run<11, 4, 3>("  wave B: 4 v_pk_mul_f32 per slot");
  run<10, 8, 3>("  wave B: 8 v_fma_f32 per slot");
  hipDeviceSynchronize();
  unsigned long long h[64]; hipMemcpy(h, c, 64 * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < slot; ++i) printf("%-60s %7.1f cycles per slot\n", names[i], (double)h[i] / (200.0 * 16));
  return 0;
}
