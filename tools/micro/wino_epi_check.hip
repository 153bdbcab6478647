// Checks the packed-f32 forms of conv_wino.hip's tile epilogue against the plain ones on the GPU (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/wino_epi_check.hip -o tools/micro/bin/wino_epi_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float wf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wf2 pk_add(wf2 a, wf2 b) { wf2 d; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ wf2 pk_sub(wf2 a, wf2 b) { wf2 d; asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ wf2 pk_spm(wf2 a) { wf2 d; asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ wf2 pk_add_nh(wf2 a, wf2 b) { wf2 d; asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ wf2 pk_mul(wf2 a, wf2 b) { wf2 d; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ wf2 pk_fma(wf2 a, wf2 b, wf2 c) { wf2 d; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

__global__ void k(const float* in, float* out) {
  const int l = threadIdx.x;
  wf2 a = {in[2 * l], in[2 * l + 1]}, b = {in[128 + 2 * l], in[129 + 2 * l]}, c = {in[256 + 2 * l], in[257 + 2 * l]};
  wf2 r;
  float* o = out + 16 * l;
  r = pk_add(a, b); o[0] = r.x; o[1] = r.y;
  r = pk_sub(a, b); o[2] = r.x; o[3] = r.y;
  r = pk_spm(a); o[4] = r.x; o[5] = r.y;
  r = pk_add_nh(a, b); o[6] = r.x; o[7] = r.y;
  r = pk_mul(a, b); o[8] = r.x; o[9] = r.y;
  r = pk_fma(a, b, c); o[10] = r.x; o[11] = r.y;
  r = pk_fma(b, b, pk_mul(a, a)); o[12] = r.x; o[13] = r.y;
}
int main() {
  float h[384], *d, *o, ho[1024];
  for (int i = 0; i < 384; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  const char* nm[7] = {"pk_add", "pk_sub", "pk_spm", "pk_add_nh", "pk_mul", "pk_fma", "pk_fma(b,b,a*a)"};
  double worst[7] = {0};
  for (int l = 0; l < 64; ++l) {
    const float a0 = h[2 * l], a1 = h[2 * l + 1], b0 = h[128 + 2 * l], b1 = h[129 + 2 * l], c0 = h[256 + 2 * l], c1 = h[257 + 2 * l];
    const float want[14] = {a0 + b0, a1 + b1, a0 - b0, a1 - b1, a0 + a1, a0 - a1, a0 + b0, a1 - b1, a0 * b0, a1 * b1,
                            fmaf(a0, b0, c0), fmaf(a1, b1, c1), fmaf(b0, b0, a0 * a0), fmaf(b1, b1, a1 * a1)};
    for (int i = 0; i < 14; ++i) worst[i / 2] = fmax(worst[i / 2], fabs((double)want[i] - ho[16 * l + i]));
  }
  for (int i = 0; i < 7; ++i) printf("%-18s max |diff| = %.3e\n", nm[i], worst[i]);
  return 0;
}
