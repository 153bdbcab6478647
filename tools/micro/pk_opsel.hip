// semantics check of the packed-f32 forms conv_wino.hip uses (op_sel / op_sel_hi / neg_lo / neg_hi on v_pk_add_f32, v_pk_fma_f32)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float wf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wf2 pk_t01(wf2 a, wf2 b) { wf2 d; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ wf2 pk_t23(wf2 a, wf2 b) { wf2 d; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ wf2 pk_sub(wf2 a, wf2 b) { wf2 d; asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ wf2 pk_nrm(wf2 x, wf2 nr) { wf2 d; asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(x), "v"(nr)); return d; }
__global__ void k(float* o) {
  wf2 a = {1.f + threadIdx.x, 10.f}, b = {100.f, 1000.f};
  wf2 r0 = pk_t01(a, b), r1 = pk_t23(a, b), r2 = pk_sub(a, b), r3 = pk_nrm(a, b);
  if (threadIdx.x == 0) { o[0] = r0.x; o[1] = r0.y; o[2] = r1.x; o[3] = r1.y; o[4] = r2.x; o[5] = r2.y; o[6] = r3.x; o[7] = r3.y; }
}
int main() {
  float* o; hipMalloc(&o, 64);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
  float h[8]; hipMemcpy(h, o, 32, hipMemcpyDeviceToHost);
  // a = (1, 10), b = (100, 1000)
  printf("t01 = (%g, %g)   expect (a0 - b0, a1 + b0) = (-99, 110)\n", h[0], h[1]);
  printf("t23 = (%g, %g)   expect (a1 - b0, a1 - b1) = (-90, -990)\n", h[2], h[3]);
  printf("sub = (%g, %g)   expect (-99, -990)\n", h[4], h[5]);
  printf("nrm = (%g, %g)   expect (a0 * b0 + b1, a1 * b0 + b1) = (1100, 2000)\n", h[6], h[7]);
  return 0;
}
