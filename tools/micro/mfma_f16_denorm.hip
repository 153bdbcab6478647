// Does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (needed by an fp16 hi/lo operand split: the lo piece of a
// value below 2^-3 is subnormal)?  A = I-like (a[m][k] = 1 for k == m % 16), B[k][n] = 2^-(15 + k % 10): rows k >= 0 are
// subnormal (fp16 normal minimum 2^-14).  Prints the products.  hipcc --offload-arch=gfx950 mfma_f16_denorm.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
  const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
  h8 a, b;
  for (int e = 0; e < 8; ++e) {
    const int kk = half * 8 + e;
    a[e] = (l31 % 16 == kk) ? (_Float16)1.0f : (_Float16)0.0f;         // A[m = l31][k = kk]
    b[e] = (_Float16)ldexpf(1.0f, -(15 + kk % 10));                     // B[k = kk][n = l31]
  }
  f16v c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}
int main() {
  float* d; hipMalloc(&d, 64 * 16 * 4);
  k<<<1, 64>>>(d);
  float h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // C layout: lane l, reg r: row m = (r&3) + 8*(r>>2) + 4*(l>>5), col n = l & 31.  Row m picks k = m % 16 -> 2^-(15 + k%10)
  int ok = 1;
  for (int m = 0; m < 16; ++m) {
    const int half = (m >> 2) & 1, r = (m & 3) + 4 * (m >> 3);
    const float got = h[(half * 32 + 0) * 16 + r], want = ldexpf(1.0f, -(15 + m % 10));
    printf("row %2d: got %.6e want %.6e %s\n", m, got, want, got == want ? "ok" : "DIFF");
    ok &= got == want;
  }
  printf(ok ? "fp16 subnormal inputs are PRESERVED by the MFMA\n" : "fp16 subnormal inputs are NOT preserved\n");
  return 0;
}
