// Operand / result layout of v_mfma_f32_16x16x4_f32 on gfx950, checked on the GPU: which (row, column, k) a lane's A, B and D
// registers hold.  Expected: A[m = lane % 16][k = lane / 16], B[k = lane / 16][n = lane % 16], D reg r = D[m = 4 (lane / 16) + r][n = lane % 16].
// build + run on the box: hipcc --offload-arch=gfx950 -O2 tools/micro/mfma16_layout.hip -o /tmp/m16 && /tmp/m16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {      // A [16][4], B [4][16], D [16][16] row-major
  const int l = threadIdx.x;
  const float a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = acc[r];
}
int main() {
  float hA[64], hB[64], hD[256], *dA, *dB, *dD;
  for (int i = 0; i < 64; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
  (void)hipMalloc(&dA, 256); (void)hipMalloc(&dB, 256); (void)hipMalloc(&dD, 1024);
  (void)hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD);
  (void)hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
    float w = 0; for (int kk = 0; kk < 4; ++kk) w += hA[m * 4 + kk] * hB[kk * 16 + n];
    double d = hD[m * 16 + n] - w; if (d < 0) d = -d; if (d > worst) worst = d;
  }
  printf("v_mfma_f32_16x16x4_f32 layout check: max |diff| = %g (0 = the layout in the header comment holds)\n", worst);
  return worst != 0;
}
