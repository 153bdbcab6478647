// micro-benchmark behind DESIGN.md section 5.3: what the WHOLE CHIP sustains, in wall time, for synthetic loads that have the
// composition of the bf16x6 conv kernel -- so that the kernel's 0.44 of the 2.5 PF / 6 spec roofline can be priced against what
// the part delivers at its power limit instead of against the 2.4 GHz data-sheet clock.
//
// One persistent 512-thread workgroup per CU, as the conv kernel: waves 0-3 ("consumers", one per SIMD) issue
// v_mfma_f32_32x32x16_bf16 on random operands in periods of 64, optionally with `lds` ds_read_b128 per period (the kernel:
// 77 fragment reads per 216 MFMAs = 23 per 64) and `idle` s_sleep units per period (duty cycle < 1: barrier waits, operand
// fill, the tile epilogue); waves 4-7 ("producers") optionally stream `kb` of HBM per period into registers, one workgroup
// barrier per period keeps the two halves in step (the kernel: 76 KB of LDS-DMA per 216 MFMAs = 22 KB per 64, about half of
// it from HBM; one barrier per 216 MFMAs).
//   usage: power_mix <seconds> <idle> <lds> <kb_per_period> [zero]
// Prints: MFMA duty cycle (32 cycles x MFMAs / elapsed shader cycles of a consumer wave), shader clock (clock64 / wall),
// issued TF/s.  tools/gpu_power_table.sh samples rocm-smi (socket power, sclk) while it runs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ inline bf16x8 rnd8(unsigned& s, float scale) {
  bf16x8 v;
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    v[i] = (__bf16)(scale * ((float)(int)(s >> 8) * (1.f / 8388608.f) - 1.f));
  }
  return v;
}

__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, const u32x4* hbm, unsigned long long hbm_units,
                                            long long periods, int idle, int lds, int units_per_period, float scale) {
  __shared__ u32x4 s_x[4096];                                  // 64 KB of operand fragments
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  unsigned s = 1234567u + tid * 7919u + blockIdx.x * 104729u;
  for (int i = tid; i < 4096; i += 512) { bf16x8 v = rnd8(s, scale); s_x[i] = __builtin_bit_cast(u32x4, v); }
  __syncthreads();
  if (wave < 4) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd8(s, scale); b[i] = rnd8(s, scale); }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned long long t0 = clock64();
    unsigned rd = lane + wave * 64;
#pragma unroll 1
    for (long long it = 0; it < periods; ++it) {
      for (int q = 0; q < lds; q += 4) {                       // operand fragments from LDS (conflict-free, lane-linear),
        b[0] = __builtin_bit_cast(bf16x8, s_x[rd & 4095]);     // four per trip: constant register indices
        b[1] = __builtin_bit_cast(bf16x8, s_x[(rd + 256) & 4095]);
        b[2] = __builtin_bit_cast(bf16x8, s_x[(rd + 512) & 4095]);
        b[3] = __builtin_bit_cast(bf16x8, s_x[(rd + 768) & 4095]);
        rd += 1024;
      }
#pragma unroll
      for (int u = 0; u < 64; ++u)
        acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 3], b[(u >> 2) & 3], acc[u & 3], 0, 0, 0);
      for (int q = 0; q < idle; ++q) __builtin_amdgcn_s_sleep(1);   // ~64 cycles each
      if (units_per_period > 0) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = clock64();
    float sum = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  } else if (units_per_period > 0) {
    // 256 producer threads: each period they fetch units_per_period 16-byte units per workgroup, striding through a
    // buffer far larger than the 256 MB Infinity Cache
    const int ptid = tid - 256;
    const unsigned long long mask = hbm_units - 1;             // power of two
    unsigned long long pos = ((unsigned long long)blockIdx.x * 1315423911ull) & mask;
    u32x4 sink = {0, 0, 0, 0};
#pragma unroll 1
    for (long long it = 0; it < periods; ++it) {
      // units_per_period 16-byte units per workgroup: 256 lanes x nper loads, all issued before the first use
      const unsigned long long p0 = (pos & ~255ull) + (unsigned long long)ptid;
#pragma unroll 1
      for (int u = 0; u < units_per_period; u += 256 * 4) {
        const u32x4 v0 = hbm[(p0 + u + 0) & mask], v1 = hbm[(p0 + u + 256) & mask], v2 = hbm[(p0 + u + 512) & mask],
                    v3 = hbm[(p0 + u + 768) & mask];
        sink[0] ^= v0[0] ^ v1[1] ^ v2[2] ^ v3[3];
      }
      pos = (pos + 65536ull * 257ull) & mask;                  // next period: another region (every CU its own walk)
      __builtin_amdgcn_s_barrier();
    }
    if (sink[0] == 0x12345678u && sink[1] == 0x9abcdef0u) out[tid] = 1.f;          // keep the loads
  }
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  const int idle = argc > 2 ? atoi(argv[2]) : 0;
  const int lds = argc > 3 ? atoi(argv[3]) : 0;
  const double kb = argc > 4 ? atof(argv[4]) : 0.0;
  const float scale = (argc > 5 && atoi(argv[5])) ? 0.f : 1.f;
  const int units = ((int)(kb * 1024.0 / 16.0) + 1023) / 1024 * 1024;   // whole trips of 256 lanes x 4 loads
  float* o; unsigned long long* c; u32x4* hbm;
  const unsigned long long hbm_bytes = 4ull << 30;             // 4 GB (a power of two; >> the 256 MB Infinity Cache)
  hipMalloc(&o, 1 << 20); hipMalloc(&c, 64); hipMalloc(&hbm, hbm_bytes);
  hipMemset(hbm, 1, hbm_bytes);
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](long long periods) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(cus), dim3(512), 0, 0, o, c, hbm, hbm_bytes / 16, periods, idle, lds, units, scale);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)ms;
  };
  run(500);                                                    // warm-up
  const double ms_cal = run(5000);                             // calibrate the period length
  const long long periods = (long long)(5000.0 * seconds * 1e3 / ms_cal);
  const double ms = run(periods);
  unsigned long long h = 0;
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  const double nm = (double)periods * 64.0;                     // MFMAs per consumer wave
  const double flops = nm * 2.0 * 32 * 32 * 16 * 4.0 * cus;
  printf("idle=%d lds=%d hbm_kb_per_period=%.1f%s | %.2f s | MFMA duty %.3f | shader clock %.3f GHz | issued %.1f TF/s | HBM %.2f TB/s\n",
         idle, lds, kb, scale == 0.f ? " ZERO-operands" : "", ms * 1e-3, nm * 32.0 / (double)h, (double)h / (ms * 1e6),
         flops / (ms * 1e-3) / 1e12, (double)periods * units * 16.0 * cus / (ms * 1e-3) / 1e12);
  (void)kb;
  return 0;
}
