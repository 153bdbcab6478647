// Shared pieces of the bf16x3 conv kernels (conv_bf16.hip, conv_bf16_dma.hip): vector types, the per-chunk MFMA loop
// over the LDS operand images, and the f32 -> (bf16 hi, bf16 lo) split.
#pragma once
#include "kernels.hpp"

namespace mn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int CKB = 16;

template <int NCO, int NR, int SF, bool TR2, int KFMASK>
__device__ __forceinline__ void chunk_mfma_bf16(f32x16 (&acc)[NCO][4], const bf16x8* s_xhi, const bf16x8* s_xlo,
                                                const bf16x8* s_whi, const bf16x8* s_wlo, int frel, int half,
                                                int l31) {
  constexpr int COP = NCO * 32;
  constexpr int NKF = ((KFMASK >> 0) & 1) + ((KFMASK >> 1) & 1) + ((KFMASK >> 2) & 1);
  constexpr int NTAP = 3 * NKF;
  // per-lane bases (units of bf16x8 = 16 bytes)
  const int wb = half * COP + l31;                       // + tap*2*COP + j*32
  int ib[3];
#pragma unroll
  for (int kf = 0; kf < 3; ++kf) {
    const int rl = TR2 ? ((frel + kf) >> 1) : (SF * frel + kf);
    ib[kf] = (rl * 2 + half) * TW + l31 + 3;             // + seg*32 + kt
  }
  // explicit two-deep pipeline over steps = (tap, frame-tile pair)
  bf16x8 ah[2][NCO], al[2][NCO], bh[2][2], bl[2][2];
  constexpr int NSTEP = NTAP * 2;
#pragma unroll
  for (int st = -1; st < NSTEP; ++st) {
    const int cur = st & 1;
    if (st + 1 < NSTEP) {
      const int nx = st + 1;
      const int tap_ = nx >> 1, sp_ = nx & 1;
      const int kt_ = tap_ / NKF, ks_ = tap_ % NKF;
      const int kf_ = (NKF == 3) ? ks_ : ((KFMASK == 2) ? 1 : (ks_ == 0 ? 0 : 2));
      const int nb = (st + 1) & 1;
      if (sp_ == 0) {
#pragma unroll
        for (int j = 0; j < NCO; ++j) {
          ah[(tap_ & 1)][j] = s_whi[wb + ((kt_ * 3 + kf_) * 2) * COP + j * 32];
          al[(tap_ & 1)][j] = s_wlo[wb + ((kt_ * 3 + kf_) * 2) * COP + j * 32];
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bh[nb][q] = s_xhi[ib[kf_] + (sp_ * 2 + q) * 32 + kt_];
        bl[nb][q] = s_xlo[ib[kf_] + (sp_ * 2 + q) * 32 + kt_];
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // keep the next step's ds_reads AHEAD of this step's MFMAs
    if (st >= 0) {
      const int tap_ = st >> 1, sp_ = st & 1;
      const int ta = tap_ & 1;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < NCO; ++j) {
          f32x16 c = acc[j][sp_ * 2 + q];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ta][j], bh[cur][q], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ta][j], bl[cur][q], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ta][j], bh[cur][q], c, 0, 0, 0);
          acc[j][sp_ * 2 + q] = c;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// two values -> packed (hi, hi) and (lo, lo) bf16 pairs: 1 cvt_pk + 2 unpack + 2 sub + 1 cvt_pk
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  f32x2 x = {x0, x1};
  const bf16x2 h = __builtin_convertvector(x, bf16x2);
  const unsigned hu = __builtin_bit_cast(unsigned, h);
  f32x2 r;
  r.x = x0 - __builtin_bit_cast(float, hu << 16);
  r.y = x1 - __builtin_bit_cast(float, hu & 0xffff0000u);
  const bf16x2 l = __builtin_convertvector(r, bf16x2);
  hi = hu;
  lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ void split2(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

}  // namespace mn
