// Shared tile epilogue of the 3x3 conv kernels: + bias, ELU, raw store, per-channel statistics partials.
//
// One wave owns one output row f and NCO x 4 accumulator tiles of 32 channels x 32 frames in the MFMA C/D layout
// (register r of lane l: channel (r&3) + 8*(r>>2) + 4*(l>>5), frame l&31).  Everything that is uniform goes through
// the scalar unit:
//   * stores are buffer stores: one descriptor per sample (readfirstlane'd, so no waterfall loop), the per-lane part
//     (row, frame, half-wave channel offset) is ONE VGPR byte offset per frame tile plus the (uniform) channel-plane
//     offset of the accumulator register -- added in the VGPR, because the SGPR soffset of a raw buffer is not
//     bounds-checked.  Channels >= Cout fall outside num_records and frames >= T get an out-of-range offset: the
//     hardware drops those stores, no exec masking;
//   * the 16 bias values of a lane are loaded once, before the first store (a per-element bias load would serialise
//     on vmcnt behind the stores);
//   * the 32-lane reductions of the statistics use DPP (quad_perm, row_half_mirror, row_mirror, row_bcast15): no LDS
//     traffic and no lgkmcnt waits.
#pragma once
#include "kernels.hpp"

namespace mn {

typedef float f32x16_t __attribute__((ext_vector_type(16)));

// ELU(alpha = 1) = x > 0 ? x : exp(x) - 1 (reference model.py:412,429,444) on the hardware exp2: absolute error
// ~1e-7, far below the path's tolerance; ocml expm1f costs ~50 instructions per element.
__device__ __forceinline__ float elu_fast(float v) { return v > 0.f ? v : (__expf(v) - 1.f); }

// sum over the 32 lanes of each half-wave; the result is valid in lanes 16..31 (lower half) and 48..63 (upper half)
__device__ __forceinline__ float half_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false)); // row_bcast15 -> rows 1, 3
  return v;
}

// s_red: [COP][2] floats of THIS wave's row (the caller adds the rows of a tile).  COP = NCO * 32.
template <int NCO>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16_t (&acc)[NCO][4], int n, int cg, int f, int t0,
                                              bool row_ok, int lane, float* s_red) {
  constexpr int COP = NCO * 32;
  const int half = lane >> 5, l31 = lane & 31;
  const int T = a.T, Tp = a.Tp;
  const unsigned P4 = (unsigned)a.Fout * (unsigned)Tp * 4u;                 // bytes per channel plane
  const int cbase = cg * COP;                                               // first channel of this group
  // ---- descriptor of this sample's output slice [out_c0, out_c0 + Cout) ----
  const float* ob = a.out + (long long)n * a.out_bstride + (long long)a.out_c0 * a.Fout * Tp;
  const unsigned long long pa = reinterpret_cast<unsigned long long>(ob);
  const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa);
  const unsigned phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)phi << 32) | plo), 0,
      __builtin_amdgcn_readfirstlane((int)((unsigned)a.Cout * P4)), 0x00020000);
  // ---- per-lane byte offsets, one per frame tile; out-of-range when the frame is >= T or the row is off ----
  unsigned voff[4];
  bool tm[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int t = t0 + s * 32 + l31;
    tm[s] = row_ok && (t < T);
    voff[s] = tm[s] ? ((unsigned)(f * Tp + t) * 4u + (unsigned)(4 * half) * P4) : 0x80000000u;
  }
  const bool full_c = (cbase + COP <= a.Cout);                              // uniform: every channel of the group exists
  const int cmax = a.Cout - cbase - 4 * half;                               // lane's channel k_r + 32j is valid iff < cmax

#pragma unroll
  for (int j = 0; j < NCO; ++j) {
    float bs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bs[r] = a.bias[cbase + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kr = j * 32 + (r & 3) + 8 * (r >> 2);
      const unsigned coff = (unsigned)(cbase + kr) * P4;                      // uniform plane offset
      const bool cok = full_c || (kr < cmax);
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float v = acc[j][s][r] + bs[r];
        if (a.act) v = elu_fast(v);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff[s] + coff, 0, 0);
        const float vm = (tm[s] && cok) ? v : 0.f;
        a1 += vm;
        a2 = fmaf(vm, vm, a2);
      }
      s1[r] = a1;
      s2[r] = a2;
    }
    if (a.act) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x1 = half_wave_sum(s1[r]);
        const float x2 = half_wave_sum(s2[r]);
        if (l31 == 31) {
          const int co_l = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          s_red[co_l * 2 + 0] = x1;
          s_red[co_l * 2 + 1] = x2;
        }
      }
    }
  }
}

}  // namespace mn
