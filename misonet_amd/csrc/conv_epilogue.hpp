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
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

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

template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// 16 per-lane partial sums v[0..15] -> the sum of v[q] over the 32 lanes of this half-wave, returned in the lane whose
// (lane & 15) == q (both 16-lane rows of the half-wave return it).  Reduce-scatter over DPP pairings (row_mirror,
// row_half_mirror, quad_perm xor 2, xor 1): each step halves the registers a lane is responsible for, so the whole
// reduction is 15 DPP moves + 15 adds + 30 selects instead of 16 x 5 dependent DPP adds.
__device__ __forceinline__ float reduce16_halfwave(const float (&v)[16], int lane) {
  const bool bA = (lane & 8) != 0, bB = (lane & 4) != 0, bC = (lane & 2) != 0, bD = (lane & 1) != 0;
  float a[8], b[4], c[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float keep = bA ? v[k + 8] : v[k];
    const float send = bA ? v[k] : v[k + 8];
    a[k] = keep + dpp_get<0x140>(send);            // row_mirror: q <-> 15 - q
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float keep = bB ? a[k + 4] : a[k];
    const float send = bB ? a[k] : a[k + 4];
    b[k] = keep + dpp_get<0x141>(send);            // row_half_mirror: q <-> q ^ 7
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float keep = bC ? b[k + 2] : b[k];
    const float send = bC ? b[k] : b[k + 2];
    c[k] = keep + dpp_get<0x4E>(send);             // quad_perm [2,3,0,1]: q <-> q ^ 2
  }
  const float keep = bD ? c[1] : c[0];
  const float send = bD ? c[0] : c[1];
  float r = keep + dpp_get<0xB1>(send);            // quad_perm [1,0,3,2]: q <-> q ^ 1
  r += __shfl_xor(r, 16, 64);                      // the other 16-lane row of this half-wave
  return r;
}

// In-register 4x4 transpose inside every lane quad (two xor butterflies over DPP quad_perm): on entry lane i of a quad
// holds x[k] = value of channel k at frame 4q+i; on exit x[j] = value of channel i at frame 4q+j, i.e. 4 consecutive
// frames of ONE channel -> one 16-byte store per lane instead of four 4-byte stores (the epilogue was store-issue bound).
__device__ __forceinline__ void quad_transpose4(float (&x)[4], int lane) {
  const bool o1 = (lane & 1) != 0, o2 = (lane & 2) != 0;
  {  // xor 1: 2x2 blocks (x0,x1) and (x2,x3)
    const float g0 = dpp_get<0xB1>(o1 ? x[0] : x[1]);
    const float g1 = dpp_get<0xB1>(o1 ? x[2] : x[3]);
    if (o1) { x[0] = g0; x[2] = g1; } else { x[1] = g0; x[3] = g1; }
  }
  {  // xor 2: 2x2 blocks (x0,x2) and (x1,x3)
    const float g0 = dpp_get<0x4E>(o2 ? x[0] : x[2]);
    const float g1 = dpp_get<0x4E>(o2 ? x[1] : x[3]);
    if (o2) { x[0] = g0; x[1] = g1; } else { x[2] = g0; x[3] = g1; }
  }
}

// s_red: [COP][2] floats of THIS wave's row (the caller adds the rows of a tile).  COP = NCO * 32.
template <int NCO, int NSEG = 4>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16_t (&acc)[NCO][NSEG], int n, int cg, int f,
                                              int t0, bool row_ok, int lane, float* s_red,
                                              const float* s_bias = nullptr) {
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
  // ---- per-lane masks / byte offsets, one per frame tile; an offset is out of range (store dropped) when the frame is
  // >= T or the row is off ----
  bool tm[NSEG];
  unsigned voff[NSEG];
#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    const int t = t0 + s * 32 + l31;
    tm[s] = row_ok && (t < T);
    voff[s] = tm[s] ? ((unsigned)(f * Tp + t) * 4u + (unsigned)(4 * half) * P4) : 0x80000000u;
  }
  const bool all_t = row_ok && (t0 + NSEG * 32 <= T);                                  // uniform: every frame of the tile exists
  const bool full_c = (cbase + COP <= a.Cout);                              // uniform: every channel of the group exists
  const int cmax = a.Cout - cbase - 4 * half;                               // lane's channel k_r + 32j is valid iff < cmax
  const bool unmasked = all_t && full_c;

#pragma unroll
  for (int j = 0; j < NCO; ++j) {
    float bs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bi = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      bs[r] = s_bias ? s_bias[bi] : a.bias[cbase + bi];      // LDS copy made at kernel start, or global
    }
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kr = j * 32 + (r & 3) + 8 * (r >> 2);
      const unsigned coff = (unsigned)(cbase + kr) * P4;                      // uniform plane offset
      float a1 = 0.f, a2 = 0.f;
      if (unmasked) {                                                         // the common case: no masks at all
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
          float v = acc[j][s][r] + bs[r];
          if (a.act) v = elu_fast(v);
          if (!(a.dbg & 8)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff[s] + coff, 0, 0);
          a1 += v;
          a2 = fmaf(v, v, a2);
        }
      } else {
        const bool cok = full_c || (kr < cmax);
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
          float v = acc[j][s][r] + bs[r];
          if (a.act) v = elu_fast(v);
          if (!(a.dbg & 8)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff[s] + coff, 0, 0);
          const float vm = (tm[s] && cok) ? v : 0.f;
          a1 += vm;
          a2 = fmaf(vm, vm, a2);
        }
      }
      s1[r] = a1;
      s2[r] = a2;
    }
    if (a.act && !(a.dbg & 16)) {
      const float x1 = reduce16_halfwave(s1, lane);
      const float x2 = reduce16_halfwave(s2, lane);
      if ((lane & 16) == 0) {
        const int q = lane & 15;
        const int co_l = j * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
        s_red[co_l * 2 + 0] = x1;
        s_red[co_l * 2 + 1] = x2;
      }
    }
  }
}

}  // namespace mn
