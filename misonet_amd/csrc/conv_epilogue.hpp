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
// x >= +0 ? x : e  as a bit select on the sign of x (v_ashrrev_i32 + v_bfi_b32): no compare, so no VCC / SGPR round
// trip between the two VALU instructions (v_cmp -> v_cndmask costs 2 wait states on gfx950, ~30 s_nops per 16-value
// row in the tile epilogue), and NaN-transparent like the compare form (x = NaN selects x or e = NaN): a v_med3_f32
// would be one instruction but returns min3 = 0 for NaN inputs, i.e. it would swallow a NaN instead of handing it on to
// the output check (reference model.py:109-110).  x = -0 selects e = exp(-0) - 1 = 0.
__device__ __forceinline__ float elu_select(float x, float e) {
  const int m = __builtin_bit_cast(int, x) >> 31;
  float r;
  // (e & m) | (x & ~m) as ONE v_bfi_b32.  Inline asm: written in C, LLVM folds any form of it back into a select on
  // x < 0, i.e. into the v_cmp + v_cndmask pair this function is here to avoid.
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(e), "v"(x));
  return r;
}

__device__ __forceinline__ float elu_fast(float v) { return elu_select(v, __expf(v) - 1.f); }

// two ELUs at once: the scale and the - 1 are packed (v_pk_mul_f32 / v_pk_add_f32); bitwise the pair (elu_fast(x), elu_fast(y))
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t elu_fast2(f32x2_t v) {
  f32x2_t e = v * 1.44269504088896341f;                    // (__expf(v) = v_exp_f32(v * log2 e): the same product, packed)
  e.x = __builtin_amdgcn_exp2f(e.x);
  e.y = __builtin_amdgcn_exp2f(e.y);
  e = e - 1.f;
  f32x2_t r;
  r.x = elu_select(v.x, e.x);
  r.y = elu_select(v.y, e.y);
  return r;
}

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

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// two values -> packed (hi, hi) and (lo, lo) bf16 pairs (x = hi + lo + O(2^-17 |x|)); element 0 in the low half
__device__ __forceinline__ void split_pair_t(float x0, float x1, unsigned& hi, unsigned& lo) {
  f32x2_t x = {x0, x1};
  const bf16x2_t h = __builtin_convertvector(x, bf16x2_t);
  const unsigned hu = __builtin_bit_cast(unsigned, h);
  f32x2_t r;
  r.x = x0 - __builtin_bit_cast(float, hu << 16);
  r.y = x1 - __builtin_bit_cast(float, hu & 0xffff0000u);
  const bf16x2_t l = __builtin_convertvector(r, bf16x2_t);
  hi = hu;
  lo = __builtin_bit_cast(unsigned, l);
}

// fp16 flavour (f16x3 mode): x = hi + lo + O(2^-22 |x|) for |x| in fp16's normal range (11-bit pieces; the MFMA keeps
// fp16 subnormals, tools/micro/mfma_f16_denorm.hip, so smaller values degrade gracefully to an absolute 2^-25)
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_h(float x0, float x1, unsigned& hi, unsigned& lo) {
  f32x2_t x = {x0, x1};
  const f16x2_t h = __builtin_convertvector(x, f16x2_t);
  f32x2_t r;
  r.x = x0 - (float)h.x;
  r.y = x1 - (float)h.y;
  const f16x2_t l = __builtin_convertvector(r, f16x2_t);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
template <bool F16>
__device__ __forceinline__ void split_pair_x(float x0, float x1, unsigned& hi, unsigned& lo) {
  if (F16) split_pair_h(x0, x1, hi, lo);
  else split_pair_t(x0, x1, hi, lo);
}

// two values -> three packed bf16 pairs with x = hi + mid + lo EXACTLY (bf16x6 mode): hi = rne(x), mid = rne(x - hi),
// lo = x - hi - mid.  x - hi is exact in float32 and has at most 16 significant bits, x - hi - mid at most 8, so the
// last conversion does not round (float32 has 24 significant bits = 3 x 8); bf16 has the exponent range of float32.
__device__ __forceinline__ void split3_pair_t(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  f32x2_t x = {x0, x1};
  const bf16x2_t h = __builtin_convertvector(x, bf16x2_t);
  const unsigned hu = __builtin_bit_cast(unsigned, h);
  // residuals as 2-wide vector subtractions (v_pk_add_f32 with negated source)
  const f32x2_t hf = {__builtin_bit_cast(float, hu << 16), __builtin_bit_cast(float, hu & 0xffff0000u)};
  const f32x2_t r = x - hf;
  const bf16x2_t m = __builtin_convertvector(r, bf16x2_t);
  const unsigned mu = __builtin_bit_cast(unsigned, m);
  const f32x2_t mf = {__builtin_bit_cast(float, mu << 16), __builtin_bit_cast(float, mu & 0xffff0000u)};
  const f32x2_t q = r - mf;
  const bf16x2_t l = __builtin_convertvector(q, bf16x2_t);
  hi = hu;
  mid = mu;
  lo = __builtin_bit_cast(unsigned, l);
}

// split3_pair_t for two pairs at once, the two dependency chains interleaved statement by statement: every link of a chain
// (v_cvt_pk_bf16_f32 -> shift/and -> v_pk_add_f32 -> v_cvt_pk_bf16_f32 ...) needs one independent instruction before its
// consumer on gfx950, and written pair after pair the compiler pads the links with s_nops.
__device__ __forceinline__ void split3_quad_t(float x0, float x1, float x2, float x3, unsigned (&hi)[2], unsigned (&mid)[2],
                                              unsigned (&lo)[2]) {
  const f32x2_t xa = {x0, x1}, xb = {x2, x3};
  const unsigned ha = __builtin_bit_cast(unsigned, __builtin_convertvector(xa, bf16x2_t));
  const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(xb, bf16x2_t));
  const f32x2_t haf = {__builtin_bit_cast(float, ha << 16), __builtin_bit_cast(float, ha & 0xffff0000u)};
  const f32x2_t hbf = {__builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xffff0000u)};
  const f32x2_t ra = xa - haf;
  const f32x2_t rb = xb - hbf;
  const unsigned ma = __builtin_bit_cast(unsigned, __builtin_convertvector(ra, bf16x2_t));
  const unsigned mb = __builtin_bit_cast(unsigned, __builtin_convertvector(rb, bf16x2_t));
  const f32x2_t maf = {__builtin_bit_cast(float, ma << 16), __builtin_bit_cast(float, ma & 0xffff0000u)};
  const f32x2_t mbf = {__builtin_bit_cast(float, mb << 16), __builtin_bit_cast(float, mb & 0xffff0000u)};
  const f32x2_t qa = ra - maf;
  const f32x2_t qb = rb - mbf;
  hi[0] = ha; hi[1] = hb;
  mid[0] = ma; mid[1] = mb;
  lo[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(qa, bf16x2_t));
  lo[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(qb, bf16x2_t));
}

// One accumulator row (16 values of one lane: channel (i&3) + 8*(i>>2) + 4*half of a 32-channel group at one frame)
// -> NP bf16 parts in the oct layout: part p of octet pair k is one 16-byte store per lane (lanes 0-31 store octet 2k,
// lanes 32-63 octet 2k + 1, after a half-wave swap).  rs[p]: descriptor of part p; vo: byte offset of (row, frame) in
// the plane of octet (group base + half); ok0 / ok1: this lane's octet of pair 0 / 1 exists and its frame is valid.
// NQ: register quads (= octets of this half-wave pair) that can hold channels: 3 for a group of <= 24 output channels, whose
// fourth octet is not split (it does not exist: its lanes' stores fall outside num_records).
template <int NP, bool F16 = false, int NQ = 4>
__device__ __forceinline__ void store_oct_row(const float (&v)[16], const __amdgpu_buffer_rsrc_t (&rs)[3], unsigned vo,
                                              unsigned P16, bool ok0, bool ok1) {
  unsigned P[3][4][2];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    if (o >= NQ) {
#pragma unroll
      for (int p = 0; p < 3; ++p) { P[p][o][0] = 0u; P[p][o][1] = 0u; }
    } else if (NP == 3) {
      split3_quad_t(v[4 * o + 0], v[4 * o + 1], v[4 * o + 2], v[4 * o + 3], P[0][o], P[1][o], P[2][o]);
    } else {
      split_pair_x<F16>(v[4 * o + 0], v[4 * o + 1], P[0][o][0], P[1][o][0]);
      split_pair_x<F16>(v[4 * o + 2], v[4 * o + 3], P[0][o][1], P[1][o][1]);
    }
  }
#pragma unroll
  for (int k = 0; k < (NQ <= 2 ? 1 : 2); ++k) {          // NQ <= 2 (a 16-channel group): octets 2, 3 do not exist
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        auto rr = __builtin_amdgcn_permlane32_swap(P[p][2 * k][d], P[p][2 * k + 1][d], false, false);
        P[p][2 * k][d] = rr[0]; P[p][2 * k + 1][d] = rr[1];
      }
    }
    if (k == 0 ? ok0 : ok1) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const u32x4_t u = {P[p][2 * k][0], P[p][2 * k][1], P[p][2 * k + 1][0], P[p][2 * k + 1][1]};
        // non-temporal: a layer's output (2-10 GB) is read back by the NEXT launch, long after it left the L2; streaming
        // stores measured +0.3 % on the whole step (A/B on one box: 144.6 -> 145.1 utt/s; sc0 / sc0+sc1: +-0)
        __builtin_amdgcn_raw_buffer_store_b128(u, rs[p], vo + (unsigned)(2 * k) * P16, 0, 2);
      }
    }
  }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_e(unsigned long long pa, unsigned bytes) {
  // NB: readfirstlane returns int -- unsigned temporaries, an int OR-ed into the 64-bit pointer would be sign-extended
  const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa);
  const unsigned phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)phi << 32) | plo), 0,
                                           __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// s_red: [COP][2] floats of THIS wave's row (the caller adds the rows of a tile).  COP = NCO * 32.
// s_bias: optional LDS copy of this group's bias [COP].
// s_b4:   optional LDS table [COP][4] = (bL, bC, bR, bL + bC + bR) of THIS wave's row: the bias plus the folded
//         instance-norm shift of the DMA dataflow, split by time tap so that the first / last frame of the utterance
//         (whose left / right taps fall into the zero padding) can drop their share.
// OCTP:   0, or the number of bf16 parts of the oct-layout output path to compile: 2 (bf16x3: hi | lo) or 3 (bf16x6:
//         hi | mid | lo) or 4 (f16x3: fp16 hi | lo); a.out_oct selects it at run time.
// CENTRE: the activation is STORED centred, y = ELU(conv + bias) - ELU(bias) (ELU(bias) is the activation at the exact
//         mean of the pre-activation when the inputs are instance-normalised), and the statistics are those of y.  Every
//         consumer of an activated conv output instance-normalises it, and the instance norm does not see a per-channel
//         constant, so nothing downstream changes mathematically -- but (i) the one-pass float32 partial sums of y^2 no
//         longer lose mean^2 / var of their accuracy and (ii) the folded modes' products W' * y cancel against a much
//         smaller shift.  (Raw, un-normalised consumers only ever read act = 0 outputs and the TCN output.)
template <int NCO, int NSEG, int OCTP, bool ACT, bool CENTRE = false>
__device__ __forceinline__ void conv_epilogue_impl(const ConvArgs& a, f32x16_t (&acc)[NCO][NSEG], int n, int cg, int f,
                                                   int t0, bool row_ok, int lane, float* s_red,
                                                   const float* s_bias, const float* s_b4) {
  constexpr int COP = NCO * 32;
  const int half = lane >> 5, l31 = lane & 31;
  const int T = a.T, Tp = a.Tp;
  const int cbase = cg * COP;                                               // first channel of this group
  bool tm[NSEG];
#pragma unroll
  for (int s = 0; s < NSEG; ++s) tm[s] = row_ok && (t0 + s * 32 + l31 < T);
  const bool all_t = row_ok && (t0 + NSEG * 32 <= T);                       // uniform: every frame of the tile exists
  const bool full_c = (cbase + COP <= a.Cout);                              // uniform: every channel of the group exists
  const int cmax = a.Cout - cbase - 4 * half;                               // lane's channel k_r + 32j is valid iff < cmax
  const bool unmasked = all_t && full_c;
  const bool t_edge = s_b4 && (t0 == 0 || t0 + NSEG * 32 >= T);             // uniform: tile holds frame 0 or T - 1

  if (OCTP && a.out_oct) {
    // ---- oct layout: per (octet, frame) one 16-byte unit in each of the OCTP parts ----
    constexpr int NP = OCTP == 3 ? 3 : 2;                                   // OCTP 2: bf16 hi|lo, 3: bf16 hi|mid|lo, 4: fp16 hi|lo
    constexpr bool F16 = OCTP == 4;
    const unsigned P16 = (unsigned)a.Fout * (unsigned)Tp * 16u;             // bytes per octet plane
    const unsigned long long pa = reinterpret_cast<unsigned long long>(a.out) + (unsigned long long)n * a.out_bstride * 4ull +
                                  (unsigned long long)(a.out_c0 >> 3) * P16;
    const unsigned long long part_b = (unsigned long long)(a.out_sstride >> 3) * P16;
    const unsigned nrec = (unsigned)(a.Cout >> 3) * P16;
    __amdgpu_buffer_rsrc_t rs[3];
    rs[0] = make_rsrc_e(pa, nrec);
    rs[1] = make_rsrc_e(pa + part_b, nrec);
    rs[2] = NP == 3 ? make_rsrc_e(pa + 2 * part_b, nrec) : rs[1];
    unsigned voff[NSEG];
#pragma unroll
    for (int s = 0; s < NSEG; ++s)
      voff[s] = (unsigned)(f * Tp + t0 + s * 32 + l31) * 16u + (unsigned)half * P16;
#pragma unroll
    for (int j = 0; j < NCO; ++j) {
      float bs[16], bl[16], br[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int bi = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (s_b4) {
          const float4 q = reinterpret_cast<const float4*>(s_b4)[bi];
          bs[r] = q.w; bl[r] = q.x; br[r] = q.z;
        } else {
          bs[r] = s_bias ? s_bias[bi] : a.bias[cbase + bi];
          bl[r] = 0.f; br[r] = 0.f;
        }
      }
      float s1[16], s2[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < NSEG; ++s) {
        float v[16];
        const int t = t0 + s * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float x = acc[j][s][r] + bs[r];
          if (t_edge) x -= (t == 0 ? bl[r] : 0.f) + (t == T - 1 ? br[r] : 0.f);
          if (ACT) x = elu_fast(x);
          // stored CENTRED about ELU(bias) (see conv_epilogue_impl's header); with s_b4 bs[r] also carries the folded
          // shift of this row, which is not a per-channel constant: the centre always comes from the plain bias
          if (CENTRE && ACT) x -= elu_fast(s_b4 ? a.bias[cbase + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] : bs[r]);
          v[r] = x;
          const int kr = j * 32 + (r & 3) + 8 * (r >> 2);
          const float vm = (unmasked || (tm[s] && (full_c || kr < cmax))) ? x : 0.f;
          s1[r] += vm;
          s2[r] = fmaf(vm, vm, s2[r]);
        }
        // this lane's octet (2k + half of group j) exists and its frame is inside the utterance
        const bool ok0 = !(a.dbg & 8) && (unmasked || (tm[s] && (cbase + j * 32 + (0 + half) * 8 < a.Cout)));
        const bool ok1 = !(a.dbg & 8) && (unmasked || (tm[s] && (cbase + j * 32 + (2 + half) * 8 < a.Cout)));
        store_oct_row<NP, F16>(v, rs, voff[s] + (unsigned)((cbase >> 3) + j * 4) * P16, P16, ok0, ok1);
      }
      if (ACT && !(a.dbg & 16)) {
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
    return;
  }

  const unsigned P4 = (unsigned)a.Fout * (unsigned)Tp * 4u;                 // bytes per channel plane
  // ---- descriptor of this sample's output slice [out_c0, out_c0 + Cout) ----
  const float* ob = a.out + (long long)n * a.out_bstride + (long long)a.out_c0 * a.Fout * Tp;
  const unsigned long long pa = reinterpret_cast<unsigned long long>(ob);
  const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa);
  const unsigned phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)phi << 32) | plo), 0,
      __builtin_amdgcn_readfirstlane((int)((unsigned)a.Cout * P4)), 0x00020000);
  // ---- per-lane byte offsets, one per frame tile; an offset is out of range (store dropped) when the frame is >= T or
  // the row is off ----
  unsigned voff[NSEG];
#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    const int t = t0 + s * 32 + l31;
    voff[s] = tm[s] ? ((unsigned)(f * Tp + t) * 4u + (unsigned)(4 * half) * P4) : 0x80000000u;
  }

#pragma unroll
  for (int j = 0; j < NCO; ++j) {
    float bs[16], bl[16], br[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bi = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (s_b4) {
        const float4 q = reinterpret_cast<const float4*>(s_b4)[bi];
        bs[r] = q.w; bl[r] = q.x; br[r] = q.z;
      } else {
        bs[r] = s_bias ? s_bias[bi] : a.bias[cbase + bi];      // LDS copy made at kernel start, or global
        bl[r] = 0.f; br[r] = 0.f;
      }
    }
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kr = j * 32 + (r & 3) + 8 * (r >> 2);
      const unsigned coff = (unsigned)(cbase + kr) * P4;                      // uniform plane offset
      float a1 = 0.f, a2 = 0.f;
      const float cr = (CENTRE && ACT) ? elu_fast(s_b4 ? a.bias[cbase + kr + 4 * half] : bs[r]) : 0.f;
      if (unmasked && !t_edge) {                                              // the common case: no masks at all
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
          float v = acc[j][s][r] + bs[r];
          if (ACT) v = elu_fast(v) - cr;
          if (!(a.dbg & 8)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff[s] + coff, 0, 0);
          a1 += v;
          a2 = fmaf(v, v, a2);
        }
      } else {
        const bool cok = full_c || (kr < cmax);
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
          const int t = t0 + s * 32 + l31;
          float v = acc[j][s][r] + bs[r];
          if (t_edge) v -= (t == 0 ? bl[r] : 0.f) + (t == T - 1 ? br[r] : 0.f);
          if (ACT) v = elu_fast(v) - cr;
          if (!(a.dbg & 8)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff[s] + coff, 0, 0);
          const float vm = (tm[s] && cok) ? v : 0.f;
          a1 += vm;
          a2 = fmaf(vm, vm, a2);
        }
      }
      s1[r] = a1;
      s2[r] = a2;
    }
    if (ACT && !(a.dbg & 16)) {
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

// ACT (ELU + statistics) is a compile-time parameter of the implementation: a run-time test inside the unrolled element
// loops turns into a branch per element and serialises the exp latency.
template <int NCO, int NSEG = 4, int OCTP = 0, bool CENTRE = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16_t (&acc)[NCO][NSEG], int n, int cg, int f,
                                              int t0, bool row_ok, int lane, float* s_red,
                                              const float* s_bias = nullptr, const float* s_b4 = nullptr) {
  if (a.act) conv_epilogue_impl<NCO, NSEG, OCTP, true, CENTRE>(a, acc, n, cg, f, t0, row_ok, lane, s_red, s_bias, s_b4);
  else conv_epilogue_impl<NCO, NSEG, OCTP, false, CENTRE>(a, acc, n, cg, f, t0, row_ok, lane, s_red, s_bias, s_b4);
}

// Tile epilogue of the row-reuse mapping (conv_bf16_dma.hip): one wave owns 32 frames [tw, tw + 32) of FOUR output rows
// f0 .. f0 + 3 (acc[r] = 32 channels x 32 frames of row f0 + r).
//   s_bs / s_bl / s_br: LDS tables [4 rows][2 half-waves][16] in ACCUMULATOR order (entry i of half h = channel
//   (i&3) + 8*(i>>2) + 4*h): bias + folded instance-norm shift summed over all in-range taps (s_bs), and the shares of
//   the left / right time tap (s_bl, s_br) that the first / last frame of the utterance must drop (their tap falls into
//   the zero padding).  A lane reads its 16 values with four 16-byte broadcast reads.
//   s_red: [32][2] floats of THIS wave (the caller adds the waves of a tile).  Output layout by a.out_oct.
// Channels >= Cout need no masking in the statistics: their weights and bias are zero-padded, so the value is
// ELU(0) = 0 exactly.  Everything that depends only on the wave (tile edges) selects between a branch-free fast path
// and a masked path; nothing is decided per element.
template <bool MASKED, bool ACT, bool F16 = false>
__device__ __forceinline__ void conv_epilogue_rows_impl(const ConvArgs& a, f32x16_t (&acc)[4], int cg, int f0, int tw,
                                                        int lane, const float* s_bs, const float* s_bl,
                                                        const float* s_br, const __amdgpu_buffer_rsrc_t rs_h,
                                                        const __amdgpu_buffer_rsrc_t rs_l, float (&s1)[16],
                                                        float (&s2)[16]) {
  constexpr int COP = 32;
  const int half = lane >> 5, l31 = lane & 31;
  const int T = a.T, Tp = a.Tp;
  const int cbase = cg * COP;
  const int t = tw + l31;
  const unsigned P16 = (unsigned)a.Fout * (unsigned)Tp * 16u;               // bytes per octet plane (oct)
  const unsigned P4 = (unsigned)a.Fout * (unsigned)Tp * 4u;                 // bytes per channel plane (planar)
  const bool oct_ok0 = cbase + (0 + half) * 8 < a.Cout;                      // this lane's octet of pair 0 / 1 exists
  const bool oct_ok1 = cbase + (2 + half) * 8 < a.Cout;
  const float e0 = (MASKED && t == 0) ? 1.f : 0.f;                          // drop the left / right tap share
  const float e1 = (MASKED && t == T - 1) ? 1.f : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = f0 + r;
    const bool ok = !MASKED || ((f < a.Fout) && (t < T));                   // this lane's (row, frame) exists
    const float m = ok ? 1.f : 0.f;
    float bs[16];
    {
      const float4* pb = reinterpret_cast<const float4*>(s_bs + (r * 2 + half) * 16);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 q = pb[q4];
        bs[4 * q4 + 0] = q.x; bs[4 * q4 + 1] = q.y; bs[4 * q4 + 2] = q.z; bs[4 * q4 + 3] = q.w;
      }
      if (MASKED) {
        const float4* pl = reinterpret_cast<const float4*>(s_bl + (r * 2 + half) * 16);
        const float4* pr = reinterpret_cast<const float4*>(s_br + (r * 2 + half) * 16);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 ql = pl[q4], qr = pr[q4];
          bs[4 * q4 + 0] -= e0 * ql.x + e1 * qr.x; bs[4 * q4 + 1] -= e0 * ql.y + e1 * qr.y;
          bs[4 * q4 + 2] -= e0 * ql.z + e1 * qr.z; bs[4 * q4 + 3] -= e0 * ql.w + e1 * qr.w;
        }
      }
    }
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float x = (F16 ? acc[r][i] * a.descale : acc[r][i]) + bs[i];       // f16x3: the weights carry a power-of-two scale
      if (ACT) x = elu_fast(x) - elu_fast(a.bias[cbase + (i & 3) + 8 * (i >> 2) + 4 * half]);   // stored centred
      v[i] = x;
      const float vm = MASKED ? x * m : x;
      s1[i] += vm;
      s2[i] = fmaf(vm, vm, s2[i]);
    }
    if (a.dbg & 8) continue;
    if (a.out_oct) {
      const __amdgpu_buffer_rsrc_t rs2[3] = {rs_h, rs_l, rs_l};
      const unsigned vo = (unsigned)(f * Tp + t) * 16u + (unsigned)half * P16 + (unsigned)(cbase >> 3) * P16;
      store_oct_row<2, F16>(v, rs2, vo, P16, ok && oct_ok0, ok && oct_ok1);
    } else {
      // planar: channel (i&3) + 8*(i>>2) + 4*half of the group; out-of-range channels fall outside num_records,
      // missing frames / rows get an out-of-range offset (4-byte stores: dropped by the hardware)
      const unsigned vo = ok ? ((unsigned)(f * Tp + t) * 4u + (unsigned)(4 * half) * P4) : 0x80000000u;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kr = (i & 3) + 8 * (i >> 2);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[i]), rs_h, vo + (unsigned)(cbase + kr) * P4, 0, 0);
      }
    }
  }
}

template <bool F16 = false>
__device__ __forceinline__ void conv_epilogue_rows(const ConvArgs& a, f32x16_t (&acc)[4], int n, int cg, int f0, int tw,
                                                   int lane, float* s_red, const float* s_bs, const float* s_bl,
                                                   const float* s_br) {
  const int half = lane >> 5;
  const int T = a.T, Tp = a.Tp;
  // uniform per wave: all 32 frames and 4 rows exist and none of the frames is the first / last of the utterance
  const bool fast = (tw > 0) && (tw + 32 < T) && (f0 + 4 <= a.Fout);
  float s1[16], s2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { s1[i] = 0.f; s2[i] = 0.f; }

  // descriptors
  const unsigned P16 = (unsigned)a.Fout * (unsigned)Tp * 16u;
  const unsigned P4 = (unsigned)a.Fout * (unsigned)Tp * 4u;
  unsigned long long pa, pb;
  unsigned nrec;
  if (a.out_oct) {
    pa = reinterpret_cast<unsigned long long>(a.out) + (unsigned long long)n * a.out_bstride * 4ull +
         (unsigned long long)(a.out_c0 >> 3) * P16;
    pb = pa + (unsigned long long)(a.out_sstride >> 3) * P16;
    nrec = (unsigned)(a.Cout >> 3) * P16;
  } else {
    pa = reinterpret_cast<unsigned long long>(a.out + (long long)n * a.out_bstride + (long long)a.out_c0 * a.Fout * Tp);
    pb = pa;
    nrec = (unsigned)a.Cout * P4;
  }
  // NB: readfirstlane returns int -- unsigned temporaries, or the OR below sign-extends the low half into the high half
  const unsigned pa_lo = __builtin_amdgcn_readfirstlane((unsigned)pa), pa_hi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
  const unsigned pb_lo = __builtin_amdgcn_readfirstlane((unsigned)pb), pb_hi = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32));
  const int nrec_s = __builtin_amdgcn_readfirstlane((int)nrec);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)pa_hi << 32) | pa_lo), 0, nrec_s, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)pb_hi << 32) | pb_lo), 0, nrec_s, 0x00020000);

  if (a.act) {
    if (fast) conv_epilogue_rows_impl<false, true, F16>(a, acc, cg, f0, tw, lane, s_bs, s_bl, s_br, rs_h, rs_l, s1, s2);
    else conv_epilogue_rows_impl<true, true, F16>(a, acc, cg, f0, tw, lane, s_bs, s_bl, s_br, rs_h, rs_l, s1, s2);
  } else {
    conv_epilogue_rows_impl<true, false, F16>(a, acc, cg, f0, tw, lane, s_bs, s_bl, s_br, rs_h, rs_l, s1, s2);
  }

  if (a.act && !(a.dbg & 16)) {
    const float x1 = reduce16_halfwave(s1, lane);
    const float x2 = reduce16_halfwave(s2, lane);
    if ((lane & 16) == 0) {
      const int q = lane & 15;
      const int co_l = (q & 3) + 8 * (q >> 2) + 4 * half;
      s_red[co_l * 2 + 0] = x1;
      s_red[co_l * 2 + 1] = x2;
    }
  }
}

// ---- epilogue of the persistent DMA kernel: the accumulators were INITIALISED with the bias (conv_acc_init_rows), so
// the element work is ELU + statistics only, written with 2-wide vector types so that the multiplies, adds and FMAs
// become v_pk_*_f32 (two elements per instruction).
typedef float f32x2_e __attribute__((ext_vector_type(2)));

// acc[r][i] = bias + folded shift of (row f0 + r, channel of accumulator slot i); tables as in conv_epilogue_rows.
template <int NROW>
__device__ __forceinline__ void conv_acc_init_rows(f32x16_t (&acc)[NROW], int tw, int T, int lane, const float* s_bs,
                                                   const float* s_bl, const float* s_br) {
  const int half = lane >> 5, l31 = lane & 31;
  const int t = tw + l31;
  const bool t_edge = (tw == 0 || tw + 32 >= T);                            // uniform
#pragma unroll
  for (int r = 0; r < NROW; ++r) {
    const float4* pb = reinterpret_cast<const float4*>(s_bs + (r * 2 + half) * 16);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 q = pb[q4];
      acc[r][4 * q4 + 0] = q.x; acc[r][4 * q4 + 1] = q.y; acc[r][4 * q4 + 2] = q.z; acc[r][4 * q4 + 3] = q.w;
    }
  }
  if (t_edge) {
    const float e0 = (t == 0) ? 1.f : 0.f, e1 = (t == T - 1) ? 1.f : 0.f;
#pragma unroll
    for (int r = 0; r < NROW; ++r) {
      const float4* pl = reinterpret_cast<const float4*>(s_bl + (r * 2 + half) * 16);
      const float4* pr = reinterpret_cast<const float4*>(s_br + (r * 2 + half) * 16);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 ql = pl[q4], qr = pr[q4];
        acc[r][4 * q4 + 0] -= e0 * ql.x + e1 * qr.x; acc[r][4 * q4 + 1] -= e0 * ql.y + e1 * qr.y;
        acc[r][4 * q4 + 2] -= e0 * ql.z + e1 * qr.z; acc[r][4 * q4 + 3] -= e0 * ql.w + e1 * qr.w;
      }
    }
  }
}

// The same for a "two rows in M" tile (conv_bf16x6.hip, chunk_mfma6_rm2): accumulator j holds output rows 2j (registers
// 0-7) and 2j + 1 (registers 8-15) of a 16-channel group, register i & 7 = table slot i & 7 of that row (the first two
// register quads of the 32-channel order are exactly the 16 channels of such a group).
template <int NROW>
__device__ __forceinline__ void conv_acc_init_rows_rm2(f32x16_t (&acc)[NROW], int tw, int T, int lane, const float* s_bs,
                                                       const float* s_bl, const float* s_br) {
  const int half = lane >> 5, l31 = lane & 31;
  const int t = tw + l31;
  const bool t_edge = (tw == 0 || tw + 32 >= T);                            // uniform
#pragma unroll
  for (int r = 0; r < NROW; ++r) {
    const float4* pb = reinterpret_cast<const float4*>(s_bs + (r * 2 + half) * 16);
#pragma unroll
    for (int q4 = 0; q4 < 2; ++q4) {
      const float4 q = pb[q4];
      const int b = 8 * (r & 1) + 4 * q4;
      acc[r >> 1][b + 0] = q.x; acc[r >> 1][b + 1] = q.y; acc[r >> 1][b + 2] = q.z; acc[r >> 1][b + 3] = q.w;
    }
  }
  if (t_edge) {
    const float e0 = (t == 0) ? 1.f : 0.f, e1 = (t == T - 1) ? 1.f : 0.f;
#pragma unroll
    for (int r = 0; r < NROW; ++r) {
      const float4* pl = reinterpret_cast<const float4*>(s_bl + (r * 2 + half) * 16);
      const float4* pr = reinterpret_cast<const float4*>(s_br + (r * 2 + half) * 16);
#pragma unroll
      for (int q4 = 0; q4 < 2; ++q4) {
        const float4 ql = pl[q4], qr = pr[q4];
        const int b = 8 * (r & 1) + 4 * q4;
        acc[r >> 1][b + 0] -= e0 * ql.x + e1 * qr.x; acc[r >> 1][b + 1] -= e0 * ql.y + e1 * qr.y;
        acc[r >> 1][b + 2] -= e0 * ql.z + e1 * qr.z; acc[r >> 1][b + 3] -= e0 * ql.w + e1 * qr.w;
      }
    }
  }
}

// s_ctr (optional): 16 floats per half-wave in accumulator order, the centre c = ELU(bias) per channel: the row is stored
// as x - c and the statistics are those of the stored values (conv_epilogue_impl's CENTRE note).
// RM2: the accumulators are those of a two-rows-in-M tile (row r = registers 8 (r & 1) .. + 7 of accumulator r >> 1: the first
// NROW / 2 of the array are in use; NQ must be 2).
// RBEG, REND: the rows of the tile this call handles (all of them by default; the two-unit statistics call it per half).
template <bool MASKED, bool ACT, int NP, bool F16 = false, int NROW = 4, int NQ = 4, bool RM2 = false, int RBEG = 0,
          int REND = NROW>
__device__ __forceinline__ void conv_epilogue_rows_nb_impl(const ConvArgs& a, f32x16_t (&acc)[NROW], int cg, int f0, int tw,
                                                           int lane, const __amdgpu_buffer_rsrc_t (&rs)[3],
                                                           f32x2_e (&s1)[8], f32x2_e (&s2)[8], int rows,
                                                           const float* s_ctr = nullptr) {
  constexpr int COP = 32;
  const int half = lane >> 5, l31 = lane & 31;
  const int T = a.T, Tp = a.Tp;
  const int cbase = cg * COP;
  const int t = tw + l31;
  const unsigned P16 = (unsigned)a.Fout * (unsigned)Tp * 16u;
  const unsigned P4 = (unsigned)a.Fout * (unsigned)Tp * 4u;
  const bool oct_ok0 = cbase + (0 + half) * 8 < a.Cout;
  const bool oct_ok1 = cbase + (2 + half) * 8 < a.Cout;
  const f32x2_e kl2e = {1.4426950408889634f, 1.4426950408889634f};
  const f32x2_e kone = {1.f, 1.f};
  f32x2_e ctr[8];
#pragma unroll
  for (int i2 = 0; i2 < 8; ++i2) ctr[i2] = f32x2_e{0.f, 0.f};
  if (s_ctr) {
    const float4* pc = reinterpret_cast<const float4*>(s_ctr + half * 16);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 q = pc[q4];
      ctr[2 * q4] = f32x2_e{q.x, q.y};
      ctr[2 * q4 + 1] = f32x2_e{q.z, q.w};
    }
  }
#pragma unroll
  for (int r = RBEG; r < REND; ++r) {
    const int f = f0 + r;
    const bool ok = !MASKED || ((f < a.Fout) && (t < T) && (r < rows));
    const float mf = ok ? 1.f : 0.f;
    const f32x2_e m2 = {mf, mf};
    float v[16];
#pragma unroll
    for (int i = 4 * NQ; i < 16; ++i) v[i] = 0.f;      // NQ < 4: channels that do not exist (<= 24-channel groups) cost nothing
    // two element pairs (A, B) per step, their instructions interleaved: on gfx950 a v_pk_*_f32 or v_exp_f32 result
    // needs one independent instruction before its consumer, and a dependent chain written pair by pair is padded with an
    // s_nop at every link (~30 wasted issue slots per 16-value row; the tile epilogue is VALU-issue bound)
#pragma unroll
    for (int i4 = 0; i4 < NQ; ++i4) {
      constexpr int AR_SHIFT = RM2 ? 1 : 0;
      const int ar = r >> AR_SHIFT, ab = RM2 ? 8 * (r & 1) : 0;            // compile-time after unrolling
      f32x2_e xa = {acc[ar][ab + 4 * i4], acc[ar][ab + 4 * i4 + 1]};
      f32x2_e xb = {acc[ar][ab + 4 * i4 + 2], acc[ar][ab + 4 * i4 + 3]};
      if (F16) { xa = xa * f32x2_e{a.descale, a.descale}; xb = xb * f32x2_e{a.descale, a.descale}; }
      if (ACT) {                                   // compile-time: a run-time test here becomes a branch per pair and
                                                   // serialises the exp latency of the eight pairs
        f32x2_e ea = xa * kl2e;
        f32x2_e eb = xb * kl2e;
        ea.x = __builtin_amdgcn_exp2f(ea.x);
        ea.y = __builtin_amdgcn_exp2f(ea.y);
        eb.x = __builtin_amdgcn_exp2f(eb.x);
        eb.y = __builtin_amdgcn_exp2f(eb.y);
        ea = ea - kone;
        eb = eb - kone;
        xa.x = elu_select(xa.x, ea.x);
        xa.y = elu_select(xa.y, ea.y);
        xb.x = elu_select(xb.x, eb.x);
        xb.y = elu_select(xb.y, eb.y);
      }
      xa = xa - ctr[2 * i4];                           // stored centred (zeros without s_ctr)
      xb = xb - ctr[2 * i4 + 1];
      v[4 * i4] = xa.x; v[4 * i4 + 1] = xa.y; v[4 * i4 + 2] = xb.x; v[4 * i4 + 3] = xb.y;
      const f32x2_e va = MASKED ? xa * m2 : xa;
      const f32x2_e vb = MASKED ? xb * m2 : xb;
      s1[2 * i4] = s1[2 * i4] + va;
      s1[2 * i4 + 1] = s1[2 * i4 + 1] + vb;
      s2[2 * i4] = __builtin_elementwise_fma(va, va, s2[2 * i4]);   // one v_pk_fma_f32 (the two fused multiply-adds, bit for bit)
      s2[2 * i4 + 1] = __builtin_elementwise_fma(vb, vb, s2[2 * i4 + 1]);
    }
    if (a.dbg & 8) continue;
    if (a.out_oct) {
      const unsigned vo = (unsigned)(f * Tp + t) * 16u + (unsigned)half * P16 + (unsigned)(cbase >> 3) * P16;
      // un-masked tiles: no predicates at all -- octets past Cout lie outside num_records of the part descriptors and
      // are dropped by the hardware (the masked path still needs the per-lane frame / row test)
      if (MASKED) store_oct_row<NP, F16, NQ>(v, rs, vo, P16, ok && oct_ok0, ok && oct_ok1);
      else store_oct_row<NP, F16, NQ>(v, rs, vo, P16, true, true);
    } else {
      const unsigned vo = ok ? ((unsigned)(f * Tp + t) * 4u + (unsigned)(4 * half) * P4) : 0x80000000u;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kr = (i & 3) + 8 * (i >> 2);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[i]), rs[0], vo + (unsigned)(cbase + kr) * P4, 0, 0);
      }
    }
  }
}

// rows: output rows of the tile (4, or 2 for the stride-2 layers of the bf16x3 kernel); NP: bf16 parts of an oct output
// U2 ("two statistic units", NROW = 8): the float partial sums of the statistics are formed per HALF tile (rows 0-3, rows 4-7:
// two reduce-scatters, two partial sets at s_red and s_red + u2_stride) and each half is added to the exact accumulator on
// its own.  An 8-row tile then contributes EXACTLY what two 4-row tiles at the same rows contribute, so a layer may run on
// 4-row tiles when it has fewer 8-row tiles than the chip has CUs (a single utterance) without changing one bit of the result
// (the batch-invariance tests compare B = 1 with B = 9 bit for bit).  Used for the F <= 31 stride-1 layers only: the
// second pair of reductions costs ~1.5 % of a tile.
// VR: output rows of the tile that can exist at all (default: all): rows >= VR are not post-processed (the F = 1 bottleneck
// layers run 4-row tiles of which 1 / 3 rows exist).
template <int NP = 2, bool F16 = false, int NQ = 4, bool RM2 = false, bool U2 = false, int VR = 0, int NROW>
__device__ __forceinline__ void conv_epilogue_rows_nb(const ConvArgs& a, f32x16_t (&acc)[NROW], int n, int cg, int f0, int tw,
                                                      int lane, float* s_red, int rows = NROW, const float* s_ctr = nullptr,
                                                      int u2_stride = 0) {
  static_assert(!RM2 || NQ == 2, "two-rows-in-M tiles hold 16-channel groups");
  static_assert(!U2 || (NROW == 8 && !RM2), "two statistic units: 8-row tiles");
  const int half = lane >> 5;
  const int T = a.T, Tp = a.Tp;
  const bool fast = (tw + 32 <= T) && (f0 + NROW <= a.Fout) && rows == NROW;   // uniform: all 32 frames and all rows exist
  f32x2_e s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s1[i] = f32x2_e{0.f, 0.f}; s2[i] = f32x2_e{0.f, 0.f}; }
  const unsigned P16 = (unsigned)a.Fout * (unsigned)Tp * 16u;
  const unsigned P4 = (unsigned)a.Fout * (unsigned)Tp * 4u;
  __amdgpu_buffer_rsrc_t rs[3];
  if (a.out_oct) {
    const unsigned long long pa = reinterpret_cast<unsigned long long>(a.out) + (unsigned long long)n * a.out_bstride * 4ull +
                                  (unsigned long long)(a.out_c0 >> 3) * P16;
    const unsigned long long part_b = (unsigned long long)(a.out_sstride >> 3) * P16;
    const unsigned nrec = (unsigned)(a.Cout >> 3) * P16;
    rs[0] = make_rsrc_e(pa, nrec);
    rs[1] = make_rsrc_e(pa + part_b, nrec);
    rs[2] = NP == 3 ? make_rsrc_e(pa + 2 * part_b, nrec) : rs[1];
  } else {
    rs[0] = make_rsrc_e(reinterpret_cast<unsigned long long>(a.out + (long long)n * a.out_bstride + (long long)a.out_c0 * a.Fout * Tp),
                        (unsigned)a.Cout * P4);
    rs[1] = rs[0];
    rs[2] = rs[0];
  }
  auto reduce_to = [&](float* dst) {
    float f1[16], f2[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { f1[2 * i] = s1[i].x; f1[2 * i + 1] = s1[i].y; f2[2 * i] = s2[i].x; f2[2 * i + 1] = s2[i].y; }
    const float x1 = reduce16_halfwave(f1, lane);
    const float x2 = reduce16_halfwave(f2, lane);
    if ((lane & 16) == 0) {
      const int q = lane & 15;
      const int co_l = (q & 3) + 8 * (q >> 2) + 4 * half;
      dst[co_l * 2 + 0] = x1;
      dst[co_l * 2 + 1] = x2;
    }
  };
  constexpr int RMID = U2 ? NROW / 2 : (VR > 0 ? VR : NROW);
  static_assert(!(U2 && VR > 0), "either two statistic units or a reduced row count");

  if (a.act) {
    if (fast) conv_epilogue_rows_nb_impl<false, true, NP, F16, NROW, NQ, RM2, 0, RMID>(a, acc, cg, f0, tw, lane, rs, s1, s2, rows, s_ctr);
    else conv_epilogue_rows_nb_impl<true, true, NP, F16, NROW, NQ, RM2, 0, RMID>(a, acc, cg, f0, tw, lane, rs, s1, s2, rows, s_ctr);
  } else {
    conv_epilogue_rows_nb_impl<true, false, NP, F16, NROW, NQ, RM2, 0, RMID>(a, acc, cg, f0, tw, lane, rs, s1, s2, rows);
  }
  if (a.act && !(a.dbg & 16)) reduce_to(s_red);
  if constexpr (U2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[i] = f32x2_e{0.f, 0.f}; s2[i] = f32x2_e{0.f, 0.f}; }
    if (a.act) {
      if (fast) conv_epilogue_rows_nb_impl<false, true, NP, F16, NROW, NQ, RM2, RMID, NROW>(a, acc, cg, f0, tw, lane, rs, s1, s2, rows, s_ctr);
      else conv_epilogue_rows_nb_impl<true, true, NP, F16, NROW, NQ, RM2, RMID, NROW>(a, acc, cg, f0, tw, lane, rs, s1, s2, rows, s_ctr);
    } else {
      conv_epilogue_rows_nb_impl<true, false, NP, F16, NROW, NQ, RM2, RMID, NROW>(a, acc, cg, f0, tw, lane, rs, s1, s2, rows);
    }
    if (a.act && !(a.dbg & 16)) reduce_to(s_red + u2_stride);
  }
}


// ---- deferred tile epilogue of the persistent kernel: ELU + statistics + bf16 split + stores of the PREVIOUS tile, cut
// into pieces that run between the MFMAs of the current tile (one call per MFMA step, st compile-time after unrolling).
// Row ROW of the previous tile is handled during K-chunk ROW of the current one:
//   pieces 0-7  : element pair q: ELU in place, statistics
//   piece  8/10 : bf16 hi/lo split of elements 0-7 / 8-15      piece 9/11: half-wave swaps + the two 16-byte stores
//   pieces 12-14 (row 3 only): the two reduce-scatters of the tile's statistics, partials to LDS
// State lives in the caller: prev (accumulators of the previous tile, bias included), s1/s2, per-row mask pm and store
// offset pvo, and the pack registers PH/PL.  Everything is branch-free except the store predicates.
struct EpiState {
  f32x2_e s1[8], s2[8];
  float pmt;                // 1 if this lane's frame of the previous tile exists (t < T), else 0
  unsigned pvo0;            // byte offset of (row 0, this lane's frame) in the octet plane of the lane's half-wave
  int prows;                // rows of the previous tile that exist (uniform: min(4, Fout - f0), 0 = no previous tile)
  unsigned prow_b;          // bytes per output row (Tp * 16)
  unsigned PH[2][2], PL[2][2];
  bool okk0, okk1;          // this lane's octet of pair 0 / 1 exists (Cout)
  float dsc;                // f16x3: 2^-k of the layer's weight scale (1 otherwise)
  const float* ctr;         // LDS: ELU(bias) per channel in accumulator order [2 half-waves][16]; rows are stored centred
};

template <int ROW, int NSTEP, bool F16 = false>
__device__ __forceinline__ void conv_epi_step(int st, f32x16_t (&prev)[4], EpiState& e, const __amdgpu_buffer_rsrc_t rs_h,
                                              const __amdgpu_buffer_rsrc_t rs_l, unsigned P16, int lane) {
  constexpr int NPIECE = 12;
  constexpr int PP = (NPIECE + NSTEP - 1) / NSTEP;
  const f32x2_e kl2e = {1.4426950408889634f, 1.4426950408889634f};
  const f32x2_e kone = {1.f, 1.f};
#pragma unroll
  for (int pi = 0; pi < PP; ++pi) {
    const int q = st * PP + pi;
    if (q < 8) {
      f32x2_e x = {prev[ROW][2 * q], prev[ROW][2 * q + 1]};
      if (F16) x = x * f32x2_e{e.dsc, e.dsc};
      f32x2_e ex = x * kl2e;
      ex.x = __builtin_amdgcn_exp2f(ex.x);
      ex.y = __builtin_amdgcn_exp2f(ex.y);
      ex = ex - kone;
      x.x = x.x > 0.f ? x.x : ex.x;
      x.y = x.y > 0.f ? x.y : ex.y;
      {                                                  // stored centred (conv_epilogue_impl's CENTRE note)
        const float2 c2 = reinterpret_cast<const float2*>(e.ctr + (lane >> 5) * 16)[q];
        x.x -= c2.x; x.y -= c2.y;
      }
      prev[ROW][2 * q] = x.x; prev[ROW][2 * q + 1] = x.y;
      const float mr = (ROW < e.prows) ? e.pmt : 0.f;
      const f32x2_e m2 = {mr, mr};
      const f32x2_e vm = x * m2;
      e.s1[q] = e.s1[q] + vm;
      // explicit FMAs: with contraction left to the compiler the interleaved and the flush instantiation of this piece
      // rounded differently, which made a tile's sum of squares depend on whether it was the last tile of its workgroup
      e.s2[q].x = fmaf(vm.x, vm.x, e.s2[q].x);
      e.s2[q].y = fmaf(vm.y, vm.y, e.s2[q].y);
    } else if (q == 8 || q == 10) {
      const int b = (q == 8) ? 0 : 8;
      split_pair_x<F16>(prev[ROW][b + 0], prev[ROW][b + 1], e.PH[0][0], e.PL[0][0]);
      split_pair_x<F16>(prev[ROW][b + 2], prev[ROW][b + 3], e.PH[0][1], e.PL[0][1]);
      split_pair_x<F16>(prev[ROW][b + 4], prev[ROW][b + 5], e.PH[1][0], e.PL[1][0]);
      split_pair_x<F16>(prev[ROW][b + 6], prev[ROW][b + 7], e.PH[1][1], e.PL[1][1]);
    } else if (q == 9 || q == 11) {
      const int k = (q == 9) ? 0 : 1;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        auto rh = __builtin_amdgcn_permlane32_swap(e.PH[0][d], e.PH[1][d], false, false);
        e.PH[0][d] = rh[0]; e.PH[1][d] = rh[1];
        auto rl = __builtin_amdgcn_permlane32_swap(e.PL[0][d], e.PL[1][d], false, false);
        e.PL[0][d] = rl[0]; e.PL[1][d] = rl[1];
      }
      const u32x4_t uh = {e.PH[0][0], e.PH[0][1], e.PH[1][0], e.PH[1][1]};
      const u32x4_t ul = {e.PL[0][0], e.PL[0][1], e.PL[1][0], e.PL[1][1]};
      if (ROW < e.prows && e.pmt != 0.f && (k == 0 ? e.okk0 : e.okk1)) {
        const unsigned off = e.pvo0 + (unsigned)ROW * e.prow_b + (unsigned)(2 * k) * P16;
        __builtin_amdgcn_raw_buffer_store_b128(uh, rs_h, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(ul, rs_l, off, 0, 0);
      }
    }
  }
}

// statistics of the previous tile: reduce-scatter over the half-waves, partials to LDS, accumulators reset.  Called once
// per tile right after the MFMA loop of K-chunk 3 (not interleaved: its temporaries would not fit the register budget).
__device__ __forceinline__ void conv_epi_reduce(EpiState& e, float* s_red_w, int lane) {
  float f1[16], f2[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f1[2 * i] = e.s1[i].x; f1[2 * i + 1] = e.s1[i].y;
    f2[2 * i] = e.s2[i].x; f2[2 * i + 1] = e.s2[i].y;
  }
  const float x1 = reduce16_halfwave(f1, lane);
  const float x2 = reduce16_halfwave(f2, lane);
  if ((lane & 16) == 0) {
    const int qq = lane & 15;
    const int co_l = (qq & 3) + 8 * (qq >> 2) + 4 * (lane >> 5);
    s_red_w[co_l * 2 + 0] = x1;
    s_red_w[co_l * 2 + 1] = x2;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { e.s1[i] = f32x2_e{0.f, 0.f}; e.s2[i] = f32x2_e{0.f, 0.f}; }
}


}  // namespace mn
