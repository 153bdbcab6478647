// conv3x3_bf16x6: the 3x3 convolution family (reference model.py:401-482) in FP32-FAITHFUL arithmetic on the bf16 matrix
// cores.  The reference computes these layers in float32 (model.py:77-80 `.float()`, nn.Conv2d / nn.ConvTranspose2d
// model.py:401-466); this kernel keeps that arithmetic class at 2.67x the rate of the fp32 MFMA:
//
//   * both operands are represented EXACTLY as three bf16 pieces, x = x_h + x_m + x_l (float32 has 24 significant
//     bits = 3 x 8, bf16 has float32's exponent range; split3_pair_t in conv_epilogue.hpp);
//   * a product is evaluated as the six leading partial products
//         w_h x_h + (w_h x_m + w_m x_h) + (w_h x_l + w_m x_m + w_l x_h)
//     each on v_mfma_f32_32x32x16_bf16 with float32 accumulation; every partial product of two bf16 numbers is exact in
//     float32, and the three dropped ones (w_m x_l, w_l x_m, w_l x_l) are below 2^-23 |w x| -- the size of ONE float32
//     rounding of the product, of which the fp32 FMA chain of the reference makes one per accumulation step;
//   * 6 x 32 cycles per 32x32x16 block against 8 x 64 on v_mfma_f32_32x32x2_f32.
//
// Data flow = the DMA dataflow of conv_bf16_dma.hip with three parts instead of two:
//   * activations travel in the "oct3" layout: per sample [hi | mid | lo] parts, each [c/8][f][Tp][8] bf16 (8 channels
//     of one frame = one 16-byte unit = one lane's MFMA operand), values RAW (bias + ELU, no instance norm), written
//     pre-split by the PRODUCER's epilogue;
//   * the instance norm of a layer's input is folded into per-sample weights W'[n] = W * rstd[n][ci] (float32, then split
//     exactly into three bf16 pieces by conv_wprep6_k) plus the border-aware shift table of conv_epilogue.hpp;
//   * staging is `buffer_load_dwordx4 ... lds` (LDS-DMA) issued by four producer waves of a persistent 8-wave workgroup
//     (one per CU, XCD-local tile walk), two complete stages in LDS, one workgroup barrier per K-chunk.
// What is different from the bf16x3 kernel besides the third part:
//   * a K-chunk is 8 input channels (one octet), so NO layer pads K (Cin = 24 / 72 / 120 cost 25 / 10 / 6 % there);
//   * the 16-deep K of an MFMA is filled by PAIRING TIME TAPS: lanes 0-31 (k = 0..7) read the 8 channels at time tap
//     kt = 0 and lanes 32-63 (k = 8..15) the same 8 channels at kt = 1 -- for the B operand that is just a per-lane LDS
//     address (frame + half), for the A operand a [kt0 | kt1] weight image.  The odd tap kt = 2 is paired ACROSS PARTS:
//     B = [x_h | x_m] with A = [w_h | w_h] gives hh + hm, A = [w_m | w_m] gives mh + mm, and B = [x_l | x_h] with
//     A = [w_h | w_l] gives hl + lh.  27 MFMAs per (output row, chunk) = 9 taps x 6 terms x 8 channels / 16: no padding;
//   * row reuse as in the bf16x3 kernel: an input fragment of staged row R serves the output rows f' with f' + kf = R,
//     and the 18 weight fragments of a phase stay in registers: 48 LDS fragment reads per 108 MFMAs;
//   * tiles are 128 frames x 8 output rows for the stride-1 and transposed layers (10 resp. 5 staged rows; 152 KB for two
//     stride-1 stages, 249 VGPRs with 128 accumulator registers) and x 4 rows for the stride-2 layers (9 staged rows)
//     and for F <= 4.  The 8-row tile stages 25 % fewer bytes per MFMA than the 4-row tile of the first version (10
//     rows per 8 instead of 6 per 4, one weight image per 216 instead of 108 MFMAs), which is what makes a chunk
//     MFMA-bound (see below).
#include "kernels.hpp"
#include "conv_epilogue.hpp"
#include "conv_bf16_core.hpp"
#include <stdio.h>
#include <stdlib.h>

namespace mn {

#define MN_LDS(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int X6_TW = 130;                  // staged frames per input row: slot j <-> frame t0 - 1 + j
constexpr int X6_WPAIR = 3 * 3 * 2 * 32;    // units of the [kf][part][kt0 | kt1][co] image
constexpr int X6_WU = X6_WPAIR + 3 * 3 * 32;   // + [kf][part][co] of kt = 2: 864 x 16 bytes per (chunk, group)

template <int SF, bool TR2>
__device__ __forceinline__ constexpr bool use6(int fr, int kf, int R) {
  return TR2 ? ((((fr + kf) & 1) == 0) && (((fr + kf) >> 1) == R)) : (SF * fr + kf == R);
}

// MFMA work of one K-chunk (8 input channels) for one consumer wave: 32 frames x 4 output rows x 32 output channels.
//   sx: the three input part images [part][NR][X6_TW] (16-byte units), sw: the weight image of the chunk.
// Phase A (time taps 0|1 paired in K): per staged row R three B fragments (h, m, l) and per (fr, kf) six MFMAs;
// phase B (time tap 2, parts paired in K): two B fragments and three MFMAs per (fr, kf).  Small terms first.
// RMASK / VR (the F = 1 bottleneck pair, round 4): bit R of RMASK = staged row R holds an input row that EXISTS in every tile
// of the layer, VR = output rows that exist.  Steps of staged rows outside the mask and output rows >= VR are compiled
// out: the products they would add are exact zeros (zero-filled rows) or belong to rows nobody stores, so the results are
// bit-identical to the full tile's.  Defaults = the full tile.
template <int NR, int SF, bool TR2, int NROW, unsigned RMASK = 0xffffffffu, int VR = NROW>
__device__ __forceinline__ void chunk_mfma6(f32x16 (&acc)[NROW], const bf16x8* sx, const bf16x8* sw, int wave, int half,
                                            int l31) {
  constexpr int XN = NR * X6_TW;
  const int wa = half * 32 + l31;                          // + ((kf * 3 + p) * 2) * 32
  const int xa = 32 * wave + l31 + half;                   // + p * XN + R * X6_TW     (kt = half)
  bf16x8 A[3][3], B[2][3];
#pragma unroll
  for (int kf = 0; kf < 3; ++kf)
#pragma unroll
    for (int p = 0; p < 3; ++p) A[kf][p] = sw[((kf * 3 + p) * 2) * 32 + wa];
  // phase-B operands: A2[kf][0] = [w_h | w_h], [1] = [w_m | w_m], [2] = [w_h | w_l]; B2[0] = [x_h | x_m], [1] = [x_l | x_h]
  const int w2 = X6_WPAIR + l31;                           // + (kf * 3 + p) * 32
  const int p2 = half ? 2 : 0;
  const int xb0 = (half ? XN : 0) + 32 * wave + l31 + 2;
  const int xb1 = (half ? 0 : 2 * XN) + 32 * wave + l31 + 2;
  bf16x8 A2[3][3], B2[2][2];
  constexpr int NSTEP = 2 * NR;
#pragma unroll
  for (int st = -1; st < NSTEP; ++st) {
    if (st + 1 < NR) {
      if ((RMASK >> (st + 1)) & 1u) {
#pragma unroll
        for (int p = 0; p < 3; ++p) B[(st + 1) & 1][p] = sx[p * XN + (st + 1) * X6_TW + xa];
      }
    } else if (st + 1 < NSTEP) {
      const int R_ = st + 1 - NR;
      if ((RMASK >> R_) & 1u) {
        B2[(st + 1) & 1][0] = sx[xb0 + R_ * X6_TW];
        B2[(st + 1) & 1][1] = sx[xb1 + R_ * X6_TW];
      }
    }
    if (st == NR - 2 || (NR == 1 && st == -1)) {
      // the weight fragments of phase B, one step ahead of their first use
#pragma unroll
      for (int kf = 0; kf < 3; ++kf) {
        A2[kf][0] = sw[w2 + (kf * 3 + 0) * 32];
        A2[kf][1] = sw[w2 + (kf * 3 + 1) * 32];
        A2[kf][2] = sw[w2 + (kf * 3 + p2) * 32];
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // keep the next step's ds_reads AHEAD of this step's MFMAs
    if (st >= 0 && st < NR) {
      const int R = st, cur = st & 1;
      // terms lh, hl, mm, mh, hm, hh; inside a term consecutive MFMAs hit different accumulators
#pragma unroll
      for (int term = 0; term < 6; ++term) {
        const int ap = term == 0 ? 2 : ((term == 2 || term == 3) ? 1 : 0);
        const int bp = term == 1 ? 2 : ((term == 2 || term == 4) ? 1 : 0);
#pragma unroll
        for (int fr = 0; fr < VR; ++fr)
#pragma unroll
          for (int kf = 0; kf < 3; ++kf)
            if (use6<SF, TR2>(fr, kf, R) && ((RMASK >> R) & 1u))
              acc[fr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[kf][ap], B[cur][bp], acc[fr], 0, 0, 0);
      }
    } else if (st >= NR) {
      const int R = st - NR, cur = st & 1;
#pragma unroll
      for (int term = 0; term < 3; ++term) {               // hl + lh, mh + mm, hh + hm
        const int aq = 2 - term;
        const int bq = term == 0 ? 1 : 0;
#pragma unroll
        for (int fr = 0; fr < VR; ++fr)
#pragma unroll
          for (int kf = 0; kf < 3; ++kf)
            if (use6<SF, TR2>(fr, kf, R) && ((RMASK >> R) & 1u))
              acc[fr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[kf][aq], B2[cur][bq], acc[fr], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// "Two rows in M" (G16): the LAST output-channel group of a stride-1 layer with Cout % 32 == 16 -- the 144 -> 48 conv that
// closes the F = 127 dense block of the last decoder (reference model.py:437-482 with de_channels[6] = 24: 10 % of a
// step).  As a 32-channel group its tile pads 16 of the 32 MFMA rows.  Here the rows are (output row d = 0, 1) x (channel
// co = 0..15): accumulator j holds the output rows 2j and 2j + 1 (registers 0-7: row 2j, 8-15: row 2j + 1 -- the first two
// register quads of the 32-channel order ARE the 16 channels of a 16-channel group, so tables, statistics and the oct
// stores keep their layout), and a staged input row R is multiplied by the BANDED weight fragment
// A_dR[(d, co)] = W[kf = dR - d][co], dR = R - 2j in 0..3 (zero outside 0 <= kf <= 2): 4 staged rows x 9 MFMAs per row pair
// instead of 2 x 3 x 9 = 144 instead of 216 MFMAs per chunk and wave.
//   sw: the STANDARD weight image of the group (channels 16-31 of it are zero padding): a lane outside the band reads the
//       unit of padded channel co + 16, i.e. zeros -- no masking instructions.
template <int NR, int NROW>
__device__ __forceinline__ void chunk_mfma6_rm2(f32x16 (&acc)[NROW], const bf16x8* sx, const bf16x8* sw, int wave, int half,
                                                int l31) {
  static_assert(NR == NROW + 2, "stride-1 tiles only");
  constexpr int XN = NR * X6_TW;
  constexpr int NP2 = NROW / 2;
  // an opaque copy of the lane index: everything derived from it below (11 per-lane LDS addresses) is otherwise hoisted out
  // of the persistent tile loop and held in registers across the standard-group tiles too (34 spilled VGPRs)
  asm volatile("" : "+v"(l31));
  const int d = l31 >> 4, co = l31 & 15;
  const int xa = 32 * wave + l31 + half;                   // + p * XN + R * X6_TW     (kt = half)
  const int p2 = half ? 2 : 0;
  const int xb0 = (half ? XN : 0) + 32 * wave + l31 + 2;
  const int xb1 = (half ? 0 : 2 * XN) + 32 * wave + l31 + 2;
  // per-lane units of the banded fragments: kf = dR - d inside the band, else the zero-padded channel co + 16 of kf = 0
  int wa[4], w2[4];
#pragma unroll
  for (int dR = 0; dR < 4; ++dR) {
    const int kf = dR - d;
    const bool in_band = (unsigned)kf < 3u;
    wa[dR] = in_band ? (kf * 3 * 2) * 32 + half * 32 + co : half * 32 + co + 16;         // + (p * 2) * 32
    w2[dR] = X6_WPAIR + (in_band ? (kf * 3) * 32 + co : co + 16);                         // + p * 32
  }
  bf16x8 A[4][3], A2[4][3], B[2][3], B2[2][2];
#pragma unroll
  for (int dR = 0; dR < 4; ++dR)
#pragma unroll
    for (int p = 0; p < 3; ++p) A[dR][p] = sw[wa[dR] + (p * 2) * 32];
  constexpr int NSTEP = 2 * NR;
#pragma unroll
  for (int st = -1; st < NSTEP; ++st) {
    if (st + 1 < NR) {
#pragma unroll
      for (int p = 0; p < 3; ++p) B[(st + 1) & 1][p] = sx[p * XN + (st + 1) * X6_TW + xa];
    } else if (st + 1 < NSTEP) {
      const int R_ = st + 1 - NR;
      B2[(st + 1) & 1][0] = sx[xb0 + R_ * X6_TW];
      B2[(st + 1) & 1][1] = sx[xb1 + R_ * X6_TW];
    }
    if (st == NR - 2) {
      // the weight fragments of phase B: [w_h | w_h], [w_m | w_m], [w_h | w_l], one step ahead of their first use
#pragma unroll
      for (int dR = 0; dR < 4; ++dR) {
        A2[dR][0] = sw[w2[dR] + 0 * 32];
        A2[dR][1] = sw[w2[dR] + 1 * 32];
        A2[dR][2] = sw[w2[dR] + p2 * 32];
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // keep the next step's ds_reads AHEAD of this step's MFMAs
    if (st >= 0 && st < NR) {
      const int R = st, cur = st & 1;
#pragma unroll
      for (int term = 0; term < 6; ++term) {               // lh, hl, mm, mh, hm, hh
        const int ap = term == 0 ? 2 : ((term == 2 || term == 3) ? 1 : 0);
        const int bp = term == 1 ? 2 : ((term == 2 || term == 4) ? 1 : 0);
#pragma unroll
        for (int j = 0; j < NP2; ++j)
#pragma unroll
          for (int dR = 0; dR < 4; ++dR)
            if (2 * j + dR == R)
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[dR][ap], B[cur][bp], acc[j], 0, 0, 0);
      }
    } else if (st >= NR) {
      const int R = st - NR, cur = st & 1;
#pragma unroll
      for (int term = 0; term < 3; ++term) {               // hl + lh, mh + mm, hh + hm
        const int aq = 2 - term;
        const int bq = term == 0 ? 1 : 0;
#pragma unroll
        for (int j = 0; j < NP2; ++j)
#pragma unroll
          for (int dR = 0; dR < 4; ++dR)
            if (2 * j + dR == R)
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[dR][aq], B2[cur][bq], acc[j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Same-padded stride-1 8-row tiles (padf = 1): the first row tile's staged row 0 and the last row tile's staged rows 8, 9
// are the zero padding of the image, and with F = 8 k + 7 (every dense-block level: F = 127 ... 7) the last tile's output
// row 7 does not exist.  One body per tile position (chosen per tile, wave-uniform) with those steps compiled out: 4 % /
// 17 % / 21 % fewer MFMAs in a first / last / only tile -- 1.3 % of a level's MFMAs at F = 127, 10 % at F = 15 -- and
// bit-identical results (the skipped products are exact zeros or belong to the missing row).  padf != 1 (the stride-1
// transposed layers: padf = 2) runs the full body.
template <int NR, int SF, bool TR2, int NROW, bool EDGE>
__device__ __forceinline__ void chunk_mfma6_pos(f32x16 (&acc)[NROW], const bf16x8* sx, const bf16x8* sw, int wave, int half,
                                                int l31, int tpos) {
  if constexpr (EDGE && NROW == 8 && NR == 10) {
    if (tpos == 0) chunk_mfma6<NR, SF, TR2, NROW>(acc, sx, sw, wave, half, l31);
    else if (tpos == 1) chunk_mfma6<NR, SF, TR2, NROW, 0x3FEu, 8>(acc, sx, sw, wave, half, l31);
    else if (tpos == 2) chunk_mfma6<NR, SF, TR2, NROW, 0x0FFu, 7>(acc, sx, sw, wave, half, l31);
    else chunk_mfma6<NR, SF, TR2, NROW, 0x0FEu, 7>(acc, sx, sw, wave, half, l31);
  } else {
    chunk_mfma6<NR, SF, TR2, NROW>(acc, sx, sw, wave, half, l31);
  }
}

// MODE 3, "rows in M": a stride-1 layer with Cout <= 4 -- the network's last transposed conv, 48 -> 2 * num_spks channels at
// F = 129 (reference model.py:418-423, 64).  With channels on the 32 MFMA rows it uses 4 of them.  Here the rows are
// (output row d = 0..7) x (channel co = 0..3): ONE accumulator tile holds a wave's whole 8-row x 4-channel x 32-frame
// output, and a staged input row R is multiplied ONCE per chunk (9 MFMAs: 6 for the paired time taps 0|1, 3 for tap 2)
// by the BANDED weight fragment A_R[(d, co)] = W[kf = R - d][co] (zero outside 0 <= kf <= 2): 90 MFMAs per chunk and wave
// instead of 216 -- the layer becomes bound by its input bytes (6 per element, read once), as a 4-channel layer should be.
//   sw: COMPACT weight image of the chunk: unit ((kf * 3 + p) * 3 + kt) * 4 + co, 108 units (gathered by two LDS-DMA
//       instructions); a lane whose output row is outside the band of R zeroes its fragment in registers.
[[maybe_unused]] constexpr int X6_RM_UNITS = 3 * 3 * 3 * 4;
template <int NR>
__device__ __forceinline__ void chunk_mfma6_rm(f32x16& acc, const bf16x8* sx, const bf16x8* sw, int wave, int half, int l31) {
  constexpr int XN = NR * X6_TW;
  const int d = l31 >> 2, co = l31 & 3;
  const int a0 = half * 4 + co - d * 36;                   // + R * 36 + p * 12: unit of (kf = R - d, p, kt = half, co)
  const int a2 = 2 * 4 + co - d * 36;                      // kt = 2
  const int p2 = half ? 2 : 0;
  const int xa = 32 * wave + l31 + half;                   // + p * XN + R * X6_TW     (kt = half)
  const int xb0 = (half ? XN : 0) + 32 * wave + l31 + 2;
  const int xb1 = (half ? 0 : 2 * XN) + 32 * wave + l31 + 2;
  bf16x8 A[2][3], A2[2][3], B[2][3], B2[2][2];
#pragma unroll
  for (int R = -1; R < NR; ++R) {
    if (R + 1 < NR) {
      const int nx = (R + 1) & 1;
      const bool in_band = (unsigned)(R + 1 - d) < 3u;     // 0 <= kf <= 2 for this lane's output row
      const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
      const int ar = in_band ? a0 + (R + 1) * 36 : half * 4 + co;           // out of the band: any valid unit, then zeroed
      const int ar2 = in_band ? a2 + (R + 1) * 36 : 2 * 4 + co;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const bf16x8 w = sw[ar + p * 12];
        A[nx][p] = in_band ? w : zero;
        B[nx][p] = sx[p * XN + (R + 1) * X6_TW + xa];
      }
      {
        const bf16x8 w0 = sw[ar2], w1 = sw[ar2 + 12], w2 = sw[ar2 + p2 * 12];
        A2[nx][0] = in_band ? w0 : zero;                                       // [w_h | w_h]
        A2[nx][1] = in_band ? w1 : zero;                                       // [w_m | w_m]
        A2[nx][2] = in_band ? w2 : zero;                                       // [w_h | w_l]
      }
      B2[nx][0] = sx[xb0 + (R + 1) * X6_TW];                                   // [x_h | x_m]
      B2[nx][1] = sx[xb1 + (R + 1) * X6_TW];                                   // [x_l | x_h]
    }
    __builtin_amdgcn_sched_barrier(0);     // keep the next row's ds_reads AHEAD of this row's MFMAs
    if (R >= 0) {
      const int cur = R & 1;
      // small terms first: time tap 2 (hl + lh, mh + mm), then lh, hl, mm, mh, hm of taps 0|1, the large ones last
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[cur][2], B2[cur][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cur][2], B[cur][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cur][0], B[cur][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[cur][1], B2[cur][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cur][1], B[cur][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cur][1], B[cur][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cur][0], B[cur][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[cur][0], B2[cur][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cur][0], B[cur][0], acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// tile epilogue of MODE 3: raw planar float32 output (act = 0, no statistics): register i of half-wave h holds channel
// i & 3 of output row f0 + 2 * (i >> 2) + h; rows / frames / channels that do not exist get an out-of-range offset.
__device__ __forceinline__ void conv_epilogue_rm(const ConvArgs& a, const f32x16& acc, int n, int f0, int tw, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const int t = tw + l31;
  const unsigned P4 = (unsigned)a.Fout * (unsigned)a.Tp * 4u;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc_e(
      reinterpret_cast<unsigned long long>(a.out + (long long)n * a.out_bstride + (long long)a.out_c0 * a.Fout * a.Tp),
      (unsigned)a.Cout * P4);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int co = i & 3, f = f0 + 2 * (i >> 2) + half;
    const bool ok = (t < a.T) && (f < a.Fout) && (co < a.Cout) && !(a.dbg & 8);
    const unsigned vo = ok ? (unsigned)co * P4 + (unsigned)(f * a.Tp + t) * 4u : 0x80000000u;
    const float v = acc[i];     // (a scalar temporary: __builtin_bit_cast on a vector-element lvalue reads element 0, hipcc 7.2)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, vo, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// ONE persistent workgroup per CU, 8 waves: waves 0-3 = consumers (MFMAs + tile epilogue), waves 4-7 = producers (LDS-DMA,
// epilogue tables of the coming tile, float64 statistics atomics of finished tiles).  Workgroup b belongs to XCD b & 7 and
// walks that XCD's own units (samples n % 8 == xcd, or (sample, frame tile) columns) tile by tile, row tile fastest.  K-chunks are numbered across tiles; chunk
// c lives in stage c & 1.  ONE workgroup barrier per chunk: barrier b separates chunk b - 1 from chunk b; producers
// arrive at it when chunk b has landed, consumers when they are done reading chunk b - 1; behind it the producers put
// chunk b + 1 into the stage that chunk b - 1 occupied (the first chunk of the next tile is in flight during an epilogue).
//
// Where a chunk period goes (MISONET_TIMELINE=96 with MISONET_WS_DEBUG ablations, tools/gpu_x6_ablate.sh; Cin = 96,
// F = 63, all 256 CUs busy).  4-ROW tiles (MISONET_X6_ROWS8=0): 4.6k cycles = 3.9k of MFMA phase (108 MFMAs at 36 instead
// of 32 cycles: operand fill behind the barrier + LDS waits) + 0.7k at the barrier waiting for the stage.  With the DMA
// off the period is 4.1k, with the MFMAs off 4.1-4.4k: the LDS-DMA stream of a CU (51 KB per chunk) runs at 17 B/clk when
// 8 CUs are active and at ~12 B/clk when all 256 are (tools/gpu_x6_slots.sh), i.e. that tile is CO-LIMITED by the matrix
// pipe and by bytes through the texture path.  8-ROW tiles: 76 KB per 216 MFMAs = 10 B/clk at the MFMA rate, the
// producers have slack, the barrier wait is 0.17k and a chunk takes 7.6k cycles = 35 per MFMA (tile incl. epilogue:
// 35.5 instead of 43.4).  (What the stamps showed as a 5.6-8.9k wait at the LAST barrier of a tile was the producers'
// set-up of the coming tile -- three integer divisions, 36 dependent table loads behind a full DMA queue, ~500
// instructions at one per 12 cycles; it is now spread over three iterations, see ISSUE_NEXT: +2-3 %.)  The shader clock (s_memtime against s_memrealtime inside the kernel) answers with 1.56 instead
// of 1.68 GHz -- the part is power-limited, a third of the cycle gain goes back -- so a layer gains 6-10 % in time.
//
// Measured and NOT kept (git history):
//   (i)   a third stage, (a) with the first operands of chunk b fetched during the last steps of chunk b - 1 (10 % slower)
//         and (b) with `s_waitcnt vmcnt(N)` leaving the newest batch in flight across the barrier (no change): the DMA
//         stream is throughput-bound, not latency-bound, so a deeper ring buys nothing;
//   (ii)  the tile epilogue on the producer waves through an LDS mailbox -- 4 % slower;
//   (iii) phases A and B of a staged row merged into one step (longer steps, first operands first): MFMA phase 4.14k
//         instead of 3.91k cycles, 2 % slower;
//   (v)   the tile epilogue on the producer waves through a GLOBAL-memory hand-over of the raw accumulators (consumers
//         dump 32 x 16 bytes per lane and go on; producers post-process one row per chunk of the next tile, loads one
//         chunk ahead, stores ahead of the next DMA batch; commit 6797d3f): parity-green, 4-8 % slower in three variants
//         (row after the DMA batch, row steps between the DMA instructions, row written phase-wise for ILP) -- beside the
//         MFMA wave of its SIMD a producer wave issues ONE INSTRUCTION PER 10-13 CYCLES whatever its priority or
//         dependencies, i.e. ~600 instructions per 7.6k-cycle chunk, of which the DMA issue takes ~370 and a row ~330.
//   (iv)  PLANAR float32 activations (4 instead of 6 bytes per element in HBM and through the texture path), split into
//         the three bf16 pieces by the producer waves (8 buffer_load_dwordx4 per lane and chunk, ~180 VALU, 12
//         ds_write_b128): parity-green, 6 % slower -- beside an MFMA-saturating wave a producer instruction issues every
//         ~10 cycles, so ~300 instructions per chunk take 3k+ cycles; raising the producers' s_setprio changes nothing.
template <int N>
__device__ __forceinline__ void x6_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// OUT16: the output goes to a buffer of the f16x3 mode (two fp16 pieces per value): the f16x3 networks run the layers that
// read UN-NORMALISED data -- the first dense block, whose input is the raw first-layer output (model.py:44, 401-406) -- in
// this exact arithmetic (net.hip, buf_oct); its last conv hands over to the fp16 dataflow.
// NQ = 3: an instantiation for layers of <= 24 output channels (the F = 127 dense blocks: 29 % of the conv time): the tile
// epilogue skips the fourth register quad of every row (channels 24-31 of the 32-row MFMA tile are padding).
// G16: the instantiation for layers whose last output-channel group has 16 channels (Cout % 32 == 16): the tiles of that group
// run the two-rows-in-M mapping (chunk_mfma6_rm2), the other groups the standard one.
// U2: two statistic units per 8-row tile (conv_epilogue_rows_nb): the instantiation for the F <= 31 stride-1 layers, which run
// on 4-row tiles (<0, 4>) instead when the launch has fewer 8-row tiles than CUs -- bit-identical either way.
// BN: the two stride-1 layers around the F = 1 bottleneck (reference model.py:50-52, 64: encoder 6, 3 -> 1 bins; decoder 0's
// transposed conv, 1 -> 3 bins) on 4-row tiles of which only some staged rows hold input rows and some output rows exist:
// BN = 1: staged rows 0-2 real, output row 0 exists; BN = 2: staged row 2 real, output rows 0-2 exist.  The MFMA steps, the
// operand reads and the LDS-DMA pieces of everything else are compiled out (chunk_mfma6's RMASK / VR): 27 instead of 108
// MFMAs and 33 / 20 instead of 51 KB per chunk.
template <int MODE, int FTR, bool OUT16 = false, int NQ = 4, bool G16 = false, bool U2 = false, int BN = 0>
__global__ __launch_bounds__(512, 1) void conv3x3_bf16x6(const ConvArgs a, int nslots) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (the LDS-DMA builtin has no host form)
  constexpr int COP = 32;
  constexpr int SF = MODE == 1 ? 2 : 1;
  constexpr bool TR2 = MODE == 2;
  constexpr bool RM = MODE == 3;                               // rows in M (chunk_mfma6_rm): Cout <= 4, act = 0, planar output
  static_assert(FTR == 4 || (FTR == 8 && MODE != 1), "8-row tiles: not for the stride-2 layers (17 staged rows)");
  static_assert(!RM || FTR == 8, "rows-in-M tiles are 8 output rows x 4 channels");
  static_assert(!G16 || (MODE == 0 && FTR == 8 && !OUT16 && NQ == 4), "two-rows-in-M groups: stride-1 8-row tiles, oct3 output");
  static_assert(!U2 || (MODE == 0 && FTR == 8 && !OUT16 && NQ == 4 && !G16), "two statistic units: stride-1 8-row tiles, oct3 output");
  constexpr int NU = U2 ? 2 : 1;                               // statistic units (partial sets) per tile
  static_assert(BN == 0 || (MODE == 0 && FTR == 4 && !OUT16 && NQ == 4 && !G16 && !U2), "bottleneck variants: stride-1 4-row tiles");
  constexpr unsigned RMASK = BN == 1 ? 0x7u : (BN == 2 ? 0x4u : 0xffffffffu);   // staged rows that hold input rows
  // same-padded stride-1 8-row tiles (the dense blocks: padf = 1, F = 2^k - 1): edge tiles skip their padding rows
  constexpr bool EDGE = MODE == 0 && FTR == 8 && !OUT16 && BN == 0;
  constexpr int VR = BN == 1 ? 1 : (BN == 2 ? 3 : FTR);        // output rows that exist
  constexpr int NR = (MODE == 0 || RM) ? FTR + 2 : (MODE == 1 ? 9 : FTR / 2 + 1);   // staged input rows of an FTR-row tile
  constexpr int NS = 2;                                        // stages
  constexpr int XN = NR * X6_TW;                               // units per input part image
  constexpr int SN = 3 * XN + X6_WU;                           // units per stage: [x_h | x_m | x_l | w]
  constexpr int NXI = (XN + 255) / 256;
  constexpr int NWI = (X6_WU + 255) / 256;
  extern __shared__ __align__(16) unsigned char smem_b[];
  bf16x8* s_stage = reinterpret_cast<bf16x8*>(smem_b);         // [NS][SN]
  float* s_tab = reinterpret_cast<float*>(s_stage + NS * SN);  // [NS sets][bs | bl | br][FTR][2][16] (written NS - 1 chunks ahead)
  float* s_red = s_tab + NS * 3 * FTR * COP;                   // [2 sets][NU units][4 waves][COP][2]
  float* s_ctr = s_red + 2 * NU * 4 * COP * 2;                      // [4 sets][2 half-waves][16]: ELU(bias) per channel, the centre the
                                                               // activations are stored about

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int rw = wave & 3;                                     // index inside the role
  const int half = lane >> 5, l31 = lane & 31;
  const int T = a.T, Tp = a.Tp, Fin = a.Fin, Cin = a.Cin;
  const int nchunk = Cin >> 3;

  // MISONET_TIMELINE=<Cin>: clock64() stamps of the third tile of workgroup 8 (consumer wave 0 -> slots 0.., producer
  // wave 4 -> slots 32..); experiments only
  unsigned long long* const tl = (a.dbg_buf && blockIdx.x == 8 && lane == 0 && (wave == 0 || wave == 4)) ? a.dbg_buf + (wave ? 32 : 0) : nullptr;
  int tl_i = 0;
  const unsigned tl_tile = (a.dbg >> 16) ? (unsigned)(a.dbg >> 16) : 2u;   // MISONET_WS_DEBUG bits 16+: the tile to stamp
  unsigned long long tl_base = 0;
#define STAMP(TI) do { if (tl && ((TI) == tl_tile || ((TI) == tl_tile + 1 && tl_i == 2 * nchunk)) && tl_i < 28) { const unsigned long long c_ = clock64(); if (!tl_i) { tl_base = c_; tl[28] = wall_clock64(); } tl[tl_i++] = c_ - tl_base; tl[31] = tl_i; tl[29] = wall_clock64(); } } while (0)

  // Unit of XCD locality (unit u belongs to XCD u % 8): a whole SAMPLE when the number of samples is a multiple of 8 (a.xcd
  // == 1: the per-sample folded weights and every halo are fetched into one L2 only), else a (sample, frame tile) COLUMN
  // (a.xcd == 2, u = n * ntx + tt): all row tiles and channel groups of a column share their halo rows / re-read the same
  // input, columns of one sample share only 2 of 130 staged frames -- so every XCD has work for any number of samples
  // (B = 1 of the reference harness: 6 and 2 samples; launch_conv_bf16x6 picks the mode).
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const unsigned ux = a.xcd == 2 ? (unsigned)a.ntx : 1u;                     // units per sample
  const unsigned per = (unsigned)(a.nty * a.ncg) * (a.xcd == 2 ? 1u : (unsigned)a.ntx);   // tiles per unit
  const unsigned nunit = (unsigned)a.nsamp * ux;
  const unsigned nk = ((nunit + 7u - xcd) / 8u) * per;                       // tiles of this XCD's units
  if (slot >= nk) return;
  const unsigned ntile = (nk - slot + (unsigned)nslots - 1u) / (unsigned)nslots;   // tiles of this workgroup
  const unsigned G = ntile * (unsigned)nchunk;                               // its chunks

  // G16: the channel group of a tile rotates with the workgroup's iteration (K / nslots), so that every workgroup gets
  // tiles of BOTH groups -- with the plain order a workgroup's tiles all have one (row tile, group) when nslots is a multiple of
  // nty * ncg (the 48-channel layer: 16 x 2 = 32 slots per XCD), and the two-rows-in-M tiles of the 16-channel group are a
  // third cheaper: half the workgroups would finish early and wait.  A bijection as long as a block of nslots consecutive
  // tiles holds whole (group, row tile) sets; results do not depend on which workgroup computes a tile.
  const unsigned cg_mix = (G16 && ((unsigned)nslots % (unsigned)(a.nty * a.ncg) == 0u)) ? 1u : 0u;
  int t0, f0, n, cg;
#define TILE_COORDS(K)                                                                                          \
  {                                                                                                             \
    const unsigned grp_ = (K) / per;                                                                            \
    unsigned tile_ = (K) - grp_ * per;                                                                          \
    const unsigned unit_ = grp_ * 8u + xcd;                                                                     \
    n = (int)(unit_ / ux);                                                                                      \
    f0 = (int)(tile_ % (unsigned)a.nty) * FTR;                                                                  \
    tile_ /= (unsigned)a.nty;                                                                                   \
    cg = (int)((tile_ + ((K) / (unsigned)nslots) * cg_mix) % (unsigned)a.ncg);                                   \
    t0 = (int)(tile_ / (unsigned)a.ncg + (unit_ - (unsigned)n * ux)) * TT;                                      \
  }

  if (producer) {
    // =============================================== producers ===============================================
    const unsigned P16 = (unsigned)Fin * (unsigned)Tp * 16u;                 // bytes per octet plane
    const unsigned in_rec = (unsigned)((a.in_c0 + Cin) >> 3) * P16;
    const unsigned long long part_b = (unsigned long long)(a.in_sstride >> 3) * P16;   // bytes between the parts
    const unsigned wbytes = (unsigned)nchunk * (unsigned)X6_WU * 16u;        // one (sample, group) weight image set
    const unsigned wo = (unsigned)(tid & 255) * 16u;
    unsigned wo_rm = 0x80000000u;                              // MODE 3: byte offset of this lane's compact unit in the image
    if (RM) {
      const int c = rw * 64 + lane;                            // ((kf * 3 + p) * 3 + kt) * 4 + co
      if (rw < 2 && c < X6_RM_UNITS) {
        const int co = c & 3, kt = (c >> 2) % 3, kfp = c / 12;
        wo_rm = (unsigned)(kt < 2 ? (kfp * 2 + kt) * 32 + co : X6_WPAIR + kfp * 32 + co) * 16u;
      }
    }
    const int btab_parts = nchunk >= 8 ? 4 : (nchunk >= 4 ? 2 : 1);          // conv_bf16x6_btab_parts
    __amdgpu_buffer_rsrc_t rs_x0, rs_x1, rs_x2, rs_w;
    unsigned xo[NXI];
    int xr_[NXI], xj_[NXI];                                    // staged row / slot of this lane's units (the same in every tile)
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int u = (i * 4 + rw) * 64 + lane;
      xr_[i] = u / X6_TW;
      xj_[i] = u - xr_[i] * X6_TW;
    }

#define TILE_SETUP()                                                                                            \
  {                                                                                                             \
    const int fin0_ = TR2 ? (f0 >> 1) - 1 : SF * f0 - a.padf;                                                   \
    const unsigned long long in_b_ =                                                                            \
        reinterpret_cast<unsigned long long>(a.in) + (unsigned long long)n * a.in_bstride * 4ull;               \
    rs_x0 = make_rsrc_e(in_b_, in_rec);                                                                         \
    rs_x1 = make_rsrc_e(in_b_ + part_b, in_rec);                                                                \
    rs_x2 = make_rsrc_e(in_b_ + 2 * part_b, in_rec);                                                            \
    rs_w = make_rsrc_e(reinterpret_cast<unsigned long long>(a.wps) + (unsigned long long)n * a.wps_nstride +    \
                           (unsigned long long)cg * wbytes, wbytes);                                            \
    _Pragma("unroll") for (int i = 0; i < NXI; ++i) {                                                           \
      const int fin = fin0_ + xr_[i];                                                                           \
      const int t = t0 - 1 + xj_[i];                                                                            \
      const bool ok = xr_[i] < NR && fin >= 0 && fin < Fin && t >= 0 && t < T;                                  \
      xo[i] = ok ? ((unsigned)((a.in_c0 >> 3) * Fin + fin) * (unsigned)Tp + (unsigned)t) * 16u : 0x80000000u;   \
    }                                                                                                           \
  }

    // chunk KC of the cursor's tile -> stage SB (xo walks one octet plane per chunk)
#define DMA_STAGE(KC, SB)                                                                                       \
  {                                                                                                             \
    bf16x8* st_ = s_stage + (SB) * SN;                                                                          \
    _Pragma("unroll") for (int i = 0; i < NXI; ++i) {                                                           \
      const int ub = (i * 4 + rw) * 64;                                                                         \
      /* (bottleneck variants: pieces that lie entirely in staged rows nobody reads are not fetched) */         \
      const bool need_ = BN == 0 || (((RMASK >> (ub / X6_TW)) | (RMASK >> ((ub + 63) / X6_TW))) & 1u);          \
      if (ub < XN && need_) {                                                                                   \
        if (ub + 64 <= XN || ub + lane < XN) {                                                                  \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x0, MN_LDS(st_ + ub), 16, xo[i], 0, 0, 0);                \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x1, MN_LDS(st_ + XN + ub), 16, xo[i], 0, 0, 0);           \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, MN_LDS(st_ + 2 * XN + ub), 16, xo[i], 0, 0, 0);       \
        }                                                                                                       \
      }                                                                                                         \
      if (xo[i] != 0x80000000u) xo[i] += P16;                                                                   \
    }                                                                                                           \
    const unsigned wsoff_ = (unsigned)(KC) * (unsigned)X6_WU * 16u;                                             \
    if (RM) {   /* gather the 108 units of the 4 channels into the compact image; lanes past it fetch zeros */   \
      if (rw < 2 && !(a.dbg & 128))                                                                             \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, MN_LDS(st_ + 3 * XN + rw * 64), 16, wo_rm, wsoff_, 0, 0); \
    } else                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < NWI; ++i) {                                                           \
      const int ub = (i * 4 + rw) * 64;                                                                         \
      if (ub < X6_WU && !(a.dbg & 128)) {                                                                       \
        if (ub + 64 <= X6_WU || ub + lane < X6_WU)                                                              \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, MN_LDS(st_ + 3 * XN + ub), 16, wo + (unsigned)i * 4096u, \
                                                   wsoff_, 0, 0);                                               \
      }                                                                                                         \
    }                                                                                                           \
  }

    // epilogue tables of the cursor's tile, set TS: half-wave h of producer wave rw builds output row f0 + rw + 4 h, lane & 31 =
    // output channel
    // in two parts, so that the table loads can be put in flight AHEAD of a DMA batch and consumed behind it
    float tv_[4][9], tbias_ = 0.f;
#pragma unroll
    for (int p_ = 0; p_ < 4; ++p_)
#pragma unroll
      for (int j_ = 0; j_ < 9; ++j_) tv_[p_][j_] = 0.f;
#define TABLES_LOAD()                                                                                           \
  {                                                                                                             \
    const int rr_ = rw + 4 * (lane >> 5), lc_ = lane & 31;                                                      \
    if (rr_ < FTR) {                                                                                            \
      tbias_ = a.bias[cg * COP + lc_];                                                                          \
      if (a.btab) {                                                                                             \
        const float* bt_ = a.btab + (long long)n * a.btab_nstride + (long long)(cg * COP + lc_) * 9;            \
        const long long pst_ = (long long)a.ncg * COP * 9;          /* floats between the shares of the table */  \
        /* all shares of the table in flight together (36 independent loads), summed in the fixed order p = 0..3 */ \
        _Pragma("unroll") for (int p_ = 0; p_ < 4; ++p_) {                                                      \
          const int pc_ = p_ < btab_parts ? p_ : btab_parts - 1;                                                \
          _Pragma("unroll") for (int j_ = 0; j_ < 9; ++j_) tv_[p_][j_] = bt_[pc_ * pst_ + j_];                  \
        }                                                                                                       \
      }                                                                                                         \
    }                                                                                                           \
  }
#define TABLES_FINISH(TS, CS)                                                                                   \
  {                                                                                                             \
    const int rr_ = rw + 4 * (lane >> 5), lc_ = lane & 31;                                                      \
    if (rr_ < FTR) {                                                                                            \
      const int f_ = f0 + rr_;                                                                                  \
      float b3[3] = {0.f, 0.f, 0.f};                                                                            \
      if (a.btab) {                                                                                             \
        _Pragma("unroll") for (int kf = 0; kf < 3; ++kf) {                                                      \
          bool ok_;                                                                                             \
          if (TR2) {                                                                                            \
            const int q_ = f_ + kf - 2;                                                                         \
            ok_ = (q_ >= 0) && !(q_ & 1) && (q_ >> 1) < Fin;                                                    \
          } else {                                                                                              \
            const int fi_ = SF * f_ + kf - a.padf;                                                              \
            ok_ = fi_ >= 0 && fi_ < Fin;                                                                        \
          }                                                                                                     \
          _Pragma("unroll") for (int kt = 0; kt < 3; ++kt) {                                                    \
            float v_ = tv_[0][kt * 3 + kf];                                                                     \
            _Pragma("unroll") for (int p_ = 1; p_ < 4; ++p_) v_ += p_ < btab_parts ? tv_[p_][kt * 3 + kf] : 0.f; \
            b3[kt] += ok_ ? v_ : 0.f;                                                                           \
          }                                                                                                     \
        }                                                                                                       \
      }                                                                                                         \
      b3[1] += tbias_;                                                                                          \
      /* accumulator order: channel co = (i&3) + 8*(i>>2) + 4*h  ->  h = (co>>2)&1, i = (co&3) + 4*(co>>3) */    \
      /* (MODE 3: ONE tile, register i of half h = channel i & 3 of row 2 * (i >> 2) + h; channels >= 4 go to a    */ \
      /* scratch slot of the second row set, which that mode never reads)                                          */ \
      const int slot_ = RM ? (lc_ < 4 ? (rr_ & 1) * 16 + lc_ + 4 * (rr_ >> 1) : COP + lc_)                      \
                           : (rr_ * 2 + ((lc_ >> 2) & 1)) * 16 + (lc_ & 3) + 4 * (lc_ >> 3);                    \
      float* tb_ = s_tab + (TS) * (3 * FTR * COP);                                                              \
      tb_[slot_] = b3[0] + b3[1] + b3[2];                                                                       \
      tb_[FTR * COP + slot_] = b3[0];                                                                           \
      tb_[2 * FTR * COP + slot_] = b3[2];                                                                       \
      /* centre of the stored activations per channel: ELU(bias) -- the pre-activation of normalised inputs has mean   */ \
      /* bias exactly.  (NOT the accumulator start value: bias - sum W' mean_in is far from the output when |mean_in|  */ \
      /* >> std_in.)                                                                                                    */ \
      if (rr_ == 0 && !RM) s_ctr[(CS) * COP + slot_] = elu_fast(tbias_);                                        \
    }                                                                                                           \
  }
#define TILE_TABLES(TS, CS) { TABLES_LOAD() TABLES_FINISH(TS, CS) }

    // float64 statistics of finished tile number J of this workgroup (producer wave 0): sum of the four consumer partials
    // (the partials are sums of the STORED, centred values: conv_epilogue_rows_nb's s_ctr)
#define TILE_STATS(J)                                                                                           \
  {                                                                                                             \
    if (a.act && rw == 0) {                                                                                     \
      const unsigned kj_ = slot + (unsigned)(J) * (unsigned)nslots;                                             \
      const unsigned grp_ = kj_ / per;                                                                          \
      const unsigned tile_ = kj_ - grp_ * per;                                                                  \
      const int pn_ = (int)((grp_ * 8u + xcd) / ux);                                                            \
      const int pcg_ = (int)((tile_ / (unsigned)a.nty + (kj_ / (unsigned)nslots) * cg_mix) % (unsigned)a.ncg);  \
      const int co_l = lane >> 1, which = lane & 1;                                                             \
      const int co = pcg_ * COP + co_l;                                                                         \
      if (co < a.Cout) {                                                                                        \
        _Pragma("unroll") for (int u_ = 0; u_ < NU; ++u_) {     /* each statistic unit is added exactly, on its own */ \
          const float* sr_ = s_red + (((J) & 1) * NU + u_) * (4 * COP * 2);                                     \
          float tot = 0.f;                                                                                      \
          for (int w = 0; w < 4; ++w) tot += sr_[(w * COP + co_l) * 2 + which];                                 \
          dstat_add(a.out_stats + (((long long)pn_ * a.out_sstride + a.out_c0 + co) * 2 + which) * DS_NL, (double)tot); \
        }                                                                                                       \
      }                                                                                                         \
    }                                                                                                           \
  }

    // issue cursor: the next chunk to put in flight = chunk ci_kc of this workgroup's tile number ci_t
    unsigned ci_g = 0, ci_t = 0;
    int ci_kc = 0;
    // The set-up of the COMING tile (~500 producer instructions) is spread over the three iterations before the switch:
    // the loads of its epilogue tables behind the third-to-last DMA batch of the current tile (they stay in flight across
    // the barrier), the tables themselves ahead of the second-to-last batch, its descriptors and offsets behind the last
    // one (in one piece at the switch it made the consumers wait 5.6-8.9k cycles per tile).
    bool tab_loaded = false;                                   // table loads of the coming tile are in flight
#define ISSUE_NEXT()                                                                                            \
  {                                                                                                             \
    if (ci_g < G) {                                                                                             \
      const bool nxt_ = ci_t + 1 < ntile;                                                                       \
      if (tab_loaded) {                                    /* loaded behind the previous DMA batch */           \
        TABLES_FINISH((ci_t + 1) % NS, (ci_t + 1) & 3)                                                          \
        tab_loaded = false;                                                                                     \
      }                                                                                                         \
      /* one-chunk layers: the tables of the tile being issued (first used behind the NEXT barrier).  Building the   */ \
      /* following tile's here, as the longer layers do, would overwrite the set the consumers are reading (two sets) */ \
      if (nchunk == 1 && ci_t >= 1) TILE_TABLES(ci_t % NS, ci_t & 3)                                            \
      if (!(a.dbg & 64)) DMA_STAGE(ci_kc, ci_g % NS)                                                            \
      ++ci_g;                                                                                                   \
      ++ci_kc;                                                                                                  \
      if (nxt_) {                                                                                               \
        if (nchunk >= 3 && ci_kc == nchunk - 2) {          /* behind the third-to-last DMA batch */             \
          x6_wait_vm<0>();                                 /* the DMA batch has landed: nothing to count at the barrier */ \
          TILE_COORDS(slot + (ci_t + 1) * (unsigned)nslots)                                                     \
          TABLES_LOAD()                                                                                         \
          tab_loaded = true;                                                                                    \
        }                                                                                                       \
        if (ci_kc == nchunk) {                                                                                  \
          TILE_COORDS(slot + (ci_t + 1) * (unsigned)nslots)                                                     \
          if (nchunk == 2) TILE_TABLES((ci_t + 1) % NS, (ci_t + 1) & 3)                                         \
          TILE_SETUP()                                                                                          \
        }                                                                                                       \
      }                                                                                                         \
      if (ci_kc == nchunk) { ci_kc = 0; ++ci_t; }                                                               \
    }                                                                                                           \
  }
    TILE_COORDS(slot)
    TILE_SETUP()
    TILE_TABLES(0, 0)
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) ISSUE_NEXT()
    unsigned bt = 0;                                           // tile of chunk b - 1
    int bkc = -1;                                              // its chunk index (-1 before the first barrier)
    for (unsigned b = 0; b <= G; ++b) {
      STAMP(bt);
      // everything the DMA issued has landed (hipcc does not count LDS-DMA loads); when the table loads of the coming
      // tile are in flight, the DMA batch was waited for before they were issued and they may cross the barrier
      if (!tab_loaded) x6_wait_vm<0>();
      STAMP(bt);
      __syncthreads();                                         // barrier b
      STAMP(bt);
      // the consumers finished the epilogue of tile bt - 1 before they entered the first chunk of tile bt
      if (bkc == 0 && bt >= 1) TILE_STATS(bt - 1)
      ISSUE_NEXT()                                             // chunk b + NS - 1
      STAMP(bt);
      if (bkc >= 0 && ++bkc == nchunk) { bkc = 0; ++bt; } else if (bkc < 0) bkc = 0;
    }
    __syncthreads();                                           // final barrier: the last epilogue is done
    TILE_STATS(ntile - 1)
#undef TILE_SETUP
#undef DMA_STAGE
#undef TILE_TABLES
#undef TABLES_LOAD
#undef TABLES_FINISH
#undef TILE_STATS
#undef ISSUE_NEXT
  } else {
    // =============================================== consumers ===============================================
    unsigned k = slot;
    TILE_COORDS(k)
    unsigned g = 0, ti = 0;
    __syncthreads();                                           // barrier 0: chunk 0 has landed
    for (;;) {
      f32x16 acc[RM ? 1 : FTR];
      const bool wave_live = (t0 + 32 * wave < T) && !(a.dbg & 1);   // this consumer's frames exist (ragged last tile)
      const bool g16 = G16 && (cg == a.ncg - 1);                    // uniform: this tile is the 16-channel group
      // position of the tile in the frequency axis (uniform): bit 0 = it holds output row 0 (its staged row 0 is the zero
      // padding above the image), bit 1 = it is the last row tile of an F = 8 k + 7 layer (output row 7 does not exist and
      // staged rows 8, 9 are the padding below) -- chunk_mfma6_pos compiles those steps out
      // (only for padf = 1, Fin = Fout: the same-padded convs; anything else runs the full body)
      const int tpos = (EDGE && a.padf == 1 && a.Fin == a.Fout) ? ((f0 == 0 ? 1 : 0) | ((f0 + FTR == a.Fout + 1) ? 2 : 0)) : 0;
      // (two-rows-in-M tiles use the first FTR / 2 accumulators)
      {
        const float* tb = s_tab + (ti % NS) * (3 * FTR * COP);  // accumulators start at bias + folded shift
        if constexpr (G16) {
          if (g16) conv_acc_init_rows_rm2<FTR>(acc, t0 + 32 * wave, T, lane, tb, tb + FTR * COP, tb + 2 * FTR * COP);
          else conv_acc_init_rows(acc, t0 + 32 * wave, T, lane, tb, tb + FTR * COP, tb + 2 * FTR * COP);
        } else conv_acc_init_rows(acc, t0 + 32 * wave, T, lane, tb, tb + FTR * COP, tb + 2 * FTR * COP);
      }
      for (int kc = 0; kc < nchunk; ++kc, ++g) {
        if (wave_live) {
          const bf16x8* st = s_stage + (g % NS) * SN;
          __builtin_amdgcn_s_setprio(1);
          if constexpr (RM) chunk_mfma6_rm<NR>(acc[0], st, st + 3 * XN, wave, half, l31);
          else if constexpr (G16) {
            if (g16) chunk_mfma6_rm2<NR, FTR>(acc, st, st + 3 * XN, wave, half, l31);
            else chunk_mfma6_pos<NR, SF, TR2, FTR, EDGE>(acc, st, st + 3 * XN, wave, half, l31, tpos);
          } else if constexpr (BN != 0) chunk_mfma6<NR, SF, TR2, FTR, RMASK, VR>(acc, st, st + 3 * XN, wave, half, l31);
          else chunk_mfma6_pos<NR, SF, TR2, FTR, EDGE>(acc, st, st + 3 * XN, wave, half, l31, tpos);
          __builtin_amdgcn_s_setprio(0);
        }
        STAMP(ti);
        __syncthreads();                                       // barrier g + 1: done reading chunk g
        STAMP(ti);
      }
      if constexpr (RM) {
        if (!(a.dbg & 4)) conv_epilogue_rm(a, acc[0], n, f0, t0 + 32 * wave, lane);
      } else if (!(a.dbg & 4)) {
        float* sr = s_red + (ti & 1) * (NU * 4 * COP * 2) + wave * (COP * 2);
        const float* sc = a.act ? s_ctr + (ti & 3) * COP : nullptr;
        if constexpr (OUT16)
          conv_epilogue_rows_nb<2, true>(a, acc, n, cg, f0, t0 + 32 * wave, lane, sr, FTR, sc);
        else if constexpr (G16) {
          if (g16) conv_epilogue_rows_nb<3, false, 2, true>(a, acc, n, cg, f0, t0 + 32 * wave, lane, sr, FTR, sc);
          else conv_epilogue_rows_nb<3, false, NQ>(a, acc, n, cg, f0, t0 + 32 * wave, lane, sr, FTR, sc);
        } else if constexpr (U2)
          conv_epilogue_rows_nb<3, false, NQ, false, true>(a, acc, n, cg, f0, t0 + 32 * wave, lane, sr, FTR, sc, 4 * COP * 2);
        else
          conv_epilogue_rows_nb<3, false, NQ, false, false, (BN ? VR : 0)>(a, acc, n, cg, f0, t0 + 32 * wave, lane, sr, FTR, sc);
      }
      ++ti;
      k += (unsigned)nslots;
      if (k >= nk) break;
      TILE_COORDS(k)
    }
    __syncthreads();                                           // final barrier
  }
#undef STAMP
#undef TILE_COORDS
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// conv3x3_x6_first: the network's FIRST layer (reference model.py:44, init_Conv2d_ model.py:401-406: 3x3 conv, padding
// (1, 0), no activation, no norm; 2 num_ch (+4) -> 24 channels at F = 129 -> 127) in the bf16x6 arithmetic.  Its input is the
// PLANAR float32 network input (the STFT planes that MVDR and the MISO3 assembly also read), so nothing can be staged by
// LDS-DMA: the workgroup loads the patch with plain buffer loads, splits every value EXACTLY into three bf16 pieces on the way
// into the LDS (split3_pair_t) and then runs the same tap-paired MFMA mapping as the persistent kernel (chunk_mfma6 on a 4-row
// x 128-frame tile: 108 MFMAs per 8-channel chunk and wave).  Rounds 1-3 ran this layer on the exact-f32 MFMA kernel
// (conv3x3_mfma<1, 0, 3>: 64-cycle MFMAs, 55 % busy, 1.06 + 0.41 ms per step).  The weights are NOT per sample (the input is
// consumed as it is: ident_c = Cin), so one 3-part image per chunk is packed on the host at commit (net.hip, pack_conv_w6s).
//   wimg: [nchunk][X6_WU] 16-byte units in conv_wprep6_k's image order.
// 256 threads = 4 waves x (4 output rows x 32 frames); single-buffered, 53 KB of LDS: 2-3 workgroups per CU overlap one
// workgroup's staging with another's MFMAs.  Output: oct3 (or planar) through the tile epilogue of the persistent kernel.
template <int NQ>
__global__ __launch_bounds__(256, 3) void conv3x3_x6_first(const ConvArgs a, const u32x4_t* wimg) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NR = 6, FTR = 4, COP = 32;
  constexpr int XN = NR * X6_TW;
  extern __shared__ __align__(16) unsigned char smem_b[];
  bf16x8* s_x = reinterpret_cast<bf16x8*>(smem_b);             // [3 parts][NR][X6_TW]
  bf16x8* s_w = s_x + 3 * XN;                                  // [X6_WU]
  float* s_bs = reinterpret_cast<float*>(s_w + X6_WU);         // [FTR][2][16] bias in accumulator order
  float* s_z = s_bs + FTR * COP;                               // [FTR][2][16] zeros (no folded shift: the input is not normalised)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int t0 = blockIdx.x * TT, f0 = blockIdx.y * FTR, n = blockIdx.z;
  const int T = a.T, Tp = a.Tp, Fin = a.Fin, Cin = a.Cin;
  const int nchunk = (Cin + 7) >> 3;
  if (tid < FTR * COP) {
    // slot (r, h, i) <-> channel (i & 3) + 8 (i >> 2) + 4 h of row r
    const int i = tid & 15, h = (tid >> 4) & 1;
    const int co = (i & 3) + 8 * (i >> 2) + 4 * h;
    s_bs[tid] = co < a.Cout ? a.bias[co] : 0.f;
    s_z[tid] = 0.f;
  }
  const float* in_n = a.in + (long long)n * a.in_bstride + (long long)a.in_c0 * Fin * Tp;
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc_e(reinterpret_cast<unsigned long long>(in_n), (unsigned)Cin * (unsigned)Fin * (unsigned)Tp * 4u);
  const unsigned plane_b = (unsigned)Fin * (unsigned)Tp * 4u;
  f32x16 acc[FTR];
  __syncthreads();
  conv_acc_init_rows(acc, t0 + 32 * wave, T, lane, s_bs, s_z, s_z);
  const bool wave_live = t0 + 32 * wave < T;
  for (int kc = 0; kc < nchunk; ++kc) {
    if (kc) __syncthreads();                                   // the previous chunk's images are consumed
    // ---- stage: 780 (row, frame) units x 8 channels -> three bf16 part images; out-of-image units are zeros ----
    // (round 6: the chunk's weight image goes in flight first, all pieces at once, and is written to the LDS last -- as "load, wait,
    // store" per piece behind the patch it was four more serial memory round trips per chunk: 1.18 -> 1.09 ms per step.  Double-
    // buffering the patch loads as well does not fit the 168 registers of three workgroups per CU: measured at two per CU, 1.12 ms.
    // The 16-24 bytes of scratch this costs hold staging temporaries outside the MFMA loop)
    constexpr int NIT = (XN + 255) / 256, NWI = (X6_WU + 255) / 256;
    const u32x4_t* wsrc = wimg + (long long)kc * X6_WU;
    u32x4_t wreg[NWI];
#pragma unroll
    for (int iw = 0; iw < NWI; ++iw) {                       // (in flight under the staging of the patch)
      const int uw = tid + 256 * iw;
      wreg[iw] = wsrc[uw < X6_WU ? uw : X6_WU - 1];
    }
    float v[1][8];
#define X6F_LOAD(IT, SL)                                                                                         \
    {                                                                                                            \
      const int u_ = tid + 256 * (IT);                                                                           \
      const int r_ = u_ / X6_TW, j_ = u_ - r_ * X6_TW;                                                           \
      const int fin_ = f0 - a.padf + r_, t_ = t0 - 1 + j_;                                                       \
      const bool ok_ = u_ < XN && fin_ >= 0 && fin_ < Fin && t_ >= 0 && t_ < T;                                  \
      const unsigned vo_ = ok_ ? ((unsigned)fin_ * (unsigned)Tp + (unsigned)t_) * 4u : 0x80000000u;              \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                            \
        const int c_ = kc * 8 + e;                          /* (the SGPR offset is not bounds-checked: clamp the channel) */ \
        const int cc_ = c_ < Cin ? c_ : Cin - 1;                                                                 \
        v[SL][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, vo_, (unsigned)cc_ * plane_b, 0)); \
      }                                                                                                          \
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      X6F_LOAD(it, 0)
      const int u = tid + 256 * it;
      if (u < XN) {
        u32x4_t ph, pm, pl;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          const float x0 = (kc * 8 + 2 * e2 < Cin) ? v[0][2 * e2] : 0.f, x1 = (kc * 8 + 2 * e2 + 1 < Cin) ? v[0][2 * e2 + 1] : 0.f;
          unsigned a_, b_, c_;
          split3_pair_t(x0, x1, a_, b_, c_);
          ph[e2] = a_; pm[e2] = b_; pl[e2] = c_;
        }
        reinterpret_cast<u32x4_t*>(s_x)[u] = ph;
        reinterpret_cast<u32x4_t*>(s_x)[XN + u] = pm;
        reinterpret_cast<u32x4_t*>(s_x)[2 * XN + u] = pl;
      }
    }
#undef X6F_LOAD
#pragma unroll
    for (int it = 0; it < NWI; ++it) {
      const int u = tid + 256 * it;
      if (u < X6_WU) reinterpret_cast<u32x4_t*>(s_w)[u] = wreg[it];
    }
    __syncthreads();
    if (wave_live) chunk_mfma6<NR, 1, false, FTR>(acc, s_x, s_w, wave, half, l31);
  }
  if (wave_live && !(a.dbg & 4))
    conv_epilogue_rows_nb<3, false, NQ>(a, acc, n, 0, f0, t0 + 32 * wave, lane, s_z /*unused: act = 0*/, FTR, nullptr);
#endif
}

hipError_t launch_conv_x6_first(const ConvArgs& a_in, const void* wimg, int n_samples, hipStream_t s) {
  ConvArgs a = a_in;
  if (a.in_oct || a.sf != 1 || a.tr2 || a.act || a.Cin > 16 || a.Cout > 32 || a.ident_c < a.Cin || !wimg) return hipErrorInvalidValue;
  if (a.out_oct && a.out_oct != 3) return hipErrorInvalidValue;
  static const int dbg = exp_env("MISONET_WS_DEBUG", 0);
  a.dbg = dbg;
  const dim3 grid((a.T + TT - 1) / TT, (a.Fout + 3) / 4, n_samples);
  const size_t lds = (size_t)(3 * 6 * X6_TW + X6_WU) * 16 + (size_t)(2 * 4 * 32) * sizeof(float);
  if (a.Cout <= 24 && a.out_oct == 3) hipLaunchKernelGGL((conv3x3_x6_first<3>), grid, dim3(256), lds, s, a, reinterpret_cast<const u32x4_t*>(wimg));
  else hipLaunchKernelGGL((conv3x3_x6_first<4>), grid, dim3(256), lds, s, a, reinterpret_cast<const u32x4_t*>(wimg));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_wprep6_k: fold the instance norm of a layer's input into per-sample weights, split exactly into three bf16 parts.
//   wf   : [ncg][nchunk][9 taps][32 co][8 ci] float32 (zero padded), shared by all samples; tap = kt * 3 + kf
//   wps  : [n][ncg][nchunk][X6_WU] 16-byte units: unit ((kf*3 + p)*2 + kt)*32 + co for kt < 2, X6_WPAIR + (kf*3 + p)*32 + co
//          for kt = 2, each = the 8 channels of the chunk of part p of  wf * rstd[n][ci]
//   btab : [n][nparts][ncg*32][9] float32, share p = sum over the chunks of part p of wf[..ci..] * (-mean * rstd)[n][ci]
//          (float64 accumulation, fixed order)
// One workgroup of 288 threads per (sample, group); thread = (tap, output channel).
__global__ __launch_bounds__(288) void conv_wprep6_k(const float* wf, const dstat_t* in_stats, int in_sstride, int in_c0,
                                                     int Cin, int ident_c, int Fin, int T, int nchunk, int ncg,
                                                     u32x4_t* wps, long long wps_nstride_b, float* btab,
                                                     long long btab_nstride, int nparts) {
  extern __shared__ float2 s_nrm[];                  // [nchunk*8] (scale, shift)
  const int n = blockIdx.x / ncg, cg = blockIdx.x - n * ncg;
  const int tid = threadIdx.x;
  // blockIdx.y = part: this workgroup folds the chunks [kc_lo, kc_hi) and writes ITS share of the shift table; the conv
  // kernel adds the nparts shares in a fixed order (the launch is latency-bound: 4 x the workgroups, 1/4 of the loop)
  const int per_part = (nchunk + nparts - 1) / nparts;
  const int kc_lo = blockIdx.y * per_part;
  const int kc_hi = (kc_lo + per_part) < nchunk ? (kc_lo + per_part) : nchunk;
  // the first batch of weight loads does not depend on the statistics: put it in flight AHEAD of them (one global round trip
  // less in a kernel that is two dependent round trips and a launch long)
  const int tap = tid >> 5, co = tid & 31;
  const float* wsrc = wf + ((long long)cg * nchunk * 9 + tap) * (32 * 8) + co * 8;
  constexpr int U = 4;                               // chunks per batch: the loads of a batch are issued together
  float4 wl[U][2];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int kc = (kc_lo + u < kc_hi) ? kc_lo + u : kc_hi - 1;
    const float4* src = reinterpret_cast<const float4*>(wsrc + (long long)kc * (9 * 32 * 8));
    wl[u][0] = src[0];
    wl[u][1] = src[1];
  }
  for (int c = kc_lo * 8 + tid; c < kc_hi * 8; c += 288) {
    float mean = 0.f, rstd = (c < Cin) ? 1.f : 0.f;
    if (c >= ident_c && c < Cin) {
      const dstat_t* st = in_stats + ((long long)n * in_sstride + in_c0 + c) * (2 * DS_NL);
      const double cnt = (double)Fin * (double)T;
      const double m = dstat_read(st) / cnt;
      double var = dstat_read(st + DS_NL) / cnt - m * m;
      var = var > 0.0 ? var : 0.0;
      mean = (float)m;
      rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));
    }
    s_nrm[c] = make_float2(rstd, -mean * rstd);
  }
  __syncthreads();
  const int kt = tap / 3, kf = tap - 3 * kt;
  u32x4_t* wdst = reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(wps) + (long long)n * wps_nstride_b) +
                  (long long)cg * nchunk * X6_WU;
  const int ubase = kt < 2 ? (kf * 3 * 2 + kt) * 32 + co : X6_WPAIR + (kf * 3) * 32 + co;
  const int ustep = kt < 2 ? 2 * 32 : 32;            // units between the parts
  double bsum = 0.0;
  for (int kc0 = kc_lo; kc0 < kc_hi; kc0 += U) {
    if (kc0 != kc_lo) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int kc = (kc0 + u < kc_hi) ? kc0 + u : kc_hi - 1;
        const float4* src = reinterpret_cast<const float4*>(wsrc + (long long)kc * (9 * 32 * 8));
        wl[u][0] = src[0];
        wl[u][1] = src[1];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kc = kc0 + u;
      if (kc < kc_hi) {
        const float4 w0 = wl[u][0], w1 = wl[u][1];
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        float ws[8];
        float bs = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float2 m = s_nrm[kc * 8 + e];
          ws[e] = wv[e] * m.x;
          bs = fmaf(wv[e], m.y, bs);
        }
        bsum += (double)bs;
        u32x4_t ph, pm, pl;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          unsigned a_, b_, c_;
          split3_pair_t(ws[2 * e2], ws[2 * e2 + 1], a_, b_, c_);
          ph[e2] = a_; pm[e2] = b_; pl[e2] = c_;
        }
        u32x4_t* d = wdst + (long long)kc * X6_WU + ubase;
        d[0] = ph;
        d[ustep] = pm;
        d[2 * ustep] = pl;
      }
    }
  }
  btab[(long long)n * btab_nstride + ((long long)blockIdx.y * ncg * 32 + cg * 32 + co) * 9 + tap] = (float)bsum;
}

static size_t x6_lds_bytes(int NR, int ftr, int nu = 1) {
  const int ns = 2;                                  // two stages + epilogue tables + statistics partials (nu units per tile)
  return (size_t)(ns * (3 * NR * X6_TW + X6_WU)) * 16 + (size_t)(ns * 3 * ftr * 32 + 2 * nu * 4 * 32 * 2 + 4 * 32) * sizeof(float);
}

template <int MODE, int FTR, bool OUT16 = false, int NQ = 4, bool G16 = false, bool U2 = false, int BN = 0>
static hipError_t x6_set_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16x6<MODE, FTR, OUT16, NQ, G16, U2, BN>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t conv_bf16x6_init() {
  hipError_t e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x6_first<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x6_first<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)) != hipSuccess) return e;
  if ((e = x6_set_attr<0, 4>()) != hipSuccess) return e;
  if ((e = x6_set_attr<0, 8>()) != hipSuccess) return e;
  if ((e = x6_set_attr<1, 4>()) != hipSuccess) return e;
  if ((e = x6_set_attr<2, 8>()) != hipSuccess) return e;
  if ((e = x6_set_attr<3, 8>()) != hipSuccess) return e;
  if ((e = x6_set_attr<0, 8, true>()) != hipSuccess) return e;
  if ((e = x6_set_attr<0, 8, false, 3>()) != hipSuccess) return e;
  if ((e = x6_set_attr<0, 8, false, 4, true>()) != hipSuccess) return e;
  if ((e = x6_set_attr<0, 8, false, 4, false, true>()) != hipSuccess) return e;
  if ((e = x6_set_attr<0, 4, false, 4, false, false, 1>()) != hipSuccess) return e;
  if ((e = x6_set_attr<0, 4, false, 4, false, false, 2>()) != hipSuccess) return e;
  if ((e = x6_set_attr<0, 4, true>()) != hipSuccess) return e;
  return x6_set_attr<2, 4>();
}

long long conv_bf16x6_wps_bytes(int Cin, int Cout) {
  return (long long)((Cout + 31) / 32) * (Cin / 8) * X6_WU * 16;
}

int conv_bf16x6_btab_parts(int Cin) {                // shares of the shift table (= workgroups per (sample, group))
  const int nchunk = Cin >> 3;
  return nchunk >= 8 ? 4 : (nchunk >= 4 ? 2 : 1);
}

hipError_t launch_conv_wprep6(const ConvArgs& a, const float* wf, int n_samples, hipStream_t s) {
  const int nchunk = a.Cin >> 3;
  const int nparts = conv_bf16x6_btab_parts(a.Cin);
  hipLaunchKernelGGL(conv_wprep6_k, dim3(n_samples * a.ncg, nparts), dim3(288), (size_t)nchunk * 8 * sizeof(float2), s, wf,
                     a.in_stats, a.in_sstride, a.in_c0, a.Cin, a.ident_c, a.Fin, a.T, nchunk, a.ncg,
                     reinterpret_cast<u32x4_t*>(const_cast<void*>(a.wps)), a.wps_nstride, const_cast<float*>(a.btab),
                     a.btab_nstride, nparts);
  return hipGetLastError();
}

hipError_t launch_conv_bf16x6(const ConvArgs& a_in, int n_samples, hipStream_t s) {
  ConvArgs a = a_in;
  if (a.in_oct != 3 || !a.wps || a.cop != 32 || (a.Cin & 7) || (a.in_c0 & 7) || (a.in_sstride & 7)) return hipErrorInvalidValue;
  if (a.out_oct && ((a.out_oct != 3 && a.out_oct != 4) || (a.Cout & 7) || (a.out_c0 & 7) || (a.out_sstride & 7))) return hipErrorInvalidValue;
  if (a.out_oct == 4 && (a.tr2 || a.sf != 1 || a.descale != 1.f)) return hipErrorInvalidValue;   // stride-1 layers only (see OUT16)
  // measurement hooks: read once per process (thread-safe function-local statics)
  static const int dbg = exp_env("MISONET_WS_DEBUG", 0);
  a.dbg = dbg;
  a.dbg_buf = nullptr;
  static const int tl_env = exp_env("MISONET_TIMELINE", 0);
  // MISONET_TIMELINE_F / _MODE: which layer (defaults: F = 63, stride 1); the timeline itself is a single-threaded experiment
  static const int tl_f = exp_env("MISONET_TIMELINE_F", 63);
  static const int tl_mode = exp_env("MISONET_TIMELINE_MODE", 0);
  static int tl_done = 0;
  static unsigned long long* tl_buf = nullptr;
  const bool do_tl = tl_env && tl_done < 2 && (a.tr2 ? 2 : (a.sf == 2 ? 1 : 0)) == tl_mode && a.Cin == tl_env && a.Fout == tl_f && n_samples >= 8;
  if (do_tl) {
    if (!tl_buf && hipMalloc(reinterpret_cast<void**>(&tl_buf), 64 * 8) != hipSuccess) tl_buf = nullptr;
    if (tl_buf) { (void)hipMemsetAsync(tl_buf, 0, 64 * 8, s); a.dbg_buf = tl_buf; }
  }
  const int mode = a.tr2 ? 2 : (a.sf == 2 ? 1 : 0);
  // tile geometry: 128 frames x 8 rows for the stride-1 layers with more than 4 rows (10 staged rows per 8 instead of
  // 6 per 4 and one weight image per 216 instead of 108 MFMAs: 25 % fewer staged bytes per MFMA), else x 4 rows
  static const int ft8 = exp_env("MISONET_X6_ROWS8", 3);   // bit 0: stride-1, bit 1: transposed
  // rows-in-M tiles (MODE 3) for the raw 4-channel output layer (MISONET_X6_RM=0: the 32-channel tiles, for A/B runs)
  static const int rm_env = exp_env("MISONET_X6_RM", 1);
  const bool rows_in_m = rm_env && mode == 0 && a.padf == 2 && !a.act && !a.out_oct && a.Cout <= 4 && a.ncg == 1 && a.Fout > 4;
  int ftr = rows_in_m ? 8 : ((mode != 1 && a.Fout > 4 && (ft8 & (mode == 0 ? 1 : 2))) ? 8 : 4);
  // "flexible" layers: stride-1, oct3 in and out, 4 < F <= 31.  Their 8-row kernel forms the statistics per half tile (U2), so
  // 4-row tiles give the same bits: the launch takes 4-row tiles when 8-row tiles would leave CUs without work (one utterance:
  // 48 frame-tile columns x ceil(F / 8) row tiles).  MISONET_X6_FLEX=0: always 8-row tiles with one statistic unit (A/B runs).
  static const int flex_env = exp_env("MISONET_X6_FLEX", 1);
  const bool flex = flex_env && mode == 0 && ftr == 8 && !rows_in_m && a.out_oct == 3 && a.Fout <= 31 && (a.Cout & 31) == 0;
  if (flex) {
    const int g_cus_ = device_cus();
    const long long ntx_ = (a.T + TT - 1) / TT;
    // rounds of tiles per CU x relative cost of a tile (a 4-row tile takes ~0.54 of an 8-row tile's time: 108 instead of
    // 216 MFMAs per chunk at a quarter more staged bytes per MFMA): the geometry with the shorter critical path.  Batch 16
    // (thousands of tiles) always takes 8-row tiles; one to three utterances take 4-row tiles on most of these layers.
    const long long tiles8 = (long long)n_samples * ntx_ * ((a.Fout + 7) / 8) * a.ncg;
    const long long tiles4 = (long long)n_samples * ntx_ * ((a.Fout + 3) / 4) * a.ncg;
    if (g_cus_ > 0) {
      const long long r8 = (tiles8 + g_cus_ - 1) / g_cus_, r4 = (tiles4 + g_cus_ - 1) / g_cus_;
      if (r4 * 54 < r8 * 100) ftr = 4;
    }
  }
  (void)conv_grid(a, n_samples, TT, ftr, 1);
  if (n_samples % 8) a.xcd = 2;                                    // columns, not samples, are dealt to the XCDs
  const int g_cus = device_cus();
  if (g_cus <= 0) return hipErrorUnknown;
  // persistent launch: one workgroup per CU, capped by the largest per-XCD tile list
  const long long nk_max = a.xcd == 2 ? (long long)((n_samples * a.ntx + 7) / 8) * a.nty * a.ncg
                                      : (long long)(n_samples / 8) * a.ntx * a.nty * a.ncg;
  int nslots = g_cus / 8;
  if (nslots < 1) nslots = 1;
  if (nslots > nk_max) nslots = (int)nk_max;
  {
    static const int cap = exp_env("MISONET_X6_SLOTS", 0);   // workgroups per XCD (experiments)
    if (cap > 0 && nslots > cap) nslots = cap;
  }
  const dim3 pgrid((unsigned)(8 * nslots), 1, 1);
  // <= 24 output channels in one group: the epilogue variant that skips the padded register quad (MISONET_X6_Q3=0: A/B runs)
  static const int q3_env = exp_env("MISONET_X6_Q3", 1);
  const bool q3 = q3_env && a.ncg == 1 && a.Cout <= 24 && a.out_oct == 3;
  // the F = 1 bottleneck pair on their reduced 4-row tiles (MISONET_X6_BN=0: the full tiles, for A/B runs)
  static const int bn_env = exp_env("MISONET_X6_BN", 1);
  const int bn = (!bn_env || mode != 0 || ftr != 4 || a.out_oct != 3 || a.nty != 1) ? 0
                 : ((a.Fin == 3 && a.Fout == 1 && a.padf == 0) ? 1 : ((a.Fin == 1 && a.Fout == 3 && a.padf == 2) ? 2 : 0));
  // Cout % 32 == 16 (the 48-channel conv of the last decoder's dense block): its 16-channel group as two rows in M
  // (MISONET_X6_G16=0: the padded 32-channel tiles, for A/B runs)
  static const int g16_env = exp_env("MISONET_X6_G16", 1);
  const bool g16 = g16_env && mode == 0 && ftr == 8 && a.out_oct == 3 && a.ncg >= 2 && (a.Cout & 31) == 16;
  if (g16) hipLaunchKernelGGL((conv3x3_bf16x6<0, 8, false, 4, true>), pgrid, dim3(512), x6_lds_bytes(10, 8), s, a, nslots);
  else if (flex && ftr == 8) hipLaunchKernelGGL((conv3x3_bf16x6<0, 8, false, 4, false, true>), pgrid, dim3(512), x6_lds_bytes(10, 8, 2), s, a, nslots);
  else if (a.out_oct == 4 && ftr == 8) hipLaunchKernelGGL((conv3x3_bf16x6<0, 8, true>), pgrid, dim3(512), x6_lds_bytes(10, 8), s, a, nslots);
  else if (a.out_oct == 4) hipLaunchKernelGGL((conv3x3_bf16x6<0, 4, true>), pgrid, dim3(512), x6_lds_bytes(6, 4), s, a, nslots);
  else if (rows_in_m) hipLaunchKernelGGL((conv3x3_bf16x6<3, 8>), pgrid, dim3(512), x6_lds_bytes(10, 8), s, a, nslots);
  else if (mode == 0 && ftr == 8 && q3) hipLaunchKernelGGL((conv3x3_bf16x6<0, 8, false, 3>), pgrid, dim3(512), x6_lds_bytes(10, 8), s, a, nslots);
  else if (mode == 0 && ftr == 8) hipLaunchKernelGGL((conv3x3_bf16x6<0, 8>), pgrid, dim3(512), x6_lds_bytes(10, 8), s, a, nslots);
  else if (mode == 0 && bn == 1) hipLaunchKernelGGL((conv3x3_bf16x6<0, 4, false, 4, false, false, 1>), pgrid, dim3(512), x6_lds_bytes(6, 4), s, a, nslots);
  else if (mode == 0 && bn == 2) hipLaunchKernelGGL((conv3x3_bf16x6<0, 4, false, 4, false, false, 2>), pgrid, dim3(512), x6_lds_bytes(6, 4), s, a, nslots);
  else if (mode == 0) hipLaunchKernelGGL((conv3x3_bf16x6<0, 4>), pgrid, dim3(512), x6_lds_bytes(6, 4), s, a, nslots);
  else if (mode == 1) hipLaunchKernelGGL((conv3x3_bf16x6<1, 4>), pgrid, dim3(512), x6_lds_bytes(9, 4), s, a, nslots);
  else if (ftr == 8) hipLaunchKernelGGL((conv3x3_bf16x6<2, 8>), pgrid, dim3(512), x6_lds_bytes(5, 8), s, a, nslots);
  else hipLaunchKernelGGL((conv3x3_bf16x6<2, 4>), pgrid, dim3(512), x6_lds_bytes(3, 4), s, a, nslots);
  if (do_tl && tl_buf) {
    unsigned long long h[64];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, tl_buf, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[timeline-x6] Cin=%d Fout=%d n=%d consumer(%llu):", a.Cin, a.Fout, n_samples, h[31]);
    for (unsigned long long i = 0; i < h[31] && i < 28; ++i) fprintf(stderr, " %llu", h[i]);
    if (h[31] > 1 && h[29] > h[28])                                // shader cycles per 100 MHz wall-clock tick
      fprintf(stderr, "\n[timeline-x6] shader clock over the stamped tile: %.3f GHz", 0.1 * (double)h[h[31] - 1] / (double)(h[29] - h[28]));
    fprintf(stderr, "\n[timeline-x6] producer(%llu):", h[63]);
    for (unsigned long long i = 0; i < h[63] && i < 28; ++i) fprintf(stderr, " %llu", h[32 + i]);
    fprintf(stderr, "\n");
    ++tl_done;
  }
  return hipGetLastError();
}

}  // namespace mn
