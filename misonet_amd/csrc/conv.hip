// 3x3 convolution family of the Dense-U-Net (reference model.py:401-482) as ONE implicit-GEMM kernel on the
// gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, bitwise an fmaf chain).
//
//   out[co][f][t] = bias[co] + sum_{ci,kt,kf} W[co][ci][kt][kf] * norm(in)[ci][fin(f,kf)][t - 1 + kt]
//
// GEMM roles: M = output channels (A operand = weights), N = 32 consecutive frames t (B operand = input tile),
// K = (tap, input channel) with two adjacent input channels per MFMA.  With t on the N axis every accumulator
// register holds 32 consecutive frames of one channel, so the epilogue stores are 128-byte coalesced along T.
//
// Workgroup = 4 waves = 4 output rows (frequency bins) x 128 frames x COP output channels; wave w owns row f0+w
// and NCO x 4 accumulator tiles.  Per K-chunk (8 input channels) the block stages the normalised input patch
// [8][NR][136] and the weight slab [9][8][COP] in LDS.  Covered layers (all 3x3, frame stride 1, frame padding 1):
//   Conv2d  s(1,1) p(1,1)  dense-block convs        model.py:442-466      sf=1 padf=1
//   Conv2d  s(1,1) p(1,0)  first conv, encoder 6    model.py:44,50        sf=1 padf=0
//   Conv2d  s(1,2) p(1,0)  down-sampling            model.py:47,52        sf=2 padf=0
//   ConvTranspose2d s(1,1) p(1,0)                   model.py:64,69        conv form with flipped taps, padf=2
//   ConvTranspose2d s(1,2) p(1,0)                   model.py:67,71        tr2: flipped taps, fin=(f+kf-2)/2 if even
// Epilogue: + bias, ELU (model.py:412,429,444), raw store, per-(n,co) sum / sum^2 for the instance norm that the
// NEXT layer applies while staging (model.py:413,430,445).
#include "kernels.hpp"
#include <stdlib.h>
#include "conv_epilogue.hpp"

#include <atomic>
#include <type_traits>

namespace mn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

int conv_cop(int Cout) { return Cout <= 32 ? 32 : 64; }
int conv_rows(int sf, int tr2) { return tr2 ? 3 : sf * (FT - 1) + 3; }   // (the staged rows of the row-per-wave geometry: conv_bf16*.hip)
// staged input rows of conv3x3_mfma: MODE 2 (stride-2 transposed) runs a PAIR of output rows per wave = 8 output rows per
// workgroup, which need input rows m0 - 1 .. m0 + 3; MODE 3 (stride-1 transposed on ONE input row) stages that row only
static int f32_rows(const ConvArgs& a) { return a.tr2 ? 5 : a.sf * (FT - 1) + 3; }

// MFMA work of one K-chunk for one wave: one output row, four 32-frame column tiles, NCO 32-channel row tiles.
// The (tap, channel-pair) steps are flattened into one fully unrolled sequence with an explicit two-deep operand
// pipeline: the ds_reads of step k+1 are issued before the MFMAs of step k, and a sched_barrier per step keeps the
// compiler from hoisting more (which would spill).  All LDS addresses are two per-lane bases + immediates.
// KFMASK selects the frequency taps (all three for convs; {1} / {0,2} for the odd / even rows of a stride-2
// transposed conv).  The kernel always computes all four column tiles: frames >= T are staged as zeros, so a ragged
// last tile costs MFMAs, not correctness (T = 1001 and T = 501 both end in a tile that needs four anyway).
// NCP: channel pairs of the chunk that hold channels (4 = all of CK; 2 for a last chunk with <= 4 channels -- the 12-channel
// network input (model.py:44) leaves half of its second chunk empty: a quarter of that layer's MFMAs).
// ROW0: every selected tap reads staged row 0 (MODE 3: the layer has one input row).
template <int NCO, int NR, int SF, bool TR2, int KFMASK, int NCP = CK / 2, bool ROW0 = false>
__device__ __forceinline__ void chunk_mfma(f32x16 (&acc)[NCO][4], const float* s_in, const float* s_w, int frel,
                                           int half, int l31) {
  constexpr int COP = NCO * 32;
  constexpr int NKF = ((KFMASK >> 0) & 1) + ((KFMASK >> 1) & 1) + ((KFMASK >> 2) & 1);
  constexpr int NSTEP = 3 * NKF * NCP;
  const float* wbase = s_w + half * COP + l31;
  // one input-row base per selected kf
  const float* ibase[3];
#pragma unroll
  for (int kf = 0; kf < 3; ++kf) {
    const int rl = ROW0 ? 0 : (TR2 ? ((frel + kf) >> 1) : (SF * frel + kf));
    ibase[kf] = s_in + (half * NR + rl) * TW + l31 + 3;
  }
  float av[2][NCO], bv[2][4];
#define MN_LOAD(ST, BUF)                                                                        \
  {                                                                                             \
    constexpr int tap_ = (ST) / NCP, cp_ = (ST) % NCP;                                          \
    constexpr int kt_ = tap_ / NKF, ks_ = tap_ % NKF;                                           \
    constexpr int kf_ = (NKF == 3) ? ks_ : (NKF == 1 ? (KFMASK == 1 ? 0 : (KFMASK == 2 ? 1 : 2)) : (ks_ == 0 ? 0 : 2)); \
    _Pragma("unroll") for (int j = 0; j < NCO; ++j)                                             \
        av[BUF][j] = wbase[((kt_ * 3 + kf_) * CK + cp_ * 2) * COP + j * 32];                    \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                               \
        bv[BUF][q] = ibase[kf_][cp_ * 2 * NR * TW + q * 32 + kt_];                              \
  }
  MN_LOAD(0, 0)
#pragma unroll
  for (int st = 0; st < NSTEP; ++st) {
    const int cur = st & 1;
    if (st + 1 < NSTEP) {
      // (the macro needs a constant: unrolled loop index)
      const int nx = st + 1;
      const int tap_ = nx / NCP, cp_ = nx % NCP;
      const int kt_ = tap_ / NKF, ks_ = tap_ % NKF;
      const int kf_ = (NKF == 3) ? ks_ : (NKF == 1 ? (KFMASK == 1 ? 0 : (KFMASK == 2 ? 1 : 2)) : (ks_ == 0 ? 0 : 2));
#pragma unroll
      for (int j = 0; j < NCO; ++j) av[cur ^ 1][j] = wbase[((kt_ * 3 + kf_) * CK + cp_ * 2) * COP + j * 32];
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[cur ^ 1][q] = ibase[kf_][cp_ * 2 * NR * TW + q * 32 + kt_];
    }
    __builtin_amdgcn_sched_barrier(0);     // keep the next step's ds_reads AHEAD of this step's MFMAs
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < NCO; ++j)
        acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][j], bv[cur][q], acc[j][q], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef MN_LOAD
}

// ---- W1D: the frequency-strided layers in 1-D Winograd F(2, 3) form along T (f32w mode, round 6) --------------------------
// Every layer of the network has frame stride 1, so the three time taps of the stride-(1,2) convs and transposed convs
// (model.py:47,52,67,71; the 2-D form of conv_wino.hip needs stride 1 in both directions) cost 4 instead of 6 products per two
// output frames:  with d_i = x[2p - 1 + i],  V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3),  U = (g0, (g0 + g1 + g2) / 2,
// (g0 - g1 + g2) / 2, g2) per (co, ci, kf),  M_nu = sum U_nu V_nu,  y[2p] = M0 + M1 + M2,  y[2p + 1] = M1 - M2 - M3.
// GEMM roles as in chunk_mfma with N = 32 frame PAIRS (64 frames) per column tile: a lane reads its pair's four inputs from
// the staged tile (frame 2p - 1 sits at the ODD column 2p + 3 of the [.., TW] rows: 4-byte aligned, two ds_read2_b32), forms V in
// registers (4 adds) and feeds 4 MFMAs per (kf, channel pair).
// (The operand reads of step k + 1 -- raw frames and weights from the LDS -- are issued BEFORE the MFMAs of step k, as in
// chunk_mfma: a sched_barrier per step keeps the compiler from sinking them back behind the matrix work.)
struct W1dRaw { float d[4]; };
__device__ __forceinline__ W1dRaw w1d_raw(const float* p) { W1dRaw r; r.d[0] = p[0]; r.d[1] = p[1]; r.d[2] = p[2]; r.d[3] = p[3]; return r; }
__device__ __forceinline__ void w1d_vr(const W1dRaw& r, float (&v)[4]) {
  v[0] = r.d[0] - r.d[2]; v[1] = r.d[1] + r.d[2]; v[2] = r.d[2] - r.d[1]; v[3] = r.d[1] - r.d[3];
}
// conv with frequency stride SF (2: the down-sampling layers; 1: the network's first layer, round 6): one output row per wave, 128
// frames = 2 column tiles; s_w = [nu * 3 + kf][CK][32].  NCP: channel pairs of the chunk that hold channels (chunk_mfma)
// NKF: frequency taps the wave runs (3; 1 for MODE 3, where output row f reads the single input row through tap 2 - f only: the
// caller hands in s_w advanced to that tap)
template <int NR, int SF = 2, int NCP = CK / 2, int CP0 = 0, int NKF = 3>
__device__ __forceinline__ void chunk_w1d_s2(f32x16 (&acc)[4][2], const float* s_in, const float* s_w, int frel, int half, int l31) {
  const float* wb = s_w + half * 32 + l31;
  const float* ib0 = s_in + (half * NR + SF * frel) * TW + 2 * l31 + 3;
  constexpr int NSTEP = NKF * NCP;
  W1dRaw r0[2], r1[2];
  float u[2][4];
#define W1D_LOAD(ST, BUF)                                                                        \
  {                                                                                              \
    constexpr int kf_ = (ST) / NCP, cp_ = CP0 + (ST) % NCP;                                      \
    const float* ib_ = ib0 + kf_ * TW + cp_ * 2 * NR * TW;                                       \
    r0[BUF] = w1d_raw(ib_); r1[BUF] = w1d_raw(ib_ + 64);                                         \
    _Pragma("unroll") for (int nu = 0; nu < 4; ++nu) u[BUF][nu] = wb[((nu * 3 + kf_) * CK + cp_ * 2) * 32]; \
  }
  W1D_LOAD(0, 0)
#pragma unroll
  for (int st = 0; st < NSTEP; ++st) {
    const int cur = st & 1;
    if (st + 1 < NSTEP) {
      const int kf_ = (st + 1) / NCP, cp_ = CP0 + (st + 1) % NCP;
      const float* ib_ = ib0 + kf_ * TW + cp_ * 2 * NR * TW;
      r0[cur ^ 1] = w1d_raw(ib_); r1[cur ^ 1] = w1d_raw(ib_ + 64);
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) u[cur ^ 1][nu] = wb[((nu * 3 + kf_) * CK + cp_ * 2) * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
    float v0[4], v1[4];
    w1d_vr(r0[cur], v0);
    w1d_vr(r1[cur], v1);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      acc[nu][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[cur][nu], v0[nu], acc[nu][0], 0, 0, 0);
      acc[nu][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[cur][nu], v1[nu], acc[nu][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef W1D_LOAD
}
// stride-2 transposed conv: a wave owns the output row pair (2m, 2m + 1) over ONE column tile (64 frames): staged rows
// pr (= input row m - 1) and pr + 1 (= m); even row: taps kf = 0 / 2 of them, odd row: tap kf = 1 of row m -- 12 MFMAs per
// channel pair, the same for every wave; the V of row m serves both output rows
// (acc[nu][0]: the even row, acc[nu][1]: the odd row)
__device__ __forceinline__ void chunk_w1d_tr2(f32x16 (&acc)[4][2], const float* s_in, const float* s_w, int pr, int q, int half, int l31) {
  constexpr int NR = 3;
  const float* wb = s_w + half * 32 + l31;
  const float* ib = s_in + (half * NR + pr) * TW + 2 * l31 + 3 + 64 * q;
  W1dRaw ra[2], rb[2];
  float u0[2][4], u1[2][4], u2[2][4];
#define W1D_LOAD(CP, BUF)                                                                        \
  {                                                                                              \
    ra[BUF] = w1d_raw(ib + (CP) * 2 * NR * TW); rb[BUF] = w1d_raw(ib + (CP) * 2 * NR * TW + TW); \
    _Pragma("unroll") for (int nu = 0; nu < 4; ++nu) {                                           \
      u0[BUF][nu] = wb[((nu * 3 + 0) * CK + (CP) * 2) * 32];                                     \
      u1[BUF][nu] = wb[((nu * 3 + 1) * CK + (CP) * 2) * 32];                                     \
      u2[BUF][nu] = wb[((nu * 3 + 2) * CK + (CP) * 2) * 32];                                     \
    }                                                                                            \
  }
  W1D_LOAD(0, 0)
#pragma unroll
  for (int cp = 0; cp < CK / 2; ++cp) {
    const int cur = cp & 1;
    if (cp + 1 < CK / 2) W1D_LOAD(cp + 1, cur ^ 1)
    __builtin_amdgcn_sched_barrier(0);
    float va[4], vb[4];
    w1d_vr(ra[cur], va);
    w1d_vr(rb[cur], vb);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      acc[nu][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0[cur][nu], va[nu], acc[nu][0], 0, 0, 0);
      acc[nu][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1[cur][nu], vb[nu], acc[nu][1], 0, 0, 0);
      acc[nu][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2[cur][nu], vb[nu], acc[nu][0], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef W1D_LOAD
}
// tile epilogue of the W1D forms: inverse transform, + bias, ELU, centring (conv_epilogue.hpp), 8-byte stores of the lane's frame
// pair, statistics partials of this (row, column-tile range) into s_red [32][2].  Accumulator columns Q0 .. Q0 + NQ - 1 are
// NQ consecutive column tiles of row f starting at frame tq.  The layers are activated ones (launch_conv): ELU and statistics
// are unconditional -- a run-time `act` inside the unrolled loops is a branch per element -- and (bias, ELU(bias)) come from
// the LDS table s_bc [32] the kernel fills at its start (round 6: a global load per accumulator row, each waited for with
// vmcnt(0), was a third of these kernels' time).  Channels >= Cout need no mask: zero weights and bias, ELU(0) - ELU(0) = 0.
template <int Q0, int NQ, bool ACT = true>
__device__ __forceinline__ void w1d_epilogue(const ConvArgs& a, f32x16 (&acc)[4][2], int n, int cg, int f, int tq, bool row_ok, int lane,
                                             float* s_red, const float2* s_bc) {
  const int half = lane >> 5, l31 = lane & 31;
  const int T = a.T, Tp = a.Tp;
  const int cbase = cg * 32;
  const unsigned P4 = (unsigned)a.Fout * (unsigned)Tp * 4u;
  const float* ob = a.out + (long long)n * a.out_bstride + (long long)a.out_c0 * a.Fout * Tp;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc_e(reinterpret_cast<unsigned long long>(ob), (unsigned)a.Cout * P4);
  const bool all_t = row_ok && (tq + 64 * NQ <= T);          // uniform: every frame of the range exists
  float2 bc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bc[r] = s_bc[(r & 3) + 8 * (r >> 2) + 4 * half];
  unsigned voff[NQ];
  bool ok[NQ], ok2[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int te = tq + 64 * q + 2 * l31;
    ok[q] = row_ok && te < T;                                // (te + 1 >= T: the second word lands in the row's padding [T, Tp))
    ok2[q] = ok[q] && te + 1 < T;
    voff[q] = ok[q] ? (unsigned)(f * Tp + te) * 4u + (unsigned)(4 * half) * P4 : 0x80000000u;
  }
  float s1[16], s2[16];
  // (the mask-free body for tiles whose frames all exist -- every tile but the last of a row -- is a separate copy: a uniform
  // branch per tile instead of two selects per accumulator row)
  auto body = [&](auto masked) {
    constexpr bool MASKED = decltype(masked)::value;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kr = (r & 3) + 8 * (r >> 2);
      const unsigned coff = (unsigned)(cbase + kr) * P4;         // uniform plane offset
      const float b = bc[r].x;
      const f32x2_t cr = {bc[r].y, bc[r].y};
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float m0 = acc[0][Q0 + q][r], m1 = acc[1][Q0 + q][r], m2 = acc[2][Q0 + q][r], m3 = acc[3][Q0 + q][r];
        f32x2_t y = {(m0 + m1) + m2 + b, (m1 - m2) - m3 + b};
        if (ACT) y = elu_fast2(y) - cr;
        typedef unsigned int uu2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uu2, y), rs, voff[q] + coff, 0, 0);
        if (!ACT) continue;                                      // (a layer without activation has no statistics: its consumers read it raw)
        const float ze = (!MASKED || ok[q]) ? y.x : 0.f, zo = (!MASKED || ok2[q]) ? y.y : 0.f;
        a1 += ze + zo;
        a2 = fmaf(ze, ze, fmaf(zo, zo, a2));
      }
      s1[r] = a1;
      s2[r] = a2;
    }
  };
  if (all_t || !ACT) body(std::false_type{}); else body(std::true_type{});
  if (!ACT) return;
  const float x1 = reduce16_halfwave(s1, lane);
  const float x2 = reduce16_halfwave(s2, lane);
  if ((lane & 16) == 0) {
    const int qq = lane & 15;
    const int co_l = (qq & 3) + 8 * (qq >> 2) + 4 * half;
    s_red[co_l * 2 + 0] = x1;
    s_red[co_l * 2 + 1] = x2;
  }
}

// MODE 0: forward / stride-1-transposed conv (sf = 1, NR = 6 staged rows); MODE 1: stride-2 conv (NR = 9);
// MODE 2: stride-2 transposed conv: a wave owns the output row PAIR (2m, 2m + 1) -- the even row takes taps kf = 0, 2 of input
//   rows m - 1, m, the odd row tap kf = 1 of row m: 9 tap steps per wave and chunk, the same for every wave (one output row
//   per wave gave the even rows 6 steps and the odd rows 3: every barrier waited for the even rows, 75 % of the matrix time at
//   best).  Workgroup = 8 output rows, NR = 5 staged rows, 32-channel groups only (two rows x 4 column tiles = 128 registers).
// MODE 3: stride-1 transposed conv on ONE input row (decoder 0 behind the F = 1 bottleneck, model.py:64): output row f reads
//   the row through tap kf = 2 - f only; NR = 1, one tap step per wave instead of three (two of them on staged zeros).
// MODE 4 (W1D only, round 6): a conv whose output is ONE row (encoder 6, 64 -> 128 channels on F = 3 -> 1, model.py:50): the four
//   waves take a 32-channel group each on one staged tile -- NR = 3 staged rows = the three frequency taps, whatever the stride;
//   weight slab [4 groups][12][CK][32], statistics slot per group.  (A row per wave left three of four waves idle.)
//
// Software pipeline per K-chunk (guide T14, "issue early / write late"): the global loads of chunk k+1 (NR float4 of
// the input patch + 1 halo scalar + the weight slab share per thread) are issued into registers BEFORE the MFMA loop
// of chunk k and are normalised and written to LDS after it, so HBM/L2 latency hides under the matrix work.
// Staging roles are division-free: thread (q = tid & 31, ci = tid >> 5) owns frames t0+4q..t0+4q+3 of channel ci
// for every staged row; threads < 16*NR own the two halo frames t0-1 / t0+128 of one (row, channel).
// OCTP = 3 / 4: the instantiations that write the bf16x6 (hi | mid | lo bf16 parts) / f16x3 (hi | lo fp16 parts) oct layout:
// the planar-input layers in front of a dense block when the network runs in one of those modes.
// HALFK: the instantiation for a layer whose LAST K-chunk holds <= CK/2 channels (the 12-channel network input, model.py:44):
// that chunk runs half the channel pairs.  It is a separate instantiation because a second fully unrolled chunk_mfma body
// inside the common kernel costs every MODE 0 instantiation its register allocation (round 3: 74-241 spilled VGPRs, f32
// mode -36 %); tests/test_build_resources.py holds the hot instantiations to ScratchSize == 0.
// W1D: MODE 0 (a network's first layer: <= 16 input channels, no activation), 1, 2, 3 and 4 in the 1-D Winograd form above (a.w1d
//   image, 12 taps = 4 positions x 3 kf; 32-channel groups; planar output).  MODE 2 then keeps the 4-row tile (NR = 3): wave w owns
//   row pair w >> 1 over column tile w & 1.  MODE 0 + HALFK: the half-empty chunk runs the first of two half bodies.
template <int NCO, int MODE, int OCTP = 0, bool HALFK = false, bool W1D = false>
__global__ __launch_bounds__(256, ((NCO == 1 && MODE != 2 && OCTP == 0 && !HALFK && !W1D) ? 3 : 2)) void conv3x3_mfma(const ConvArgs a) {
  static_assert(MODE != 2 || NCO == 1, "stride-2 transposed: two output rows per wave, 32-channel groups");
  static_assert(MODE != 4 || W1D, "MODE 4 exists in the W1D form only");
  static_assert(!W1D || (NCO == 1 && OCTP == 0 && (MODE == 0 || (!HALFK && (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 4)))),
                "W1D: the frequency-strided layers and the first layer, 32-channel groups, planar output");
  constexpr int COP = NCO * 32;
  constexpr int NR = MODE == 0 ? 6 : (MODE == 1 ? 9 : (MODE == 2 ? (W1D ? 3 : 5) : (MODE == 4 ? 3 : 1)));
  constexpr int SF = (MODE == 1 || MODE == 4) ? 2 : 1;
  constexpr int NG = MODE == 4 ? 4 : 1;          // 32-channel groups per workgroup (MODE 4: one per wave)
  constexpr bool TR2 = MODE == 2;
  constexpr int FTO = (TR2 && !W1D) ? 2 * FT : FT;   // output rows per workgroup
  constexpr int NTAP = W1D ? 12 : 9;
  constexpr int NW4 = NTAP * CK * COP * NG / 4;  // float4 per weight slab
  constexpr int NWI = (NW4 + 255) / 256;
  extern __shared__ __align__(16) float smem[];
  float* s_in = smem;                         // [CK][NR][TW]: col 3 = frame t0-1, cols 4..131 = t0..t0+127, col 132 = t0+128
  float* s_w = s_in + CK * NR * TW;           // [NTAP][CK][COP]
  float2* s_nrm = reinterpret_cast<float2*>(s_w + NTAP * CK * COP * NG);   // [CinP] (mean, rstd)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ConvTile ct = conv_tile(a);
  if (!ct.valid) return;
  const int t0 = ct.t_tile * TT;
  const int f0 = ct.f_tile * FTO;
  const int n = ct.n;
  const int cg = ct.cg;
  const int T = a.T, Tp = a.Tp, Fin = a.Fin, Cin = a.Cin;
  const int nchunk = (Cin + CK - 1) / CK;
  const int fin0 = MODE == 3 ? 0 : (TR2 ? (f0 >> 1) - 1 : SF * f0 - a.padf);

  const float* in_n = a.in + (long long)n * a.in_bstride + (long long)a.in_c0 * Fin * Tp;
  const f32x4* w_g = reinterpret_cast<const f32x4*>((W1D ? a.w1d : a.w) + (long long)cg * NG * nchunk * (NTAP * CK * COP));

  // staging roles
  const int sq = tid & 31, sci = tid >> 5;
  const int tg = t0 + 4 * sq;
  const int hr = tid >> 4, hci = (tid >> 1) & 7, hside = tid & 1;
  const int htg = hside ? t0 + TT : t0 - 1;
  const bool hok = (hr < NR) && htg >= 0 && htg < T && (fin0 + hr) >= 0 && (fin0 + hr) < Fin;

  f32x4 pin[NR];
  float ph = 0.f;
  f32x4 pw[NWI];

  // All prefetch loads are unconditional buffer loads (one readfirstlane'd descriptor per sample, clamped 32-bit byte
  // offsets: no 64-bit address VGPRs, no branches); validity is re-derived in STAGE_COMMIT.
  const unsigned row_e = (unsigned)Tp;
  const unsigned plane_e = (unsigned)Fin * row_e;
  const unsigned long long pa = reinterpret_cast<unsigned long long>(in_n);
  const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa);
  const unsigned phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)phi << 32) | plo), 0,
      __builtin_amdgcn_readfirstlane((int)((unsigned)Cin * plane_e * 4u)), 0x00020000);
  const unsigned tg_e = (unsigned)(tg < Tp ? tg : Tp - 4);
  const bool full_t = (t0 + TT <= T);
  unsigned hoff_b;
  {
    int fh = fin0 + (hr < NR ? hr : NR - 1);
    fh = fh < 0 ? 0 : (fh >= Fin ? Fin - 1 : fh);
    const int th = htg < 0 ? 0 : (htg >= Tp ? Tp - 1 : htg);
    hoff_b = ((unsigned)fh * row_e + (unsigned)th) * 4u;
  }
  unsigned roff_b[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    int fin = fin0 + r;
    fin = fin < 0 ? 0 : (fin >= Fin ? Fin - 1 : fin);
    roff_b[r] = ((unsigned)fin * row_e + tg_e) * 4u;
  }

#define STAGE_ISSUE_S(KC, PIN, PH, PW)                                                                          \
  {                                                                                              \
    int c_ = (KC) * CK + sci;                                                                    \
    c_ = c_ < Cin ? c_ : Cin - 1;                                                                \
    const unsigned cb_ = (unsigned)c_ * plane_e * 4u;                                            \
    _Pragma("unroll") for (int r = 0; r < NR; ++r) PIN[r] = __builtin_bit_cast(                  \
        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, cb_ + roff_b[r], 0, 0));             \
    int c2_ = (KC) * CK + hci;                                                                   \
    c2_ = c2_ < Cin ? c2_ : Cin - 1;                                                             \
    PH = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(                         \
        rs_in, (unsigned)c2_ * plane_e * 4u + hoff_b, 0, 0));                                    \
    const f32x4* wsrc_ = w_g + (unsigned)(KC) * (unsigned)NW4;                                   \
    _Pragma("unroll") for (int i = 0; i < NWI; ++i) {                                            \
      unsigned idx_ = tid + 256 * i;                                                             \
      if (NW4 % 256 != 0) idx_ = idx_ < (unsigned)NW4 ? idx_ : (unsigned)(NW4 - 1);              \
      if (NG > 1) { /* the chunk's slab of group i / 3 (a group's slab = 768 float4 = 3 per thread) */ \
        constexpr int per_ = NTAP * CK * COP / 4;                                                \
        PW[i] = w_g[(unsigned)((i / (per_ / 256)) * nchunk + (KC)) * (unsigned)per_ + tid + 256 * (i % (per_ / 256))]; \
      } else PW[i] = wsrc_[idx_];                                                                \
    }                                                                                            \
  }

#define STAGE_COMMIT_S(KC, PIN, PH, PW)                                                                         \
  {                                                                                              \
    const float2 m_ = s_nrm[(KC) * CK + sci];                                                    \
    _Pragma("unroll") for (int r = 0; r < NR; ++r) {                                             \
      const int fin_ = fin0 + r;                                                                 \
      f32x4 v_ = {0.f, 0.f, 0.f, 0.f};                                                           \
      if (fin_ >= 0 && fin_ < Fin) {                      /* uniform */                          \
        v_.x = fmaf(PIN[r].x, m_.x, m_.y);                                                       \
        v_.y = fmaf(PIN[r].y, m_.x, m_.y);                                                       \
        v_.z = fmaf(PIN[r].z, m_.x, m_.y);                                                       \
        v_.w = fmaf(PIN[r].w, m_.x, m_.y);                                                       \
        if (!full_t) {                                                                           \
          v_.x = (tg + 0 < T) ? v_.x : 0.f;                                                      \
          v_.y = (tg + 1 < T) ? v_.y : 0.f;                                                      \
          v_.z = (tg + 2 < T) ? v_.z : 0.f;                                                      \
          v_.w = (tg + 3 < T) ? v_.w : 0.f;                                                      \
        }                                                                                        \
      }                                                                                          \
      *reinterpret_cast<f32x4*>(s_in + (sci * NR + r) * TW + 4 + 4 * sq) = v_;                   \
    }                                                                                            \
    if (hr < NR) {                                                                               \
      const float2 m2_ = s_nrm[(KC) * CK + hci];                                                 \
      s_in[(hci * NR + hr) * TW + (hside ? TT + 4 : 3)] = hok ? fmaf(PH, m2_.x, m2_.y) : 0.f;    \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NWI; ++i) {                                            \
      const int idx_ = tid + 256 * i;                                                            \
      if (NW4 % 256 == 0 || idx_ < NW4) reinterpret_cast<f32x4*>(s_w)[idx_] = PW[i];             \
    }                                                                                            \
  }
#define STAGE_ISSUE(KC) STAGE_ISSUE_S(KC, pin, ph, pw)
#define STAGE_COMMIT(KC) STAGE_COMMIT_S(KC, pin, ph, pw)

  // (TR2: the even row of the pair; f + 1 is the odd one.  W1D TR2: pair wave >> 1, column tile wave & 1)
  const int f = TR2 ? (W1D ? f0 + 2 * (wave >> 1) : f0 + 2 * wave) : (MODE == 4 ? f0 : f0 + wave);
  const bool row_ok = f < a.Fout;                       // wave-uniform
  int nseg = (T - t0 + 31) >> 5;
  nseg = nseg > 4 ? 4 : nseg;

  // W1D accumulators: [position nu][column tile of 64 frames] (MODE 1) / [nu] for the even and the odd row (MODE 2)
  f32x16 wacc[4][2];
  if (W1D) {
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[nu][q][r] = 0.f;
  }
  f32x16 acc[NCO][4];
  f32x16 acc_o[NCO][4];                                 // TR2: the odd row of the pair (never touched otherwise)
#pragma unroll
  for (int j = 0; j < NCO; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[j][s][r] = 0.f; if (TR2) acc_o[j][s][r] = 0.f; }

  const int half = lane >> 5, l31 = lane & 31;

  STAGE_ISSUE(0)
  // W1D: the group's bias for the epilogue table, in flight with the first chunk (read where the table is written it was one
  // more memory round trip in front of the first barrier)
  float bias_v = 0.f;
  if (W1D && tid < 32 * NG) bias_v = __builtin_nontemporal_load(a.bias + cg * (32 * NG) + tid);
  // (behind the first chunk's loads: the statistics reads and their float64 arithmetic run while those are in flight -- in front
  // of them every workgroup started with two serial memory latencies)
  // instance-norm parameters of the input channels (normalise-on-load)
  for (int c = tid; c < nchunk * CK; c += 256) {
    float mean = 0.f, rstd = (c < Cin) ? 1.f : 0.f;       // channels beyond Cin stage as zeros
    if (c >= a.ident_c && c < Cin) {
      const dstat_t* st = a.in_stats + ((long long)n * a.in_sstride + a.in_c0 + c) * (2 * DS_NL);
      const double cnt = (double)Fin * (double)T;
      const double m = dstat_read(st) / cnt;
      double var = dstat_read(st + DS_NL) / cnt - m * m;
      var = var > 0.0 ? var : 0.0;
      mean = (float)m;
      rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));
    }
    s_nrm[c] = make_float2(rstd, -mean * rstd);            // (scale, shift): x_norm = fma(x, scale, shift)
  }
  // W1D: (bias, ELU(bias)) of the group's 32 channels for the epilogue (the bias is zero padded to the group)
  float2* s_bc = s_nrm + nchunk * CK;
  if (W1D && tid < 32 * NG) s_bc[tid] = make_float2(bias_v, elu_fast(bias_v));

  __syncthreads();          // s_nrm visible
  STAGE_COMMIT(0)
  __syncthreads();

  // (Measured and dropped, round 6: a second register set with the loads of chunk k + 2 in flight changed nothing on the W1D
  // kernels -- they are not waiting for the chunk loads; PMC: waves spend 32 % of their time in s_waitcnt / barriers, 50 %
  // waiting for the matrix pipe the two resident workgroups share, the pipe is 54 % busy.)
  for (int kc = 0; kc < nchunk; ++kc) {
    const bool more = (kc + 1 < nchunk);
    if (more) STAGE_ISSUE(kc + 1)
    if (W1D) {
      if (row_ok) {
        if (TR2) chunk_w1d_tr2(wacc, s_in, s_w, wave >> 1, wave & 1, half, l31);
        else if (MODE == 4) chunk_w1d_s2<NR, SF>(wacc, s_in, s_w + wave * (NTAP * CK * COP), 0, half, l31);
        else if (MODE == 3) chunk_w1d_s2<NR, SF, CK / 2, 0, 1>(wacc, s_in, s_w + (2 - wave) * (CK * COP), 0, half, l31);   // (tap kf = 2 - f of staged row 0)
        else if (HALFK) {
          // (two half bodies in sequence, the second skipped for the half-empty last chunk: an if / else between a full and a half
          // body costs the instantiation its register allocation -- 432 bytes of scratch)
          chunk_w1d_s2<NR, SF, CK / 4, 0>(wacc, s_in, s_w, f - f0, half, l31);
          if (kc != nchunk - 1) chunk_w1d_s2<NR, SF, CK / 4, CK / 4>(wacc, s_in, s_w, f - f0, half, l31);
        } else chunk_w1d_s2<NR, SF>(wacc, s_in, s_w, f - f0, half, l31);
      }
    } else if (row_ok) {
      if (TR2) {
        chunk_mfma<NCO, NR, SF, TR2, 5>(acc, s_in, s_w, f - f0, half, l31);
        chunk_mfma<NCO, NR, SF, TR2, 2>(acc_o, s_in, s_w, f - f0 + 1, half, l31);
      } else if (MODE == 3) {
        // (ONE body: the tap kf = 2 - f is a wave-uniform offset into the weight slab, the staged row is row 0)
        chunk_mfma<NCO, NR, SF, TR2, 1, CK / 2, true>(acc, s_in, s_w + (2 - wave) * (CK * COP), 0, half, l31);
      } else if (HALFK && kc == nchunk - 1) {
        chunk_mfma<NCO, NR, SF, TR2, 7, CK / 4>(acc, s_in, s_w, f - f0, half, l31);      // half-empty last chunk
      } else {
        chunk_mfma<NCO, NR, SF, TR2, 7>(acc, s_in, s_w, f - f0, half, l31);
      }
    }
    __syncthreads();        // every wave is done reading this chunk
    if (more) {
      STAGE_COMMIT(kc + 1)
      __syncthreads();
    }
  }

  float* s_red = smem;   // [FTO rows][COP][2]  (safe: the loop ends with a barrier after the last reads)
  if (W1D) {
    // statistics slots: MODE 1 one per row (wave); MODE 2 [row of the tile][column tile]: 8 partial sums, added in slot order
    if (TR2) {
      const int q = wave & 1, rt = 2 * (wave >> 1);
      w1d_epilogue<0, 1>(a, wacc, n, cg, f, t0 + 64 * q, row_ok, lane, s_red + ((rt * 2 + q) * 32) * 2, s_bc);
      w1d_epilogue<1, 1>(a, wacc, n, cg, f + 1, t0 + 64 * q, f + 1 < a.Fout, lane, s_red + (((rt + 1) * 2 + q) * 32) * 2, s_bc);
    } else if (MODE == 4) {
      w1d_epilogue<0, 2>(a, wacc, n, cg * NG + wave, f, t0, row_ok, lane, s_red + wave * (32 * 2), s_bc + wave * 32);
    } else {
      // (MODE 0 = a network's first layer, which has no activation: model.py:44)
      if (MODE == 0 && !a.act) w1d_epilogue<0, 2, false>(a, wacc, n, cg, f, t0, row_ok, lane, s_red + wave * (32 * 2), s_bc);
      else w1d_epilogue<0, 2>(a, wacc, n, cg, f, t0, row_ok, lane, s_red + wave * (32 * 2), s_bc);
    }
    if (a.act) {
      __syncthreads();
      if (MODE == 4) {
        // one row, a group per wave: slot g holds the row's sums of group cg * 4 + g (0 + x, as the four-row sum of a one-row layer forms it)
        const int g = tid >> 6, co_l = (tid & 63) >> 1, which = tid & 1;
        const int co = (cg * NG + g) * 32 + co_l;
        if (co < a.Cout) {
          float tot = 0.f;
          if (f0 < a.Fout) tot += s_red[(g * 32 + co_l) * 2 + which];
          dstat_add(a.out_stats + (((long long)n * a.out_sstride + a.out_c0 + co) * 2 + which) * DS_NL, (double)tot);
        }
      } else if (tid < 64) {
        const int co_l = tid >> 1, which = tid & 1;
        const int co = cg * 32 + co_l;
        if (co < a.Cout) {
          float tot = 0.f;
          if (TR2) {
            for (int sl = 0; sl < 8; ++sl)
              if (f0 + (sl >> 1) < a.Fout) tot += s_red[(sl * 32 + co_l) * 2 + which];
          } else {
            for (int w = 0; w < FT; ++w)
              if (f0 + w < a.Fout) tot += s_red[(w * 32 + co_l) * 2 + which];
          }
          dstat_add(a.out_stats + (((long long)n * a.out_sstride + a.out_c0 + co) * 2 + which) * DS_NL, (double)tot);
        }
      }
    }
    return;
  }
  if (TR2) {
    conv_epilogue<NCO, 4, OCTP, true>(a, acc, n, cg, f, t0, row_ok, lane, s_red + (2 * wave) * (COP * 2));
    conv_epilogue<NCO, 4, OCTP, true>(a, acc_o, n, cg, f + 1, t0, f + 1 < a.Fout, lane, s_red + (2 * wave + 1) * (COP * 2));
  } else {
    conv_epilogue<NCO, 4, OCTP, true>(a, acc, n, cg, f, t0, row_ok, lane, s_red + wave * (COP * 2));
  }
  if (a.act) {
    __syncthreads();
    if (tid < COP * 2) {
      const int co_l = tid >> 1, which = tid & 1;
      const int co = cg * COP + co_l;
      if (co < a.Cout) {
        float tot = 0.f;
        for (int w = 0; w < FTO; ++w)
          if (f0 + w < a.Fout) tot += s_red[(w * COP + co_l) * 2 + which];
        dstat_add(a.out_stats + (((long long)n * a.out_sstride + a.out_c0 + co) * 2 + which) * DS_NL, (double)tot);
      }
    }
  }
}

#undef STAGE_ISSUE
#undef STAGE_COMMIT
#undef STAGE_ISSUE_S
#undef STAGE_COMMIT_S

static size_t conv_lds_bytes(int NR, int cop, int Cin, int ntap = 9) {
  const int nchunk = (Cin + CK - 1) / CK;
  return (size_t)(CK * NR * TW + ntap * CK * cop) * sizeof(float) + (size_t)(nchunk * CK + 128) * sizeof(float2);   // (+ s_bc: W1D, up to 4 groups)
}

template <int NCO, int MODE, int OCTP = 0, bool HALFK = false, bool W1D = false>
static hipError_t set_lds_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma<NCO, MODE, OCTP, HALFK, W1D>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}

hipError_t conv_init() {
  hipError_t e;
  if ((e = set_lds_attr<1, 0>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 1>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 2>()) != hipSuccess) return e;
  if ((e = set_lds_attr<2, 0>()) != hipSuccess) return e;
  if ((e = set_lds_attr<2, 1>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 1, 0, false, true>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 2, 0, false, true>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 0, 0, true, true>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 0, 0, false, true>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 4, 0, false, true>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 3, 0, false, true>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 3>()) != hipSuccess) return e;
  if ((e = set_lds_attr<2, 3>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 0, 3>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 2, 3>()) != hipSuccess) return e;
  if ((e = set_lds_attr<2, 0, 3>()) != hipSuccess) return e;
#ifdef MISONET_EXPERIMENTS                         // (fp16-piece outputs: the f16x3 mode of the experiment build)
  if ((e = set_lds_attr<1, 0, 4>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 2, 4>()) != hipSuccess) return e;
  if ((e = set_lds_attr<2, 0, 4>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 0, 4, true>()) != hipSuccess) return e;
#endif
  if ((e = set_lds_attr<1, 0, 0, true>()) != hipSuccess) return e;
  return set_lds_attr<1, 0, 3, true>();
}

int device_cus() {
  static std::atomic<int> cus[64];                   // 0 = not asked yet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  int c = cus[dev].load(std::memory_order_relaxed);
  if (!c) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) return -1;
    cus[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}

int conv_xcd_env() {
  static int v = -1;
  if (v < 0) v = exp_env("MISONET_XCD", 1);
  return v;
}

hipError_t launch_conv(const ConvArgs& a_in, int n_samples, hipStream_t s) {
  ConvArgs a = a_in;
  // XCD-aware order deals whole SAMPLES to the 8 XCDs (n % 8 == xcd): with a sample count that is not a multiple of 8 some
  // XCDs get nothing (B = 1: MISO3 runs 2 samples -> 6 of 8 XCDs idle, the first layer took 114 us instead of ~30); the
  // natural (t, f, n) grid is used then
  const int mode = a.tr2 ? 2 : (a.sf == 2 ? 1 : ((a.padf == 2 && a.Fin == 1 && !a.out_oct) ? 3 : 0));
  const bool w1d_first = mode == 0 && a.Cout <= 32 && a.Cin < 3 * CK;   // (a network's first layer: 12 / 16 -> 24 channels)
  const bool w1d_half = ((a.Cin - 1) % CK) < CK / 2;
  const bool w1d_row = (mode == 0 || mode == 1) && a.act && a.Fout == 1 && a.Fin == 3 && a.padf == 0 && a.Cout % 128 == 0;   // (encoder 6: F = 3 -> 1)
  if (a.w1d && (a.act || w1d_first) && (mode == 1 || mode == 2 || mode == 3 || w1d_first || w1d_row) && !a.out_oct && !a.in_oct) {
    // f32w: the frequency-strided layers (and the first layer) in 1-D Winograd form along T (32-channel groups, 4-row tiles)
    a.cop = 32;
    a.ncg = (a.Cout + 31) / 32;
    if (w1d_row) {
      // one output row (the encoder's last layer, 64 -> 128 channels on F = 3 -> 1, model.py:50): a row per wave would leave three of
      // four waves idle -- the waves take a 32-channel group each (MODE 4), the staged input serves all four.  The three staged rows
      // are the three frequency taps whatever the stride
      a.ncg = a.Cout / 128;
      a.NR = 3;
      const dim3 grid4 = conv_grid(a, n_samples, TT, FT, (n_samples % 8 == 0) ? conv_xcd_env() : 0);
      hipLaunchKernelGGL((conv3x3_mfma<1, 4, 0, false, true>), grid4, dim3(256), conv_lds_bytes(3, 128, a.Cin, 12), s, a);
      return hipGetLastError();
    }
    a.NR = mode == 1 ? 9 : (mode == 2 ? 3 : (mode == 3 ? 1 : 6));
    const dim3 gridw = conv_grid(a, n_samples, TT, FT, (n_samples % 8 == 0) ? conv_xcd_env() : 0);
    const size_t ldsw = conv_lds_bytes(a.NR, 32, a.Cin, 12);
    if (mode == 1) hipLaunchKernelGGL((conv3x3_mfma<1, 1, 0, false, true>), gridw, dim3(256), ldsw, s, a);
    else if (mode == 2) hipLaunchKernelGGL((conv3x3_mfma<1, 2, 0, false, true>), gridw, dim3(256), ldsw, s, a);
    else if (mode == 3) hipLaunchKernelGGL((conv3x3_mfma<1, 3, 0, false, true>), gridw, dim3(256), ldsw, s, a);
    else if (w1d_half) hipLaunchKernelGGL((conv3x3_mfma<1, 0, 0, true, true>), gridw, dim3(256), ldsw, s, a);
    else hipLaunchKernelGGL((conv3x3_mfma<1, 0, 0, false, true>), gridw, dim3(256), ldsw, s, a);
    return hipGetLastError();
  }
  a.NR = mode == 3 ? 1 : (mode == 0 ? 6 : f32_rows(a));
  if (mode == 2 && a.cop != 32) return hipErrorInvalidValue;   // (net.hip packs stride-2 transposed layers in 32-channel groups)
  const dim3 grid = conv_grid(a, n_samples, TT, mode == 2 ? 2 * FT : FT, (n_samples % 8 == 0) ? conv_xcd_env() : 0);
  const size_t lds = conv_lds_bytes(a.NR, a.cop, a.Cin);
#define MN_LAUNCH(NCO, MODE) hipLaunchKernelGGL((conv3x3_mfma<NCO, MODE>), grid, dim3(256), lds, s, a)
#define MN_LAUNCH3(NCO, MODE) hipLaunchKernelGGL((conv3x3_mfma<NCO, MODE, 3>), grid, dim3(256), lds, s, a)
#define MN_LAUNCH4(NCO, MODE) hipLaunchKernelGGL((conv3x3_mfma<NCO, MODE, 4>), grid, dim3(256), lds, s, a)
  // last K-chunk at most half full (the network's first layer): the HALFK instantiations, 32-channel groups only
  const bool halfk = mode == 0 && a.cop == 32 && ((a.Cin - 1) % CK) < CK / 2;
#define MN_LAUNCH_H(OCTP) hipLaunchKernelGGL((conv3x3_mfma<1, 0, OCTP, true>), grid, dim3(256), lds, s, a)
  if (a.out_oct) {
    // planar float32 in, oct layout out (bf16x6: three bf16 pieces; f16x3: two fp16 pieces): only the layer shapes that
    // occur in front of a dense block
    if ((a.out_oct != 3 && a.out_oct != 4) || mode == 1 || (a.Cout & 7) || (a.out_c0 & 7) || (a.out_sstride & 7)) return hipErrorInvalidValue;
    if (a.out_oct == 3) {
      if (a.cop == 32) { if (halfk) MN_LAUNCH_H(3); else if (mode == 0) MN_LAUNCH3(1, 0); else MN_LAUNCH3(1, 2); }
      else { if (mode == 0) MN_LAUNCH3(2, 0); else return hipErrorInvalidValue; }
    } else {
#ifdef MISONET_EXPERIMENTS
      if (a.cop == 32) { if (halfk) MN_LAUNCH_H(4); else if (mode == 0) MN_LAUNCH4(1, 0); else MN_LAUNCH4(1, 2); }
      else { if (mode == 0) MN_LAUNCH4(2, 0); else return hipErrorInvalidValue; }
#else
      return hipErrorInvalidValue;
#endif
    }
  } else if (a.cop == 32) {
    if (halfk) MN_LAUNCH_H(0); else if (mode == 0) MN_LAUNCH(1, 0); else if (mode == 1) MN_LAUNCH(1, 1); else if (mode == 2) MN_LAUNCH(1, 2); else MN_LAUNCH(1, 3);
  } else {
    if (mode == 0) MN_LAUNCH(2, 0); else if (mode == 1) MN_LAUNCH(2, 1); else MN_LAUNCH(2, 3);
  }
#undef MN_LAUNCH
#undef MN_LAUNCH3
#undef MN_LAUNCH4
#undef MN_LAUNCH_H
  return hipGetLastError();
}

}  // namespace mn
