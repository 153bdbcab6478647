// Winograd F(2x2, 3x3) form of the stride-1 same-padded 3x3 convs of the DenseBlocks (reference model.py:437-482: 50 of the
// 64 conv layers of a trunk, 94 % of its MACs) on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32 products
// and sums).  Precision mode "f32w" (misonet_net.precision == 5); every other layer of that mode runs on conv3x3_mfma.
//
//   out[co][f][t] = bias[co] + sum_{ci,kt,kf} W[co][ci][kt][kf] * xhat[ci][f - 1 + kf][t - 1 + kt],   xhat = instance norm
//
// A tile is 2 rows (f) x 2 frames (t) of output; it needs the 4 x 4 input patch d around it:
//   Y = A^T [ U (.) V ] A,   V = B^T d B (per input channel, per tile),   U = G g G^T (per (co, ci), packed at commit time)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
// so a 3x3 conv costs 16 instead of 36 multiplications per tile and channel pair: 2.25 x fewer MFMAs.  The 16 positions
// (xi, nu) of the transformed domain are 16 independent GEMMs  M_p[co][tile] = sum_ci U_p[co][ci] V_p[ci][tile]:
//
//   MFMA roles: M = 32 output channels (A operand = U_p), N = 32 tiles = 64 consecutive frames of one tile row (B operand =
//   V_p), K = 2 input channels per instruction (lanes 0-31: channel 2s, lanes 32-63: channel 2s + 1).  A lane computes the
//   transform of ITS (tile, channel) patch in registers (32 additions) and feeds it to 16 MFMAs: per 1024 matrix cycles a
//   wave reads 32 dwords of LDS -- the matrix pipe is the only busy unit.
//   One wave per SIMD, 16 accumulators of 16 registers = 256 AGPRs (the 512-register shape); workgroup = 4 waves = 4 tile
//   rows = 8 output rows x 64 frames x 32 output channels.
//
// PERSISTENT: one workgroup per CU walks a contiguous range of its XCD's (sample, tile) list; the K-chunks (8 input
// channels) of all its tiles form ONE stream through THREE LDS stages: while chunk g runs on the matrix pipe, chunk g + 1
// (in registers since the previous chunk) is normalised and written to its stage -- instance norm applied on the way into
// the LDS, zero padding applied after it, the reference's order -- and the global loads of chunk g + 2 are issued, also
// across a tile boundary, so a new tile starts with its operands in registers.  One workgroup barrier per chunk, after
// three of its four K-steps.
// The instruction stream is laid out by hand: a K-step is 16 SLOTS of one MFMA (64 matrix cycles) + a piece of side work
// (operand fetch of the next step, its input transform, one staging item), fenced by sched_barriers; the matrix pipe never
// waits for an LDS round trip or a global load.
// Epilogue: inverse transform per lane (the 16 positions of a (channel, tile) are 16 accumulator registers of ONE lane),
// + bias, ELU, centring, 8-byte stores along T, exact statistics (det_stats.hpp) as in conv_epilogue.hpp.
#include "kernels.hpp"
#include "conv_epilogue.hpp"
#include <stdlib.h>
#include <utility>

namespace mn {

typedef float wf16 __attribute__((ext_vector_type(16)));
typedef float wf4 __attribute__((ext_vector_type(4)));
typedef float wf2 __attribute__((ext_vector_type(2)));
typedef unsigned int wu2 __attribute__((ext_vector_type(2)));

constexpr int WCK = 8;                         // input channels per chunk
constexpr int WTT = 64;                        // output frames per workgroup (32 tiles)
constexpr int WFT = 8;                         // output rows per workgroup (4 waves x 2)
constexpr int WNR = 10;                        // staged input rows
constexpr int WTW = 68;                        // floats per staged row: column c = frame t0 - 1 + c (66 used)
constexpr int WIN_FLOATS = WCK * WNR * WTW;    // 5440
constexpr int WW_FLOATS = 16 * WCK * 32;       // 4096: [pos / 4][ci][co][pos % 4]
constexpr int WSTAGE_FLOATS = WIN_FLOATS + WW_FLOATS;
constexpr int WNSTAGE = 3;
constexpr int WNRM_MAX = 256;                  // input channels (s_nrm entries per parity)
// stages | s_nrm[2][WNRM_MAX] float2 | s_red [4][32][2] | s_dummy [256]
constexpr size_t WINO_LDS = (size_t)(WNSTAGE * WSTAGE_FLOATS) * 4 + 2 * WNRM_MAX * 8 + 4 * 64 * 4 + 256 * 4;

// ---- the 256 accumulator registers are FIXED physical AGPRs a0..a255 (position p = a[16 p : 16 p + 15]), touched only by
// inline asm.  With the MFMA builtin (or asm with "+a" operands) hipcc 7.2's allocator treats accumulators and operands as
// one either-file register class and, in this hand-ordered stream (sched_barriers between the slots), shuffles accumulator
// pieces through VGPRs and scratch every iteration (238 spills, -Rpass-analysis).  Every MFMA statement names all AGPRs as
// clobbered, so the compiler never keeps a value there; tests/test_build_resources.py holds the kernel to 0 spills (a VGPR
// spill could be parked in an AGPR between two statements) and to no compiler-generated v_accvgpr_* at all.
// The hazard recogniser does not see inside asm: the stream keeps >= 2 instructions between a VALU / LDS write of an
// operand and the MFMA that reads it (the s_waitcnt for LDS operands is still inserted by the compiler), v[15] -- read by
// the last MFMA of a step -- is rewritten one slot later, and explicit s_nops separate the last MFMA from the epilogue's
// accumulator reads and the accumulator zeroing from the next MFMA.
#define W_ACLOB \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", \
  "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", \
  "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", \
  "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", \
  "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", \
  "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", \
  "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", \
  "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", \
  "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", \
  "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", \
  "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", \
  "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", \
  "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", \
  "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", \
  "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", \
  "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", \
  "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", \
  "a253", "a254", "a255"
template <int P>
__device__ __forceinline__ void wino_mfma(float uu, float vv) {
  asm volatile("v_mfma_f32_32x32x2_f32 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(16 * P), "n"(16 * P + 15), "v"(uu), "v"(vv) : W_ACLOB);
}
template <int I>
__device__ __forceinline__ float agpr_get() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(I));
  return x;
}
template <int I>
__device__ __forceinline__ void agpr_zero() {
  asm volatile("v_accvgpr_write_b32 a[%0], 0" ::"n"(I));
}
template <class F, int... Is>
__device__ __forceinline__ void wfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wfor(F&& f) {
  wfor_impl(f, std::make_integer_sequence<int, N>{});
}

// DBG (timing experiments only, -DMISONET_EXPERIMENTS + MISONET_WINO_DBG): 1 = no staging side work in the chunk loop (wrong
// results), 2 = no epilogue arithmetic / stores.
template <int DBG>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32(const ConvArgs a) {
  extern __shared__ __align__(16) float smem[];
  float2* s_nrm = reinterpret_cast<float2*>(smem + WNSTAGE * WSTAGE_FLOATS);
  float* s_red = reinterpret_cast<float*>(s_nrm + 2 * WNRM_MAX);   // [4 waves][32][2]
  float* s_dummy = s_red + 4 * 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int T = a.T, Tp = a.Tp, F = a.Fin, Cin = a.Cin;
  const int nchunk = Cin / WCK;

  // ---- this workgroup's tile range [q0, q1) of the linear (sample slot, row tile, channel group, frame tile) list ----
  const unsigned tps = (unsigned)(a.ntx * a.nty * a.ncg);
  unsigned q0, q1, xcd_id = 0;
  if (a.xcd) {
    xcd_id = blockIdx.x & 7u;
    const unsigned wl = blockIdx.x >> 3, nwl = gridDim.x >> 3;
    const unsigned Q = (unsigned)(a.nsamp >> 3) * tps;
    const unsigned per = (Q + nwl - 1) / nwl;
    q0 = wl * per;
    q1 = q0 + per < Q ? q0 + per : Q;
  } else {
    const unsigned Q = (unsigned)a.nsamp * tps;
    const unsigned per = (Q + gridDim.x - 1) / gridDim.x;
    q0 = blockIdx.x * per;
    q1 = q0 + per < Q ? q0 + per : Q;
  }
  if (q0 >= q1) return;

  const unsigned row_e = (unsigned)Tp;
  const unsigned plane_b = (unsigned)F * row_e * 4u;

  // ---- staging roles: thread (sq, scr) owns frames t0 + 4 sq .. + 3 of the (channel, row) items scr + 16 i, i = 0..4 ----
  const int sq = tid & 15, scr = tid >> 4;
  int sch[5];                                  // channel of item i; its LDS offset is loff0 + i * 16 * WTW (item p = row p of [80][WTW])
#pragma unroll
  for (int i = 0; i < 5; ++i) sch[i] = (scr + 16 * i) / WNR;
  const int loff0 = scr * WTW + 1 + 4 * sq;
  // halo columns: frame t0 - 1 (column 0) and t0 + 64 (column 65) of the 80 (channel, row) items
  const int hp = tid >> 1, hside = tid & 1;
  const bool hrole = hp < WCK * WNR;
  const int hch = hrole ? hp / WNR : 0, hrow = hrole ? hp - hch * WNR : 0;
  const int hloff = (hch * WNR + hrow) * WTW + (hside ? WTT + 1 : 0);
  // operand fetch of K-step S (channels 2 S + half of the chunk)
  const int d_off = (half * WNR + 2 * wave) * WTW + 2 * l31;     // + S * 2 * WNR * WTW
  const int u_off = half * 32 + l31;                             // wf4 units; + S * 64, + q * 256

  // ---- load-side state: the tile and chunk of the NEXT chunk to issue ----
  unsigned ql = q0;
  int kl = 0;
  int Ln = -1, Lpar = 1;                       // sample and s_nrm parity of the load tile
  unsigned Lgoff[5], Lhoff = 0, Lrokm = 0;     // Lrokm: bit i = row of item i inside the image, bit 5 = halo element exists
  bool Ltm0 = false, Ltm1 = false, Ltm2 = false, Ltm3 = false;     // frame tg + j < T
  __amdgpu_buffer_rsrc_t rs_l;
  const wf4* w_l = nullptr;
  // commit-side copies (the chunk committed in iteration g + 1 was issued in iteration g)
  unsigned Crokm = 0;
  bool Ctm0 = false, Ctm1 = false, Ctm2 = false, Ctm3 = false;
  int Cnb = 0;

#define W_DECODE(Q, TT_, FT_, N_, CG_)                                                                \
  {                                                                                                   \
    const unsigned q_ = __builtin_amdgcn_readfirstlane(Q);                                            \
    const unsigned j_ = q_ / tps;                                                                     \
    unsigned r_ = q_ - j_ * tps;                                                                      \
    N_ = (int)(a.xcd ? j_ * 8u + xcd_id : j_);                                                        \
    TT_ = (int)(r_ % (unsigned)a.ntx);                                                                \
    r_ /= (unsigned)a.ntx;                                                                            \
    CG_ = (int)(r_ % (unsigned)a.ncg);                                                                \
    FT_ = (int)(r_ / (unsigned)a.ncg);                                                                \
  }

#define W_LOAD_SETUP(Q)                                                                               \
  {                                                                                                   \
    int tt_, ft_, n_, cg_;                                                                            \
    W_DECODE(Q, tt_, ft_, n_, cg_)                                                                    \
    const int t0_ = tt_ * WTT, fin0_ = ft_ * WFT - 1;                                                 \
    if (n_ != Ln) {                            /* new sample: its instance-norm parameters, other parity */ \
      Ln = n_;                                                                                        \
      Lpar ^= 1;                                                                                      \
      for (int c = tid; c < Cin; c += 256) {                                                          \
        float mean = 0.f, rstd = 1.f;                                                                 \
        if (c >= a.ident_c) {                                                                         \
          const dstat_t* st_ = a.in_stats + ((long long)n_ * a.in_sstride + a.in_c0 + c) * (2 * DS_NL); \
          const double cnt = (double)F * (double)T;                                                   \
          const double m = dstat_read(st_) / cnt;                                                     \
          double var = dstat_read(st_ + DS_NL) / cnt - m * m;                                         \
          var = var > 0.0 ? var : 0.0;                                                                \
          mean = (float)m;                                                                            \
          rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));                                           \
        }                                                                                             \
        s_nrm[Lpar * WNRM_MAX + c] = make_float2(rstd, -mean * rstd);                                 \
      }                                                                                               \
      const float* in_n_ = a.in + (long long)n_ * a.in_bstride + (long long)a.in_c0 * F * Tp;         \
      rs_l = make_rsrc_e(reinterpret_cast<unsigned long long>(in_n_), (unsigned)Cin * plane_b);       \
    }                                                                                                 \
    w_l = reinterpret_cast<const wf4*>(a.ww) + (long long)cg_ * nchunk * (WW_FLOATS / 4);             \
    const int tg_ = t0_ + 4 * sq;                                                                     \
    const unsigned tge_ = (unsigned)(tg_ + 4 <= Tp ? tg_ : Tp - 4);                                   \
    Ltm0 = tg_ + 0 < T; Ltm1 = tg_ + 1 < T; Ltm2 = tg_ + 2 < T; Ltm3 = tg_ + 3 < T;                   \
    Lrokm = 0;                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                                   \
      int fin = fin0_ + (scr + 16 * i) - sch[i] * WNR;                                                \
      if (fin >= 0 && fin < F) Lrokm |= 1u << i;                                                      \
      fin = fin < 0 ? 0 : (fin >= F ? F - 1 : fin);                                                   \
      Lgoff[i] = (unsigned)sch[i] * plane_b + ((unsigned)fin * row_e + tge_) * 4u;                    \
    }                                                                                                 \
    {                                                                                                 \
      const int htg_ = hside ? t0_ + WTT : t0_ - 1;                                                   \
      int fin = fin0_ + hrow;                                                                         \
      if (hrole && fin >= 0 && fin < F && htg_ >= 0 && htg_ < T) Lrokm |= 1u << 5;                    \
      fin = fin < 0 ? 0 : (fin >= F ? F - 1 : fin);                                                   \
      const int th_ = htg_ < 0 ? 0 : (htg_ >= Tp ? Tp - 1 : htg_);                                    \
      Lhoff = (unsigned)hch * plane_b + ((unsigned)fin * row_e + (unsigned)th_) * 4u;                 \
    }                                                                                                 \
  }

  // what the commit of the chunk issued LAST needs (taken before the load position advances)
#define W_LATCH                                                                                       \
  { Crokm = Lrokm; Ctm0 = Ltm0; Ctm1 = Ltm1; Ctm2 = Ltm2; Ctm3 = Ltm3; Cnb = Lpar * WNRM_MAX + kl * WCK; }
#define W_ADVANCE                                                                                     \
  {                                                                                                   \
    if (++kl == nchunk) {                                                                             \
      if (ql + 1 < q1) { kl = 0; ++ql; W_LOAD_SETUP(ql) }                                             \
      else kl = nchunk - 1;                    /* end of the stream: the last chunk again (never consumed) */ \
    }                                                                                                 \
  }

  wf4 pin[5];
  float ph = 0.f;
  wf4 pw[4];
  float2 nra, nrb;
  wf4 cv;

#define W_CB ((unsigned)kl * (unsigned)WCK * plane_b)
#define W_ISSUE_I(I) pin[I] = __builtin_bit_cast(wf4, __builtin_amdgcn_raw_buffer_load_b128(rs_l, W_CB + Lgoff[I], 0, 0));
#define W_ISSUE_H ph = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_l, W_CB + Lhoff, 0, 0));
#define W_ISSUE_W(J) pw[J] = w_l[(unsigned)kl * (unsigned)(WW_FLOATS / 4) + tid + 256 * (J)];
#define W_NR(DST, CH) DST = s_nrm[Cnb + (CH)];
  // item I of the chunk in registers: normalise, zero what lies outside the image
#define W_CC(I, NR)                                                                                   \
  {                                                                                                   \
    const bool rok_ = (Crokm >> (I)) & 1u;                                                            \
    const float sc_ = rok_ ? NR.x : 0.f, sh_ = rok_ ? NR.y : 0.f;                                     \
    cv.x = Ctm0 ? fmaf(pin[I].x, sc_, sh_) : 0.f;                                                     \
    cv.y = Ctm1 ? fmaf(pin[I].y, sc_, sh_) : 0.f;                                                     \
    cv.z = Ctm2 ? fmaf(pin[I].z, sc_, sh_) : 0.f;                                                     \
    cv.w = Ctm3 ? fmaf(pin[I].w, sc_, sh_) : 0.f;                                                     \
  }
#define W_CW(I, ST)                                                                                   \
  {                                                                                                   \
    float* si_ = smem + (ST) * WSTAGE_FLOATS + loff0 + (I) * (16 * WTW);                              \
    si_[0] = cv.x;                                                                                    \
    *reinterpret_cast<wf2*>(si_ + 1) = wf2{cv.y, cv.z};                                               \
    si_[3] = cv.w;                                                                                    \
  }
#define W_CH(ST, NR)                                                                                  \
  {                                                                                                   \
    const bool hok_ = (Crokm >> 5) & 1u;                                                              \
    float* hp_ = hrole ? smem + (ST) * WSTAGE_FLOATS + hloff : s_dummy + tid;                         \
    *hp_ = hok_ ? fmaf(ph, NR.x, NR.y) : 0.f;                                                         \
  }
#define W_CWW(J, ST) reinterpret_cast<wf4*>(smem + (ST) * WSTAGE_FLOATS + WIN_FLOATS)[tid + 256 * (J)] = pw[J];

  wf2 dd[4][2];
  float tt[4][4];
  wf4 u[4];                                    // U operands: quad q = positions 4 q .. 4 q + 3; refilled quad by quad
  float v[16];                                 // V operands of the running K-step; rewritten in its last two slots
  float v15n;                                  // ... except v[15] (operand of the step's last MFMA): one slot later
#define W_FD(I0, ST, S)                                                                               \
  {                                                                                                   \
    const float* si_ = smem + (ST) * WSTAGE_FLOATS + d_off + (S) * (2 * WNR * WTW);                   \
    dd[I0][0] = *reinterpret_cast<const wf2*>(si_ + (I0) * WTW);                                      \
    dd[I0][1] = *reinterpret_cast<const wf2*>(si_ + (I0) * WTW + 2);                                  \
    dd[I0 + 1][0] = *reinterpret_cast<const wf2*>(si_ + ((I0) + 1) * WTW);                            \
    dd[I0 + 1][1] = *reinterpret_cast<const wf2*>(si_ + ((I0) + 1) * WTW + 2);                        \
  }
#define W_FU(Q, ST, S)                                                                                \
  u[Q] = (reinterpret_cast<const wf4*>(smem + (ST) * WSTAGE_FLOATS + WIN_FLOATS) + u_off + (S) * 64)[(Q) * 256];
#define W_TROW(I)                                                                                     \
  {                                                                                                   \
    const float d0 = dd[I][0].x, d1 = dd[I][0].y, d2 = dd[I][1].x, d3 = dd[I][1].y;                   \
    tt[I][0] = d0 - d2; tt[I][1] = d1 + d2; tt[I][2] = d2 - d1; tt[I][3] = d1 - d3;                   \
  }
#define W_TCOL01                                                                                      \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) { v[j] = tt[0][j] - tt[2][j]; v[4 + j] = tt[1][j] + tt[2][j]; }
#define W_TCOL23                                                                                      \
  {                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) v[8 + j] = tt[2][j] - tt[1][j];                     \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) v[12 + j] = tt[1][j] - tt[3][j];                    \
    v15n = tt[1][3] - tt[3][3];                                                                       \
  }

#define W_SB __builtin_amdgcn_sched_barrier(0);
#define W_MF(P) wino_mfma<P>(u[(P) >> 2][(P) & 3], v[P]);
  // one K-step (CST, CS): 16 slots = MFMA + side work.  The U quads are refilled as they are consumed (quad 3 of THIS step
  // in slot 0, quads 0-2 of the next step (FST, FS) behind the slots that used them); the patch of the next step is fetched
  // in slots 0-1 and transformed in slots 10-15 into v[] (positions 0-7 after MFMA 14, 8-15 after MFMA 15).
#define W_STEP(CST, CS, FST, FS, X2, X3, X5, X6, X7, X9)                                              \
  W_MF(0) v[15] = v15n; W_FD(0, FST, FS) W_FU(3, CST, CS) W_SB                                        \
  W_MF(1) W_FD(2, FST, FS) W_SB                                                                       \
  W_MF(2) X2 W_SB                                                                                     \
  W_MF(3) X3 W_SB                                                                                     \
  W_MF(4) W_FU(0, FST, FS) W_SB                                                                       \
  W_MF(5) X5 W_SB                                                                                     \
  W_MF(6) X6 W_SB                                                                                     \
  W_MF(7) X7 W_SB                                                                                     \
  W_MF(8) W_FU(1, FST, FS) W_SB                                                                       \
  W_MF(9) X9 W_SB                                                                                     \
  W_MF(10) W_T(W_TROW(0)) W_SB                                                                        \
  W_MF(11) W_T(W_TROW(1)) W_SB                                                                        \
  W_MF(12) W_T(W_TROW(2)) W_FU(2, FST, FS) W_SB                                                       \
  W_MF(13) W_T(W_TROW(3)) W_SB                                                                        \
  W_MF(14) W_T(W_TCOL01) W_SB                                                                         \
  W_MF(15) W_T(W_TCOL23) W_SB
#define W_NONE
#define W_X(...) if (!(DBG & 1)) { __VA_ARGS__ }
#define W_T(...) if (!(DBG & 4)) { __VA_ARGS__ } else { v[(DBG >> 4) & 15] += dd[0][0].x + dd[1][1].y + dd[2][0].x + dd[3][1].y; }

  // ---- prologue: chunk 0 committed, chunk 1 in registers, operands of (chunk 0, step 0) transformed ----
  W_LOAD_SETUP(ql)
  __syncthreads();                                   // s_nrm visible
  W_ISSUE_I(0) W_ISSUE_I(1) W_ISSUE_I(2) W_ISSUE_I(3) W_ISSUE_I(4) W_ISSUE_H
  W_ISSUE_W(0) W_ISSUE_W(1) W_ISSUE_W(2) W_ISSUE_W(3)
  W_LATCH
  W_ADVANCE
  W_NR(nra, sch[0]) W_CC(0, nra) W_CW(0, 0) W_ISSUE_I(0)
  W_NR(nra, sch[1]) W_CC(1, nra) W_CW(1, 0) W_ISSUE_I(1)
  W_NR(nra, sch[2]) W_CC(2, nra) W_CW(2, 0) W_ISSUE_I(2)
  W_NR(nra, sch[3]) W_CC(3, nra) W_CW(3, 0) W_ISSUE_I(3)
  W_NR(nra, sch[4]) W_CC(4, nra) W_CW(4, 0) W_ISSUE_I(4)
  W_NR(nra, hch) W_CH(0, nra) W_ISSUE_H
  W_CWW(0, 0) W_ISSUE_W(0) W_CWW(1, 0) W_ISSUE_W(1) W_CWW(2, 0) W_ISSUE_W(2) W_CWW(3, 0) W_ISSUE_W(3)
  W_LATCH
  W_ADVANCE
  __syncthreads();
  W_FD(0, 0, 0) W_FD(2, 0, 0) W_FU(0, 0, 0) W_FU(1, 0, 0) W_FU(2, 0, 0)
  W_TROW(0) W_TROW(1) W_TROW(2) W_TROW(3) W_TCOL01 W_TCOL23
  wfor<256>([&](auto i) __attribute__((always_inline)) { agpr_zero<decltype(i)::value>(); });
  asm volatile("s_nop 4");

  unsigned qc = q0;                                  // tile of the chunk on the matrix pipe
  int kc = 0;
  int st = 0;                                        // stage of chunk g
  const unsigned G = (q1 - q0) * (unsigned)nchunk;
  for (unsigned g = 0; g < G; ++g) {
    const int stn = st == WNSTAGE - 1 ? 0 : st + 1;
    // steps 0-2: chunk g + 1 registers -> stage stn, the loads of chunk g + 2 issued item by item behind it
    W_STEP(st, 0, st, 1, W_X(W_NR(nra, sch[0]) W_NR(nrb, sch[1])), W_X(W_CC(0, nra)), W_X(W_CW(0, stn) W_ISSUE_I(0)), W_X(W_CC(1, nrb)),
           W_X(W_CW(1, stn) W_ISSUE_I(1)), W_X(W_NR(nra, sch[2]) W_NR(nrb, sch[3])))
    W_STEP(st, 1, st, 2, W_X(W_CC(2, nra)), W_X(W_CW(2, stn) W_ISSUE_I(2)), W_X(W_CC(3, nrb)), W_X(W_CW(3, stn) W_ISSUE_I(3)),
           W_X(W_NR(nra, sch[4]) W_NR(nrb, hch)), W_X(W_CC(4, nra)))
    W_STEP(st, 2, st, 3, W_X(W_CW(4, stn) W_ISSUE_I(4)), W_X(W_CH(stn, nrb) W_ISSUE_H), W_X(W_CWW(0, stn) W_ISSUE_W(0)),
           W_X(W_CWW(1, stn) W_ISSUE_W(1)), W_X(W_CWW(2, stn) W_ISSUE_W(2)), W_X(W_CWW(3, stn) W_ISSUE_W(3)))
    if (!(DBG & 8)) __syncthreads();                 // chunk g + 1 complete in stage stn; the stage of chunk g - 1 is free
    W_STEP(st, 3, stn, 0, W_NONE, W_NONE, W_NONE, W_NONE, W_NONE, W_NONE)
    W_LATCH
    W_ADVANCE
    st = stn;
    if (++kc == nchunk) {
      // ---- tile epilogue: Y = A^T M A per (channel, tile), + bias, ELU, centring, stores, statistics ----
      asm volatile("s_nop 15\n\ts_nop 7");             // the last MFMA's result (16 passes) before any accumulator read
      if (!(DBG & 2)) {
      int tt_, ft_, n, cg;
      W_DECODE(qc, tt_, ft_, n, cg)
      const int t0 = tt_ * WTT, f0 = ft_ * WFT;
      const int cbase = cg * 32;
      const int fa = f0 + 2 * wave;
      const int t = t0 + 2 * l31;
      const bool full_t = (t0 + WTT <= T);
      const bool r0ok = fa < F, r1ok = fa + 1 < F;
      const bool c0ok = t < T, c1ok = t + 1 < T;
      const unsigned P4 = (unsigned)F * (unsigned)Tp * 4u;
      const float* ob_ = a.out + (long long)n * a.out_bstride + (long long)a.out_c0 * F * Tp;
      const __amdgpu_buffer_rsrc_t rs_out = make_rsrc_e(reinterpret_cast<unsigned long long>(ob_), (unsigned)a.Cout * P4);
      const unsigned vbase = (unsigned)(fa * Tp + t) * 4u + (unsigned)(4 * half) * P4;
      const unsigned vo0 = (r0ok && c0ok) ? vbase : 0x80000000u;
      const unsigned vo1 = (r1ok && c0ok) ? vbase + (unsigned)Tp * 4u : 0x80000000u;
      const float m00 = (r0ok && c0ok) ? 1.f : 0.f, m01 = (r0ok && c1ok) ? 1.f : 0.f;
      const float m10 = (r1ok && c0ok) ? 1.f : 0.f, m11 = (r1ok && c1ok) ? 1.f : 0.f;
      const bool act = a.act != 0;
      float s1[16], s2[16];
      wfor<16>([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        constexpr int kr = (r & 3) + 8 * (r >> 2);
        const unsigned coff = (unsigned)(cbase + kr) * P4;
        const float b = a.bias[cbase + kr + 4 * half];
        const float cr = act ? elu_fast(b) : 0.f;
        float e0[4], e1[4];
        wfor<4>([&](auto xc) __attribute__((always_inline)) {
          constexpr int x = decltype(xc)::value;
          const float m0 = agpr_get<(4 * x + 0) * 16 + r>(), m1 = agpr_get<(4 * x + 1) * 16 + r>();
          const float m2 = agpr_get<(4 * x + 2) * 16 + r>(), m3 = agpr_get<(4 * x + 3) * 16 + r>();
          e0[x] = (m0 + m1) + m2;
          e1[x] = (m1 - m2) - m3;
        });
        float y00 = (e0[0] + e0[1]) + e0[2] + b, y10 = (e0[1] - e0[2]) - e0[3] + b;
        float y01 = (e1[0] + e1[1]) + e1[2] + b, y11 = (e1[1] - e1[2]) - e1[3] + b;
        if (act) {
          y00 = elu_fast(y00) - cr; y01 = elu_fast(y01) - cr;
          y10 = elu_fast(y10) - cr; y11 = elu_fast(y11) - cr;
        }
        if (full_t) {
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(wu2, wf2{y00, y01}), rs_out, vo0 + coff, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(wu2, wf2{y10, y11}), rs_out, vo1 + coff, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y00), rs_out, vo0 + coff, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y10), rs_out, vo1 + coff, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y01), rs_out, c1ok ? vo0 + coff + 4u : 0x80000000u, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y11), rs_out, c1ok ? vo1 + coff + 4u : 0x80000000u, 0, 0);
        }
        const float z00 = y00 * m00, z01 = y01 * m01, z10 = y10 * m10, z11 = y11 * m11;
        s1[r] = (z00 + z01) + (z10 + z11);
        s2[r] = fmaf(z00, z00, fmaf(z01, z01, fmaf(z10, z10, z11 * z11)));
      });
      if (act) {
        const float x1 = reduce16_halfwave(s1, lane);
        const float x2 = reduce16_halfwave(s2, lane);
        if ((lane & 16) == 0) {
          const int q = lane & 15;
          const int co_l = (q & 3) + 8 * (q >> 2) + 4 * half;
          s_red[(wave * 32 + co_l) * 2 + 0] = x1;
          s_red[(wave * 32 + co_l) * 2 + 1] = x2;
        }
        __syncthreads();
        if (tid < 64) {
          const int co_l = tid >> 1, which = tid & 1;
          const int co = cbase + co_l;
          if (co < a.Cout) {
            float tot = 0.f;
            for (int w = 0; w < 4; ++w)
              if (f0 + 2 * w < F) tot += s_red[(w * 32 + co_l) * 2 + which];
            dstat_add(a.out_stats + (((long long)n * a.out_sstride + a.out_c0 + co) * 2 + which) * DS_NL, (double)tot);
          }
        }
      }
      }
      wfor<256>([&](auto i) __attribute__((always_inline)) { agpr_zero<decltype(i)::value>(); });
      asm volatile("s_nop 4");
      kc = 0;
      ++qc;
    }
  }
}

bool conv_wino_ok(const ConvArgs& a) {
  return a.sf == 1 && a.padf == 1 && !a.tr2 && a.Fin == a.Fout && (a.Cin % WCK) == 0 && a.Cin <= WNRM_MAX && !a.in_oct &&
         !a.out_oct && a.ww != nullptr;
}

#ifdef MISONET_EXPERIMENTS
static int wino_dbg_env() {
  static const int v = [] { const char* e = getenv("MISONET_WINO_DBG"); return e ? atoi(e) : 0; }();
  return v;
}
#endif

hipError_t conv_wino_init() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_f32<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)WINO_LDS);
#ifdef MISONET_EXPERIMENTS
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_f32<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO_LDS);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_f32<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO_LDS);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_f32<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO_LDS);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_f32<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO_LDS);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_f32<11>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO_LDS);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_f32<15>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO_LDS);
#endif
  return e;
}

hipError_t launch_conv_wino(const ConvArgs& a_in, int n_samples, hipStream_t s) {
  ConvArgs a = a_in;
  if (!conv_wino_ok(a)) return hipErrorInvalidValue;
  a.cop = 32;
  a.ncg = (a.Cout + 31) / 32;
  const int cus = device_cus();
  if (cus <= 0) return hipErrorInvalidDevice;
  (void)conv_grid(a, n_samples, WTT, WFT, (n_samples % 8 == 0 && cus % 8 == 0) ? conv_xcd_env() : 0);   // ntx, nty, nsamp, xcd
  const long long tiles = (long long)n_samples * a.ntx * a.nty * a.ncg;
  const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
  const dim3 g(a.xcd ? (unsigned)cus : grid);
#ifdef MISONET_EXPERIMENTS
  switch (wino_dbg_env()) {
    case 1: hipLaunchKernelGGL(conv3x3_wino_f32<1>, g, dim3(256), WINO_LDS, s, a); return hipGetLastError();
    case 2: hipLaunchKernelGGL(conv3x3_wino_f32<2>, g, dim3(256), WINO_LDS, s, a); return hipGetLastError();
    case 3: hipLaunchKernelGGL(conv3x3_wino_f32<3>, g, dim3(256), WINO_LDS, s, a); return hipGetLastError();
    case 7: hipLaunchKernelGGL(conv3x3_wino_f32<7>, g, dim3(256), WINO_LDS, s, a); return hipGetLastError();
    case 11: hipLaunchKernelGGL(conv3x3_wino_f32<11>, g, dim3(256), WINO_LDS, s, a); return hipGetLastError();
    case 15: hipLaunchKernelGGL(conv3x3_wino_f32<15>, g, dim3(256), WINO_LDS, s, a); return hipGetLastError();
    default: break;
  }
#endif
  hipLaunchKernelGGL(conv3x3_wino_f32<0>, g, dim3(256), WINO_LDS, s, a);
  return hipGetLastError();
}

}  // namespace mn
