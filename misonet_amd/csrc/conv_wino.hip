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
// channels) of all its tiles form ONE stream, also across tile boundaries.  While chunk g runs on the matrix pipe from its
// stage, chunk g + 1 is normalised from the RAW RING into its stage -- instance norm applied on the way, zero padding after
// it, the reference's order -- the U image of chunk g + 2 and the raw input of chunk g + 3 travel global -> LDS by DMA
// (`buffer_load ... lds`: no registers hold data in flight; with the input only ONE chunk ahead, in registers, 22 % of the
// kernel was s_waitcnt on HBM latency).  A wave stages exactly the raw words its own lanes' DMA wrote, so the raw ring needs
// no barrier, only the wave's own vmcnt.  Three stages, one workgroup barrier per chunk, inside the fourth K-step.
// The instruction stream is laid out by hand.  Measured on gfx950 (tools/micro/mfma_f32_valu.hip): VALU work does NOT hide
// behind v_mfma_f32_32x32x2_f32 -- the f32 MFMA runs at the f32 vector rate and a VALU instruction between two of them costs
// its full ~4.5 cycles plus ~12 for the first one (64 -> 105 cycles per MFMA with 8 v_add behind each), while LDS, VMEM and
// SALU instructions are free.  So a K-step is 16 MFMAs with only LDS reads / writes and global loads between them, followed by
// ONE clustered VALU group: the input transform of the next step as 16 packed adds (op_sel / neg modifiers; position nu = 2
// comes out negated, the weight image carries the same sign) and the instance norm of two staging items as packed FMAs.
// Everything else that would be VALU is gone: LDS addresses are per-stage base registers + immediates (the chunk loop is
// unrolled over the three stages), the chunk offset of a global load is an SGPR, rows outside the image read a (0, 0) norm
// entry instead of being masked, the weights go global -> LDS by DMA, frame masks exist only in the two ragged column tiles.
// Epilogue: inverse transform per lane (the 16 positions of a (channel, tile) are 16 accumulator registers of ONE lane),
// + bias, ELU, centring, 8-byte stores along T, exact statistics (det_stats.hpp) as in conv_epilogue.hpp.
#include "kernels.hpp"
#include "conv_epilogue.hpp"
#include "wino_regs.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <utility>

namespace mn {

typedef float wf16 __attribute__((ext_vector_type(16)));
typedef float wf4 __attribute__((ext_vector_type(4)));
typedef float wf2 __attribute__((ext_vector_type(2)));
typedef unsigned int wu2 __attribute__((ext_vector_type(2)));
#define MN_WLDS(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int WCK = 8;                         // input channels per chunk
constexpr int WTT = 64;                        // output frames per workgroup (32 tiles)
constexpr int WFT = 8;                         // output rows per workgroup (4 waves x 2)
constexpr int WNR = 10;                        // staged input rows
constexpr int WTW = 66;                        // floats per staged row: column c = frame t0 - 1 + c
constexpr int WIN_FLOATS = WCK * WNR * WTW;    // 5280
constexpr int WW_FLOATS = 16 * WCK * 32;       // 4096: [pos / 4][ci][co][pos % 4]
constexpr int WSTAGE_FLOATS = WIN_FLOATS + WW_FLOATS;
constexpr int WNSTAGE = 3;
constexpr int WNRM_MAX = 192;                  // input channels (s_nrm entries per parity); the network's widest DenseBlock conv has 192
constexpr int WCO_MAX = 64;                    // output channels (epilogue table entries)

// Z: the accumulator STARTS here (C = 0, the first K-step of a tile): no v_accvgpr_write pass over the 256 registers per tile
template <int P, bool Z = false>
__device__ __forceinline__ void wino_mfma(float uu, float vv) {
  if (Z) asm volatile("v_mfma_f32_32x32x2_f32 a[%0:%1], %2, %3, 0" ::"n"(16 * P), "n"(16 * P + 15), "v"(uu), "v"(vv) : W_ACLOB);
  else asm volatile("v_mfma_f32_32x32x2_f32 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(16 * P), "n"(16 * P + 15), "v"(uu), "v"(vv) : W_ACLOB);
}
// G16 (round 6): the 16-channel group of a 48-channel layer on v_mfma_f32_16x16x4_f32 -- M = 16 output channels, N = 16 tiles, K = 4
// input channels; accumulator block B = a[4 B : 4 B + 3] (layout checked by tools/micro/mfma16_layout.hip)
template <int B, bool Z = false>
__device__ __forceinline__ void wino_mfma16(float uu, float vv) {
  if (Z) asm volatile("v_mfma_f32_16x16x4_f32 a[%0:%1], %2, %3, 0" ::"n"(4 * B), "n"(4 * B + 3), "v"(uu), "v"(vv) : W_ACLOB);
  else asm volatile("v_mfma_f32_16x16x4_f32 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(4 * B), "n"(4 * B + 3), "v"(uu), "v"(vv) : W_ACLOB);
}
// packed f32 forms (semantics checked on the GPU by tools/micro/pk_opsel.hip)
__device__ __forceinline__ wf2 pk_add(wf2 a, wf2 b) { wf2 d; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ wf2 pk_sub(wf2 a, wf2 b) { wf2 d; asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
// (a0 - b0, a1 + b0)
__device__ __forceinline__ wf2 pk_t01(wf2 a, wf2 b) { wf2 d; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(a), "v"(b)); return d; }
// (a1 - b0, a1 - b1)
__device__ __forceinline__ wf2 pk_t23(wf2 a, wf2 b) { wf2 d; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
// (a0 + a1, a0 - a1)
__device__ __forceinline__ wf2 pk_spm(wf2 a) { wf2 d; asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(d) : "v"(a)); return d; }
// (a0 + b0, a1 - b1)
__device__ __forceinline__ wf2 pk_add_nh(wf2 a, wf2 b) { wf2 d; asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ wf2 pk_mul(wf2 a, wf2 b) { wf2 d; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ wf2 pk_fma(wf2 a, wf2 b, wf2 c) { wf2 d; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// The hazard recogniser does not look INSIDE inline asm: a hazard between an instruction in an asm block and a compiler-generated
// neighbour is ours to cover.  Two occur in the epilogue (round 6: a packed consumer scheduled directly behind v_exp_f32 gave
// timing-dependent garbage): (i) gfx950 needs one wait state between a TRANS instruction (v_exp_f32) and a VALU instruction that
// reads its result -- the exp2 below carries it; (ii) two wait states between a VALU write and a DPP read of the register --
// elu_pick's result feeds the DPP lane exchange of the stores.
__device__ __forceinline__ float exp2_ws(float x) {
  float r;
  asm volatile("v_exp_f32 %0, %1\n\ts_nop 0" : "=v"(r) : "v"(x));
  return r;
}
// y >= +0 ? pos : neg as ONE v_bfi_b32 on the sign of y (conv_epilogue.hpp elu_select: no compare, NaN-transparent)
__device__ __forceinline__ float elu_pick(float y, float neg, float pos) {
  // (a select on the SIGN BIT: v_cmp_gt_i32 + v_cndmask, scheduled by the compiler -- four of them per channel, so the compare ->
  // select wait states are covered by the neighbours; the asm v_bfi form needed an s_nop per value in front of the DPP exchange.
  // NaN-transparent like it: y = NaN selects NaN - c or exp2(NaN) + ..)
  return __builtin_bit_cast(int, y) < 0 ? neg : pos;
}
// LDS store as inline asm (for a C++ LDS store the compiler waits for every LDS-DMA in flight); a function, because an asm operand
// inside a generic lambda may not name a captured variable
template <int OFF>
__device__ __forceinline__ void lds_write_f32(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// (x0 * n0 + n1, x1 * n0 + n1)
__device__ __forceinline__ wf2 pk_nrm(wf2 x, wf2 nr) { wf2 d; asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(x), "v"(nr)); return d; }
// LDS map: 3 stages {normalised input [8][10][WTW] | U image 16 KB} | raw ring: 2 slots {80 items x 64 frames as the DMA leaves
// them | 256 halo words} | s_nrm[2][WNRM_MAX] float2 | s_zero[WNRM_MAX] float2 | s_red [4][32][2] | 64 dummy words | epilogue table [WCO_MAX] x {(b, b), (-c, -c), (-1 - c, -1 - c)}, c = ELU(b)
constexpr int WRAW_FLOATS = WCK * WNR * WTT + 256;
constexpr unsigned WRAW_B = (unsigned)(WNSTAGE * WSTAGE_FLOATS) * 4u;
constexpr unsigned WNRM_B = WRAW_B + 2u * WRAW_FLOATS * 4u;
constexpr unsigned WZERO_B = WNRM_B + 2u * WNRM_MAX * 8u;
constexpr unsigned WRED_B = WZERO_B + WNRM_MAX * 8u;
constexpr unsigned WDUMMY_B = WRED_B + 4u * 64u * 4u;
constexpr unsigned WBIAS_B = WDUMMY_B + 64u * 4u;       // per output channel: bias and the ELU centring constants as PAIRS for the packed epilogue (LDS: the epilogue must not queue VMEM loads behind the DMA)
constexpr size_t WINO_LDS = WBIAS_B + WCO_MAX * 24;
static_assert(WINO_LDS <= 160 * 1024, "one workgroup per CU: at most 160 KB of LDS");

// DBG (timing experiments only, -DMISONET_EXPERIMENTS + MISONET_WINO_DBG): 1 = no staging side work in the chunk loop (wrong
// results), 2 = no epilogue arithmetic / stores, 4 = no input transform, 8 / 16 / 32 = no weight DMA / input DMA / staging
// arithmetic + LDS traffic (the three parts of 1), 64 / 128 / 256 = no epilogue stores / statistics / ELU.
// G16: the instantiation for a 16-channel output group (the second group of the 48-channel layer dec6.db.c5, model.py:64-73: half of
// a 32-row MFMA would be padding -- 5.9 ms per step).  Same stream, staging, DMA and bookkeeping; what differs is the operand path:
//   v_mfma_f32_16x16x4_f32: lane (kk = lane >> 4, nn = lane & 15) <-> B operand V_p[ci = 4 s + kk][tile nn + 16 g], A operand
//   U_p[co = nn][ci = 4 s + kk]; a K-step = 4 input channels = 16 positions x 2 tile groups = 32 MFMAs of 32 cycles, two K-steps
//   per chunk; a lane transforms the patches of TWO tiles per K-step; 128 accumulators (block 2 p + g), register r of a block =
//   channel 4 kk + r.  The 64 MFMA slots of a chunk carry the same side work as the 64 of the 32-row body, re-timed so that the
//   staging is complete before the chunk barrier in front of the next stage's first operand fetch (slot 36 instead of 51).
template <int DBG, bool G16 = false>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32(const ConvArgs a) {
  extern __shared__ __align__(16) float smem[];
  char* const smem_c = reinterpret_cast<char*>(smem);
  wf2* s_nrm = reinterpret_cast<wf2*>(smem_c + WNRM_B);
  wf2* s_zero = reinterpret_cast<wf2*>(smem_c + WZERO_B);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int kk = lane >> 4, nn = lane & 15;      // G16 operand roles
  const int T = a.T, Tp = a.Tp, F = a.Fin, Cin = a.Cin;
  const int nchunk = Cin / WCK;
  constexpr unsigned WIMG_B = G16 ? 2u * 4u * 4u * 16u * 16u : (unsigned)(WW_FLOATS * 4);   // U image bytes per chunk: [s][pos / 4][kk][16 co][pos % 4] / [pos / 4][ci][32 co][pos % 4]

  // ---- this workgroup's tiles: q0, q0 + qstep, ... < Q of the linear (sample slot, row tile, frame tile, channel group) list of
  // its XCD.  STRIDED, not a contiguous range: at any moment the 32 workgroups of an XCD then work on 32 NEIGHBOURING tiles --
  // the two channel groups of a 48- / 64-channel layer (same input), tiles that share halo rows and columns -- and what one of
  // them fetched the others find in the XCD's L2 (contiguous ranges: 3.12 GB of HBM traffic per launch, 2.2 x the layer's bytes)
  const unsigned tps = (unsigned)(a.ntx * a.nty * a.ncg);
  unsigned q0, qstep, Q, xcd_id = 0;
  if (a.xcd) {
    xcd_id = blockIdx.x & 7u;
    q0 = blockIdx.x >> 3;
    qstep = gridDim.x >> 3;
    Q = (unsigned)(a.nsamp >> 3) * tps;
  } else {
    q0 = blockIdx.x;
    qstep = gridDim.x;
    Q = (unsigned)a.nsamp * tps;
  }
  if (q0 >= Q) return;
  const unsigned ntile = (Q - q0 + qstep - 1) / qstep;
  for (int i = tid; i < WNRM_MAX; i += 256) s_zero[i] = wf2{0.f, 0.f};
  if (tid < WCO_MAX) {
    const float b = tid < a.ncg * 32 ? a.bias[tid] : 0.f;
    const float c = elu_fast(b);
    wf2* tb = reinterpret_cast<wf2*>(smem_c + WBIAS_B + tid * 24);
    tb[0] = wf2{b, b}; tb[1] = wf2{-c, -c}; tb[2] = wf2{-1.f - c, -1.f - c};
  }

  const unsigned row_e = (unsigned)Tp;
  const unsigned plane_b = (unsigned)F * row_e * 4u;

  // ---- staging roles: thread (sq, scr) owns frames t0 + 4 sq .. + 3 (stage columns 1 + 4 sq .. 4 + 4 sq) of the (channel,
  // row) items scr + 16 i, i = 0..4; threads 0..159 own the halo columns 0 / 65 (frames t0 - 1 / t0 + 64) of item tid >> 1 ----
  const int sq = tid & 15, scr = tid >> 4;
  int sch[5];                                  // channel of item i; its stage row is scr + 16 i of [80][WTW]
#pragma unroll
  for (int i = 0; i < 5; ++i) sch[i] = (scr + 16 * i) / WNR;
  const bool hrole = tid < 2 * WCK * WNR;
  const int hit = hrole ? tid >> 1 : 0, hside = tid & 1;
  const int hch = hit / WNR, hrow = hit - hch * WNR;
  // per-stage LDS addresses (absolute; laundered through an empty asm: the compiler must keep them in registers instead of
  // re-deriving them from one base with VALU adds; every access is then base register + immediate.  The patch rows of a
  // K-step are read with ds_read2_b64, whose offsets reach 2 KB, so every (stage, step) has its own base)
  const unsigned lds0 = (unsigned)(unsigned long long)MN_WLDS(smem);
  // Rolling stage addresses: the chunk on the matrix pipe reads stage c (patch rows of its K-steps 1-3: d1-d3, U: uc) and, in
  // its last step, the first operands of stage n (d0n, un); the chunk being staged is written to stage n (cn, hn).  After a
  // chunk every register moves one stage on with ONE v_add of an SGPR (+ stride, or - 2 strides at the wrap), so the chunk
  // body names no stage and exists once per (frame masks, epilogue distance) instead of three times.
  constexpr unsigned STAGE_B = (unsigned)WSTAGE_FLOATS * 4u, STEP_B = (unsigned)((G16 ? 4 : 2) * WNR * WTW) * 4u;
  const unsigned dbase = G16 ? lds0 + (unsigned)((kk * WNR + 2 * wave) * WTW + 2 * nn) * 4u
                             : lds0 + (unsigned)((half * WNR + 2 * wave) * WTW + 2 * l31) * 4u;
  unsigned d1 = launder(dbase + STEP_B), d2 = launder(dbase + 2 * STEP_B), d3 = launder(dbase + 3 * STEP_B);
  unsigned d0n = launder(dbase + STAGE_B);
  unsigned uc = launder(lds0 + (unsigned)WIN_FLOATS * 4u + (unsigned)(G16 ? kk * 16 + nn : half * 32 + l31) * 16u);
  unsigned un = launder(uc + STAGE_B);
  unsigned cn = launder(lds0 + STAGE_B + (unsigned)(scr * WTW + 1 + 4 * sq) * 4u);
  unsigned hn = launder(lds0 + (hrole ? STAGE_B + (unsigned)(hit * WTW + (hside ? WTT + 1 : 0)) * 4u : WDUMMY_B + (unsigned)(tid & 63) * 4u));
#define W_LP(TYPE, ADDR) (reinterpret_cast<__attribute__((address_space(3))) TYPE*>(ADDR))
  // raw ring: what THIS thread's DMA lanes wrote (slot 0; slot 1 is WRAW_FLOATS * 4 further): item i at + i * 4096
  const unsigned rofs0 = lds0 + WRAW_B + (unsigned)(scr * WTT + 4 * sq) * 4u;
  const unsigned rhofs0 = lds0 + WRAW_B + (unsigned)(WCK * WNR * WTT + tid) * 4u;
  // DMA destinations are wave-uniform LDS addresses (the hardware adds lane * size)
  const unsigned rdma0 = WRAW_B + (unsigned)wave * 1024u;                              // + slot * WRAW_FLOATS * 4 + i * 4096
  const unsigned rhdma0 = WRAW_B + (unsigned)(WCK * WNR * WTT) * 4u + (unsigned)wave * 256u;
  const unsigned wdma0 = (unsigned)(WIN_FLOATS * 4) + (unsigned)wave * 1024u;          // + stage * WSTAGE_FLOATS * 4 + j * 4096
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc_e(reinterpret_cast<unsigned long long>(a.ww),
                                                   (unsigned)(a.ncg * nchunk) * WIMG_B);
  // (G16: the image is 8 KB = two pieces; pieces 2, 3 repeat 0, 1 into the same words -- the s_waitcnt immediates count four)
  unsigned wvo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wvo[j] = (unsigned)(tid + 256 * (G16 ? (j & 1) : j)) * 16u;

  // ---- load-side state L: the tile and chunk of the NEXT raw-input DMA (three chunks ahead of the matrix pipe).  What the
  // later steps need of a chunk's tile is latched when the chunk is issued and handed down L -> D -> C.  D: the chunk whose
  // U image is DMA'd in this iteration (g + 2); C: the chunk staged in this iteration (g + 1) ----
  unsigned ql = q0;
  int kl = 0;
  int Ln = -1, Lpar = 1, Lcg = 0;              // sample, s_nrm parity and channel group of the load tile
  int lslot = 0;                               // raw slot of the chunk L points at
  unsigned Lgoff[5], Lhoff = 0;
  unsigned Lnr[6];                             // LDS address of the norm entry of item i / the halo item at chunk 0: in s_nrm, or s_zero when the row (halo: the element) is outside the image
  bool Lfull = false;                          // every frame t0 .. t0 + 63 of the tile exists (no frame masks)
  unsigned Lmask = 0;                          // bits 0-3: frame t0 + 4 sq + j exists
  __amdgpu_buffer_rsrc_t rs_l;
  // C-side state (the chunk staged in this iteration): the norm-entry addresses ROLL (+ one chunk = 64 bytes per iteration) and
  // jump to the load side's base when the staged chunk is the first of the next tile -- the load side is on that tile by then and
  // stays on it for >= nchunk - 2 more iterations (round 6; rounds 5's L -> D -> C hand-down copied 18 registers per chunk)
  unsigned Cnr[6];
  bool Cfull = false;
  unsigned Cmask = 0;
  unsigned Dwso = 0;                           // byte offset of D's chunk in the weight image
  int cslot = 0;                               // raw slot of C's chunk
#pragma unroll
  for (int i = 0; i < 6; ++i) { Lnr[i] = lds0 + WZERO_B; Cnr[i] = lds0 + WZERO_B; }

#define W_DECODE(Q, TT_, FT_, N_, CG_)                                                                \
  {                                                                                                   \
    const unsigned q_ = __builtin_amdgcn_readfirstlane(Q);                                            \
    const unsigned j_ = q_ / tps;                                                                     \
    unsigned r_ = q_ - j_ * tps;                                                                      \
    N_ = (int)(a.xcd ? j_ * 8u + xcd_id : j_);                                                        \
    CG_ = (int)(r_ % (unsigned)a.ncg);                                                                \
    r_ /= (unsigned)a.ncg;                                                                            \
    TT_ = (int)(r_ % (unsigned)a.ntx);                                                                \
    FT_ = (int)(r_ / (unsigned)a.ntx);                                                                \
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
        s_nrm[Lpar * WNRM_MAX + c] = wf2{rstd, -mean * rstd};                                         \
      }                                                                                               \
      const float* in_n_ = a.in + (long long)n_ * a.in_bstride + (long long)a.in_c0 * F * Tp;         \
      rs_l = make_rsrc_e(reinterpret_cast<unsigned long long>(in_n_), (unsigned)Cin * plane_b);       \
    }                                                                                                 \
    Lcg = cg_;                                                                                        \
    const int tg_ = t0_ + 4 * sq;                                                                     \
    const int tge_ = tg_ + 4 <= Tp ? tg_ : Tp - 4;                                                    \
    Lfull = t0_ + WTT <= T;                                                                           \
    Lmask = (tg_ < T ? 1u : 0u) | (tg_ + 1 < T ? 2u : 0u) | (tg_ + 2 < T ? 4u : 0u) | (tg_ + 3 < T ? 8u : 0u); \
    _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                                   \
      int fin = fin0_ + (scr + 16 * i) - sch[i] * WNR;                                                \
      Lnr[i] = lds0 + ((fin >= 0 && fin < F) ? WNRM_B + (unsigned)(Lpar * WNRM_MAX + sch[i]) * 8u : WZERO_B); \
      fin = fin < 0 ? 0 : (fin >= F ? F - 1 : fin);                                                   \
      Lgoff[i] = (unsigned)sch[i] * plane_b + (unsigned)((fin * (int)row_e + tge_) * 4);              \
    }                                                                                                 \
    {                                                                                                 \
      const int htg_ = hside ? t0_ + WTT : t0_ - 1;                                                   \
      int fin = fin0_ + hrow;                                                                         \
      const bool hok_ = hrole && fin >= 0 && fin < F && htg_ >= 0 && htg_ < T;                        \
      Lnr[5] = lds0 + (hok_ ? WNRM_B + (unsigned)(Lpar * WNRM_MAX + hch) * 8u : WZERO_B);             \
      fin = fin < 0 ? 0 : (fin >= F ? F - 1 : fin);                                                   \
      const int th_ = (htg_ < 0 || htg_ >= T) ? 0 : htg_;   /* an element that does not exist: a word that does, times 0 (the padding may hold NaN) */ \
      Lhoff = (unsigned)hch * plane_b + (unsigned)((fin * (int)row_e + th_) * 4);                     \
    }                                                                                                 \
  }

  // hand the per-chunk state down (after the chunk L points at has been issued, before L advances)
#define W_LATCH                                                                                       \
  {                                                                                                   \
    Dwso = (unsigned)(Lcg * nchunk + kl) * WIMG_B;                                                    \
  }
#define W_ADVANCE                                                                                     \
  {                                                                                                   \
    lslot ^= 1;                                                                                       \
    if (++kl == nchunk) {                                                                             \
      if (ql + qstep < Q) { kl = 0; ql += qstep; W_LOAD_SETUP(ql) }                                             \
      else kl = nchunk - 1;                    /* end of the stream: the last chunk again (never consumed) */ \
    }                                                                                                 \
  }

  wf4 rwa, rwb;                                // raw words of the two items being staged
  float rh = 0.f;
  wf2 nra, nrb;
  wf2 cva[2], cvb[2];                          // two normalised items waiting for their LDS writes
  float chv = 0.f;
  unsigned rofs = rofs0, rhofs = rhofs0;       // raw-ring addresses of this thread in C's slot

#define W_CB ((unsigned)kl * (unsigned)WCK * plane_b)
#define W_ISSUE_I(I)                                                                                  \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_l, MN_WLDS(smem_c + rdma0 + lslot * (WRAW_FLOATS * 4) + (I) * 4096), 16, Lgoff[I], W_CB, 0, 0);
#define W_ISSUE_H                                                                                     \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_l, MN_WLDS(smem_c + rhdma0 + lslot * (WRAW_FLOATS * 4)), 4, Lhoff, W_CB, 0, 0);
#define W_ISSUE_W(J, ST, WSO)                                                                         \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, MN_WLDS(smem_c + (ST) * (WSTAGE_FLOATS * 4) + wdma0 + (G16 ? ((J) & 1) : (J)) * 4096), 16, wvo[J], WSO, 0, 0);
#define W_WSO_L ((unsigned)(Lcg * nchunk + kl) * WIMG_B)
#define W_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");
#define W_NR(DST, I) DST = *W_LP(const wf2, Cnr[I]);
#define W_RR(DST, I) DST = *W_LP(const wf4, rofs + (I) * 4096);
#define W_RRH rh = *W_LP(const float, rhofs);
  // item I: instance norm (a row outside the image reads scale = shift = 0); frames that do not exist occur only in the last
  // column tile of an utterance
#define W_CC(CV, RW, NR)                                                                              \
  {                                                                                                   \
    CV[0] = pk_nrm(wf2{RW.x, RW.y}, NR);                                                              \
    CV[1] = pk_nrm(wf2{RW.z, RW.w}, NR);                                                              \
    if (RAG) {                                                                                        \
      CV[0].x = (Cmask & 1u) ? CV[0].x : 0.f; CV[0].y = (Cmask & 2u) ? CV[0].y : 0.f;                 \
      CV[1].x = (Cmask & 4u) ? CV[1].x : 0.f; CV[1].y = (Cmask & 8u) ? CV[1].y : 0.f;                 \
    }                                                                                                 \
  }
#define W_CCH(NR) chv = fmaf(rh, NR.x, NR.y);
  // The stage writes are inline asm: for a C++ store to LDS the compiler waits for EVERY LDS-DMA in flight (s_waitcnt vmcnt(0):
  // it cannot tell the raw ring and the U images from the staged rows), which would take the DMA's two-chunk run-ahead away.
  // Nothing reads these words before the chunk barrier, in front of which the wave waits for lgkmcnt(0) by hand.
#define W_DSW(ADDR, VAL, OFF) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(ADDR), "v"(VAL), "n"(OFF) : "memory");
#define W_CW(CV, I)                                                                                   \
  {                                                                                                   \
    W_DSW(cn, CV[0].x, (I) * (16 * WTW * 4))                                                          \
    W_DSW(cn, CV[0].y, (I) * (16 * WTW * 4) + 4)                                                      \
    W_DSW(cn, CV[1].x, (I) * (16 * WTW * 4) + 8)                                                      \
    W_DSW(cn, CV[1].y, (I) * (16 * WTW * 4) + 12)                                                     \
  }
#define W_CWH W_DSW(hn, chv, 0)
  // workgroup barrier without the compiler's fence (which is s_waitcnt vmcnt(0) while LDS-DMA is in flight): every LDS access
  // of this wave has completed (lgkmcnt(0)); the DMA the OTHER waves must see is covered by the explicit vmcnt in front of it
#define W_BARRIER { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

  wf2 dd16[2][4][2];                           // G16: the raw patches of the next step's two tile groups
  wf2 vp16[2][4][2];                           // G16: their transforms
  wf2 dd[4][2];                                // raw patch of the next step: row i, columns (0, 1) / (2, 3)
  wf4 u[4];                                    // U operands: quad q = positions 4 q .. 4 q + 3; refilled quad by quad
  wf2 vp[4][2];                                // V operands of the running K-step: position 4 x + nu = vp[x][nu >> 1][nu & 1] (nu = 2 negated)
#define W_FD(I0, DREG)                                                                                \
  {                                                                                                   \
    const unsigned si_ = DREG;                                                                        \
    dd[I0][0] = *W_LP(const wf2, si_ + (I0) * (WTW * 4));                                             \
    dd[I0][1] = *W_LP(const wf2, si_ + (I0) * (WTW * 4) + 8);                                         \
    dd[I0 + 1][0] = *W_LP(const wf2, si_ + ((I0) + 1) * (WTW * 4));                                   \
    dd[I0 + 1][1] = *W_LP(const wf2, si_ + ((I0) + 1) * (WTW * 4) + 8);                               \
  }
#define W_FU(Q, UREG, S) u[Q] = *W_LP(const wf4, UREG + ((S) * 64 + (Q) * 256) * 16);
  // V = B^T d B as 16 packed adds: rows first (pairs of columns), then the two column pairs of every row
#define W_TRANSFORM                                                                                   \
  if (!(DBG & 4)) {                                                                                   \
    const wf2 c0a = pk_sub(dd[0][0], dd[2][0]), c0b = pk_sub(dd[0][1], dd[2][1]);                     \
    const wf2 c1a = pk_add(dd[1][0], dd[2][0]), c1b = pk_add(dd[1][1], dd[2][1]);                     \
    const wf2 c2a = pk_sub(dd[2][0], dd[1][0]), c2b = pk_sub(dd[2][1], dd[1][1]);                     \
    const wf2 c3a = pk_sub(dd[1][0], dd[3][0]), c3b = pk_sub(dd[1][1], dd[3][1]);                     \
    vp[0][0] = pk_t01(c0a, c0b); vp[0][1] = pk_t23(c0a, c0b);                                         \
    vp[1][0] = pk_t01(c1a, c1b); vp[1][1] = pk_t23(c1a, c1b);                                         \
    vp[2][0] = pk_t01(c2a, c2b); vp[2][1] = pk_t23(c2a, c2b);                                         \
    vp[3][0] = pk_t01(c3a, c3b); vp[3][1] = pk_t23(c3a, c3b);                                         \
  }

#define W_SB __builtin_amdgcn_sched_barrier(0);
#ifdef MISONET_EXPERIMENTS
  // timeline of ONE wave (MISONET_WINO_TIMELINE=<Cin>): s_memtime stamps of the iterations around the first tile epilogues
  unsigned long long* const tl_buf = (blockIdx.x == 8 && wave == 1) ? a.dbg_buf : nullptr;
  int tl_n = 0;
#define W_STAMP(ID) if (tl_buf && tl_n < 250) { if (lane == 0) { tl_buf[2 * tl_n] = (ID); tl_buf[2 * tl_n + 1] = __builtin_readcyclecounter(); } ++tl_n; }
#else
#define W_STAMP(ID)
#endif
#define W_MF(P, CS) wino_mfma<P, (POST == 2 && (CS) == 0)>(u[(P) >> 2][(P) & 3], vp[(P) >> 2][((P) & 3) >> 1][(P) & 1]);
#define W_NONE
#define W_X(...) if (!(DBG & (1 | 32))) { __VA_ARGS__ }      /* staging arithmetic, norm / raw reads, LDS writes */
#define W_XL(...) if (!(DBG & (1 | 16))) { __VA_ARGS__ }     /* raw input DMA */
#define W_XW(...) if (!(DBG & (1 | 8))) { __VA_ARGS__ }      /* weight DMA */
  // One K-step (CST, CS): 16 MFMAs with only LDS / DMA work between them, then ONE VALU group.  U quads are refilled as they
  // are consumed (quad 3 of THIS step in slot 0, quads 0-2 of the next step (FST, FS) behind the slots that used them); the
  // patch of the next step is fetched in slots 5-6 and transformed in the VALU group.  X1-X3: LDS writes of the previous
  // group's items; X7, X9-X11, X13, X14: norm entries and raw words for this group's items, or DMA pieces; XB: barrier (step
  // 3 only, after slot 3, in front of every access to the next chunk's stage); XG: this step's staging arithmetic.
#define W_STEP(CS, FD_, FU_, FS, X1, X2, X3, XB, X4, X5, X6, X7, X8, X9, X10, X11, X13, X14, XG)      \
  W_MF(0, CS) W_FU(3, uc, CS) W_SB                                                                    \
  W_MF(1, CS) X1 W_SB                                                                                 \
  W_MF(2, CS) X2 W_SB                                                                                 \
  W_MF(3, CS) X3 W_SB                                                                                 \
  XB                                                                                                  \
  W_MF(4, CS) W_FU(0, FU_, FS) X4 W_SB                                                                \
  W_MF(5, CS) W_FD(0, FD_) X5 W_SB                                                                    \
  W_MF(6, CS) W_FD(2, FD_) X6 W_SB                                                                    \
  W_MF(7, CS) X7 W_SB                                                                                 \
  W_MF(8, CS) W_FU(1, FU_, FS) X8 W_SB                                                                \
  W_MF(9, CS) X9 W_SB                                                                                 \
  W_MF(10, CS) X10 W_SB                                                                               \
  W_MF(11, CS) X11 W_SB                                                                               \
  W_MF(12, CS) W_FU(2, FU_, FS) W_SB                                                                  \
  W_MF(13, CS) X13 W_SB                                                                               \
  W_MF(14, CS) X14 W_SB                                                                               \
  W_MF(15, CS) W_SB                                                                                   \
  W_TRANSFORM XG W_SB

  // Chunk g on the matrix pipe from stage ST.  Chunk g + 1: raw ring -> stage STN (the wave's own DMA of two iterations ago:
  // behind it in the wave's VMEM queue are the 10 pieces of the last iteration, hence vmcnt(10)).  After the barrier: U image
  // of chunk g + 2 -> stage STNN, raw input of chunk g + 3 -> the raw slot this iteration has emptied.  In front of the
  // barrier the U image of chunk g + 1 must have landed: behind it are the 6 raw pieces of the same iteration, vmcnt(6).
  // vmcnt counts in issue order, stores included, and a tile epilogue queues exactly 16 stores (+ the statistics atomics, which
  // only make a wait more conservative) behind the DMA of its tile's last chunk: in the two iterations after an epilogue (POST =
  // 2, 1) the waits that look across them allow 16 more.  Six chunk bodies (frame masks x POST) carry the immediates; measured
  // alternatives: waiting for the stores instead costs their HBM round trip per tile (the epilogue appeared twice as expensive),
  // branching between two s_waitcnt immediates 13 % of the chunk time in instruction fetch, polling IB_STS.VM_CNT (readable:
  // tools/micro/ibsts_vmcnt.hip) 8 %.  The immediates are tied to the issue counts here -- change a W_ISSUE_* list or the epilogue's
  // stores and these fail to compile:
  constexpr int W_NDMA_W = 4, W_NDMA_I = 6, W_NSTORE = 16;       // per wave and iteration: U-image pieces, raw-input pieces (5 + halo); stores per tile epilogue
  static_assert(W_NDMA_W * 256 * 16 == WW_FLOATS * 4, "U image = 4 pieces of 256 threads x 16 bytes");
  static_assert(5 * 256 * 16 == WCK * WNR * WTT * 4 && 2 * WCK * WNR <= 256, "raw input = 5 pieces of 256 x 16 bytes + one 4-byte halo piece");
  static_assert(W_NDMA_W + W_NDMA_I == 10 && W_NDMA_I == 6 && 10 + W_NSTORE == 26 && 6 + W_NSTORE == 22,
                "the s_waitcnt vmcnt immediates of W_CHUNK_ (10 / 26 in step 0, 6 / 22 in front of the barrier)");
  // The BOOKKEEPING of the chunk stream lives inside step 3 (round 6).  With one wave per SIMD nothing hides an instruction
  // that is issued while the matrix pipe is idle: every one -- scalar, branch, LDS -- costs its ~4-cycle issue slot, and the
  // ~110 instructions that used to sit between two chunk bodies (state hand-down L -> D -> C, stream advance, stage
  // rotation, loop control) were 12 % of the kernel (timeline: 890 cycles between the last MFMA of a chunk and the first of
  // the next, against 116 for a K-step's own VALU group).  Behind an MFMA, scalar instructions are free: the DMA issue is
  // spread one piece per slot over slots 4-14 (two per slot cost 0.4 %: a DMA instruction takes ~30 cycles to issue, measured by
  // ablation -- the 10 pieces are ~6 % of the kernel), the scalar bookkeeping (W_BK_S1 .. S3) rides in slot 14, the vector part (W_BK_V: the norm-address
  // hand-down and the stage rotation -- VALU costs its time wherever it stands) joins the transform's VALU group.  What is left
  // between two bodies: the tile-end test, the load-side tile change (once per tile) and the body dispatch.
#define W_BK_S1 /* latch, scalar part; nxt0_: the chunk staged in the NEXT iteration is the first of the next tile */ \
  const bool nxt0_ = (kc + 2 == nchunk);                                                              \
  Cfull = nxt0_ ? Lfull : Cfull;                                                                      \
  Dwso = (unsigned)(Lcg * nchunk + kl) * WIMG_B;
#define W_BK_S2 /* advance the load side (the tile change itself -- W_LOAD_SETUP -- stays behind the body) */ \
  lslot ^= 1; ++kl;
#define W_BK_S3 /* stage rotation deltas, loop state */                                               \
  const unsigned dl_n_ = (ph3 == 1) ? 0u - 2u * STAGE_B : STAGE_B;      /* stage n = ph3 + 1 -> ph3 + 2 */ \
  ph3 = ph3 == 2 ? 0 : ph3 + 1;                                                                       \
  post = post > 0 ? post - 1 : 0;                                                                     \
  cslot ^= 1;                                                                                         \
  stnn = ph3 == 0 ? 2 : ph3 - 1;                       /* stage of the NEXT body's chunk g + 2 */     \
  const unsigned rsl_ = cslot ? (unsigned)(WRAW_FLOATS * 4) : 0u;
#define W_BK_V                                                                                        \
  if (__builtin_expect(nxt0_, 0)) {                                                                   \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) Cnr[i] = Lnr[i];                                    \
    Cmask = Lmask;                                                                                    \
  } else {                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) Cnr[i] += (unsigned)(WCK * 8);                      \
  }                                                                                                   \
  d1 = d0n + STEP_B; d2 = d0n + 2 * STEP_B; d3 = d0n + 3 * STEP_B;                                    \
  uc = un;                                                                                            \
  d0n += dl_n_; un += dl_n_; cn += dl_n_;                                                             \
  hn += hrole ? dl_n_ : 0u;                                                                           \
  rofs = rofs0 + rsl_; rhofs = rhofs0 + rsl_;
#define W_CHUNK_                                                                                      \
  W_STEP(0, d1, uc, 1, W_NONE, W_NONE, W_NONE, W_NONE, W_NONE, W_NONE, W_NONE,                        \
         W_X(W_STAMP(1) if (POST) { W_VMCNT(26) } else { W_VMCNT(10) } W_STAMP(2) W_RR(rwa, 0)), W_NONE, W_X(W_RR(rwb, 1)), W_X(W_NR(nra, 0)), W_X(W_NR(nrb, 1)), W_NONE, W_NONE, \
         W_X(W_CC(cva, rwa, nra) W_CC(cvb, rwb, nrb)))                                                \
  W_STEP(1, d2, uc, 2, W_X(W_CW(cva, 0)), W_X(W_CW(cvb, 1)), W_NONE, W_NONE, W_NONE, W_NONE, W_NONE,  \
         W_X(W_RR(rwa, 2)), W_NONE, W_X(W_RR(rwb, 3)), W_X(W_NR(nra, 2)), W_X(W_NR(nrb, 3)), W_NONE, W_NONE,  \
         W_X(W_CC(cva, rwa, nra) W_CC(cvb, rwb, nrb)))                                                \
  W_STEP(2, d3, uc, 3, W_X(W_CW(cva, 2)), W_X(W_CW(cvb, 3)), W_NONE, W_NONE, W_NONE, W_NONE, W_NONE,  \
         W_X(W_RR(rwa, 4)), W_NONE, W_X(W_RRH), W_X(W_NR(nra, 4)), W_X(W_NR(nrb, 5)), W_NONE, W_NONE, \
         W_X(W_CC(cva, rwa, nra) W_CCH(nrb)))                                                         \
  W_STEP(3, d0n, un, 0, W_X(W_CW(cva, 4)), W_X(W_CWH), W_NONE,                                        \
         W_STAMP(3) W_XW(if (POST == 2) { W_VMCNT(22) } else { W_VMCNT(6) }) W_STAMP(4) W_BARRIER W_STAMP(5), \
         W_XW(W_ISSUE_W(0, stnn, Dwso)), W_XW(W_ISSUE_W(1, stnn, Dwso)), W_XW(W_ISSUE_W(2, stnn, Dwso)), W_XW(W_ISSUE_W(3, stnn, Dwso)), \
         W_XL(W_ISSUE_I(0)), W_XL(W_ISSUE_I(1)), W_XL(W_ISSUE_I(2)), W_XL(W_ISSUE_I(3)), W_XL(W_ISSUE_I(4)),  \
         W_XL(W_ISSUE_H) W_BK_S1 W_BK_S2 W_BK_S3, W_BK_V)

  // ---- G16 chunk body (see the kernel's header comment).  Slot I of a K-step: position I >> 1, tile group I & 1. ----
#define W_FD16(G, I0, DREG)                                                                           \
  {                                                                                                   \
    const unsigned si_ = (DREG) + (G) * 128u;                                                         \
    dd16[G][I0][0] = *W_LP(const wf2, si_ + (I0) * (WTW * 4));                                        \
    dd16[G][I0][1] = *W_LP(const wf2, si_ + (I0) * (WTW * 4) + 8);                                    \
    dd16[G][I0 + 1][0] = *W_LP(const wf2, si_ + ((I0) + 1) * (WTW * 4));                              \
    dd16[G][I0 + 1][1] = *W_LP(const wf2, si_ + ((I0) + 1) * (WTW * 4) + 8);                          \
  }
#define W_FU16(Q, UREG, S) u[Q] = *W_LP(const wf4, UREG + ((S) * 4 + (Q)) * 1024);
#define W_TRANSFORM16(G)                                                                              \
  {                                                                                                   \
    const wf2 c0a = pk_sub(dd16[G][0][0], dd16[G][2][0]), c0b = pk_sub(dd16[G][0][1], dd16[G][2][1]); \
    const wf2 c1a = pk_add(dd16[G][1][0], dd16[G][2][0]), c1b = pk_add(dd16[G][1][1], dd16[G][2][1]); \
    const wf2 c2a = pk_sub(dd16[G][2][0], dd16[G][1][0]), c2b = pk_sub(dd16[G][2][1], dd16[G][1][1]); \
    const wf2 c3a = pk_sub(dd16[G][1][0], dd16[G][3][0]), c3b = pk_sub(dd16[G][1][1], dd16[G][3][1]); \
    vp16[G][0][0] = pk_t01(c0a, c0b); vp16[G][0][1] = pk_t23(c0a, c0b);                               \
    vp16[G][1][0] = pk_t01(c1a, c1b); vp16[G][1][1] = pk_t23(c1a, c1b);                               \
    vp16[G][2][0] = pk_t01(c2a, c2b); vp16[G][2][1] = pk_t23(c2a, c2b);                               \
    vp16[G][3][0] = pk_t01(c3a, c3b); vp16[G][3][1] = pk_t23(c3a, c3b);                               \
  }
#define W_MF16(I, CS) wino_mfma16<(I), (POST == 2 && (CS) == 0)>(u[(I) >> 3][((I) >> 1) & 3], vp16[(I) & 1][(I) >> 3][(((I) >> 1) & 3) >> 1][((I) >> 1) & 1]);
#define W_S16(I, CS, ...) W_MF16(I, CS) __VA_ARGS__ W_SB
#define W_CHUNK16_                                                                                    \
  /* K-step 0 (channels 0-3 of the chunk); operands of K-step 1 come from the same stage (d1, uc) */  \
  W_S16(0, 0, W_FU16(3, uc, 0))                                                                       \
  W_S16(1, 0, W_X(if (POST) { W_VMCNT(26) } else { W_VMCNT(10) } W_RR(rwa, 0)))                       \
  W_S16(2, 0, W_X(W_RR(rwb, 1)))                                                                      \
  W_S16(3, 0, W_X(W_NR(nra, 0)))                                                                      \
  W_S16(4, 0, W_X(W_NR(nrb, 1)))                                                                      \
  W_S16(5, 0) W_S16(6, 0)                                                                             \
  W_S16(7, 0, W_X(W_CC(cva, rwa, nra) W_CC(cvb, rwb, nrb)))                                           \
  W_S16(8, 0, W_FU16(0, uc, 1) W_X(W_CW(cva, 0)))                                                     \
  W_S16(9, 0, W_X(W_CW(cvb, 1)))                                                                      \
  W_S16(10, 0, W_FD16(0, 0, d1) W_X(W_RR(rwa, 2)))                                                    \
  W_S16(11, 0, W_FD16(0, 2, d1) W_X(W_RR(rwb, 3)))                                                    \
  W_S16(12, 0, W_FD16(1, 0, d1) W_X(W_NR(nra, 2)))                                                    \
  W_S16(13, 0, W_FD16(1, 2, d1) W_X(W_NR(nrb, 3)))                                                    \
  W_S16(14, 0) W_S16(15, 0)                                                                           \
  W_S16(16, 0, W_FU16(1, uc, 1))                                                                      \
  W_S16(17, 0) W_S16(18, 0)                                                                           \
  W_S16(19, 0, W_X(W_CC(cva, rwa, nra) W_CC(cvb, rwb, nrb)))                                          \
  W_S16(20, 0, W_X(W_CW(cva, 2)))                                                                     \
  W_S16(21, 0, W_X(W_CW(cvb, 3)))                                                                     \
  W_S16(22, 0, W_X(W_RR(rwa, 4)))                                                                     \
  W_S16(23, 0, W_X(W_RRH))                                                                            \
  W_S16(24, 0, W_FU16(2, uc, 1) W_X(W_NR(nra, 4)))                                                    \
  W_S16(25, 0, W_X(W_NR(nrb, 5)))                                                                     \
  W_S16(26, 0) W_S16(27, 0) W_S16(28, 0) W_S16(29, 0) W_S16(30, 0) W_S16(31, 0)                       \
  W_TRANSFORM16(0) W_TRANSFORM16(1) W_X(W_CC(cva, rwa, nra) W_CCH(nrb)) W_SB                          \
  /* K-step 1; the staging of chunk g + 1 is complete behind slot 1, the barrier stands behind slot 3, then the next stage's operands */ \
  W_S16(0, 1, W_FU16(3, uc, 1) W_X(W_CW(cva, 4)))                                                     \
  W_S16(1, 1, W_X(W_CWH))                                                                             \
  W_S16(2, 1) W_S16(3, 1)                                                                             \
  W_XW(if (POST == 2) { W_VMCNT(22) } else { W_VMCNT(6) }) W_BARRIER                                  \
  W_S16(4, 1, W_XW(W_ISSUE_W(0, stnn, Dwso)))                                                         \
  W_S16(5, 1, W_XW(W_ISSUE_W(1, stnn, Dwso)))                                                         \
  W_S16(6, 1, W_XW(W_ISSUE_W(2, stnn, Dwso)))                                                         \
  W_S16(7, 1, W_XW(W_ISSUE_W(3, stnn, Dwso)))                                                         \
  W_S16(8, 1, W_FU16(0, un, 0) W_XL(W_ISSUE_I(0)))                                                    \
  W_S16(9, 1, W_XL(W_ISSUE_I(1)))                                                                     \
  W_S16(10, 1, W_FD16(0, 0, d0n) W_XL(W_ISSUE_I(2)))                                                  \
  W_S16(11, 1, W_FD16(0, 2, d0n) W_XL(W_ISSUE_I(3)))                                                  \
  W_S16(12, 1, W_FD16(1, 0, d0n) W_XL(W_ISSUE_I(4)))                                                  \
  W_S16(13, 1, W_FD16(1, 2, d0n) W_XL(W_ISSUE_H))                                                     \
  W_S16(14, 1) W_S16(15, 1)                                                                           \
  W_S16(16, 1, W_FU16(1, un, 0))                                                                      \
  W_S16(17, 1) W_S16(18, 1) W_S16(19, 1) W_S16(20, 1) W_S16(21, 1) W_S16(22, 1) W_S16(23, 1)          \
  W_S16(24, 1, W_FU16(2, un, 0))                                                                      \
  W_S16(25, 1) W_S16(26, 1) W_S16(27, 1) W_S16(28, 1) W_S16(29, 1)                                    \
  W_S16(30, 1, W_BK_S1 W_BK_S2 W_BK_S3)                                                               \
  W_S16(31, 1)                                                                                        \
  W_TRANSFORM16(0) W_TRANSFORM16(1) W_BK_V W_SB
  // the body of an iteration: one chunk on the matrix pipe in the instantiation's operand form
#define W_BODY_ if constexpr (G16) { W_CHUNK16_ } else { W_CHUNK_ }

  // ---- prologue: raw chunks 0, 1, 2 and the U images of chunks 0, 1 on their way, chunk 0 staged, operands of (chunk 0,
  // step 0) fetched and transformed ----
  constexpr bool RAG = true;                         // (the prologue always applies the frame masks)
  W_LOAD_SETUP(ql)
#pragma unroll
  for (int i = 0; i < 6; ++i) Cnr[i] = Lnr[i];       // C = chunk 0 of the first tile (staged by this prologue)
  Cmask = Lmask; Cfull = Lfull;
  __syncthreads();                                   // s_nrm, s_zero visible
  W_ISSUE_W(0, 0, W_WSO_L) W_ISSUE_W(1, 0, W_WSO_L) W_ISSUE_W(2, 0, W_WSO_L) W_ISSUE_W(3, 0, W_WSO_L)
  W_ISSUE_I(0) W_ISSUE_I(1) W_ISSUE_I(2) W_ISSUE_I(3) W_ISSUE_I(4) W_ISSUE_H
  W_LATCH
  W_ADVANCE
  W_ISSUE_W(0, 1, W_WSO_L) W_ISSUE_W(1, 1, W_WSO_L) W_ISSUE_W(2, 1, W_WSO_L) W_ISSUE_W(3, 1, W_WSO_L)
  W_ISSUE_I(0) W_ISSUE_I(1) W_ISSUE_I(2) W_ISSUE_I(3) W_ISSUE_I(4) W_ISSUE_H
  W_LATCH                                            // D = chunk 1
  W_ADVANCE
  W_VMCNT(0)
  cn -= STAGE_B;                                     // (chunk 0 is staged to stage 0; the loop stages chunk g + 1 to the stage after g's)
  if (hrole) hn -= STAGE_B;
  W_RR(rwa, 0) W_NR(nra, 0) W_CC(cva, rwa, nra) W_CW(cva, 0)
  W_RR(rwa, 1) W_NR(nra, 1) W_CC(cva, rwa, nra) W_CW(cva, 1)
  W_RR(rwa, 2) W_NR(nra, 2) W_CC(cva, rwa, nra) W_CW(cva, 2)
  W_RR(rwa, 3) W_NR(nra, 3) W_CC(cva, rwa, nra) W_CW(cva, 3)
  W_RR(rwa, 4) W_NR(nra, 4) W_CC(cva, rwa, nra) W_CW(cva, 4)
  W_RRH W_NR(nrb, 5) W_CCH(nrb) W_CWH
#pragma unroll
  for (int i = 0; i < 6; ++i) Cnr[i] += (unsigned)(WCK * 8);   // C = chunk 1 (iteration 0 stages it)
  cn += STAGE_B;
  if (hrole) hn += STAGE_B;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's reads of raw slot 0 are done: chunk 2 may land there
  W_ISSUE_I(0) W_ISSUE_I(1) W_ISSUE_I(2) W_ISSUE_I(3) W_ISSUE_I(4) W_ISSUE_H
  W_LATCH                                            // D = chunk 2
  W_ADVANCE
  cslot = 1;
  rofs = rofs0 + WRAW_FLOATS * 4;
  rhofs = rhofs0 + WRAW_FLOATS * 4;
  W_BARRIER
  if constexpr (G16) {
    W_FD16(0, 0, dbase) W_FD16(0, 2, dbase) W_FD16(1, 0, dbase) W_FD16(1, 2, dbase) W_FU16(0, uc, 0) W_FU16(1, uc, 0) W_FU16(2, uc, 0)
    W_TRANSFORM16(0) W_TRANSFORM16(1)
  } else {
    W_FD(0, dbase) W_FD(2, dbase) W_FU(0, uc, 0) W_FU(1, uc, 0) W_FU(2, uc, 0)
    W_TRANSFORM
  }
  wfor<256>([&](auto i) __attribute__((always_inline)) { agpr_zero<decltype(i)::value>(); });
  asm volatile("s_nop 4");

  unsigned qc = q0;                                  // tile of the chunk on the matrix pipe
  int kc = 0;
  int ph3 = 0;                                       // stage of chunk g = g mod 3
  int post = 0;                                      // iterations since a tile epilogue: 2, 1, then 0
  const unsigned G = ntile * (unsigned)nchunk;
  int stnn = 2;                                      // stage of chunk g + 2 (the bodies keep it current)
  for (unsigned g = 0; g < G; ++g) {
    // six bodies: frame masks in the staging arithmetic or not (only the last column tile of an utterance stages frames that
    // do not exist) x the distance to the last tile epilogue (the immediates of two waits)
    if (Cfull) {
      constexpr bool RAG = false;
      if (post == 0) { constexpr int POST = 0; W_BODY_ } else if (post == 1) { constexpr int POST = 1; W_BODY_ } else { constexpr int POST = 2; W_BODY_ }
    } else {
      constexpr bool RAG = true;
      if (post == 0) { constexpr int POST = 0; W_BODY_ } else if (post == 1) { constexpr int POST = 1; W_BODY_ } else { constexpr int POST = 2; W_BODY_ }
    }
    if (++kc == nchunk) {
      // ---- tile epilogue: Y = A^T M A per (channel, tile), + bias, ELU, centring, stores, statistics ----
      asm volatile("s_nop 15\n\ts_nop 7");             // the last MFMA's result (16 passes) before any accumulator read
      W_STAMP(10)
      if (!(DBG & 2)) {
      int tt_, ft_, n, cg;
      W_DECODE(qc, tt_, ft_, n, cg)
      const int t0 = tt_ * WTT, f0 = ft_ * WFT;
      const int cbase = cg * 32;
      const int fa = f0 + 2 * wave;
      const bool r0ok = fa < F, r1ok = fa + 1 < F;
      const unsigned P4 = (unsigned)F * (unsigned)Tp * 4u;
      const float* ob_ = a.out + (long long)n * a.out_bstride + (long long)a.out_c0 * F * Tp;
      const __amdgpu_buffer_rsrc_t rs_out = make_rsrc_e(reinterpret_cast<unsigned long long>(ob_), (unsigned)a.Cout * P4);
      const unsigned red_a = lds0 + WRED_B;
      // The epilogue is VALU work the matrix pipe waits for (one wave per SIMD, and the f32 MFMA shares the vector ALU anyway):
      // it is written in PACKED f32 -- (e0, e1) = the two output columns of a position row, (y00, y01) / (y10, y11) = the two
      // frames of an output row -- 8 + 6 packed adds instead of 16 + 16 plain ones per channel, the ELU's scale / centring and
      // the statistics packed as well; per-channel constants come as pairs from the LDS table.
      // (laundered: the compiler otherwise re-materialises the pair from its constant for every channel)
      wf2 l2e = {1.44269504088896341f, 1.44269504088896341f};
      asm volatile("" : "+v"(l2e));
      // ONE (channel, tile) item: the 16 positions m(xi, nu) of the lane -> 2 x 2 outputs, + bias, ELU, centring, the 16-byte store
      // (neighbouring lanes exchange pairs: the even lane stores 4 frames of row fa, the odd lane 4 frames of row fa + 1), the
      // masked sums.  getm(xi, nu): the accumulator register of a position; tabp: the channel's LDS table entry.
      // The packed arithmetic is written as VECTOR expressions, not asm: the compiler forms the v_pk_*_f32 itself (op_sel / neg
      // modifiers folded) and knows that no wait state is needed between two of them -- behind every asm block it pads an s_nop,
      // and outside the MFMA shadow every instruction of a lone wave costs its ~4-cycle issue slot.
      auto epi_item = [&](auto getm, unsigned tabp, unsigned voff, bool evl, const wf2& mt, const wf2& mb, float& o1, float& o2) __attribute__((always_inline)) {
        const wf2 bb = *W_LP(const wf2, tabp);
        wf2 E[4];
        wfor<4>([&](auto xc) __attribute__((always_inline)) {
          constexpr int x = decltype(xc)::value;
          const float m0 = getm(xc, std::integral_constant<int, 0>{}), m1 = getm(xc, std::integral_constant<int, 1>{});
          const float m2 = getm(xc, std::integral_constant<int, 2>{}), m3 = getm(xc, std::integral_constant<int, 3>{});
          // (e0, e1) = (m0 + m1 + m2, m1 - m2 - M3) = (m1 + m2, m1 - m2) + (m0, m3): the weight image carries a minus sign at
          // nu = 3, so m3 = -M3 (and at xi = 3, so E[3] = -E3).  (Two plain adds for the mixed-sign step: its packed op_sel form
          // needs an asm block, and every asm block is padded with an s_nop on each side.)
          E[x] = wf2{m1 + m2, m1 - m2} + wf2{m0, m3};
        });
        wf2 yt = (E[0] + E[1]) + (E[2] + bb);                         // (y00, y01): row fa, frames t, t + 1
        wf2 yb = (E[1] - E[2]) + (bb + E[3]);                         // (y10, y11): row fa + 1
        if (!(DBG & 256)) {
          // ELU(y) - c, c = ELU(bias):  y >= 0 ? y - c : exp(y) - (1 + c)   (a select on the sign of y)
          const wf2 ncr = *W_LP(const wf2, tabp + 8), nc1 = *W_LP(const wf2, tabp + 16);
          const wf2 xt = yt * l2e, xb = yb * l2e;
          const wf2 et = wf2{__builtin_amdgcn_exp2f(xt.x), __builtin_amdgcn_exp2f(xt.y)} + nc1;
          const wf2 eb = wf2{__builtin_amdgcn_exp2f(xb.x), __builtin_amdgcn_exp2f(xb.y)} + nc1;
          const wf2 at = yt + ncr, ab = yb + ncr;
          yt = wf2{elu_pick(yt.x, et.x, at.x), elu_pick(yt.y, et.y, at.y)};
          yb = wf2{elu_pick(yb.x, eb.x, ab.x), elu_pick(yb.y, eb.y, ab.y)};
        }
        if (!(DBG & 64)) {
          // (quad_perm [1,0,3,2]: the neighbour's value)
          const float n00 = dpp_get<0xB1>(yt.x), n01 = dpp_get<0xB1>(yt.y), n10 = dpp_get<0xB1>(yb.x), n11 = dpp_get<0xB1>(yb.y);
          const wf4 o = {evl ? yt.x : n10, evl ? yt.y : n11, evl ? n00 : yb.x, evl ? n01 : yb.y};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rs_out, (DBG & 512) ? 0x80000000u : voff, 0, 2);   // non-temporal: read back by the NEXT launch, long after it left the L2
        }
        const wf2 zt = yt * mt, zb = yb * mb;
        const wf2 zs = zt + zb, zq = __builtin_elementwise_fma(zb, zb, zt * zt);
        o1 = zs.x + zs.y;
        o2 = zq.x + zq.y;
      };
      // (a store that is ISSUED out of range = dropped by the hardware: the s_waitcnt vmcnt immediates of the next two chunks count
      // exactly W_NSTORE stores per epilogue)
      auto dummy_store = [&]() __attribute__((always_inline)) {
        if (!(DBG & 64)) __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{0u, 0u, 0u, 0u}, rs_out, 0x80000000u, 0, 2);
      };
      if constexpr (G16) {
        // ---- 16-channel group: accumulator block 2 p + g (position p, tile group g), register r = channel 4 kk + r, tile nn + 16 g ----
        const bool ev = (nn & 1) == 0;
        const unsigned tab16 = launder(lds0 + WBIAS_B + (unsigned)(cbase + 4 * kk) * 24u);
        float s1g[8], s2g[8];
        wfor<2>([&](auto gc) __attribute__((always_inline)) {
          constexpr int g = decltype(gc)::value;
          const int t = t0 + 2 * (nn + 16 * g);
          const bool c0ok = t < T, c1ok = t + 1 < T;
          const unsigned vbase = (unsigned)(fa * Tp + t) * 4u + (unsigned)(cbase + 4 * kk) * P4;
          const unsigned vo_x = ev ? ((r0ok && c0ok) ? vbase : 0x80000000u)
                                   : ((r1ok && t - 2 < T) ? vbase + (unsigned)Tp * 4u - 8u : 0x80000000u);
          wf2 mt = {(r0ok && c0ok) ? 1.f : 0.f, (r0ok && c1ok) ? 1.f : 0.f}, mb = {(r1ok && c0ok) ? 1.f : 0.f, (r1ok && c1ok) ? 1.f : 0.f};
          asm volatile("" : "+v"(mt), "+v"(mb));
          wfor<4>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            epi_item([&](auto xc, auto nc) __attribute__((always_inline)) { return agpr_get<((4 * decltype(xc)::value + decltype(nc)::value) * 2 + g) * 4 + r>(); },
                     tab16 + r * 24, vo_x + (unsigned)r * P4, ev, mt, mb, s1g[4 * g + r], s2g[4 * g + r]);
          });
        });
        wfor<8>([&](auto) __attribute__((always_inline)) { dummy_store(); });
        W_STAMP(11)
        if (!(DBG & 128)) {
          // sums over the 16 tiles of a lane row (DPP: xor 1, xor 2, row_half_mirror, row_mirror) and the two tile groups
          wfor<4>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            float x1 = s1g[r] + s1g[4 + r], x2 = s2g[r] + s2g[4 + r];
            x1 += dpp_get<0xB1>(x1); x2 += dpp_get<0xB1>(x2);
            x1 += dpp_get<0x4E>(x1); x2 += dpp_get<0x4E>(x2);
            x1 += dpp_get<0x141>(x1); x2 += dpp_get<0x141>(x2);
            x1 += dpp_get<0x140>(x1); x2 += dpp_get<0x140>(x2);
            if (nn == 0) {
              const unsigned ra_ = red_a + (unsigned)((wave * 32 + 4 * kk + r) * 2) * 4u;
              lds_write_f32<0>(ra_, x1);
              lds_write_f32<4>(ra_, x2);
            }
          });
        }
      } else {
      const int t = t0 + 2 * l31;
      const bool c0ok = t < T, c1ok = t + 1 < T;
      const unsigned vbase = (unsigned)(fa * Tp + t) * 4u + (unsigned)(4 * half) * P4;
      // Stores are 16 bytes per lane (the store path is ISSUE-bound: 32 8-byte stores per lane cost 17 k cycles per tile, a
      // fifth of the kernel): neighbouring lanes exchange pairs, the even lane stores frames t .. t + 3 of row fa, the odd lane
      // frames t - 2 .. t + 1 of row fa + 1.  A group whose last frames are >= T writes words of the row's padding [T, Tp),
      // which every consumer masks.
      const bool ev = (l31 & 1) == 0;
      const unsigned vo_x = ev ? ((r0ok && c0ok) ? vbase : 0x80000000u)
                               : ((r1ok && t - 2 < T) ? vbase + (unsigned)Tp * 4u - 8u : 0x80000000u);
      // explicit LDS addresses (a generic-pointer access is a FLAT op, and a flat op waits for vmcnt(0) = for every store), the
      // table base laundered: one register + immediates instead of an address add per access (WBIAS_B is beyond the 16-bit offset)
      const unsigned tab_a = launder(lds0 + WBIAS_B + (unsigned)(cbase + 4 * half) * 24u);
      wf2 mt = {(r0ok && c0ok) ? 1.f : 0.f, (r0ok && c1ok) ? 1.f : 0.f}, mb = {(r1ok && c0ok) ? 1.f : 0.f, (r1ok && c1ok) ? 1.f : 0.f};
      asm volatile("" : "+v"(mt), "+v"(mb));
      float s1[16], s2[16];
      // Accumulator quad q (registers 4 q .. 4 q + 3) holds channels cbase + 8 q + {0 .. 3} + 4 half: with 24 output channels the
      // last of the four quads is padding in BOTH half-waves -- its channels have zero weights and no memory.  Their share of the
      // epilogue (reads, transform, ELU) is skipped: ONE wave-uniform branch per quad, the valid quad the fall-through (a taken
      // branch costs a lone wave ~30 cycles: the first version tested per register and cost the 32-channel layers what it saved
      // the 24-channel ones); their statistics entries are the zeros they would have computed.
      const int nquad = (a.Cout - cbase + 7) >> 3;           // valid quads of this group (>= 4: all)
      wfor<4>([&](auto qc_) __attribute__((always_inline)) {
        constexpr int qd = decltype(qc_)::value;
        if (__builtin_expect(qd >= nquad, 0)) {
          wfor<4>([&](auto r4) __attribute__((always_inline)) {
            constexpr int r = 4 * qd + decltype(r4)::value;
            dummy_store();
            s1[r] = 0.f; s2[r] = 0.f;
          });
          return;
        }
        wfor<4>([&](auto r4) __attribute__((always_inline)) {
          constexpr int r = 4 * qd + decltype(r4)::value;
          constexpr int kr = (r & 3) + 8 * (r >> 2);
          epi_item([&](auto xc, auto nc) __attribute__((always_inline)) { return agpr_get<(4 * decltype(xc)::value + decltype(nc)::value) * 16 + r>(); },
                   tab_a + kr * 24, vo_x + (unsigned)(cbase + kr) * P4, ev, mt, mb, s1[r], s2[r]);
        });
      });
      W_STAMP(11)
      if (!(DBG & 128)) {
        const float x1 = reduce16_halfwave(s1, lane);
        const float x2 = reduce16_halfwave(s2, lane);
        if ((lane & 16) == 0) {
          const int q = lane & 15;
          const int co_l = (q & 3) + 8 * (q >> 2) + 4 * half;
          W_DSW(red_a + (unsigned)((wave * 32 + co_l) * 2) * 4u, x1, 0)
          W_DSW(red_a + (unsigned)((wave * 32 + co_l) * 2) * 4u, x2, 4)
        }
      }
      }
      if (!(DBG & 128)) {
        W_BARRIER
        if (lane < 16) {                       // 16 of the 64 (channel, statistic) pairs per wave: no wave carries the whole tail
          const int pr = wave * 16 + lane;
          const int co_l = pr >> 1, which = pr & 1;
          const int co = cbase + co_l;
          if (co < a.Cout) {
            float tot = 0.f;
            for (int w = 0; w < 4; ++w)
              if (f0 + 2 * w < F) tot += *W_LP(const float, red_a + (unsigned)((w * 32 + co_l) * 2 + which) * 4u);
            dstat_add(a.out_stats + (((long long)n * a.out_sstride + a.out_c0 + co) * 2 + which) * DS_NL, (double)tot);
          }
        }
      }
      }
      W_STAMP(12)
      // (no zeroing: the next tile's first K-step starts its accumulators with C = 0 -- the POST == 2 chunk bodies; the
      // ablation builds that skip the epilogue keep post = 0 and therefore the explicit pass)
      if (DBG & (2 | 64)) {
        wfor<256>([&](auto i) __attribute__((always_inline)) { agpr_zero<decltype(i)::value>(); });
        asm volatile("s_nop 4");
      }
      W_STAMP(13)
      kc = 0;
      qc += qstep;
      post = (DBG & (2 | 64)) ? 0 : 2;               // (the body has already counted this iteration down)
    }
    if (kl == nchunk) {                              // the load side has finished a tile (the body advanced kl)
      if (ql + qstep < Q) { kl = 0; ql += qstep; W_LOAD_SETUP(ql) }
      else kl = nchunk - 1;                          // end of the stream: the last chunk again (never consumed)
    }
  }
}

// (act: every DenseBlock conv is followed by ELU + InstanceNorm, model.py:444-445 -- the epilogue has no other form.
// Cin >= 24: every sample then contributes >= 3 consecutive chunks to a workgroup's stream, which is what the two-parity
// s_nrm table and the look-back of the DMA waits assume; the network's DenseBlock layers have 24 ... 192 input channels)
bool conv_wino_ok(const ConvArgs& a) {
  return a.act && a.Cin >= 3 * WCK && a.Cout <= WCO_MAX && a.sf == 1 && a.padf == 1 && !a.tr2 && a.Fin == a.Fout && (a.Cin % WCK) == 0 && a.Cin <= WNRM_MAX && !a.in_oct &&
         !a.out_oct && a.ww != nullptr;
}

#ifdef MISONET_EXPERIMENTS
static int wino_dbg_env() {
  static const int v = exp_env("MISONET_WINO_DBG", 0);
  return v;
}
#endif

hipError_t conv_wino_init() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_f32<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)WINO_LDS);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_f32<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO_LDS);
#ifdef MISONET_EXPERIMENTS
#define W_ATTR(D) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_f32<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO_LDS);
  W_ATTR(1) W_ATTR(2) W_ATTR(3) W_ATTR(4) W_ATTR(7) W_ATTR(10) W_ATTR(18) W_ATTR(34) W_ATTR(64) W_ATTR(128) W_ATTR(256) W_ATTR(448) W_ATTR(512)
#undef W_ATTR
#endif
  return e;
}

// one launch of the persistent kernel over the channel groups of `a` (G16: one 16-channel group)
static hipError_t launch_wino_groups(ConvArgs a, int n_samples, hipStream_t s, bool g16);

hipError_t launch_conv_wino(const ConvArgs& a_in, int n_samples, hipStream_t s) {
  if (!conv_wino_ok(a_in)) return hipErrorInvalidValue;
  // A layer whose channel count leaves a 16-channel group (the 48-channel dec6.db.c5) runs as two launches: its full 32-channel
  // groups on the 32-row body, the last 16 channels on the 16-row body (a.ww16: their own weight image) -- half of a 32-row
  // MFMA would be padding there.
  if ((a_in.Cout & 31) == 16 && a_in.ww16 != nullptr) {
    const int c32 = a_in.Cout - 16;
    if (c32 > 0) {
      ConvArgs a = a_in;
      a.Cout = c32;
      const hipError_t e = launch_wino_groups(a, n_samples, s, false);
      if (e != hipSuccess) return e;
    }
    ConvArgs b = a_in;
    b.Cout = 16;
    b.out_c0 = a_in.out_c0 + c32;
    b.bias = a_in.bias + c32;
    b.ww = a_in.ww16;
    return launch_wino_groups(b, n_samples, s, true);
  }
  return launch_wino_groups(a_in, n_samples, s, false);
}

static hipError_t launch_wino_groups(ConvArgs a, int n_samples, hipStream_t s, bool g16) {
  a.cop = 32;
  a.ncg = g16 ? 1 : (a.Cout + 31) / 32;
  const int cus = device_cus();
  if (cus <= 0) return hipErrorInvalidDevice;
  (void)conv_grid(a, n_samples, WTT, WFT, (n_samples % 8 == 0 && cus % 8 == 0) ? conv_xcd_env() : 0);   // ntx, nty, nsamp, xcd
  const long long tiles = (long long)n_samples * a.ntx * a.nty * a.ncg;
  const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
  const dim3 g(a.xcd ? (unsigned)cus : grid);
  if (g16) {
    hipLaunchKernelGGL((conv3x3_wino_f32<0, true>), g, dim3(256), WINO_LDS, s, a);
    return hipGetLastError();
  }
#ifdef MISONET_EXPERIMENTS
  {
    static const int tl_cin = exp_env("MISONET_WINO_TIMELINE", 0);
    static unsigned long long* tl_dev = nullptr;
    static int tl_done = 0;
    if (tl_cin && a.Cin == tl_cin && a.Fin == 63 && n_samples == 96 && tl_done < 2) {
      if (!tl_dev) (void)hipMalloc(reinterpret_cast<void**>(&tl_dev), 512 * 8);
      (void)hipMemsetAsync(tl_dev, 0, 512 * 8, s);
      a.dbg_buf = tl_dev;
      hipLaunchKernelGGL(conv3x3_wino_f32<0>, g, dim3(256), WINO_LDS, s, a);
      (void)hipStreamSynchronize(s);
      unsigned long long h[512];
      (void)hipMemcpy(h, tl_dev, sizeof(h), hipMemcpyDeviceToHost);
      if (++tl_done == 2) {
        fprintf(stderr, "[wino timeline] Cin=%d nchunk=%d: id, cycles since previous stamp\n", a.Cin, a.Cin / 8);
        for (int i = 0; i < 250 && h[2 * i]; ++i)
          fprintf(stderr, "  %2llu %8lld\n", h[2 * i], i ? (long long)(h[2 * i + 1] - h[2 * i - 1]) : 0ll);
      }
      return hipGetLastError();
    }
  }
  switch (wino_dbg_env()) {
#define W_CASE(D) case D: hipLaunchKernelGGL(conv3x3_wino_f32<D>, g, dim3(256), WINO_LDS, s, a); return hipGetLastError();
    W_CASE(1) W_CASE(2) W_CASE(3) W_CASE(4) W_CASE(7) W_CASE(10) W_CASE(18) W_CASE(34) W_CASE(64) W_CASE(128) W_CASE(256) W_CASE(448) W_CASE(512)
#undef W_CASE
    default: break;
  }
#endif
  hipLaunchKernelGGL(conv3x3_wino_f32<0>, g, dim3(256), WINO_LDS, s, a);
  return hipGetLastError();
}

}  // namespace mn
