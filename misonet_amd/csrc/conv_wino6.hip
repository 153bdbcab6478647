// Winograd F(2x2, 3x3) form of the stride-1 same-padded 3x3 convs of the DenseBlocks (reference model.py:437-482) in the
// FP32-FAITHFUL bf16x6 arithmetic of conv_bf16x6.hip on the gfx950 bf16 matrix cores (v_mfma_f32_32x32x16_bf16).
// Precision mode "bf16x6w" (misonet_net.precision == 6).  VERDICT r4 item 1(c), the 2-D form.
//
//   Y = A^T [ U (.) V ] A,   V = B^T d B per (input channel, tile),   U = G g G^T per (co, ci)        (conv_wino.hip)
//   M_p[co][tile] = sum_ci U_p[co][ci] V_p[ci][tile] for the 16 positions p = (xi, nu), evaluated as the six leading partial
//   products of the exact three-piece bf16 splits U = U_h + U_m + U_l, V = V_h + V_m + V_l (conv_bf16x6.hip):
//       U_l V_h + U_h V_l + U_m V_m + U_m V_h + U_h V_m + U_h V_h          (small terms first, fp32 accumulation)
//   96 MFMAs per K-step (16 input channels x 32 output channels x 32 tiles) instead of the 216 of the direct form.
//
// What is different from the f32 Winograd kernel: V exists in fp32 only inside a lane, so the SPLIT is consumer-side VALU work
// (the direct kernel's producers write their outputs pre-split).  Measured on gfx950 (tools/micro/mfma_bf16_valu.hip,
// tools/micro/wino6_kstep.hip): plain VALU instructions issue beside a bf16 MFMA (6 per MFMA are free; v_pk_*_f32 and v_dot2*
// are not: 12 cycles each), and a K-step is VALU-bound: 1120 VALU instructions at 5.1 cycles + the MFMA issue slots =
// 6500 cycles against 3072 matrix cycles.
//
//   MFMA roles: M = 32 output channels (A = U pieces), N = 32 tiles = 64 frames of one tile row (B = V pieces), K = 16 input
//   channels: lanes 0-31 hold channels 0-7 of the K-step, lanes 32-63 channels 8-15 -- a lane owns the 4 x 4 patches of 8
//   channels of ITS tile.  Position row xi + 1 is prepared (instance norm folded into the row combination
//   t = (a * ra + sa) +- (b * rb + sb), 4 adds for the nu direction, the 11-instruction split of a pair of values) while the
//   24 MFMAs of row xi run.  One wave per SIMD, 256 fixed accumulator AGPRs; workgroup = 4 waves = 8 output rows x 64 frames.
//
// PERSISTENT: one workgroup per CU walks its XCD's tile list (conv_wino.hip); the K-steps of all its tiles form one stream.
// LDS: raw input double buffer, [16 channels][10 rows][66] fp32 per K-step (column c = frame t0 - 1 + c; columns 0-63 by
// LDS-DMA row by row, 64-65 by two DMA instructions into a side array and an LDS -> LDS move), a ring of five U QUARTERS (the
// three pieces of the four positions of one position row: 12 KB; a whole K-step of U is 48 KB and two of them do not fit beside
// the input), the norm tables.  One workgroup barrier
// per position row: it publishes the next row's U quarter (and, in row 2, the next K-step's input) and frees the quarter
// before; behind it the wave queues the quarter of the same row of the next K-step (and, in row 3, the input two K-steps
// ahead).
#include "kernels.hpp"
#include "conv_epilogue.hpp"
#include "wino_regs.hpp"
#include <stdio.h>
#include <stdlib.h>

namespace mn {

typedef unsigned xu4 __attribute__((ext_vector_type(4)));
typedef float xf2 __attribute__((ext_vector_type(2)));
typedef float xf4 __attribute__((ext_vector_type(4)));
#define MN_XLDS(p) ((__attribute__((address_space(3))) void*)(p))
#define X_LP(TYPE, ADDR) (reinterpret_cast<const __attribute__((address_space(3))) TYPE*>(ADDR))

constexpr int XCK = 16;                        // input channels per K-step
constexpr int XTT = 64, XFT = 8;               // output frames / rows per workgroup
constexpr int XNR = 10, XRW = 66;              // staged rows, floats per staged row (66: the two K-halves of a wave read disjoint LDS banks: 8 channels x 10 x 66 = 32 mod 64 dwords)
constexpr unsigned XSLOT_B = XCK * XNR * XRW * 4u;          // 42240
constexpr unsigned XQ_B = 4u * 3u * 64u * 16u;              // 12288: [nu][piece][lane] x 16 bytes
constexpr int XNQ = 5;
constexpr int XNRM_MAX = 256;
constexpr unsigned XRAW_B = 0;
constexpr unsigned XU_B = 2u * XSLOT_B;
constexpr unsigned XNRM_B = XU_B + XNQ * XQ_B;
constexpr unsigned XZERO_B = XNRM_B + 2u * XNRM_MAX * 8u;
constexpr unsigned XRED_B = XZERO_B + XNRM_MAX * 8u;
constexpr unsigned XBIAS_B = XRED_B + 4u * 64u * 4u;
constexpr unsigned XHALO_B = XBIAS_B + 128u * 4u;           // [wave][column 64 | 65][64 lanes] words: the two halo columns as the DMA leaves them
constexpr size_t WINO6_LDS = XHALO_B + 4 * 2 * 64 * 4;
static_assert(WINO6_LDS <= 160 * 1024, "one workgroup per CU: at most 160 KB of LDS");

template <int P>
__device__ __forceinline__ void x6_mfma(xu4 uu, xu4 vv) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(16 * P), "n"(16 * P + 15), "v"(uu), "v"(vv) : W_ACLOB);
}
__device__ __forceinline__ unsigned xfu(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float xuf(unsigned x) { return __uint_as_float(x); }
// exact split of a pair of fp32 values into three bf16 pairs at FIXED bit positions: h = the top 8 significant bits, w = the
// top 16, mid = w - h, lo = v - w (both exact, <= 8 significant bits).  Every piece depends on v only (depth 3).
__device__ __forceinline__ void x6_split(float v0, float v1, unsigned& H, unsigned& M, unsigned& L) {
  const float h0 = xuf(xfu(v0) & 0xffff0000u), h1 = xuf(xfu(v1) & 0xffff0000u);
  const float w0 = xuf(xfu(v0) & 0xffffff00u), w1 = xuf(xfu(v1) & 0xffffff00u);
  const float m0 = w0 - h0, m1 = w1 - h1, q0 = v0 - w0, q1 = v1 - w1;
  H = __builtin_amdgcn_perm(xfu(v1), xfu(v0), 0x07060302u);
  M = __builtin_amdgcn_perm(xfu(m1), xfu(m0), 0x07060302u);
  L = __builtin_amdgcn_perm(xfu(q1), xfu(q0), 0x07060302u);
}

// position row xi combines the staged patch rows (a, b): xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
template <int XI> struct XRows {
  static constexpr int a = XI == 0 ? 0 : (XI == 2 ? 2 : 1);
  static constexpr int b = XI == 0 ? 2 : (XI == 1 ? 2 : (XI == 2 ? 1 : 3));
  static constexpr bool plus = XI == 1;
};

struct XPipe {
  xu4 B[2][4][3];              // [buffer][nu][piece]: component q = channels 2 q | 2 q + 1 of the lane's 8
  xu4 U[2][3];                 // the three pieces of one position
  float ra[2][4], rb[2][4];    // raw words of the two channels of a pair unit (patch rows a, b; columns 0-3)
  xf2 na[2], nb[2];            // their norm entries (scale, shift), (0, 0) for a row outside the image
  float v[4][2];               // V[nu][channel of the pair]
};
// the lane's view of the K-step whose rows are being prepared
struct XPrep {
  unsigned rawc[8];            // LDS address of (channel j, patch row 0, column 2 tc) in the K-step's raw slot
  unsigned nrm[4];             // LDS address of the norm entry of (channel 0, patch row i): s_nrm of the sample, or s_zero
  unsigned cmask;              // bit k: frame t0 - 1 + 2 tc + k exists
};

template <int XI, int UNIT>
__device__ __forceinline__ void x6_fetch_raw(XPipe& p, const XPrep& s) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    constexpr int ra_ = XRows<XI>::a, rb_ = XRows<XI>::b;
    const int j = 2 * UNIT + c;
    const xf2 a0 = *X_LP(xf2, s.rawc[j] + ra_ * (XRW * 4)), a1 = *X_LP(xf2, s.rawc[j] + ra_ * (XRW * 4) + 8);
    const xf2 b0 = *X_LP(xf2, s.rawc[j] + rb_ * (XRW * 4)), b1 = *X_LP(xf2, s.rawc[j] + rb_ * (XRW * 4) + 8);
    p.ra[c][0] = a0.x; p.ra[c][1] = a0.y; p.ra[c][2] = a1.x; p.ra[c][3] = a1.y;
    p.rb[c][0] = b0.x; p.rb[c][1] = b0.y; p.rb[c][2] = b1.x; p.rb[c][3] = b1.y;
    p.na[c] = *X_LP(xf2, s.nrm[ra_] + j * 8);
    p.nb[c] = *X_LP(xf2, s.nrm[rb_] + j * 8);
  }
}
template <int XI, bool RAG>
__device__ __forceinline__ void x6_transform(XPipe& p, const XPrep& s, int c) {
  const float sh = XRows<XI>::plus ? p.na[c].y + p.nb[c].y : p.na[c].y - p.nb[c].y;
  const float rb = XRows<XI>::plus ? p.nb[c].x : -p.nb[c].x;
  float t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    t[k] = fmaf(p.rb[c][k], rb, fmaf(p.ra[c][k], p.na[c].x, sh));
    if (RAG) t[k] = (s.cmask >> k) & 1u ? t[k] : 0.f;       // a frame that does not exist is a ZERO of the normalised input
  }
  p.v[0][c] = t[0] - t[2]; p.v[1][c] = t[1] + t[2]; p.v[2][c] = t[2] - t[1]; p.v[3][c] = t[1] - t[3];
}
template <int POS>
__device__ __forceinline__ void x6_fetch_u(XPipe& p, int buf, unsigned uq) {
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) p.U[buf][pc] = *X_LP(xu4, uq + (unsigned)(((POS & 3) * 3 + pc) * 1024));
}

// DBG (timing experiments, -DMISONET_EXPERIMENTS + MISONET_WINO6_DBG): 1 = no preparation of the next row (wrong results),
// 2 = no epilogue, 4 = no DMA, 8 = no waits and barriers in the row loop, 16 / 32 = no input / weight DMA
template <int DBG>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_x6(const ConvArgs a) {
  extern __shared__ __align__(16) float smem[];
  char* const smem_c = reinterpret_cast<char*>(smem);
  xf2* s_nrm = reinterpret_cast<xf2*>(smem_c + XNRM_B);
  xf2* s_zero = reinterpret_cast<xf2*>(smem_c + XZERO_B);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int T = a.T, Tp = a.Tp, F = a.Fin, Cin = a.Cin;
  const int nk = (Cin + XCK - 1) / XCK;

  // ---- this workgroup's tiles (conv_wino.hip): q0, q0 + qstep, ... < Q of its XCD's (sample, row tile, frame tile, group) list
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
  for (int i = tid; i < XNRM_MAX; i += 256) s_zero[i] = xf2{0.f, 0.f};
  float* s_bias = reinterpret_cast<float*>(smem_c + XBIAS_B);
  if (tid < 128) s_bias[tid] = tid < a.ncg * 32 ? a.bias[tid] : 0.f;

  const unsigned plane_b = (unsigned)F * (unsigned)Tp * 4u;
  const unsigned lds0 = (unsigned)(unsigned long long)MN_XLDS(smem);
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc_e(reinterpret_cast<unsigned long long>(a.ww6), (unsigned)(a.ncg * nk) * (4u * XQ_B));
  const unsigned uvo = (unsigned)wave * 3072u + (unsigned)lane * 16u;

  // ---- a cursor of the K-step stream: tile q (decoded), K-step kk of it, s_nrm parity of its sample ----
  struct Cur { unsigned q; int kk, n, cg, t0, f0, par; };
  auto decode = [&](Cur& c) __attribute__((always_inline)) {
    const unsigned q_ = __builtin_amdgcn_readfirstlane(c.q);
    const unsigned j_ = q_ / tps;
    unsigned r_ = q_ - j_ * tps;
    const int n_ = (int)(a.xcd ? j_ * 8u + xcd_id : j_);
    c.cg = (int)(r_ % (unsigned)a.ncg);
    r_ /= (unsigned)a.ncg;
    c.t0 = (int)(r_ % (unsigned)a.ntx) * XTT;
    c.f0 = (int)(r_ / (unsigned)a.ntx) * XFT;
    if (n_ != c.n) { c.n = n_; c.par ^= 1; }
  };
  // the next K-step of the stream (the last one repeats: its loads are never consumed)
  auto advance = [&](Cur& c) __attribute__((always_inline)) {
    if (c.kk + 1 < nk) { ++c.kk; return false; }
    if (c.q + qstep >= Q) return false;
    c.q += qstep; c.kk = 0;
    const int n_old = c.n;
    decode(c);
    return c.n != n_old;
  };
  // instance-norm table of a sample (the load cursor fills it two K-steps before the first row of the sample is prepared)
  auto fill_nrm = [&](const Cur& c) __attribute__((always_inline)) {
    for (int ch = tid; ch < nk * XCK; ch += 256) {
      float mean = 0.f, rstd = ch < Cin ? 1.f : 0.f;
      if (ch >= a.ident_c && ch < Cin) {
        const dstat_t* st_ = a.in_stats + ((long long)c.n * a.in_sstride + a.in_c0 + ch) * (2 * DS_NL);
        const double cnt = (double)F * (double)T;
        const double m = dstat_read(st_) / cnt;
        double var = dstat_read(st_ + DS_NL) / cnt - m * m;
        var = var > 0.0 ? var : 0.0;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));
      }
      s_nrm[c.par * XNRM_MAX + ch] = xf2{rstd, -mean * rstd};
    }
  };
  auto rsrc_of = [&](const Cur& c) __attribute__((always_inline)) {
    const float* in_n_ = a.in + (long long)c.n * a.in_bstride + (long long)a.in_c0 * F * Tp;
    return make_rsrc_e(reinterpret_cast<unsigned long long>(in_n_), (unsigned)Cin * plane_b);
  };
  // ---- DMA of one K-step's raw input into raw slot `slot`: wave w owns channels 4 w .. 4 w + 3 of the K-step, one
  // 64-frame row (columns 0-63 = frames t0 - 1 .. t0 + 62) per instruction; rows and frames outside the image read a word
  // that exists (the zero norm entry / the frame mask removes it; the padding may hold anything).  Columns 64, 65 (frames
  // t0 + 63, t0 + 64) go through a side array (two more DMA instructions) and are moved into their rows in front of the publishing barrier.
  unsigned hla = 0;
  unsigned rowoff_v = 0;                     // lane r < 10: byte offset of staged row r of the load cursor's tile inside a channel plane
  auto row_setup = [&](const Cur& c) __attribute__((always_inline)) {
    int f = c.f0 - 1 + (lane < XNR ? lane : 0);
    f = f < 0 ? 0 : (f >= F ? F - 1 : f);
    rowoff_v = (unsigned)f * (unsigned)Tp * 4u;
  };
  // part 0..3: the ten rows of the wave's channel `part`; part 4: columns 64, 65 of the wave's 40 rows
  auto issue_raw = [&](const Cur& c, int slot, auto part_) __attribute__((always_inline)) {
    constexpr int part = decltype(part_)::value;
    if (DBG & (4 | 16)) return;
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(c);
    if constexpr (part < 4) {
      int fr = c.t0 - 1 + lane;
      fr = fr < 0 ? 0 : (fr >= Tp ? Tp - 1 : fr);
      const unsigned dvo = (unsigned)fr * 4u;
      const unsigned dst0 = XRAW_B + (unsigned)slot * XSLOT_B + (unsigned)(4 * wave + part) * (XNR * XRW * 4u);
      int ch = c.kk * XCK + 4 * wave + part;
      ch = ch < Cin ? ch : Cin - 1;
      const unsigned cho = (unsigned)ch * plane_b;
#pragma unroll
      for (int r = 0; r < XNR; ++r) {      // (row offsets of the tile: lane r of rowoff_v -- recomputing the clamp per row cost 12 SALU per DMA)
        const unsigned so = cho + (unsigned)__builtin_amdgcn_readlane((int)rowoff_v, r);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, MN_XLDS(smem_c + dst0 + (unsigned)r * (XRW * 4u)), 4, dvo, so, 0, 0);
      }
    } else {
      const int l = lane < 40 ? lane : 39;
      const int cc = l / XNR, r = l - cc * XNR;
      int ch = c.kk * XCK + 4 * wave + cc;
      ch = ch < Cin ? ch : Cin - 1;
      int f = c.f0 - 1 + r;
      f = f < 0 ? 0 : (f >= F ? F - 1 : f);
      int f1 = c.t0 + 63, f2 = c.t0 + 64;
      f1 = f1 >= Tp ? Tp - 1 : f1; f2 = f2 >= Tp ? Tp - 1 : f2;
      const unsigned ro = (unsigned)ch * plane_b + (unsigned)f * (unsigned)Tp * 4u;
      // lane l < 40 <-> (channel l / 10, row l % 10) of the wave's 40 rows: a DMA instruction writes 64 CONSECUTIVE words, so the
      // two columns land in a side array and are moved into their rows (LDS -> LDS, write_halo) in front of the publishing barrier
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, MN_XLDS(smem_c + XHALO_B + (unsigned)wave * 512u), 4, ro + (unsigned)f1 * 4u, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, MN_XLDS(smem_c + XHALO_B + (unsigned)wave * 512u + 256u), 4, ro + (unsigned)f2 * 4u, 0, 0, 0);
      hla = lds0 + XRAW_B + (unsigned)slot * XSLOT_B + (unsigned)((4 * wave + cc) * XNR + r) * (XRW * 4u) + 64u * 4u;
    }
  };
  auto issue_raw_all = [&](const Cur& c, int slot) __attribute__((always_inline)) {
    wfor<5>([&](auto pp) __attribute__((always_inline)) { issue_raw(c, slot, pp); });
  };
  // (asm: a C++ LDS access would make the compiler wait for every DMA in flight)
  const unsigned hsrc = lds0 + XHALO_B + (unsigned)wave * 512u + (unsigned)lane * 4u;
  auto write_halo = [&]() __attribute__((always_inline)) {
    if (DBG & (4 | 16)) return;
    unsigned h0, h1;
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:256\n\ts_waitcnt lgkmcnt(0)" : "=&v"(h0), "=&v"(h1) : "v"(hsrc) : "memory");
    if (lane < 40) {
      asm volatile("ds_write_b32 %0, %1" ::"v"(hla), "v"(h0) : "memory");
      asm volatile("ds_write_b32 %0, %1 offset:4" ::"v"(hla), "v"(h1) : "memory");
    }
  };
  // U quarter (position row xi) of the K-step of cursor c -> ring slot rq
  auto issue_u = [&](const Cur& c, int xi, int rq) __attribute__((always_inline)) {
    if (DBG & (4 | 32)) return;
    const unsigned so = (unsigned)((c.cg * nk + c.kk) * 4 + xi) * XQ_B;
    const unsigned dst = XU_B + (unsigned)rq * XQ_B + (unsigned)wave * 3072u;
    // (the instruction offset moves BOTH addresses: LDS address = M0 + offset + lane * 16)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, MN_XLDS(smem_c + dst), 16, uvo, so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, MN_XLDS(smem_c + dst), 16, uvo, so, 1024, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, MN_XLDS(smem_c + dst), 16, uvo, so, 2048, 0);
  };
  // the lane's view of a K-step for the preparation of its rows
  XPrep P;
  auto prep_of = [&](const Cur& c, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      P.rawc[j] = launder(lds0 + XRAW_B + (unsigned)slot * XSLOT_B + (unsigned)(((half * 8 + j) * XNR + 2 * wave) * XRW + 2 * l31) * 4u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = c.f0 - 1 + 2 * wave + i;
      const bool ok = f >= 0 && f < F;
      P.nrm[i] = launder(lds0 + (ok ? XNRM_B + (unsigned)(c.par * XNRM_MAX) * 8u : XZERO_B) + (unsigned)(c.kk * XCK + half * 8) * 8u);
    }
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int fr = c.t0 - 1 + 2 * l31 + k;
      m |= (fr >= 0 && fr < T) ? (1u << k) : 0u;
    }
    P.cmask = m;
  };
  auto full_of = [&](const Cur& c) __attribute__((always_inline)) { return c.t0 >= 1 && c.t0 + XTT + 1 <= T; };
#define X_BARRIER { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
#define X_VMCNT0 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- prologue: cursors C (K-step g on the matrix pipe), D = g + 1, L = g + 2 ----
  Cur C;
  C.q = q0; C.kk = 0; C.n = -1; C.par = 1;
  decode(C);
  Cur D = C, L;
  fill_nrm(C);
  if (advance(D)) fill_nrm(D);
  L = D;
  const bool l_new = advance(L);
  __syncthreads();                                   // s_zero, s_nrm of C (and D) visible
  row_setup(C);
  issue_raw_all(C, 0);
  X_VMCNT0
  write_halo();
  row_setup(D);
  issue_raw_all(D, 1);
  row_setup(L);
  issue_u(C, 0, 0); issue_u(C, 1, 1); issue_u(C, 2, 2); issue_u(C, 3, 3);
  X_VMCNT0
  // (slot 1's columns 64, 65 are written in front of the barrier of row 2 of the first K-step, like every later one's)
  X_BARRIER
  if (l_new) fill_nrm(L);                            // (two samples in flight: C's and L's; D's is one of them)

  XPipe p;
  prep_of(C, 0);
  bool Pfull = full_of(C);
  {
    constexpr bool RAG = true;
    wfor<4>([&](auto u) __attribute__((always_inline)) {
      constexpr int unit = decltype(u)::value;
      x6_fetch_raw<0, unit>(p, P);
      x6_transform<0, RAG>(p, P, 0); x6_transform<0, RAG>(p, P, 1);
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        unsigned H, M, Lo;
        x6_split(p.v[nu][0], p.v[nu][1], H, M, Lo);
        p.B[0][nu][0][unit] = H; p.B[0][nu][1][unit] = M; p.B[0][nu][2][unit] = Lo;
      }
    });
  }
  int uq_cur = 0;                                    // ring slot of the quarter on the matrix pipe
  unsigned uq_a = launder(lds0 + XU_B + (unsigned)lane * 16u);       // its LDS address for this lane
  x6_fetch_u<0>(p, 0, uq_a);
  wfor<256>([&](auto i) __attribute__((always_inline)) { agpr_zero<decltype(i)::value>(); });
  asm volatile("s_nop 4");

  const unsigned G = ntile * (unsigned)nk;
  // One position row XI of K-step g.  Slot s = MFMA (position 4 XI + s / 6, term s % 6).  Behind the MFMAs: the preparation of
  // row XN = XI + 1 (of K-step g + 1 when XI = 3) -- slot 0: raw words of pair unit 0; slots 3 + 5 u .. 7 + 5 u: unit u
  // (transform of both channels + the next unit's raw words, then the four splits); slot 11: the row's barrier and DMA.
#define X_ROW(XI, RAG_)                                                                                                   \
  wfor<24>([&](auto s_) __attribute__((always_inline)) {                                                                 \
    constexpr int S = decltype(s_)::value, nu = S / 6, term = S % 6, PP = (XI) * 4 + nu, cb = (XI) & 1, nb_ = cb ^ 1;     \
    constexpr int XN = ((XI) + 1) & 3, ub_ = PP & 1;                                                                      \
    constexpr int ap = term == 0 ? 2 : ((term == 2 || term == 3) ? 1 : 0);                                                \
    constexpr int bp = term == 1 ? 2 : ((term == 2 || term == 4) ? 1 : 0);                                                \
    x6_mfma<PP>(p.U[ub_][ap], p.B[cb][nu][bp]);                                                                           \
    if constexpr (term == 1) {                           /* the next position's pieces (first use 5 slots away) */       \
      if constexpr (nu < 3) x6_fetch_u<PP + 1>(p, ub_ ^ 1, uq_a);                                                                   \
      else x6_fetch_u<PP + 1>(p, ub_ ^ 1, uq_next);                                                                       \
    }                                                                                                                     \
    if (!(DBG & 1)) {                                                                                                     \
      if constexpr (S == 0) x6_fetch_raw<XN, 0>(p, P);                                                                    \
      if constexpr (S >= 3 && S < 23) {                                                                                           \
        constexpr int unit = (S - 3) / 5, k = (S - 3) % 5;                                                                \
        if constexpr (k == 0) {                                                                                           \
          x6_transform<XN, RAG_>(p, P, 0); x6_transform<XN, RAG_>(p, P, 1);                                               \
          if constexpr (unit < 3) x6_fetch_raw<XN, (unit + 1) & 3>(p, P);                                                         \
        } else {                                                                                                          \
          unsigned H, M, Lo;                                                                                              \
          x6_split(p.v[k - 1][0], p.v[k - 1][1], H, M, Lo);                                                               \
          p.B[nb_][k - 1][0][unit] = H; p.B[nb_][k - 1][1][unit] = M; p.B[nb_][k - 1][2][unit] = Lo;                      \
        }                                                                                                                 \
      }                                                                                                                   \
    }                                                                                                                     \
    if constexpr (S == 11) {                                                                                              \
      /* VMEM queue of a wave, in issue order, per K-step g: [row 0] U(g+1,0) x3  [row 1] U(g+1,1) x3  [row 2] U(g+1,2) x3   \
         [row 3] U(g+1,3) x3, input(g+2) x42.  The barrier of row xi publishes the quarter of row xi + 1 (queued one K-step  \
         ago) and, in row 2, the input of K-step g + 1: what may still be in flight behind them is 48 / 48 / 6 / 6          \
         instructions (a tile epilogue's stores, queued in front of row 0, only make rows 0 and 1 wait longer).  Then the   \
         quarter of this row of K-step g + 1 goes into the ring slot of the row before this one */                          \
      if (!(DBG & 8)) {                                                                                                   \
      if constexpr ((XI) < 2) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");                                           \
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                               \
      if constexpr ((XI) == 2) write_halo();                                                                              \
      X_BARRIER                                                                                                           \
      }                                                                                                                   \
      issue_u(D, (XI), uq_cur == 0 ? XNQ - 1 : uq_cur - 1);                                                               \
    }                                                                                                                     \
    /* row 3: the input of K-step g + 2 into the slot K-step g has emptied, spread over five slots */                     \
    if constexpr ((XI) == 3 && S >= 12 && S <= 20 && (S & 1) == 0)                                                        \
      issue_raw(L, (int)(g & 1u), std::integral_constant<int, (S - 12) / 2>{});                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  });

  for (unsigned g = 0; g < G; ++g) {
    unsigned uq_next;
#define X_ROWS(XI)                                                                                                        \
    uq_next = launder(lds0 + XU_B + (unsigned)(uq_cur == XNQ - 1 ? 0 : uq_cur + 1) * XQ_B + (unsigned)lane * 16u);       \
    if (Pfull) { X_ROW(XI, false) } else { X_ROW(XI, true) }                                                              \
    uq_cur = uq_cur == XNQ - 1 ? 0 : uq_cur + 1; uq_a = uq_next;
    X_ROWS(0)
    X_ROWS(1)
    X_ROWS(2)
    // row 3 prepares row 0 of K-step g + 1: the lane's view moves on (its input was published by the barrier of row 2)
    prep_of(D, (int)((g + 1u) & 1u));
    Pfull = full_of(D);
    X_ROWS(3)

    if (C.kk == nk - 1) {
      // ---- tile epilogue (conv_wino.hip): Y = A^T M A per (channel, tile), + bias, ELU, centring, stores, statistics ----
      asm volatile("s_nop 15\n\ts_nop 7");
      if (!(DBG & 2)) {
        const int t0 = C.t0, f0 = C.f0, n = C.n;
        const int cbase = C.cg * 32;
        const int fa = f0 + 2 * wave;
        const int t = t0 + 2 * l31;
        const bool r0ok = fa < F, r1ok = fa + 1 < F;
        const bool c0ok = t < T, c1ok = t + 1 < T;
        const unsigned P4 = (unsigned)F * (unsigned)Tp * 4u;
        const float* ob_ = a.out + (long long)n * a.out_bstride + (long long)a.out_c0 * F * Tp;
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc_e(reinterpret_cast<unsigned long long>(ob_), (unsigned)a.Cout * P4);
        const unsigned vbase = (unsigned)(fa * Tp + t) * 4u + (unsigned)(4 * half) * P4;
        const bool ev = (l31 & 1) == 0;
        const unsigned vo_x = ev ? ((r0ok && c0ok) ? vbase : 0x80000000u)
                                 : ((r1ok && t - 2 < T) ? vbase + (unsigned)Tp * 4u - 8u : 0x80000000u);
        const float m00 = (r0ok && c0ok) ? 1.f : 0.f, m01 = (r0ok && c1ok) ? 1.f : 0.f;
        const float m10 = (r1ok && c0ok) ? 1.f : 0.f, m11 = (r1ok && c1ok) ? 1.f : 0.f;
        const bool act = a.act != 0;
        const unsigned bias_a = lds0 + XBIAS_B + (unsigned)(4 * half) * 4u;
        const unsigned red_a = lds0 + XRED_B;
        float s1[16], s2[16];
        wfor<16>([&](auto rc) __attribute__((always_inline)) {
          constexpr int r = decltype(rc)::value;
          constexpr int kr = (r & 3) + 8 * (r >> 2);
          const unsigned coff = (unsigned)(cbase + kr) * P4;
          const float b = *X_LP(float, bias_a + (unsigned)(cbase + kr) * 4u);
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
          const float n00 = dpp_get<0xB1>(y00), n01 = dpp_get<0xB1>(y01), n10 = dpp_get<0xB1>(y10), n11 = dpp_get<0xB1>(y11);
          const xf4 o = {ev ? y00 : n10, ev ? y01 : n11, ev ? n00 : y10, ev ? n01 : y11};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rs_out, vo_x + coff, 0, 2);
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
            asm volatile("ds_write_b32 %0, %1" ::"v"(red_a + (unsigned)((wave * 32 + co_l) * 2) * 4u), "v"(x1) : "memory");
            asm volatile("ds_write_b32 %0, %1 offset:4" ::"v"(red_a + (unsigned)((wave * 32 + co_l) * 2) * 4u), "v"(x2) : "memory");
          }
          X_BARRIER
          if (lane < 16) {
            const int pr = wave * 16 + lane;
            const int co_l = pr >> 1, which = pr & 1;
            const int co = cbase + co_l;
            if (co < a.Cout) {
              float tot = 0.f;
              for (int w = 0; w < 4; ++w)
                if (f0 + 2 * w < F) tot += *X_LP(float, red_a + (unsigned)((w * 32 + co_l) * 2 + which) * 4u);
              dstat_add(a.out_stats + (((long long)n * a.out_sstride + a.out_c0 + co) * 2 + which) * DS_NL, (double)tot);
            }
          }
        }
      }
      wfor<256>([&](auto i) __attribute__((always_inline)) { agpr_zero<decltype(i)::value>(); });
      asm volatile("s_nop 4");
    }
    C = D;
    D = L;
    {
      const unsigned q_old = L.q;
      if (advance(L)) fill_nrm(L);
      if (L.q != q_old) row_setup(L);
    }
  }
}

bool conv_wino6_ok(const ConvArgs& a) {
  return a.Cin >= 24 && a.Cout <= 128 && a.sf == 1 && a.padf == 1 && !a.tr2 && a.Fin == a.Fout && (a.Cin % 8) == 0 && a.Cin <= XNRM_MAX && !a.in_oct &&
         !a.out_oct && a.ww6 != nullptr;
}

#ifdef MISONET_EXPERIMENTS
static int wino6_dbg_env() {
  static const int v = exp_env("MISONET_WINO6_DBG", 0);
  return v;
}
#define X6_DBGS(M) M(1) M(2) M(4) M(8) M(12) M(13) M(14) M(15) M(3) M(16) M(32) M(24) M(40)
#endif

hipError_t conv_wino6_init() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_x6<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO6_LDS);
#ifdef MISONET_EXPERIMENTS
#define X_ATTR(D) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_x6<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO6_LDS);
  X6_DBGS(X_ATTR)
#undef X_ATTR
#endif
  return e;
}

hipError_t launch_conv_wino6(const ConvArgs& a_in, int n_samples, hipStream_t s) {
  ConvArgs a = a_in;
  if (!conv_wino6_ok(a)) return hipErrorInvalidValue;
  a.cop = 32;
  a.ncg = (a.Cout + 31) / 32;
  const int cus = device_cus();
  if (cus <= 0) return hipErrorInvalidDevice;
  (void)conv_grid(a, n_samples, XTT, XFT, (n_samples % 8 == 0 && cus % 8 == 0) ? conv_xcd_env() : 0);
  const long long tiles = (long long)n_samples * a.ntx * a.nty * a.ncg;
  const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
  const dim3 g(a.xcd ? (unsigned)cus : grid);
#ifdef MISONET_EXPERIMENTS
  switch (wino6_dbg_env()) {
#define X_CASE(D) case D: hipLaunchKernelGGL(conv3x3_wino_x6<D>, g, dim3(256), WINO6_LDS, s, a); return hipGetLastError();
    X6_DBGS(X_CASE)
#undef X_CASE
    default: break;
  }
#endif
  hipLaunchKernelGGL(conv3x3_wino_x6<0>, g, dim3(256), WINO6_LDS, s, a);
  return hipGetLastError();
}

}  // namespace mn
