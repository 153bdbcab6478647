// conv3x3_bf16x3: the same 3x3 convolution family as conv.hip (reference model.py:401-482), computed on the gfx950
// bf16 matrix cores with a three-term split so that the result keeps float32-class accuracy:
//
//     x = x_hi + x_lo (+ O(2^-18 |x|)),  x_hi = bf16(x),  x_lo = bf16(x - x_hi)          (same for the weights)
//     w * x  ~=  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi          (dropped terms <= 3 * 2^-18 relative)
//
// Each product term is one v_mfma_f32_32x32x16_bf16 (f32 accumulate), i.e. 3 bf16 MFMAs replace 8 f32 MFMAs of
// conv.hip for the same 32x32x16 block: 5.3x the matrix-core rate at ~1e-5 relative error per layer (tolerance of
// the path: 1e-3).  Same GEMM roles, tile shape, normalise-on-load, statistics epilogue and launch geometry as
// conv3x3_mfma; differences:
//   * K-chunk = 16 input channels = one MFMA K; per lane an operand is 8 consecutive channels (16 bytes), so the
//     LDS images are [row][channel-octet h][frame][8 x bf16] (input, hi and lo) and [tap][h][cout][8 x bf16]
//     (weights, pre-split and pre-packed on the host) -- every ds_read_b128 / ds_write_b128 is conflict-free;
//   * staging: wave w owns (row, octet) pairs w, w+4, ...; lane l owns frames t0+2l, t0+2l+1 and loads them for the
//     8 channels of the octet (8 x 8-byte loads, coalesced along T), normalises, splits, and writes 4 x 16 bytes.
#include "kernels.hpp"
#include "conv_epilogue.hpp"
#include "conv_bf16_core.hpp"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <stdio.h>
#include <stdlib.h>

namespace mn {

template <int NCO, int MODE>
__global__ __launch_bounds__(256, ((NCO == 1 && MODE != 1) ? 2 : 1)) void conv3x3_bf16x3(const ConvArgs a) {
  constexpr int COP = NCO * 32;
  constexpr int NR = MODE == 0 ? 6 : (MODE == 1 ? 9 : 3);
  constexpr int SF = MODE == 1 ? 2 : 1;
  constexpr bool TR2 = MODE == 2;
  constexpr int NPAIR = 2 * NR;                      // (row, channel-octet) pairs per chunk
  constexpr int NPW = (NPAIR + 3) / 4;               // pairs per wave
  constexpr int NHT = (NPAIR * 16 + 255) / 256;      // halo scalars per thread
  constexpr int XN = NR * 2 * TW;                    // bf16x8 units per input image (hi or lo)
  constexpr int WN = 9 * 2 * COP;                    // bf16x8 units per weight image (hi or lo)
  constexpr int NWI = (2 * WN + 255) / 256;          // 16-byte weight units per thread
  extern __shared__ __align__(16) unsigned char smem_b[];
  bf16x8* s_xhi = reinterpret_cast<bf16x8*>(smem_b);
  bf16x8* s_xlo = s_xhi + XN;
  bf16x8* s_whi = s_xlo + XN;                        // hi image followed by lo image (as packed in HBM)
  bf16x8* s_wlo = s_whi + WN;
  float2* s_nrm = reinterpret_cast<float2*>(s_wlo + WN);
  float* s_bias = reinterpret_cast<float*>(s_nrm + ((a.Cin + CKB - 1) / CKB) * CKB);   // [COP]

  const unsigned long long ts0 = clock64();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ConvTile ct = conv_tile(a);
  if (!ct.valid) return;
  const int t0 = ct.t_tile * TT;
  const int f0 = ct.f_tile * FT;
  const int n = ct.n;
  const int cg = ct.cg;
  const int T = a.T, Tp = a.Tp, Fin = a.Fin, Cin = a.Cin;
  const int nchunk = (Cin + CKB - 1) / CKB;
  const int fin0 = TR2 ? (f0 >> 1) - 1 : SF * f0 - a.padf;

  for (int c = tid; c < nchunk * CKB; c += 256) {
    float mean = 0.f, rstd = (c < Cin) ? 1.f : 0.f;       // channels beyond Cin stage as zeros
    if (c >= a.ident_c && c < Cin) {
      const double* st = a.in_stats + ((long long)n * a.in_sstride + a.in_c0 + c) * 2;
      const double cnt = (double)Fin * (double)T;
      const double m = st[0] / cnt;
      double var = st[1] / cnt - m * m;
      var = var > 0.0 ? var : 0.0;
      mean = (float)m;
      rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));
    }
    s_nrm[c] = make_float2(rstd, -mean * rstd);            // (scale, shift): x_norm = fma(x, scale, shift)
  }

  if (tid < COP) s_bias[tid] = a.bias[cg * COP + tid];

  const float* in_n = a.in + (long long)n * a.in_bstride + (long long)a.in_c0 * Fin * Tp;
  const u32x4* w_g = reinterpret_cast<const u32x4*>(a.w16) + (long long)cg * nchunk * (2 * WN);

  // ---- staging roles: uniform buffer descriptor + per-lane byte offset (VGPR) + per-channel byte offset (SGPR) ----
  const unsigned row_e = (unsigned)Tp;
  const unsigned plane_e = (unsigned)Fin * row_e;
  const unsigned long long pa = reinterpret_cast<unsigned long long>(in_n);
  const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa);
  const unsigned phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)phi << 32) | plo), 0,
      __builtin_amdgcn_readfirstlane((int)((unsigned)Cin * plane_e * 4u)), 0x00020000);
  const int tl = t0 + 2 * lane;                                  // this lane's first frame
  const unsigned tl_e = (unsigned)(tl < Tp ? tl : Tp - 2);
  const bool full_t = (t0 + TT <= T);
  unsigned poff_b[NPW];                                          // byte offset of each owned pair's row (clamped)
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    int p = wave + 4 * i;
    p = p < NPAIR ? p : NPAIR - 1;
    int fin = fin0 + (p >> 1);
    fin = fin < 0 ? 0 : (fin >= Fin ? Fin - 1 : fin);
    poff_b[i] = ((unsigned)fin * row_e + tl_e) * 4u;
  }
  f32x2 pf[NPW][8];
  float ph[NHT];
  u32x4 pw[NWI];

#define BF_ISSUE(KC)                                                                                   \
  {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                  \
      const int p_ = wave + 4 * i;                                                                     \
      const int h_ = (p_ < NPAIR ? p_ : NPAIR - 1) & 1;                                                \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                  \
        int c_ = (KC) * CKB + 8 * h_ + e;                                                              \
        c_ = c_ < Cin ? c_ : Cin - 1;                                                                  \
        const u32x2 v_ = __builtin_amdgcn_raw_buffer_load_b64(                                         \
            rs_in, poff_b[i], __builtin_amdgcn_readfirstlane((unsigned)c_ * plane_e * 4u), 0);         \
        pf[i][e] = __builtin_bit_cast(f32x2, v_);                                                      \
      }                                                                                                \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NHT; ++i) {                                                  \
      const int k_ = tid + 256 * i;                                                                    \
      int p_ = k_ >> 4;                                                                                \
      p_ = p_ < NPAIR ? p_ : NPAIR - 1;                                                                \
      const int side_ = (k_ >> 3) & 1, e_ = k_ & 7;                                                    \
      int c_ = (KC) * CKB + 8 * (p_ & 1) + e_;                                                         \
      c_ = c_ < Cin ? c_ : Cin - 1;                                                                    \
      int fin_ = fin0 + (p_ >> 1);                                                                     \
      fin_ = fin_ < 0 ? 0 : (fin_ >= Fin ? Fin - 1 : fin_);                                            \
      int th_ = side_ ? t0 + TT : t0 - 1;                                                              \
      th_ = th_ < 0 ? 0 : (th_ >= Tp ? Tp - 1 : th_);                                                  \
      const unsigned ho_ = ((unsigned)c_ * plane_e + (unsigned)fin_ * row_e + (unsigned)th_) * 4u;     \
      ph[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, ho_, 0, 0));       \
    }                                                                                                  \
    const u32x4* wsrc_ = w_g + (unsigned)(KC) * (unsigned)(2 * WN);                                    \
    _Pragma("unroll") for (int i = 0; i < NWI; ++i) {                                                  \
      unsigned idx_ = tid + 256 * i;                                                                   \
      if ((2 * WN) % 256 != 0) idx_ = idx_ < (unsigned)(2 * WN) ? idx_ : (unsigned)(2 * WN - 1);       \
      pw[i] = wsrc_[idx_];                                                                             \
    }                                                                                                  \
  }

#define BF_COMMIT(KC)                                                                                  \
  {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                  \
      const int p_ = wave + 4 * i;                                                                     \
      if (p_ < NPAIR) {                                                                                \
        const int r_ = p_ >> 1, h_ = p_ & 1;                                                           \
        const int fin_ = fin0 + r_;                                                                    \
        const bool rok_ = fin_ >= 0 && fin_ < Fin;                                                     \
        const int o_ = (r_ * 2 + h_) * TW + 4 + 2 * lane;                                              \
        if (rok_) {                                                                                    \
          float y0_[8], y1_[8];                                                                        \
          _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                              \
            const float2 m_ = s_nrm[(KC) * CKB + 8 * h_ + e];                                          \
            y0_[e] = fmaf(pf[i][e].x, m_.x, m_.y);                                                     \
            y1_[e] = fmaf(pf[i][e].y, m_.x, m_.y);                                                     \
          }                                                                                            \
          if (!full_t) {                                                                               \
            const bool k0_ = tl + 0 < T, k1_ = tl + 1 < T;                                             \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                            \
              y0_[e] = k0_ ? y0_[e] : 0.f;                                                             \
              y1_[e] = k1_ ? y1_[e] : 0.f;                                                             \
            }                                                                                          \
          }                                                                                            \
          u32x4 h0_, l0_, h1_, l1_;                                                                    \
          _Pragma("unroll") for (int e2 = 0; e2 < 4; ++e2) {                                           \
            unsigned a_, b_;                                                                           \
            split_pair(y0_[2 * e2], y0_[2 * e2 + 1], a_, b_); h0_[e2] = a_; l0_[e2] = b_;              \
            split_pair(y1_[2 * e2], y1_[2 * e2 + 1], a_, b_); h1_[e2] = a_; l1_[e2] = b_;              \
          }                                                                                            \
          reinterpret_cast<u32x4*>(s_xhi)[o_] = h0_; reinterpret_cast<u32x4*>(s_xhi)[o_ + 1] = h1_;    \
          reinterpret_cast<u32x4*>(s_xlo)[o_] = l0_; reinterpret_cast<u32x4*>(s_xlo)[o_ + 1] = l1_;    \
        } else {                                                                                       \
          const u32x4 z_ = {0u, 0u, 0u, 0u};                                                           \
          reinterpret_cast<u32x4*>(s_xhi)[o_] = z_; reinterpret_cast<u32x4*>(s_xhi)[o_ + 1] = z_;      \
          reinterpret_cast<u32x4*>(s_xlo)[o_] = z_; reinterpret_cast<u32x4*>(s_xlo)[o_ + 1] = z_;      \
        }                                                                                              \
      }                                                                                                \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NHT; ++i) {                                                  \
      const int k_ = tid + 256 * i;                                                                    \
      const int p_ = k_ >> 4;                                                                          \
      if (p_ < NPAIR) {                                                                                \
        const int side_ = (k_ >> 3) & 1, e_ = k_ & 7;                                                  \
        const int r_ = p_ >> 1, h_ = p_ & 1;                                                           \
        const int c_ = (KC) * CKB + 8 * h_ + e_;                                                       \
        const int fin_ = fin0 + r_;                                                                    \
        const int th_ = side_ ? t0 + TT : t0 - 1;                                                      \
        const bool ok_ = fin_ >= 0 && fin_ < Fin && th_ >= 0 && th_ < T;                               \
        const float2 m_ = s_nrm[c_];                                                                   \
        const float x_ = ok_ ? fmaf(ph[i], m_.x, m_.y) : 0.f;                                          \
        __bf16 a_, b_;                                                                                 \
        split2(x_, a_, b_);                                                                            \
        const int o_ = ((r_ * 2 + h_) * TW + (side_ ? TT + 4 : 3)) * 8 + e_;                           \
        reinterpret_cast<__bf16*>(s_xhi)[o_] = a_;                                                     \
        reinterpret_cast<__bf16*>(s_xlo)[o_] = b_;                                                     \
      }                                                                                                \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NWI; ++i) {                                                  \
      const int idx_ = tid + 256 * i;                                                                  \
      if ((2 * WN) % 256 == 0 || idx_ < 2 * WN) reinterpret_cast<u32x4*>(s_whi)[idx_] = pw[i];         \
    }                                                                                                  \
  }

  const int f = f0 + wave;
  const bool row_ok = f < a.Fout;

  f32x16 acc[NCO][4];
#pragma unroll
  for (int j = 0; j < NCO; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][s][r] = 0.f;

  const int half = lane >> 5, l31 = lane & 31;

  const bool stamp = a.dbg_buf && tid == 0 && ct.t_tile == 3 && ct.f_tile == 5 && n == 7 && cg == 0;
  int si = 0;
#define STAMP() do { if (stamp && si < 60) a.dbg_buf[si++] = clock64() - ts0; } while (0)
  STAMP();
  BF_ISSUE(0)
  __syncthreads();          // s_nrm visible
  STAMP();
  BF_COMMIT(0)
  __syncthreads();
  STAMP();

  for (int kc = 0; kc < nchunk; ++kc) {
    const bool more = (kc + 1 < nchunk);
    if (more) BF_ISSUE(kc + 1)
    STAMP();
    if (row_ok) {
      __builtin_amdgcn_s_setprio(1);       // MFMA phase outranks the co-resident block's staging phase (+1 % measured)
      if (TR2) {
        if ((f - f0) & 1) chunk_mfma_bf16<NCO, NR, SF, TR2, 2>(acc, s_xhi, s_xlo, s_whi, s_wlo, f - f0, half, l31);
        else chunk_mfma_bf16<NCO, NR, SF, TR2, 5>(acc, s_xhi, s_xlo, s_whi, s_wlo, f - f0, half, l31);
      } else {
        chunk_mfma_bf16<NCO, NR, SF, TR2, 7>(acc, s_xhi, s_xlo, s_whi, s_wlo, f - f0, half, l31);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    STAMP();
    __syncthreads();
    STAMP();
    if (more) {
      BF_COMMIT(kc + 1)
      STAMP();
      __syncthreads();
      STAMP();
    }
  }
#undef BF_ISSUE
#undef BF_COMMIT

  // ---- epilogue (conv_epilogue.hpp) ----
  float* s_red = reinterpret_cast<float*>(smem_b);   // [FT][COP][2]
  conv_epilogue<NCO, 4, true>(a, acc, n, cg, f, t0, row_ok, lane, s_red + wave * (COP * 2), s_bias);
  STAMP();
  if (a.act) {
    __syncthreads();
    if (tid < COP * 2) {
      const int co_l = tid >> 1, which = tid & 1;
      const int co = cg * COP + co_l;
      if (co < a.Cout) {
        float tot = 0.f;
        for (int w = 0; w < FT; ++w)
          if (f0 + w < a.Fout) tot += s_red[(w * COP + co_l) * 2 + which];
        unsafeAtomicAdd(a.out_stats + ((long long)n * a.out_sstride + a.out_c0 + co) * 2 + which, (double)tot);
      }
    }
  }
  STAMP();
  if (stamp) a.dbg_buf[63] = si;
#undef STAMP
}

// =====================================================================================================================
// Persistent, warp-specialised variant (MODE 0 and 2, 32 output channels per pass):
//   * one workgroup of 8 waves per CU, looping over output tiles (tile = blockIdx.x + i * gridDim.x);
//   * waves 0-3 = consumers: one output row each, MFMAs only (chunk_mfma_bf16), plus the tile epilogue;
//   * waves 4-7 = producers: global loads -> normalise -> bf16 split -> LDS for the NEXT (tile, K-chunk) job, running
//     one job ahead across tile boundaries, so staging, tile prologues and most of the epilogue overlap the MFMAs;
//   * LDS holds two complete stage buffers; ONE workgroup barrier per job.
// Job j = (tile j / nchunk, chunk j % nchunk).  Step j: producers commit job j+1 into buffer (j+1)&1 and issue the
// loads of job j+2; consumers run the MFMAs of job j from buffer j&1 (and the epilogue if it closes a tile).
// Per-tile instance-norm parameters live in s_nrm[tile & 1]; per-tile statistics partials in s_red[tile & 1] are
// flushed (summed over the 4 rows, one float64 atomic per channel) by a producer wave one step later.
// Workgroup barrier that waits only for this wave's LDS traffic: global loads (producer prefetch) and global stores
// (consumer epilogue) stay in flight across it -- __syncthreads() would drain vmcnt(0) every step.
#define WS_BARRIER()                                      \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("" ::: "memory");                        \
  } while (0)

struct TileInfo { int t0, f0, n, cg, fin0; };

template <bool TR2>
__device__ __forceinline__ TileInfo decode_tile(int tile, int ntt, int ntf, int ncg, int padf) {
  TileInfo ti;
  const int bt = tile % ntt;
  const int r1 = tile / ntt;
  const int bf = r1 % ntf;
  const int z = r1 / ntf;
  ti.t0 = bt * TT;
  ti.f0 = bf * FT;
  ti.n = z / ncg;
  ti.cg = z - ti.n * ncg;
  ti.fin0 = TR2 ? (ti.f0 >> 1) - 1 : ti.f0 - padf;
  return ti;
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void conv3x3_bf16x3_ws(const ConvArgs a, int ntt, int ntf, int ntiles) {
  constexpr int COP = 32;
  constexpr int NR = MODE == 0 ? 6 : 3;
  constexpr bool TR2 = MODE == 2;
  constexpr int NPAIR = 2 * NR;
  constexpr int NPW = (NPAIR + 3) / 4;
  constexpr int NHT = (NPAIR * 16 + 255) / 256;
  constexpr int XN = NR * 2 * TW;
  constexpr int WN = 9 * 2 * COP;
  constexpr int NWI = (2 * WN + 255) / 256;
  constexpr int BUFN = 2 * XN + 2 * WN;              // bf16x8 units per stage buffer
  constexpr int MAXC = 256;
  extern __shared__ __align__(16) unsigned char smem_b[];
  bf16x8* s_buf = reinterpret_cast<bf16x8*>(smem_b);                     // [2][BUFN]
  float2* s_nrm = reinterpret_cast<float2*>(s_buf + 2 * BUFN);           // [2][MAXC]
  float* s_red = reinterpret_cast<float*>(s_nrm + 2 * MAXC);             // [2][FT][COP][2]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int T = a.T, Tp = a.Tp, Fin = a.Fin, Cin = a.Cin;
  const int nchunk = (Cin + CKB - 1) / CKB;
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int J = my_tiles * nchunk;
  const unsigned row_e = (unsigned)Tp;
  const unsigned plane_e = (unsigned)Fin * row_e;
  const long long plane_ll = (long long)Fin * Tp;

  auto tile_of = [&](int j) { return (int)blockIdx.x + (j / nchunk) * (int)gridDim.x; };

  // instance-norm parameters of the input channels of tile `tile` -> s_nrm[par]   (producer threads, ptid 0..255)
  auto compute_nrm = [&](int tile, int par, int ptid) {
    const TileInfo ti = decode_tile<TR2>(tile, ntt, ntf, a.ncg, a.padf);
    for (int c = ptid; c < nchunk * CKB; c += 256) {
      float mean = 0.f, rstd = (c < Cin) ? 1.f : 0.f;     // channels beyond Cin stage as zeros
      if (c >= a.ident_c && c < Cin) {
        const double* st = a.in_stats + ((long long)ti.n * a.in_sstride + a.in_c0 + c) * 2;
        const double cnt = (double)Fin * (double)T;
        const double m = st[0] / cnt;
        double var = st[1] / cnt - m * m;
        var = var > 0.0 ? var : 0.0;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));
      }
      s_nrm[par * MAXC + c] = make_float2(rstd, -mean * rstd);   // (scale, shift): x_norm = fma(x, scale, shift)
    }
  };

  if (producer) {
    // ================================================= producers ==================================================
    const int ptid = tid - 256;
    const int pwv = wave - 4;
    // two prefetch register sets: the loads of job j+2 are issued BEFORE job j+1 is committed, so every load has a
    // full step of latency budget even when the producers are the critical path
    f32x2 pfA[NPW][8], pfB[NPW][8];
    float phA[NHT], phB[NHT];
    u32x4 pwA[NWI], pwB[NWI];

#define WS_ISSUE(JOB, pf, ph, pw)                                                                                \
  {                                                                                                    \
    const int tile_ = tile_of(JOB);                                                                    \
    const int kc_ = (JOB) % nchunk;                                                                    \
    const TileInfo ti_ = decode_tile<TR2>(tile_, ntt, ntf, a.ncg, a.padf);                             \
    const float* in_n_ = a.in + (long long)ti_.n * a.in_bstride + (long long)a.in_c0 * plane_ll;       \
    /* buffer addressing: uniform descriptor + per-lane byte offset (VGPR) + per-channel byte offset (SGPR) */ \
    const unsigned long long pa_ = reinterpret_cast<unsigned long long>(in_n_);                        \
    const unsigned plo_ = __builtin_amdgcn_readfirstlane((unsigned)pa_);                               \
    const unsigned phi_ = __builtin_amdgcn_readfirstlane((unsigned)(pa_ >> 32));                       \
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                              \
        reinterpret_cast<void*>(((unsigned long long)phi_ << 32) | plo_), 0,                           \
        __builtin_amdgcn_readfirstlane((int)((unsigned)Cin * plane_e * 4u)), 0x00020000);              \
    const int tl_ = ti_.t0 + 2 * lane;                                                                 \
    const unsigned tl_e_ = (unsigned)(tl_ < Tp ? tl_ : Tp - 2);                                        \
    _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                  \
      int p_ = pwv + 4 * i;                                                                            \
      p_ = p_ < NPAIR ? p_ : NPAIR - 1;                                                                \
      int fin_ = ti_.fin0 + (p_ >> 1);                                                                 \
      fin_ = fin_ < 0 ? 0 : (fin_ >= Fin ? Fin - 1 : fin_);                                            \
      const unsigned po_ = ((unsigned)fin_ * row_e + tl_e_) * 4u;                                      \
      int c0_ = kc_ * CKB + 8 * (p_ & 1);                                                              \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                  \
        int c_ = c0_ + e;                                                                              \
        c_ = c_ < Cin ? c_ : Cin - 1;                                                                  \
        const u32x2 v_ = __builtin_amdgcn_raw_buffer_load_b64(                                       \
            rs_, po_, __builtin_amdgcn_readfirstlane((unsigned)c_ * plane_e * 4u), 0);                 \
        pf[i][e] = __builtin_bit_cast(f32x2, v_);                                                      \
      }                                                                                                \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NHT; ++i) {                                                  \
      const int k_ = ptid + 256 * i;                                                                   \
      int p_ = k_ >> 4;                                                                                \
      p_ = p_ < NPAIR ? p_ : NPAIR - 1;                                                                \
      const int side_ = (k_ >> 3) & 1, e_ = k_ & 7;                                                    \
      int c_ = kc_ * CKB + 8 * (p_ & 1) + e_;                                                          \
      c_ = c_ < Cin ? c_ : Cin - 1;                                                                    \
      int fin_ = ti_.fin0 + (p_ >> 1);                                                                 \
      fin_ = fin_ < 0 ? 0 : (fin_ >= Fin ? Fin - 1 : fin_);                                            \
      int th_ = side_ ? ti_.t0 + TT : ti_.t0 - 1;                                                      \
      th_ = th_ < 0 ? 0 : (th_ >= Tp ? Tp - 1 : th_);                                                  \
      const unsigned ho_ = ((unsigned)c_ * plane_e + (unsigned)fin_ * row_e + (unsigned)th_) * 4u;     \
      ph[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_, ho_, 0, 0));         \
    }                                                                                                  \
    const u32x4* wsrc_ = reinterpret_cast<const u32x4*>(a.w16) +                                       \
                         ((long long)ti_.cg * nchunk + kc_) * (2 * WN);                                \
    _Pragma("unroll") for (int i = 0; i < NWI; ++i) {                                                  \
      unsigned idx_ = ptid + 256 * i;                                                                  \
      if ((2 * WN) % 256 != 0) idx_ = idx_ < (unsigned)(2 * WN) ? idx_ : (unsigned)(2 * WN - 1);       \
      pw[i] = wsrc_[idx_];                                                                             \
    }                                                                                                  \
  }

#define WS_COMMIT(JOB, pf, ph, pw)                                                                               \
  {                                                                                                    \
    const int tidx_ = (JOB) / nchunk;                                                                  \
    const int kc_ = (JOB) - tidx_ * nchunk;                                                            \
    const int tile_ = (int)blockIdx.x + tidx_ * (int)gridDim.x;                                        \
    const TileInfo ti_ = decode_tile<TR2>(tile_, ntt, ntf, a.ncg, a.padf);                             \
    const float2* nrm_ = s_nrm + (tidx_ & 1) * MAXC;                                                   \
    bf16x8* xhi_ = s_buf + ((JOB) & 1) * BUFN;                                                         \
    bf16x8* xlo_ = xhi_ + XN;                                                                          \
    bf16x8* wdst_ = xlo_ + XN;                                                                         \
    const int tl_ = ti_.t0 + 2 * lane;                                                                 \
    const bool full_ = (ti_.t0 + TT <= T);                                                             \
    _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                  \
      const int p_ = pwv + 4 * i;                                                                      \
      if (p_ < NPAIR) {                                                                                \
        const int r_ = p_ >> 1, h_ = p_ & 1;                                                           \
        const int fin_ = ti_.fin0 + r_;                                                                \
        const bool rok_ = fin_ >= 0 && fin_ < Fin;                                                     \
        const int o_ = (r_ * 2 + h_) * TW + 4 + 2 * lane;                                              \
        if (rok_) {                                                                                    \
          float y0_[8], y1_[8];                                                                        \
          _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                              \
            const float2 m_ = nrm_[kc_ * CKB + 8 * h_ + e];                                            \
            y0_[e] = fmaf(pf[i][e].x, m_.x, m_.y);                                                     \
            y1_[e] = fmaf(pf[i][e].y, m_.x, m_.y);                                                     \
          }                                                                                            \
          if (!full_) {                                                                                \
            const bool k0_ = tl_ + 0 < T, k1_ = tl_ + 1 < T;                                           \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                            \
              y0_[e] = k0_ ? y0_[e] : 0.f;                                                             \
              y1_[e] = k1_ ? y1_[e] : 0.f;                                                             \
            }                                                                                          \
          }                                                                                            \
          u32x4 h0_, l0_, h1_, l1_;                                                                    \
          _Pragma("unroll") for (int e2 = 0; e2 < 4; ++e2) {                                           \
            unsigned a_, b_;                                                                           \
            split_pair(y0_[2 * e2], y0_[2 * e2 + 1], a_, b_); h0_[e2] = a_; l0_[e2] = b_;              \
            split_pair(y1_[2 * e2], y1_[2 * e2 + 1], a_, b_); h1_[e2] = a_; l1_[e2] = b_;              \
          }                                                                                            \
          reinterpret_cast<u32x4*>(xhi_)[o_] = h0_; reinterpret_cast<u32x4*>(xhi_)[o_ + 1] = h1_;      \
          reinterpret_cast<u32x4*>(xlo_)[o_] = l0_; reinterpret_cast<u32x4*>(xlo_)[o_ + 1] = l1_;      \
        } else {                                                                                       \
          const u32x4 z_ = {0u, 0u, 0u, 0u};                                                           \
          reinterpret_cast<u32x4*>(xhi_)[o_] = z_; reinterpret_cast<u32x4*>(xhi_)[o_ + 1] = z_;        \
          reinterpret_cast<u32x4*>(xlo_)[o_] = z_; reinterpret_cast<u32x4*>(xlo_)[o_ + 1] = z_;        \
        }                                                                                              \
      }                                                                                                \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NHT; ++i) {                                                  \
      const int k_ = ptid + 256 * i;                                                                   \
      const int p_ = k_ >> 4;                                                                          \
      if (p_ < NPAIR) {                                                                                \
        const int side_ = (k_ >> 3) & 1, e_ = k_ & 7;                                                  \
        const int r_ = p_ >> 1, h_ = p_ & 1;                                                           \
        const int c_ = kc_ * CKB + 8 * h_ + e_;                                                        \
        const int fin_ = ti_.fin0 + r_;                                                                \
        const int th_ = side_ ? ti_.t0 + TT : ti_.t0 - 1;                                              \
        const bool ok_ = fin_ >= 0 && fin_ < Fin && th_ >= 0 && th_ < T;                               \
        const float2 m_ = nrm_[c_];                                                                    \
        const float x_ = ok_ ? fmaf(ph[i], m_.x, m_.y) : 0.f;                                          \
        __bf16 a_, b_;                                                                                 \
        split2(x_, a_, b_);                                                                            \
        const int o_ = ((r_ * 2 + h_) * TW + (side_ ? TT + 4 : 3)) * 8 + e_;                           \
        reinterpret_cast<__bf16*>(xhi_)[o_] = a_;                                                      \
        reinterpret_cast<__bf16*>(xlo_)[o_] = b_;                                                      \
      }                                                                                                \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NWI; ++i) {                                                  \
      const int idx_ = ptid + 256 * i;                                                                 \
      if ((2 * WN) % 256 == 0 || idx_ < 2 * WN) reinterpret_cast<u32x4*>(wdst_)[idx_] = pw[i];         \
    }                                                                                                  \
  }

    // statistics of the tile closed at job JB (sum over the 4 rows -> one float64 atomic per channel)
    auto flush_stats = [&](int jb) {
      if (!a.act || pwv != 0) return;
      const int tidx = jb / nchunk;
      const TileInfo ti = decode_tile<TR2>((int)blockIdx.x + tidx * (int)gridDim.x, ntt, ntf, a.ncg, a.padf);
      const float* red = s_red + (tidx & 1) * (FT * COP * 2);
      const int co_l = lane >> 1, which = lane & 1;
      const int co = ti.cg * COP + co_l;
      if (co < a.Cout) {
        float tot = 0.f;
        for (int w = 0; w < FT; ++w)
          if (ti.f0 + w < a.Fout) tot += red[(w * COP + co_l) * 2 + which];
        unsafeAtomicAdd(a.out_stats + ((long long)ti.n * a.out_sstride + a.out_c0 + co) * 2 + which, (double)tot);
      }
    };

    compute_nrm(tile_of(0), 0, ptid);
    if (nchunk == 1 && J > 1) compute_nrm(tile_of(1), 1, ptid);
    WS_ISSUE(0, pfA, phA, pwA)
    if (J > 1) WS_ISSUE(1, pfB, phB, pwB)
    __syncthreads();                                   // (P1) s_nrm visible
    WS_COMMIT(0, pfA, phA, pwA)
    __syncthreads();                                   // (P2) job 0 staged
    // step j: issue job j+2 into the set job j used, commit job j+1 from the other set, one barrier
#define WS_STEP(JJ, pfI, phI, pwI, pfC, phC, pwC)                                                      \
  {                                                                                                    \
    const int j_ = (JJ);                                                                               \
    if (!(a.dbg & 2)) {                                                                                \
      if (j_ + 2 < J) WS_ISSUE(j_ + 2, pfI, phI, pwI)                                                  \
      if (j_ + 1 < J) WS_COMMIT(j_ + 1, pfC, phC, pwC)                                                 \
      if (j_ + 2 < J && (j_ + 2) % nchunk == 0) compute_nrm(tile_of(j_ + 2), ((j_ + 2) / nchunk) & 1, ptid); \
    }                                                                                                  \
    if (j_ >= 1 && (j_ % nchunk) == 0) flush_stats(j_ - 1);                                            \
    WS_BARRIER();                                                                                      \
  }
    for (int j = 0; j < J; j += 2) {
      WS_STEP(j, pfA, phA, pwA, pfB, phB, pwB)
      if (j + 1 < J) WS_STEP(j + 1, pfB, phB, pwB, pfA, phA, pwA)
    }
    flush_stats(J - 1);
#undef WS_STEP
#undef WS_ISSUE
#undef WS_COMMIT
  } else {
    // ================================================= consumers ==================================================
    const int half = lane >> 5, l31 = lane & 31;
    f32x16 acc[1][4];
    __syncthreads();                                   // (P1)
    __syncthreads();                                   // (P2)
    for (int j = 0; j < J; ++j) {
      const int tidx = j / nchunk;
      const int kc = j - tidx * nchunk;
      const TileInfo ti = decode_tile<TR2>((int)blockIdx.x + tidx * (int)gridDim.x, ntt, ntf, a.ncg, a.padf);
      const int f = ti.f0 + wave;
      const bool row_ok = f < a.Fout;
      if (kc == 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[0][s][r] = 0.f;
      }
      const bf16x8* xhi = s_buf + (j & 1) * BUFN;
      const bf16x8* xlo = xhi + XN;
      const bf16x8* whi = xlo + XN;
      const bf16x8* wlo = whi + WN;
      if (row_ok && !(a.dbg & 1)) {
        if (TR2) {
          if (wave & 1) chunk_mfma_bf16<1, NR, 1, TR2, 2>(acc, xhi, xlo, whi, wlo, wave, half, l31);
          else chunk_mfma_bf16<1, NR, 1, TR2, 5>(acc, xhi, xlo, whi, wlo, wave, half, l31);
        } else {
          chunk_mfma_bf16<1, NR, 1, TR2, 7>(acc, xhi, xlo, whi, wlo, wave, half, l31);
        }
      }
      if (kc == nchunk - 1 && !(a.dbg & 4)) {
        // ---- tile epilogue: bias, ELU, raw store, per-row statistics partials (conv_epilogue.hpp) ----
        float* red = s_red + (tidx & 1) * (FT * COP * 2) + wave * (COP * 2);
        conv_epilogue<1>(a, acc, ti.n, ti.cg, f, ti.t0, row_ok, lane, red);
      }
      WS_BARRIER();
    }
  }
}
#undef WS_BARRIER

template <int MODE>
static hipError_t ws_set_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16x3_ws<MODE>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

static size_t ws_lds_bytes(int NR) {
  const size_t bufn = 2 * (size_t)NR * 2 * TW + 2 * 9 * 2 * 32;
  return 2 * bufn * 16 + 2 * 256 * sizeof(float2) + 2 * FT * 32 * 2 * sizeof(float);
}

static int g_ws_cus = 0;
static int g_ws_enabled = 0;   // measured slower than the 2-blocks-per-CU kernel (profiles/): opt-in via MISONET_WS=1
void conv_bf16_ws_enable(int on) { g_ws_enabled = on; }

static size_t bf_lds_bytes(int NR, int cop, int Cin) {
  const int nchunk = (Cin + CKB - 1) / CKB;
  return (size_t)(2 * NR * 2 * TW + 2 * 9 * 2 * cop) * 16 + (size_t)nchunk * CKB * sizeof(float2) + (size_t)cop * sizeof(float);
}

template <int NCO, int MODE>
static hipError_t bf_set_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16x3<NCO, MODE>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
}

hipError_t conv_bf16_init() {
  hipError_t e;
  if ((e = bf_set_attr<1, 0>()) != hipSuccess) return e;
  if ((e = bf_set_attr<1, 1>()) != hipSuccess) return e;
  if ((e = bf_set_attr<1, 2>()) != hipSuccess) return e;
  if ((e = bf_set_attr<2, 0>()) != hipSuccess) return e;
  if ((e = bf_set_attr<2, 1>()) != hipSuccess) return e;
  if ((e = bf_set_attr<2, 2>()) != hipSuccess) return e;
  if ((e = ws_set_attr<0>()) != hipSuccess) return e;
  if ((e = ws_set_attr<2>()) != hipSuccess) return e;
  if ((e = conv_bf16_r8_init()) != hipSuccess) return e;
  int dev = 0;
  if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
  g_ws_cus = prop.multiProcessorCount;
  return hipSuccess;
}

hipError_t launch_conv_bf16(const ConvArgs& a_in, int n_samples, hipStream_t s) {
  ConvArgs a = a_in;
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("MISONET_WS_DEBUG"); dbg = e ? atoi(e) : 0; }
    a.dbg = dbg;
    static int ws_env = -1;
    if (ws_env < 0) { const char* e = getenv("MISONET_WS"); ws_env = e ? atoi(e) : 0; g_ws_enabled = ws_env; }
  }
  const dim3 grid = conv_grid(a, n_samples, TT, FT, conv_xcd_env());
  const size_t lds = bf_lds_bytes(a.NR, a.cop, a.Cin);
  const int mode = a.tr2 ? 2 : (a.sf == 2 ? 1 : 0);
  if (a.NR != conv_rows(a.sf, a.tr2) || !a.w16) return hipErrorInvalidValue;
  {
    static int r8_env = -1;
    if (r8_env < 0) { const char* e = getenv("MISONET_R8"); r8_env = e ? atoi(e) : 0; }
    if (r8_env && mode == 0 && a.cop == 32 && !g_ws_enabled && !a.out_oct) return launch_conv_bf16_r8(a, n_samples, s);
  }
  static int tl_env = -1;
  static int tl_done = 0;
  static unsigned long long* tl_buf = nullptr;
  if (tl_env < 0) { const char* e = getenv("MISONET_TIMELINE"); tl_env = e ? atoi(e) : 0; }
  const bool do_tl = tl_env && tl_done < 3 && mode == 0 && a.Cin == 96 && a.Fout == 63 && n_samples >= 8;
  if (do_tl) {
    if (!tl_buf && hipMalloc(reinterpret_cast<void**>(&tl_buf), 64 * 8) != hipSuccess) tl_buf = nullptr;
    if (tl_buf) { (void)hipMemsetAsync(tl_buf, 0, 64 * 8, s); a.dbg_buf = tl_buf; }
  }
  if (g_ws_enabled && mode != 1 && a.cop == 32 && a.Cin <= 256 && g_ws_cus > 0 && !a.out_oct) {
    const int ntt = (a.T + TT - 1) / TT, ntf = (a.Fout + FT - 1) / FT;
    const int ntiles = ntt * ntf * n_samples * a.ncg;
    const int nblk = ntiles < g_ws_cus ? ntiles : g_ws_cus;
    const size_t l2 = ws_lds_bytes(a.NR);
    if (mode == 0) hipLaunchKernelGGL((conv3x3_bf16x3_ws<0>), dim3(nblk), dim3(512), l2, s, a, ntt, ntf, ntiles);
    else hipLaunchKernelGGL((conv3x3_bf16x3_ws<2>), dim3(nblk), dim3(512), l2, s, a, ntt, ntf, ntiles);
    return hipGetLastError();
  }
#define MN_LAUNCH(NCO, MODE) hipLaunchKernelGGL((conv3x3_bf16x3<NCO, MODE>), grid, dim3(256), lds, s, a)
  if (a.cop == 32) {
    if (mode == 0) MN_LAUNCH(1, 0); else if (mode == 1) MN_LAUNCH(1, 1); else MN_LAUNCH(1, 2);
  } else {
    if (mode == 0) MN_LAUNCH(2, 0); else if (mode == 1) MN_LAUNCH(2, 1); else MN_LAUNCH(2, 2);
  }
#undef MN_LAUNCH
  if (do_tl && tl_buf) {
    unsigned long long h[64];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, tl_buf, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[timeline] Cin=%d Fout=%d n=%d stamps=%llu:", a.Cin, a.Fout, n_samples, h[63]);
    for (unsigned long long i = 0; i < h[63] && i < 40; ++i) fprintf(stderr, " %llu", h[i]);
    fprintf(stderr, " | epi: %llu %llu %llu", h[41] - h[40], h[42] - h[41], h[43] - h[42]);
    fprintf(stderr, "\n");
    ++tl_done;
  }
  return hipGetLastError();
}

}  // namespace mn
