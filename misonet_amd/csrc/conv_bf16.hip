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
  conv_epilogue<NCO, 4, 2>(a, acc, n, cg, f, t0, row_ok, lane, s_red + wave * (COP * 2), s_bias);
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
        dstat_add(a.out_stats + (((long long)n * a.out_sstride + a.out_c0 + co) * 2 + which) * DS_NL, (double)tot);
      }
    }
  }
  STAMP();
  if (stamp) a.dbg_buf[63] = si;
#undef STAMP
}

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
  return hipSuccess;
}

hipError_t launch_conv_bf16(const ConvArgs& a_in, int n_samples, hipStream_t s) {
  ConvArgs a = a_in;
  {
    static int dbg = -1;
    if (dbg < 0) dbg = exp_env("MISONET_WS_DEBUG", 0);
    a.dbg = dbg;
  }
  const dim3 grid = conv_grid(a, n_samples, TT, FT, conv_xcd_env());
  const size_t lds = bf_lds_bytes(a.NR, a.cop, a.Cin);
  const int mode = a.tr2 ? 2 : (a.sf == 2 ? 1 : 0);
  if (a.NR != conv_rows(a.sf, a.tr2) || !a.w16) return hipErrorInvalidValue;
  static int tl_env = -1;
  static int tl_done = 0;
  static unsigned long long* tl_buf = nullptr;
  if (tl_env < 0) tl_env = exp_env("MISONET_TIMELINE", 0);
  const bool do_tl = tl_env && tl_done < 3 && mode == 0 && a.Cin == 96 && a.Fout == 63 && n_samples >= 8;
  if (do_tl) {
    if (!tl_buf && hipMalloc(reinterpret_cast<void**>(&tl_buf), 64 * 8) != hipSuccess) tl_buf = nullptr;
    if (tl_buf) { (void)hipMemsetAsync(tl_buf, 0, 64 * 8, s); a.dbg_buf = tl_buf; }
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
