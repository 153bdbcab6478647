// conv3x3_bf16x3_r8: the bf16x3 (three-term bf16 split, see conv_bf16.hip) kernel for the stride-1 layers with an
// 8-row x 64-frame workgroup tile.  The bf16x3 convs are HBM-bound (ablation in DESIGN.md 3.1: loads + barriers alone
// take 36 of 76 ms), so the tile is shaped to move fewer bytes per output:
//   * 8 output rows per workgroup need 10 staged input rows (1.25x halo instead of 1.5x for 4 rows);
//   * a wave owns TWO adjacent output rows x two 32-frame tiles, so every B fragment (input row, frame shift) feeds up to
//     two (output row, frequency tap) pairs: 66 ds_read_b128 per 108 MFMAs instead of 90;
//   * LDS: input hi+lo [10][2][72] x 16 B = 45 KB + weights hi+lo 18 KB -> 2 workgroups per CU as before.
// Staging: wave-round q <-> staged row q (waves 0,1: rows w, w+4, w+8; waves 2,3: rows w, w+4); lanes 0-31 <-> channel
// octet 0, lanes 32-63 <-> octet 1; lane & 31 <-> frames t0+2l, t0+2l+1.  Everything else (normalise-on-load, FMA +
// v_cvt_pk split, buffer loads, epilogue) as in conv_bf16.hip.
#include "kernels.hpp"
#include "conv_epilogue.hpp"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace mn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int R8_FT = 8;          // output rows per workgroup
constexpr int R8_TT = 64;         // output frames per workgroup
constexpr int R8_TW = 72;         // staged frame slots: col 3 = t0-1, cols 4..67 = t0..t0+63, col 68 = t0+64
constexpr int R8_NR = 10;         // staged rows
constexpr int R8_CK = 16;

__device__ __forceinline__ void r8_split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  f32x2 x = {x0, x1};
  const bf16x2 h = __builtin_convertvector(x, bf16x2);
  const unsigned hu = __builtin_bit_cast(unsigned, h);
  f32x2 r;
  r.x = x0 - __builtin_bit_cast(float, hu << 16);
  r.y = x1 - __builtin_bit_cast(float, hu & 0xffff0000u);
  const bf16x2 l = __builtin_convertvector(r, bf16x2);
  hi = hu;
  lo = __builtin_bit_cast(unsigned, l);
}

// MFMA work of one K-chunk: wave rows (2w, 2w+1) x 2 frame tiles.  Steps = (kt, local input row ri in 0..3); the B
// fragments of a step feed (out row 0, kf = ri) when ri <= 2 and (out row 1, kf = ri - 1) when ri >= 1.
__device__ __forceinline__ void r8_chunk_mfma(f32x16 (&acc)[2][2], const bf16x8* s_xhi, const bf16x8* s_xlo,
                                              const bf16x8* s_whi, const bf16x8* s_wlo, int wave, int half, int l31) {
  constexpr int COP = 32;
  const int wb = half * COP + l31;                                   // + (tap*2)*COP
  const int ib = ((2 * wave) * 2 + half) * R8_TW + l31 + 3;           // + ri*2*TW + seg*32 + kt
  bf16x8 ah[2][3], al[2][3];                                         // weights of kt (double-buffered), kf = 0..2
  bf16x8 bh[2][2], bl[2][2];
  constexpr int NSTEP = 12;
#pragma unroll
  for (int st = -1; st < NSTEP; ++st) {
    const int cur = st & 1;
    if (st + 1 < NSTEP) {
      const int nx = st + 1;
      const int kt_ = nx >> 2, ri_ = nx & 3;
      const int nb = nx & 1;
      if (ri_ == 0) {
#pragma unroll
        for (int kf = 0; kf < 3; ++kf) {
          ah[kt_ & 1][kf] = s_whi[wb + ((kt_ * 3 + kf) * 2) * COP];
          al[kt_ & 1][kf] = s_wlo[wb + ((kt_ * 3 + kf) * 2) * COP];
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bh[nb][q] = s_xhi[ib + ri_ * 2 * R8_TW + q * 32 + kt_];
        bl[nb][q] = s_xlo[ib + ri_ * 2 * R8_TW + q * 32 + kt_];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (st >= 0) {
      const int kt_ = st >> 2, ri_ = st & 3;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int kf = ri_ - r;
        if (kf >= 0 && kf <= 2) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            f32x16 c = acc[r][q];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kt_ & 1][kf], bh[cur][q], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kt_ & 1][kf], bl[cur][q], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kt_ & 1][kf], bh[cur][q], c, 0, 0, 0);
            acc[r][q] = c;
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(256, 2) void conv3x3_bf16x3_r8(const ConvArgs a) {
  constexpr int COP = 32;
  constexpr int NPW = 3;                                // staged rows per wave (waves 2,3 use 2)
  constexpr int NHT = 2;                                // halo scalars per thread: 10 rows x 2 octets x 2 sides x 8 ch = 320
  constexpr int XN = R8_NR * 2 * R8_TW;                 // bf16x8 units per input image
  constexpr int WN = 9 * 2 * COP;
  constexpr int NWI = (2 * WN + 255) / 256;
  extern __shared__ __align__(16) unsigned char smem_b[];
  bf16x8* s_xhi = reinterpret_cast<bf16x8*>(smem_b);
  bf16x8* s_xlo = s_xhi + XN;
  bf16x8* s_whi = s_xlo + XN;
  bf16x8* s_wlo = s_whi + WN;
  float2* s_nrm = reinterpret_cast<float2*>(s_wlo + WN);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int t0 = blockIdx.x * R8_TT;
  const int f0 = blockIdx.y * R8_FT;
  const int n = blockIdx.z / a.ncg;
  const int cg = blockIdx.z - n * a.ncg;
  const int T = a.T, Tp = a.Tp, Fin = a.Fin, Cin = a.Cin;
  const int nchunk = (Cin + R8_CK - 1) / R8_CK;
  const int fin0 = f0 - a.padf;

  for (int c = tid; c < nchunk * R8_CK; c += 256) {
    float mean = 0.f, rstd = (c < Cin) ? 1.f : 0.f;
    if (c >= a.ident_c && c < Cin) {
      const double* st = a.in_stats + ((long long)n * a.in_sstride + a.in_c0 + c) * 2;
      const double cnt = (double)Fin * (double)T;
      const double m = st[0] / cnt;
      double var = st[1] / cnt - m * m;
      var = var > 0.0 ? var : 0.0;
      mean = (float)m;
      rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));
    }
    s_nrm[c] = make_float2(rstd, -mean * rstd);
  }

  const float* in_n = a.in + (long long)n * a.in_bstride + (long long)a.in_c0 * Fin * Tp;
  const u32x4* w_g = reinterpret_cast<const u32x4*>(a.w16) + (long long)cg * nchunk * (2 * WN);

  const unsigned row_e = (unsigned)Tp;
  const unsigned plane_e = (unsigned)Fin * row_e;
  const unsigned plane_b = plane_e * 4u;
  const unsigned long long pa = reinterpret_cast<unsigned long long>(in_n);
  const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa);
  const unsigned phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)phi << 32) | plo), 0,
      __builtin_amdgcn_readfirstlane((int)((unsigned)Cin * plane_b)), 0x00020000);
  const int tl = t0 + 2 * l31;                                   // this lane's first frame
  const unsigned tl_e = (unsigned)(tl < Tp ? tl : Tp - 2);
  const bool full_t = (t0 + R8_TT <= T);
  const int nrow = (wave < 2) ? 3 : 2;                           // staged rows owned by this wave: wave, wave+4, (wave+8)
  unsigned roff_b[NPW];                                          // row/frame byte offset (+ this lane's octet plane offset)
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    int r = wave + 4 * i;
    r = r < R8_NR ? r : R8_NR - 1;
    int fin = fin0 + r;
    fin = fin < 0 ? 0 : (fin >= Fin ? Fin - 1 : fin);
    roff_b[i] = ((unsigned)fin * row_e + tl_e) * 4u;
  }
  f32x2 pf[NPW][8];
  float ph[NHT];
  u32x4 pw[NWI];

  // channel of (chunk kc, octet `half`, element e) = kc*16 + 8*half + e, clamped to Cin-1 (channels >= Cin have
  // (scale, shift) = (0, 0) in s_nrm, so whatever finite value is read stages as zero).  The whole offset is per-lane.
#define R8_ISSUE(KC)                                                                                   \
  {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                  \
      if (i < nrow) {                                                                                  \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                \
          int c_ = (KC) * R8_CK + 8 * half + e;                                                        \
          c_ = c_ < Cin ? c_ : Cin - 1;                                                                \
          const u32x2 v_ = __builtin_amdgcn_raw_buffer_load_b64(                                       \
              rs_in, roff_b[i] + (unsigned)c_ * plane_b, 0, 0);                                        \
          pf[i][e] = __builtin_bit_cast(f32x2, v_);                                                    \
        }                                                                                              \
      }                                                                                                \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NHT; ++i) {                                                  \
      int k_ = tid + 256 * i;                          /* (row, octet, side, e): 320 scalars */       \
      k_ = k_ < R8_NR * 32 ? k_ : R8_NR * 32 - 1;                                                      \
      const int e_ = k_ & 7, side_ = (k_ >> 3) & 1, h_ = (k_ >> 4) & 1, r_ = k_ >> 5;                  \
      int c_ = (KC) * R8_CK + 8 * h_ + e_;                                                             \
      c_ = c_ < Cin ? c_ : Cin - 1;                                                                    \
      int fin_ = fin0 + r_;                                                                            \
      fin_ = fin_ < 0 ? 0 : (fin_ >= Fin ? Fin - 1 : fin_);                                            \
      int th_ = side_ ? t0 + R8_TT : t0 - 1;                                                           \
      th_ = th_ < 0 ? 0 : (th_ >= Tp ? Tp - 1 : th_);                                                  \
      const unsigned ho_ = ((unsigned)c_ * plane_e + (unsigned)fin_ * row_e + (unsigned)th_) * 4u;     \
      ph[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, ho_, 0, 0));       \
    }                                                                                                  \
    const u32x4* wsrc_ = w_g + (unsigned)(KC) * (unsigned)(2 * WN);                                    \
    _Pragma("unroll") for (int i = 0; i < NWI; ++i) {                                                  \
      unsigned idx_ = tid + 256 * i;                                                                   \
      idx_ = idx_ < (unsigned)(2 * WN) ? idx_ : (unsigned)(2 * WN - 1);                                \
      pw[i] = wsrc_[idx_];                                                                             \
    }                                                                                                  \
  }

#define R8_COMMIT(KC)                                                                                  \
  {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                  \
      if (i < nrow) {                                                                                  \
        const int r_ = wave + 4 * i;                                                                   \
        const int fin_ = fin0 + r_;                                                                    \
        const bool rok_ = fin_ >= 0 && fin_ < Fin;                                                     \
        const int o_ = (r_ * 2 + half) * R8_TW + 4 + 2 * l31;                                          \
        u32x4 h0_ = {0u, 0u, 0u, 0u}, l0_ = h0_, h1_ = h0_, l1_ = h0_;                                 \
        if (rok_) {                                                                                    \
          float y0_[8], y1_[8];                                                                        \
          _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                              \
            const float2 m_ = s_nrm[(KC) * R8_CK + 8 * half + e];                                      \
            y0_[e] = fmaf(pf[i][e].x, m_.x, m_.y);                                                     \
            y1_[e] = fmaf(pf[i][e].y, m_.x, m_.y);                                                     \
          }                                                                                            \
          if (!full_t) {                                                                               \
            const bool k0_ = (tl + 0 < T), k1_ = (tl + 1 < T);                                         \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                            \
              y0_[e] = k0_ ? y0_[e] : 0.f;                                                             \
              y1_[e] = k1_ ? y1_[e] : 0.f;                                                             \
            }                                                                                          \
          }                                                                                            \
          _Pragma("unroll") for (int e2 = 0; e2 < 4; ++e2) {                                           \
            unsigned a_, b_;                                                                           \
            r8_split_pair(y0_[2 * e2], y0_[2 * e2 + 1], a_, b_); h0_[e2] = a_; l0_[e2] = b_;           \
            r8_split_pair(y1_[2 * e2], y1_[2 * e2 + 1], a_, b_); h1_[e2] = a_; l1_[e2] = b_;           \
          }                                                                                            \
        }                                                                                              \
        reinterpret_cast<u32x4*>(s_xhi)[o_] = h0_; reinterpret_cast<u32x4*>(s_xhi)[o_ + 1] = h1_;      \
        reinterpret_cast<u32x4*>(s_xlo)[o_] = l0_; reinterpret_cast<u32x4*>(s_xlo)[o_ + 1] = l1_;      \
      }                                                                                                \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NHT; ++i) {                                                  \
      const int k_ = tid + 256 * i;                                                                    \
      if (k_ < R8_NR * 32) {                                                                           \
        const int e_ = k_ & 7, side_ = (k_ >> 3) & 1, h_ = (k_ >> 4) & 1, r_ = k_ >> 5;                \
        const int c_ = (KC) * R8_CK + 8 * h_ + e_;                                                     \
        const int fin_ = fin0 + r_;                                                                    \
        const int th_ = side_ ? t0 + R8_TT : t0 - 1;                                                   \
        const bool ok_ = fin_ >= 0 && fin_ < Fin && th_ >= 0 && th_ < T && c_ < Cin;                   \
        const float2 m_ = s_nrm[c_];                                                                   \
        const float x_ = ok_ ? fmaf(ph[i], m_.x, m_.y) : 0.f;                                          \
        const __bf16 hi_ = (__bf16)x_;                                                                 \
        const __bf16 lo_ = (__bf16)(x_ - (float)hi_);                                                  \
        const int o_ = ((r_ * 2 + h_) * R8_TW + (side_ ? R8_TT + 4 : 3)) * 8 + e_;                     \
        reinterpret_cast<__bf16*>(s_xhi)[o_] = hi_;                                                    \
        reinterpret_cast<__bf16*>(s_xlo)[o_] = lo_;                                                    \
      }                                                                                                \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NWI; ++i) {                                                  \
      const int idx_ = tid + 256 * i;                                                                  \
      if (idx_ < 2 * WN) reinterpret_cast<u32x4*>(s_whi)[idx_] = pw[i];                                \
    }                                                                                                  \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[r][s][k] = 0.f;

  const bool wave_ok = (f0 + 2 * wave) < a.Fout;          // at least the first of the wave's two rows exists

  R8_ISSUE(0)
  __syncthreads();
  R8_COMMIT(0)
  __syncthreads();
  for (int kc = 0; kc < nchunk; ++kc) {
    const bool more = (kc + 1 < nchunk);
    if (more) R8_ISSUE(kc + 1)
    if (wave_ok) {
      __builtin_amdgcn_s_setprio(1);
      r8_chunk_mfma(acc, s_xhi, s_xlo, s_whi, s_wlo, wave, half, l31);
      __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    if (more) {
      R8_COMMIT(kc + 1)
      __syncthreads();
    }
  }
#undef R8_ISSUE
#undef R8_COMMIT

  // ---- epilogue: per output row, two frame tiles (conv_epilogue.hpp) ----
  float* s_red = reinterpret_cast<float*>(smem_b);          // [8 rows][COP][2]
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int f = f0 + 2 * wave + r;
    f32x16 (&arow)[1][2] = *reinterpret_cast<f32x16 (*)[1][2]>(&acc[r]);
    conv_epilogue<1, 2>(a, arow, n, cg, f, t0, f < a.Fout, lane, s_red + (2 * wave + r) * (COP * 2));
  }
  if (a.act) {
    __syncthreads();
    if (tid < COP * 2) {
      const int co_l = tid >> 1, which = tid & 1;
      const int co = cg * COP + co_l;
      if (co < a.Cout) {
        float tot = 0.f;
        for (int w = 0; w < R8_FT; ++w)
          if (f0 + w < a.Fout) tot += s_red[(w * COP + co_l) * 2 + which];
        unsafeAtomicAdd(a.out_stats + ((long long)n * a.out_sstride + a.out_c0 + co) * 2 + which, (double)tot);
      }
    }
  }
}

static size_t r8_lds_bytes(int Cin) {
  const int nchunk = (Cin + R8_CK - 1) / R8_CK;
  return (size_t)(2 * R8_NR * 2 * R8_TW + 2 * 9 * 2 * 32) * 16 + (size_t)nchunk * R8_CK * sizeof(float2);
}

hipError_t conv_bf16_r8_init() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16x3_r8),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}

// stride-1 layers only (a.sf == 1, !a.tr2), 32-channel output groups (a.cop == 32)
hipError_t launch_conv_bf16_r8(const ConvArgs& a, int n_samples, hipStream_t s) {
  if (a.sf != 1 || a.tr2 || a.cop != 32 || !a.w16) return hipErrorInvalidValue;
  dim3 grid((a.T + R8_TT - 1) / R8_TT, (a.Fout + R8_FT - 1) / R8_FT, n_samples * a.ncg);
  hipLaunchKernelGGL(conv3x3_bf16x3_r8, grid, dim3(256), r8_lds_bytes(a.Cin), s, a);
  return hipGetLastError();
}

}  // namespace mn
