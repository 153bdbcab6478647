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
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace mn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int CKB = 16;

template <int NCO, int NR, int SF, bool TR2, int KFMASK>
__device__ __forceinline__ void chunk_mfma_bf16(f32x16 (&acc)[NCO][4], const bf16x8* s_xhi, const bf16x8* s_xlo,
                                                const bf16x8* s_whi, const bf16x8* s_wlo, int frel, int half,
                                                int l31) {
  constexpr int COP = NCO * 32;
  constexpr int NKF = ((KFMASK >> 0) & 1) + ((KFMASK >> 1) & 1) + ((KFMASK >> 2) & 1);
  constexpr int NTAP = 3 * NKF;
  // per-lane bases (units of bf16x8 = 16 bytes)
  const int wb = half * COP + l31;                       // + tap*2*COP + j*32
  int ib[3];
#pragma unroll
  for (int kf = 0; kf < 3; ++kf) {
    const int rl = TR2 ? ((frel + kf) >> 1) : (SF * frel + kf);
    ib[kf] = (rl * 2 + half) * TW + l31 + 3;             // + seg*32 + kt
  }
  // explicit two-deep pipeline over steps = (tap, frame-tile pair)
  bf16x8 ah[2][NCO], al[2][NCO], bh[2][2], bl[2][2];
  constexpr int NSTEP = NTAP * 2;
#pragma unroll
  for (int st = -1; st < NSTEP; ++st) {
    const int cur = st & 1;
    if (st + 1 < NSTEP) {
      const int nx = st + 1;
      const int tap_ = nx >> 1, sp_ = nx & 1;
      const int kt_ = tap_ / NKF, ks_ = tap_ % NKF;
      const int kf_ = (NKF == 3) ? ks_ : ((KFMASK == 2) ? 1 : (ks_ == 0 ? 0 : 2));
      const int nb = (st + 1) & 1;
      if (sp_ == 0) {
#pragma unroll
        for (int j = 0; j < NCO; ++j) {
          ah[(tap_ & 1)][j] = s_whi[wb + ((kt_ * 3 + kf_) * 2) * COP + j * 32];
          al[(tap_ & 1)][j] = s_wlo[wb + ((kt_ * 3 + kf_) * 2) * COP + j * 32];
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bh[nb][q] = s_xhi[ib[kf_] + (sp_ * 2 + q) * 32 + kt_];
        bl[nb][q] = s_xlo[ib[kf_] + (sp_ * 2 + q) * 32 + kt_];
      }
    }
    if (st >= 0) {
      const int tap_ = st >> 1, sp_ = st & 1;
      const int ta = tap_ & 1;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < NCO; ++j) {
          f32x16 c = acc[j][sp_ * 2 + q];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ta][j], bh[cur][q], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ta][j], bl[cur][q], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ta][j], bh[cur][q], c, 0, 0, 0);
          acc[j][sp_ * 2 + q] = c;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void split2(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

template <int NCO, int MODE>
__global__ __launch_bounds__(256, (NCO == 1 ? 2 : 1)) void conv3x3_bf16x3(const ConvArgs a) {
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

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t0 = blockIdx.x * TT;
  const int f0 = blockIdx.y * FT;
  const int n = blockIdx.z / a.ncg;
  const int cg = blockIdx.z - n * a.ncg;
  const int T = a.T, Tp = a.Tp, Fin = a.Fin, Cin = a.Cin;
  const int nchunk = (Cin + CKB - 1) / CKB;
  const int fin0 = TR2 ? (f0 >> 1) - 1 : SF * f0 - a.padf;

  for (int c = tid; c < nchunk * CKB; c += 256) {
    float mean = 0.f, rstd = 1.f;
    if (c >= a.ident_c && c < Cin) {
      const double* st = a.in_stats + ((long long)n * a.in_sstride + a.in_c0 + c) * 2;
      const double cnt = (double)Fin * (double)T;
      const double m = st[0] / cnt;
      double var = st[1] / cnt - m * m;
      var = var > 0.0 ? var : 0.0;
      mean = (float)m;
      rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));
    }
    s_nrm[c] = make_float2(mean, rstd);
  }

  const float* in_n = a.in + (long long)n * a.in_bstride + (long long)a.in_c0 * Fin * Tp;
  const u32x4* w_g = reinterpret_cast<const u32x4*>(a.w16) + (long long)cg * nchunk * (2 * WN);

  // ---- staging roles ----
  const unsigned row_e = (unsigned)Tp;
  const unsigned plane_e = (unsigned)Fin * row_e;
  const int tl = t0 + 2 * lane;                                  // this lane's first frame
  const unsigned tl_e = (unsigned)(tl < Tp ? tl : Tp - 2);
  unsigned poff_e[NPW];                                          // row offset of each owned pair (clamped)
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    int p = wave + 4 * i;
    p = p < NPAIR ? p : NPAIR - 1;
    int fin = fin0 + (p >> 1);
    fin = fin < 0 ? 0 : (fin >= Fin ? Fin - 1 : fin);
    poff_e[i] = (unsigned)fin * row_e + tl_e;
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
        pf[i][e] = *reinterpret_cast<const f32x2*>(in_n + ((unsigned)c_ * plane_e + poff_e[i]));       \
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
      ph[i] = in_n[(unsigned)c_ * plane_e + (unsigned)fin_ * row_e + (unsigned)th_];                   \
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
        bf16x8 h0_, l0_, h1_, l1_;                                                                     \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                \
          const int c_ = (KC) * CKB + 8 * h_ + e;                                                      \
          const float2 m_ = s_nrm[c_];                                                                 \
          const bool ok_ = rok_ && c_ < Cin;                                                           \
          const float x0_ = (ok_ && tl + 0 < T) ? (pf[i][e].x - m_.x) * m_.y : 0.f;                    \
          const float x1_ = (ok_ && tl + 1 < T) ? (pf[i][e].y - m_.x) * m_.y : 0.f;                    \
          __bf16 a_, b_;                                                                               \
          split2(x0_, a_, b_); h0_[e] = a_; l0_[e] = b_;                                               \
          split2(x1_, a_, b_); h1_[e] = a_; l1_[e] = b_;                                               \
        }                                                                                              \
        const int o_ = (r_ * 2 + h_) * TW + 4 + 2 * lane;                                              \
        s_xhi[o_] = h0_; s_xhi[o_ + 1] = h1_;                                                          \
        s_xlo[o_] = l0_; s_xlo[o_ + 1] = l1_;                                                          \
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
        const bool ok_ = fin_ >= 0 && fin_ < Fin && c_ < Cin && th_ >= 0 && th_ < T;                   \
        const float2 m_ = s_nrm[c_];                                                                   \
        const float x_ = ok_ ? (ph[i] - m_.x) * m_.y : 0.f;                                            \
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

  BF_ISSUE(0)
  __syncthreads();          // s_nrm visible
  BF_COMMIT(0)
  __syncthreads();

  for (int kc = 0; kc < nchunk; ++kc) {
    const bool more = (kc + 1 < nchunk);
    if (more) BF_ISSUE(kc + 1)
    if (row_ok) {
      if (TR2) {
        if ((f - f0) & 1) chunk_mfma_bf16<NCO, NR, SF, TR2, 2>(acc, s_xhi, s_xlo, s_whi, s_wlo, f - f0, half, l31);
        else chunk_mfma_bf16<NCO, NR, SF, TR2, 5>(acc, s_xhi, s_xlo, s_whi, s_wlo, f - f0, half, l31);
      } else {
        chunk_mfma_bf16<NCO, NR, SF, TR2, 7>(acc, s_xhi, s_xlo, s_whi, s_wlo, f - f0, half, l31);
      }
    }
    __syncthreads();
    if (more) {
      BF_COMMIT(kc + 1)
      __syncthreads();
    }
  }
#undef BF_ISSUE
#undef BF_COMMIT

  // ---- epilogue: identical to conv3x3_mfma ----
  float* s_red = reinterpret_cast<float*>(smem_b);   // [FT][COP][2]
  const int nseg = (T - t0 + 31) >> 5;
  float* out_n = a.out + (long long)n * a.out_bstride + (long long)a.out_c0 * a.Fout * Tp;
#pragma unroll
  for (int j = 0; j < NCO; ++j) {
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
    if (row_ok) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < nseg) {
          const int t = t0 + s * 32 + l31;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co_l = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int co = cg * COP + co_l;
            float v = acc[j][s][r] + a.bias[co];
            if (a.act) v = v > 0.f ? v : expm1f(v);
            const bool ok = (co < a.Cout) && (t < T);
            if (ok) {
              out_n[((long long)co * a.Fout + f) * Tp + t] = v;
              s1[r] += v;
              s2[r] += v * v;
            }
          }
        }
      }
    }
    if (a.act) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float x1 = s1[r], x2 = s2[r];
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) {
          x1 += __shfl_xor(x1, m, 64);
          x2 += __shfl_xor(x2, m, 64);
        }
        if (l31 == 0) {
          const int co_l = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          s_red[(wave * COP + co_l) * 2 + 0] = x1;
          s_red[(wave * COP + co_l) * 2 + 1] = x2;
        }
      }
    }
  }
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
}

static size_t bf_lds_bytes(int NR, int cop, int Cin) {
  const int nchunk = (Cin + CKB - 1) / CKB;
  return (size_t)(2 * NR * 2 * TW + 2 * 9 * 2 * cop) * 16 + (size_t)nchunk * CKB * sizeof(float2);
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
  return bf_set_attr<2, 2>();
}

hipError_t launch_conv_bf16(const ConvArgs& a, int n_samples, hipStream_t s) {
  dim3 grid((a.T + TT - 1) / TT, (a.Fout + FT - 1) / FT, n_samples * a.ncg);
  const size_t lds = bf_lds_bytes(a.NR, a.cop, a.Cin);
  const int mode = a.tr2 ? 2 : (a.sf == 2 ? 1 : 0);
  if (a.NR != conv_rows(a.sf, a.tr2) || !a.w16) return hipErrorInvalidValue;
#define MN_LAUNCH(NCO, MODE) hipLaunchKernelGGL((conv3x3_bf16x3<NCO, MODE>), grid, dim3(256), lds, s, a)
  if (a.cop == 32) {
    if (mode == 0) MN_LAUNCH(1, 0); else if (mode == 1) MN_LAUNCH(1, 1); else MN_LAUNCH(1, 2);
  } else {
    if (mode == 0) MN_LAUNCH(2, 0); else if (mode == 1) MN_LAUNCH(2, 1); else MN_LAUNCH(2, 2);
  }
#undef MN_LAUNCH
  return hipGetLastError();
}

}  // namespace mn
