// 3x3 stride-1 convolution with a FEW output channels (<= 4) on the vector ALU: the network's last layer
// (last_Deconv2d, reference model.py:73,419-435 -- ConvTranspose2d 2*de_ch[6] -> out_ch = 2 * num_spks, no activation, no norm
// behind it; MISO1: 48 -> 4, MISO3: 48 -> 2, F = 127 -> 129).
//
//   out[co][f][t] = bias[co] + sum_{ci,kt,kf} W[co][ci][kt][kf] * xhat[ci][f - padf + kf][t - 1 + kt],   xhat = instance norm
//
// Why not the matrix cores: with 4 output channels an MFMA tile is 4 of 32 rows useful.  conv3x3_mfma ran this layer at
// 16 TF/s -- 3.5 ms per step, the most expensive layer outside the dense blocks -- against an HBM floor of 0.55 ms (48
// channels x 127 rows x 4 KB per sample read once).  The gfx950 vector ALU has the SAME fp32 rate as its fp32 matrix
// pipe (v_pk_fma_f32: 2 FMAs per lane and issue), and on it no multiplier is wasted: a lane owns two consecutive frames of
// four output rows and all output channels, the weights are wave-uniform scalars (s_load), a staged input value is used
// for 3 x 3 x COUT FMAs.  Exact float32 FMA chains like the MFMA form (the summation order differs, as between any two
// tilings).
//
// Workgroup = 4 waves = 16 output rows x 128 frames; wave w owns rows 4 w .. 4 w + 3, lane l frames 2 l, 2 l + 1.  K-chunks
// of 4 input channels are staged [4][18][132] (instance norm applied on the way, zero padding after it -- the reference's
// order) into a double-buffered LDS tile; the global loads of chunk k + 1 are in flight while chunk k is computed.
#include "kernels.hpp"
#include "conv_epilogue.hpp"

namespace mn {

typedef float ff2 __attribute__((ext_vector_type(2)));
typedef float ff4 __attribute__((ext_vector_type(4)));

constexpr int FW_CK = 4;                 // input channels per chunk
constexpr int FW_RT = 16;                // output rows per workgroup
constexpr int FW_NR = FW_RT + 2;         // staged input rows
constexpr int FW_LW = 132;               // floats per staged row: column c = frame t0 - 1 + c (130 used)
constexpr int FW_ITEMS = FW_CK * FW_NR;  // (channel, row) items per chunk: 72 = 9 per thread group
constexpr int FW_NI = FW_ITEMS / 8;
constexpr int FW_STAGE = FW_CK * FW_NR * FW_LW;
constexpr int FW_NRM_MAX = 256;

template <int COUT>
__global__ __launch_bounds__(256, 2) void conv3x3_few(const ConvArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* s_in = smem;                                              // [2][FW_CK][FW_NR][FW_LW]
  float2* s_nrm = reinterpret_cast<float2*>(smem + 2 * FW_STAGE);  // [Cin] (scale, shift)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ConvTile ct = conv_tile(a);
  if (!ct.valid) return;
  const int t0 = ct.t_tile * TT;
  const int f0 = ct.f_tile * FW_RT;
  const int n = ct.n;
  const int T = a.T, Tp = a.Tp, Fin = a.Fin, Cin = a.Cin;
  const int nchunk = Cin / FW_CK;
  const int fin0 = f0 - a.padf;

  for (int c = tid; c < Cin; c += 256) {
    float mean = 0.f, rstd = 1.f;
    if (c >= a.ident_c) {
      const dstat_t* st = a.in_stats + ((long long)n * a.in_sstride + a.in_c0 + c) * (2 * DS_NL);
      const double cnt = (double)Fin * (double)T;
      const double m = dstat_read(st) / cnt;
      double var = dstat_read(st + DS_NL) / cnt - m * m;
      var = var > 0.0 ? var : 0.0;
      mean = (float)m;
      rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));
    }
    s_nrm[c] = make_float2(rstd, -mean * rstd);
  }

  // ---- staging roles (division-free, as conv3x3_mfma): thread (sq, sg) owns frames t0 + 4 sq .. + 3 of the items sg + 8 i;
  // threads < 2 * FW_ITEMS own the halo frames t0 - 1 / t0 + 128 of item tid >> 1 ----
  const int sq = tid & 31, sg = tid >> 5;
  const int tg = t0 + 4 * sq;
  const bool hrole = tid < 2 * FW_ITEMS;
  const int hit = hrole ? tid >> 1 : 0, hside = tid & 1;
  const int hch = hit / FW_NR, hrow = hit - hch * FW_NR;
  const int htg = hside ? t0 + TT : t0 - 1;
  const bool hok = hrole && htg >= 0 && htg < T && (fin0 + hrow) >= 0 && (fin0 + hrow) < Fin;
  const unsigned row_e = (unsigned)Tp, plane_e = (unsigned)Fin * row_e;
  const float* in_n = a.in + (long long)n * a.in_bstride + (long long)a.in_c0 * Fin * Tp;
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc_e(reinterpret_cast<unsigned long long>(in_n), (unsigned)Cin * plane_e * 4u);
  const unsigned tg_e = (unsigned)(tg + 4 <= Tp ? tg : Tp - 4);
  const bool full_t = (t0 + TT <= T);
  int ich[FW_NI], irow[FW_NI];
  unsigned ioff[FW_NI];
  bool iok[FW_NI];
#pragma unroll
  for (int i = 0; i < FW_NI; ++i) {
    const int it = sg + 8 * i;
    ich[i] = it / FW_NR;
    irow[i] = it - ich[i] * FW_NR;
    int fin = fin0 + irow[i];
    iok[i] = fin >= 0 && fin < Fin;
    fin = fin < 0 ? 0 : (fin >= Fin ? Fin - 1 : fin);
    ioff[i] = (unsigned)ich[i] * plane_e * 4u + ((unsigned)fin * row_e + tg_e) * 4u;
  }
  unsigned hoff;
  {
    int fh = fin0 + hrow;
    fh = fh < 0 ? 0 : (fh >= Fin ? Fin - 1 : fh);
    const int th = htg < 0 ? 0 : (htg >= Tp ? Tp - 1 : htg);
    hoff = (unsigned)hch * plane_e * 4u + ((unsigned)fh * row_e + (unsigned)th) * 4u;
  }
  ff4 pin[FW_NI];
  float ph = 0.f;
#define FW_ISSUE(KC)                                                                                   \
  {                                                                                                    \
    const unsigned cb_ = (unsigned)((KC) * FW_CK) * plane_e * 4u;                                      \
    _Pragma("unroll") for (int i = 0; i < FW_NI; ++i)                                                  \
        pin[i] = __builtin_bit_cast(ff4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, ioff[i], cb_, 0)); \
    ph = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, hoff, cb_, 0));         \
  }
#define FW_COMMIT(KC, BUF)                                                                             \
  {                                                                                                    \
    float* dst_ = s_in + (BUF) * FW_STAGE;                                                             \
    _Pragma("unroll") for (int i = 0; i < FW_NI; ++i) {                                                \
      float2 m_ = s_nrm[(KC) * FW_CK + ich[i]];                                                        \
      m_ = iok[i] ? m_ : make_float2(0.f, 0.f);      /* a row outside the image: zero padding AFTER the norm */ \
      ff4 v_;                                                                                          \
      v_.x = fmaf(pin[i].x, m_.x, m_.y); v_.y = fmaf(pin[i].y, m_.x, m_.y);                            \
      v_.z = fmaf(pin[i].z, m_.x, m_.y); v_.w = fmaf(pin[i].w, m_.x, m_.y);                            \
      if (!full_t) {                                 /* uniform: only the last frame tile of an utterance */ \
        v_.x = (tg + 0 < T) ? v_.x : 0.f; v_.y = (tg + 1 < T) ? v_.y : 0.f;                            \
        v_.z = (tg + 2 < T) ? v_.z : 0.f; v_.w = (tg + 3 < T) ? v_.w : 0.f;                            \
      }                                                                                                \
      float* d_ = dst_ + (ich[i] * FW_NR + irow[i]) * FW_LW + 1 + 4 * sq;                              \
      d_[0] = v_.x; d_[1] = v_.y; d_[2] = v_.z; d_[3] = v_.w;                                          \
    }                                                                                                  \
    if (hrole) {                                                                                       \
      const float2 m2_ = s_nrm[(KC) * FW_CK + hch];                                                    \
      dst_[(hch * FW_NR + hrow) * FW_LW + (hside ? TT + 1 : 0)] = hok ? fmaf(ph, m2_.x, m2_.y) : 0.f;  \
    }                                                                                                  \
  }

  ff2 acc[4][COUT];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[r][co] = ff2{0.f, 0.f};

  const bool wave_ok = f0 + 4 * wave < a.Fout;                       // wave-uniform
  const float* __restrict__ wsm = a.wsm;                             // [Cin][9][4]: (kt * 3 + kf) x co, conv-form taps

  FW_ISSUE(0)
  __syncthreads();                                                   // s_nrm visible
  FW_COMMIT(0, 0)
  __syncthreads();

  for (int kc = 0; kc < nchunk; ++kc) {
    const bool more = kc + 1 < nchunk;
    if (more) FW_ISSUE(kc + 1)
    if (wave_ok) {
      const float* sb = s_in + (kc & 1) * FW_STAGE + (4 * wave) * FW_LW + 2 * lane;
#pragma unroll
      for (int cl = 0; cl < FW_CK; ++cl) {
        const float* __restrict__ wc = wsm + (long long)(kc * FW_CK + cl) * 36;
        // the 6 input rows of this lane's 4 output rows: columns 2 l .. 2 l + 3 = frames t - 1 .. t + 2 (t = t0 + 2 l)
        ff2 xa[6], xm[6], xb[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const float* p = sb + (cl * FW_NR + i) * FW_LW;
          xa[i] = *reinterpret_cast<const ff2*>(p);                  // (t - 1, t)
          xb[i] = *reinterpret_cast<const ff2*>(p + 2);              // (t + 1, t + 2)
          xm[i] = ff2{xa[i].y, xb[i].x};                             // (t, t + 1)
        }
#pragma unroll
        for (int kf = 0; kf < 3; ++kf)
#pragma unroll
          for (int co = 0; co < COUT; ++co) {
            const float w0 = wc[(0 * 3 + kf) * 4 + co], w1 = wc[(1 * 3 + kf) * 4 + co], w2 = wc[(2 * 3 + kf) * 4 + co];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              ff2 v = acc[r][co];
              v = __builtin_elementwise_fma(ff2{w0, w0}, xa[r + kf], v);
              v = __builtin_elementwise_fma(ff2{w1, w1}, xm[r + kf], v);
              v = __builtin_elementwise_fma(ff2{w2, w2}, xb[r + kf], v);
              acc[r][co] = v;
            }
          }
      }
    }
    if (more) FW_COMMIT(kc + 1, (kc + 1) & 1)
    __syncthreads();
  }
#undef FW_ISSUE
#undef FW_COMMIT

  // ---- + bias, raw store: 8 bytes per lane, 512 contiguous bytes per (wave, row, channel) ----
  if (!wave_ok) return;
  const unsigned P4 = (unsigned)a.Fout * (unsigned)Tp * 4u;
  const float* ob = a.out + (long long)n * a.out_bstride + (long long)a.out_c0 * a.Fout * Tp;
  const __amdgpu_buffer_rsrc_t rs_out = make_rsrc_e(reinterpret_cast<unsigned long long>(ob), (unsigned)a.Cout * P4);
  const int t = t0 + 2 * lane;
  // (the biases BEFORE the stores: read inside the loop they are re-loaded after every store -- the stores may alias them for all the
  // compiler knows -- each with its own s_waitcnt vmcnt(0): 4 x COUT serial memory round trips at the end of every workgroup)
  float bv[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) bv[co] = a.bias[co];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = f0 + 4 * wave + r;
    if (f >= a.Fout) break;                                          // wave-uniform
    // (a pair whose second frame is >= T writes one word of the row's padding [T, Tp), which every consumer masks)
    const unsigned vo = t < T ? (unsigned)(f * Tp + t) * 4u : 0x80000000u;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float b = bv[co];
      const ff2 v = {acc[r][co].x + b, acc[r][co].y + b};
      typedef unsigned int uu2 __attribute__((ext_vector_type(2)));
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uu2, v), rs_out, vo + (unsigned)co * P4, 0, 0);
    }
  }
}

bool conv_few_ok(const ConvArgs& a) {
  return a.wsm != nullptr && !a.act && (a.Cout == 2 || a.Cout == 4) && a.sf == 1 && !a.tr2 && (a.Cin % FW_CK) == 0 && a.Cin <= FW_NRM_MAX &&
         !a.in_oct && !a.out_oct && a.Fout == a.Fin + 2 * a.padf - 2;
}

static size_t few_lds_bytes(int Cin) { return (size_t)(2 * FW_STAGE) * sizeof(float) + (size_t)Cin * sizeof(float2); }

hipError_t conv_few_init() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_few<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_few<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
}

hipError_t launch_conv_few(const ConvArgs& a_in, int n_samples, hipStream_t s) {
  ConvArgs a = a_in;
  if (!conv_few_ok(a)) return hipErrorInvalidValue;
  a.ncg = 1;
  const dim3 grid = conv_grid(a, n_samples, TT, FW_RT, (n_samples % 8 == 0) ? conv_xcd_env() : 0);
  const size_t lds = few_lds_bytes(a.Cin);
  if (a.Cout == 2) hipLaunchKernelGGL(conv3x3_few<2>, grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL(conv3x3_few<4>, grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

}  // namespace mn
