// Temporal convolution network kernels (reference model.py:486-632): TemporalConvNet(2,7,128,128,128,"IN").
//
// One TemporalBlock (model.py:517-550) =  x + DS2(ELU(IN1d(DS1(ELU(IN1d(x))))))  with
// DS(y) = pwconv(gLN(PReLU(dwconv_dilated(y))))  (model.py:553-567).  Every norm is a reduction over the whole
// utterance (IN1d: per (n,c) over T; gLN: per n over (C,T)), so a DS conv is two launches:
//   tcn_dw : a = ELU(IN1d(x)) on the fly -> depth-wise dilated conv -> PReLU -> d ; gLN sums of d
//   tcn_pw : g = gLN(d) on the fly -> 128x128 point-wise conv on the fp32 matrix cores (+ residual) ; IN1d sums
// All sums are float64 and bit-reproducible WITHOUT atomics: a producer writes one partial (sum, sum of squares) per
// workgroup -- IN1d: one per 128-frame tile of a (n, c) row, gLN: one per 4-channel group of a sample -- and the consumer
// adds the partials of its row / sample in index order (tpart_sum).  Activations are planar [n][c][Tp] (F = 1).
#include "kernels.hpp"
#include "conv_epilogue.hpp"

namespace mn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline double block_sum_256(double v, double* s_tmp /*[4]*/) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) s_tmp[tid >> 6] = v;
  __syncthreads();
  return s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
}

// fixed-order sum of the np partials of one statistic pair
__device__ inline double2 tpart_sum(const double2* p, int np) {
  double2 s = p[0];
#pragma unroll 8      // (fully unrolled, the 32 uniform partials of a gLN sum are 128 SGPRs of scalar loads at once)
  for (int i = 1; i < np; ++i) { const double2 q = p[i]; s.x += q.x; s.y += q.y; }
  return s;
}

__device__ inline void in_params(double2 st, int T, float& mean, float& rstd) {
  const double m = st.x / (double)T;
  double var = st.y / (double)T - m * m;
  var = var > 0.0 ? var : 0.0;
  mean = (float)m;
  rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));
}

// x[n][c][t] = IN2d(raw)[n][c][t] (the encoder's last Conv2d_ output, F = 1; model.py:89) + sums of x
// raw_oct3: the source buffer is in the oct3 layout of the bf16x6 mode (three bf16 parts [c/8][f = 0][Tp][8], value = their
// exact sum); raw_sstride = channels of that buffer.
__global__ __launch_bounds__(256) void tcn_prepare_k(const float* raw, long long raw_bstride, int raw_c0,
                                                     const dstat_t* raw_stats, int raw_sstride, float* x,
                                                     double2* x_part, int nps, int C, int T, int Tp, int raw_oct3) {
  __shared__ double s_tmp[4];
  const int c = blockIdx.x, n = blockIdx.y;
  float mean, rstd;
  {
    const dstat_t* st = raw_stats + ((long long)n * raw_sstride + raw_c0 + c) * (2 * DS_NL);
    in_params(make_double2(dstat_read(st), dstat_read(st + DS_NL)), T, mean, rstd);
  }
  const float* src = raw + (long long)n * raw_bstride + (long long)(raw_c0 + c) * Tp;
  const unsigned short* so = reinterpret_cast<const unsigned short*>(raw + (long long)n * raw_bstride);
  const int ch = raw_c0 + c;
  const long long part_e = (long long)(raw_sstride >> 3) * Tp * 8;      // bf16 elements per part (F = 1)
  float* dst = x + ((long long)n * C + c) * Tp;
  double s1 = 0.0, s2 = 0.0;
  for (int t = threadIdx.x; t < T; t += 256) {
    float r;
    if (raw_oct3) {
      const long long e = ((long long)(ch >> 3) * Tp + t) * 8 + (ch & 7);
      r = (__uint_as_float((unsigned)so[e] << 16) + __uint_as_float((unsigned)so[part_e + e] << 16)) +
          __uint_as_float((unsigned)so[2 * part_e + e] << 16);
    } else {
      r = src[t];
    }
    const float v = (r - mean) * rstd;
    dst[t] = v;
    s1 += v;
    s2 += (double)v * v;
  }
  s1 = block_sum_256(s1, s_tmp);
  s2 = block_sum_256(s2, s_tmp);
  if (threadIdx.x == 0) x_part[((long long)n * C + c) * nps] = make_double2(s1, s2);     // the row's only partial
}

// d = PReLU(dwconv(ELU(IN1d(x)))), kernel 3, dilation = padding = dil, no bias (model.py:556-558)
// One workgroup per 4 channels of one sample, one WAVE per (n, c) row: a = ELU(IN1d(x)) is computed once per frame
// into the wave's LDS row (zero outside [0, T): Conv1d pads its input, i.e. the ELU output), then every lane produces
// 4 consecutive frames per pass from LDS.  Rows longer than the LDS row are walked in segments of DW_SEG frames with a
// DW_HALO-frame halo on both sides (dil <= DW_HALO), so there is no limit on T; T = 1001 is one segment.  One float64
// partial pair per workgroup for the gLN sums.
constexpr int DW_MAXT = 2048;                      // frames per row kept in LDS (4 rows x 8 KB)
constexpr int DW_HALO = 64;                        // largest dilation of TemporalConvNet(2, 7, ...): 2^6
constexpr int DW_SEG = DW_MAXT - 2 * DW_HALO;      // output frames per segment
// NORM: the outer norm in front of the ELU (launch_tcn_dw): 0 IN, 1 gLN, 2 cLN, 3 BN (eval).  Every variant is
// a[t] = ELU(x[t] * sc + sh) with a per-row (sc, sh), except cLN whose (mean, rstd) change per frame.
template <int NORM>
__global__ __launch_bounds__(256) void tcn_dw_k(const float* x, const double2* x_part, int x_np, int nps,
                                                const float* wdw, const float* prelu, float* d, double2* gln_part,
                                                int C, int T, int Tp, int dil, int seg_f, const float* nsc, const float* nsh,
                                                const float2* fstat) {
  __shared__ double s_tmp[4][2];
  extern __shared__ __align__(16) float s_a_dyn[];             // [4 rows][row_f]: row_f = frames of a segment + 2 halos
  const int row_f = seg_f + 2 * DW_HALO;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 4 + wave, n = blockIdx.y;
  // a[t] = ELU((x[t] - mu) * sc + sh).  The row mean is SUBTRACTED FIRST, as the reference does ((x - mean) * rstd): the
  // folded form x * rstd - mean * rstd loses eps32 * |mean| / std of its result, which compounds over the 28 instance norms
  // of the TCN when a row is a few nearly equal frames (T = 2, 3: tests/test_gpu_parity.py, shortest inputs).
  float sc, sh, mu = 0.f;
  if (NORM == 0) {
    float mean, rstd;
    in_params(tpart_sum(x_part + ((long long)n * C + c) * nps, x_np), T, mean, rstd);
    sc = rstd; sh = 0.f; mu = mean;
  } else if (NORM == 1) {
    // gLN (model.py:609-632): mean / biased variance over all (C, T) of the sample, eps 1e-8.  The sample's C * x_np
    // partials are added in a FIXED order: thread i takes rows i, i + 256, ... (their x_np partials in index order), then
    // the deterministic shuffle / LDS tree of block_sum_256.
    __shared__ double s_g[4];
    double a1 = 0.0, a2 = 0.0;
    for (int r = threadIdx.x; r < C; r += 256) {
      const double2 q = tpart_sum(x_part + ((long long)n * C + r) * nps, x_np);
      a1 += q.x; a2 += q.y;
    }
    a1 = block_sum_256(a1, s_g);
    a2 = block_sum_256(a2, s_g);
    const double cnt = (double)C * (double)T;
    const double gm = a1 / cnt;
    double gv = a2 / cnt - gm * gm;
    gv = gv > 0.0 ? gv : 0.0;
    const float rstd = (float)(1.0 / sqrt(gv + (double)GLN_EPS));
    sc = nsc[c] * rstd;
    sh = nsh[c];
    mu = (float)gm;
  } else if (NORM == 3) {
    sc = nsc[c]; sh = nsh[c];                                  // BatchNorm1d in eval mode, folded on the host
  } else {
    sc = nsc[c]; sh = nsh[c];                                  // cLN: gamma, beta (the per-frame statistics come from fstat)
  }
  const float2* fs = NORM == 2 ? fstat + (long long)n * Tp : nullptr;
  const float w0 = wdw[c * 3 + 0], w1 = wdw[c * 3 + 1], w2 = wdw[c * 3 + 2];
  const float slope = prelu[0];
  const float* src = x + ((long long)n * C + c) * Tp;
  float* dst = d + ((long long)n * C + c) * Tp;
  float* a = s_a_dyn + wave * row_f;
  const int Tq = (T + 3) & ~3;
  double r1 = 0.0, r2 = 0.0;
  for (int ts = 0; ts < Tq; ts += seg_f) {
    if (ts) __syncthreads();                                             // the previous segment is consumed
    // LDS slot j holds frame ts - DW_HALO + j
    for (int j = lane * 4; j < row_f; j += 256) {
      const int t = ts - DW_HALO + j;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t >= 0 && t < Tq) {                                            // Tp is a multiple of 32 >= Tq
        float4 v = *reinterpret_cast<const float4*>(src + t);
        if (NORM == 2) {                                       // (x - mean_t) * rstd_t, then gamma_c, beta_c
          const float4 f01 = *reinterpret_cast<const float4*>(fs + t), f23 = *reinterpret_cast<const float4*>(fs + t + 2);
          v.x = (v.x - f01.x) * f01.y; v.y = (v.y - f01.z) * f01.w;
          v.z = (v.z - f23.x) * f23.y; v.w = (v.w - f23.z) * f23.w;
        }
        o.x = (t + 0 < T) ? elu_fast(fmaf(v.x - mu, sc, sh)) : 0.f;
        o.y = (t + 1 < T) ? elu_fast(fmaf(v.y - mu, sc, sh)) : 0.f;
        o.z = (t + 2 < T) ? elu_fast(fmaf(v.z - mu, sc, sh)) : 0.f;
        o.w = (t + 3 < T) ? elu_fast(fmaf(v.w - mu, sc, sh)) : 0.f;
      }
      *reinterpret_cast<float4*>(a + j) = o;
    }
    __syncthreads();
    const int te = (ts + seg_f < Tq) ? ts + seg_f : Tq;
    float s1 = 0.f, s2 = 0.f;
    for (int t = ts + lane * 4; t < te; t += 256) {
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tc = t + i;
        const int j = tc - ts + DW_HALO;
        float acc = w1 * a[j];
        acc = fmaf(w0, a[j - dil], acc);                                 // zero outside [0, T)
        acc = fmaf(w2, a[j + dil], acc);
        acc = acc > 0.f ? acc : slope * acc;
        o[i] = acc;
        if (tc < T) { s1 += acc; s2 = fmaf(acc, acc, s2); }
      }
      *reinterpret_cast<float4*>(dst + t) = make_float4(o[0], o[1], o[2], o[3]);
    }
    // per-lane partials of a segment hold <= 32 terms in float; accumulate in float64
    r1 += (double)s1;
    r2 += (double)s2;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    r1 += __shfl_xor(r1, m, 64);
    r2 += __shfl_xor(r2, m, 64);
  }
  if (lane == 0) { s_tmp[wave][0] = r1; s_tmp[wave][1] = r2; }
  __syncthreads();
  if (threadIdx.x == 0)      // this 4-channel group's gLN partial; tcn_pw_k adds the C / 4 of a sample in group order
    gln_part[(long long)n * gridDim.x + blockIdx.x] = make_double2((s_tmp[0][0] + s_tmp[1][0]) + (s_tmp[2][0] + s_tmp[3][0]),
                                                                   (s_tmp[0][1] + s_tmp[1][1]) + (s_tmp[2][1] + s_tmp[3][1]));
}

// y[co][t] = sum_ci W[co][ci] * (gamma[ci] * (d[ci][t] - mean_n) * rstd_n + beta[ci])  (+ residual[co][t])
// C must be 128.  Workgroup: 128 output channels x 128 frames; wave w owns channels 32w..32w+31 and four 32-frame
// tiles (fp32 MFMA 32x32x2, exact).  K is consumed in 8 chunks of 16 input channels through two LDS buffers: the global
// loads of chunk k+1 are issued before the MFMAs of chunk k and committed (gLN applied) after them, one barrier per
// chunk (32 KB of LDS, <= 170 VGPRs: three workgroups per CU).  Epilogue: the residual tile is loaded with 64 buffer
// loads issued together, stores are buffer stores with an
// out-of-range offset for frames >= T (dropped by the hardware), statistics go through the reduce-scatter of
// conv_epilogue.hpp.
constexpr int PW_TT = 128;
constexpr int PW_KC = 16;
// Y_OCT3: y is an oct3 buffer of the bf16x6 mode (the TCN output feeds decoder 0 there): the accumulator layout is the
// conv kernels', so the row goes out through store_oct_row<3>; y_bstride in floats, y_cbuf = channels of that buffer.
// X6: the 128x128 product in the bf16x6 arithmetic of conv_bf16x6.hip (both operands split exactly into three bf16
// pieces, six leading partial products on v_mfma_f32_32x32x16_bf16, float32 accumulation) instead of the fp32 MFMA: the
// kernel is bound by the matrix pipe otherwise (3.2 GFLOP per launch at 52 TF/s).  The split of the normalised input
// happens once per workgroup on the way into the LDS, the split of a wave's weight fragment in its registers.
template <bool Y_OCT3, bool X6>
__global__ __launch_bounds__(256) void tcn_pw_k(const float* d, const double2* gln_part, const float* gamma,
                                                const float* beta, const float* wt /*[ci][co]*/,
                                                const float* residual, float* y, long long y_bstride, int y_c0,
                                                double2* y_part, int nps, int T, int Tp, int y_cbuf) {
  constexpr int C = 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int t0 = blockIdx.x * PW_TT, n = blockIdx.y;
  const double cnt = (double)C * (double)T;
  const double2 gs = tpart_sum(gln_part + (long long)n * (C / 4), C / 4);
  const double gm = gs.x / cnt;
  double gv = gs.y / cnt - gm * gm;
  gv = gv > 0.0 ? gv : 0.0;
  const float mean = (float)gm;
  const float rstd = (float)(1.0 / sqrt(gv + (double)GLN_EPS));
  const float* dn = d + (long long)n * C * Tp;

  f32x16 acc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

  if constexpr (X6) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    __shared__ __align__(16) u32x4_t s_x[2][3][2][PW_TT];      // [buffer][piece][octet of the chunk][frame]
    __shared__ float s_ga[C], s_be[C];
    if (tid < C) {
      const float g_ = gamma[tid] * rstd;
      s_ga[tid] = g_;
      s_be[tid] = beta[tid];                 // the sample mean is subtracted FIRST, as the reference does (model.py:609-632)
    }
    __syncthreads();
    // staging role: thread <-> (octet so of the 16-channel chunk, frame sf); A fragment: lane (l31, half) <-> output
    // channel wave * 32 + l31, input channels k0 + 8 * half .. + 7
    const int so = tid >> 7, sf = tid & 127;
    const int ts = t0 + sf;
    const bool tok = ts < T;
    const float* xl = dn + (long long)(8 * so) * Tp + ts;
    const float* wl = wt + (long long)(8 * half) * C + wave * 32 + l31;
    float xin[2][8], win[2][8];                                // loads run two chunks ahead of their use
    u32x4_t A[3];
#define X6_ISSUE(SL, K0)                                                                                          \
  _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                \
    xin[SL][c] = tok ? xl[((K0) + c) * Tp] : 0.f;          /* 32-bit offsets: C * Tp elements per sample */      \
    win[SL][c] = wl[((K0) + c) * C];                                                                             \
  }
#define X6_COMMIT(BUF, SL, K0)                                                                                   \
  {                                                                                                              \
    u32x4_t h_, m_, l_;                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
      const int c0_ = (K0) + 8 * so + 2 * i;                                                                     \
      const float v0_ = tok ? fmaf(xin[SL][2 * i] - mean, s_ga[c0_], s_be[c0_]) : 0.f;                              \
      const float v1_ = tok ? fmaf(xin[SL][2 * i + 1] - mean, s_ga[c0_ + 1], s_be[c0_ + 1]) : 0.f;                  \
      unsigned hh_, mm_, ll_;                                                                                    \
      split3_pair_t(v0_, v1_, hh_, mm_, ll_);                                                                    \
      h_[i] = hh_; m_[i] = mm_; l_[i] = ll_;                                                                     \
    }                                                                                                            \
    s_x[BUF][0][so][sf] = h_; s_x[BUF][1][so][sf] = m_; s_x[BUF][2][so][sf] = l_;                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
      unsigned hh_, mm_, ll_;                                                                                    \
      split3_pair_t(win[SL][2 * i], win[SL][2 * i + 1], hh_, mm_, ll_);                                                \
      A[0][i] = hh_; A[1][i] = mm_; A[2][i] = ll_;                                                               \
    }                                                                                                            \
  }
    X6_ISSUE(0, 0)
    X6_ISSUE(1, 16)
    X6_COMMIT(0, 0, 0)
    __syncthreads();
    // the slot / buffer index of a chunk is a LITERAL in each half of the pair below: indexed by `kc & 1` the two-slot
    // arrays stayed an alloca (68 bytes of scratch per lane: SROA runs before the loop is unrolled)
#define X6_STEP(KC, SL)                                                                                          \
    {                                                                                                            \
      if ((KC) + 2 < C / 16) X6_ISSUE(SL, ((KC) + 2) * 16)      /* slot SL was consumed by the commit of chunk KC */ \
      const bf16x8_t Ah = __builtin_bit_cast(bf16x8_t, A[0]), Am = __builtin_bit_cast(bf16x8_t, A[1]),           \
                     Al = __builtin_bit_cast(bf16x8_t, A[2]);                                                    \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                            \
        const bf16x8_t Bh = __builtin_bit_cast(bf16x8_t, s_x[SL][0][half][s * 32 + l31]);                        \
        const bf16x8_t Bm = __builtin_bit_cast(bf16x8_t, s_x[SL][1][half][s * 32 + l31]);                        \
        const bf16x8_t Bl = __builtin_bit_cast(bf16x8_t, s_x[SL][2][half][s * 32 + l31]);                        \
        /* small terms first: lh, hl, mm, mh, hm, hh */                                                          \
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[s], 0, 0, 0);                               \
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc[s], 0, 0, 0);                               \
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, acc[s], 0, 0, 0);                               \
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, acc[s], 0, 0, 0);                               \
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, acc[s], 0, 0, 0);                               \
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[s], 0, 0, 0);                               \
      }                                                                                                          \
      if ((KC) + 1 < C / 16) {                                                                                   \
        X6_COMMIT((SL) ^ 1, (SL) ^ 1, ((KC) + 1) * 16)                                                           \
        __syncthreads();                                                                                         \
      }                                                                                                          \
    }
#pragma unroll
    for (int kp = 0; kp < C / 32; ++kp) {
      X6_STEP(2 * kp, 0)
      X6_STEP(2 * kp + 1, 1)
    }
#undef X6_STEP
#undef X6_ISSUE
#undef X6_COMMIT
  } else {
  __shared__ __align__(16) float s_g[2][PW_KC][PW_TT];
  __shared__ __align__(16) float s_w[2][PW_KC][C];
  // staging roles: thread (q = tid & 31 -> frames 4q..4q+3, g = tid >> 5 -> channels g, g+8 of the chunk)
  const int sq = tid & 31, sg = tid >> 5;
  const int tg = t0 + 4 * sq;
  const bool tok = tg < Tp;
  float4 gi[2], wi[2];
  float ga[2], be[2];
#define PW_ISSUE(K0)                                                                                             \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
    const int ci = (K0) + sg + 8 * i;                                                                            \
    gi[i] = tok ? *reinterpret_cast<const float4*>(dn + (long long)ci * Tp + tg) : make_float4(0.f, 0.f, 0.f, 0.f); \
    wi[i] = *reinterpret_cast<const float4*>(wt + (long long)ci * C + 4 * sq);                                   \
    ga[i] = gamma[ci] * rstd;                                                                                    \
    be[i] = beta[ci];                                                                                            \
  }
#define PW_COMMIT(BUF)                                                                                           \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
    float4 v;                                                                                                    \
    v.x = (tg + 0 < T) ? fmaf(gi[i].x - mean, ga[i], be[i]) : 0.f;                                                      \
    v.y = (tg + 1 < T) ? fmaf(gi[i].y - mean, ga[i], be[i]) : 0.f;                                                      \
    v.z = (tg + 2 < T) ? fmaf(gi[i].z - mean, ga[i], be[i]) : 0.f;                                                      \
    v.w = (tg + 3 < T) ? fmaf(gi[i].w - mean, ga[i], be[i]) : 0.f;                                                      \
    *reinterpret_cast<float4*>(&s_g[BUF][sg + 8 * i][4 * sq]) = v;                                               \
    *reinterpret_cast<float4*>(&s_w[BUF][sg + 8 * i][4 * sq]) = wi[i];                                           \
  }
  PW_ISSUE(0)
  PW_COMMIT(0)
  __syncthreads();
#pragma unroll
  for (int kc = 0; kc < C / PW_KC; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < C / PW_KC) PW_ISSUE((kc + 1) * PW_KC)
#pragma unroll
    for (int kk = 0; kk < PW_KC; kk += 2) {
      const float av = s_w[buf][kk + half][wave * 32 + l31];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float bv = s_g[buf][kk + half][s * 32 + l31];
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[s], 0, 0, 0);
      }
    }
    if (kc + 1 < C / PW_KC) {
      PW_COMMIT(buf ^ 1)
      __syncthreads();
    }
  }
#undef PW_ISSUE
#undef PW_COMMIT

  }

  // ---- epilogue ----
  // residual tile: this lane's (channel, frame) elements in accumulator order, all loads issued together
  float rv[4][16];
  const float* rn = residual ? residual + (long long)n * C * Tp : nullptr;
  {
    const unsigned long long pa = reinterpret_cast<unsigned long long>(rn ? rn : dn);
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa);
    const unsigned phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long long)phi << 32) | plo), 0,
        __builtin_amdgcn_readfirstlane(rn ? (int)((unsigned)C * (unsigned)Tp * 4u) : 0), 0x00020000);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int t = t0 + s * 32 + l31;
      const unsigned vo = (t < T) ? (unsigned)((wave * 32 + 4 * half) * Tp + t) * 4u : 0x80000000u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co_off = (r & 3) + 8 * (r >> 2);
        rv[s][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, vo + (unsigned)(co_off * Tp) * 4u, 0, 0));
      }
    }
  }

  float* yn = y + (long long)n * y_bstride + (long long)y_c0 * Tp;
  const unsigned long long pa = reinterpret_cast<unsigned long long>(yn);
  const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa);
  const unsigned phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)phi << 32) | plo), 0,
      __builtin_amdgcn_readfirstlane((int)((unsigned)C * (unsigned)Tp * 4u)), 0x00020000);
  // oct3 destination: three parts, each [y_cbuf / 8][F = 1][Tp] 16-byte units, this kernel's channels at octet y_c0 / 8
  const unsigned OP16 = (unsigned)Tp * 16u;
  __amdgpu_buffer_rsrc_t rs_o[3];
  {
    const unsigned long long po = reinterpret_cast<unsigned long long>(y) + (unsigned long long)n * y_bstride * 4ull +
                                  (unsigned long long)(y_c0 >> 3) * OP16;
    const unsigned long long pb = (unsigned long long)(y_cbuf >> 3) * OP16;
    const unsigned nrec = Y_OCT3 ? (unsigned)(C >> 3) * OP16 : 0u;
    rs_o[0] = make_rsrc_e(po, nrec);
    rs_o[1] = make_rsrc_e(po + pb, nrec);
    rs_o[2] = make_rsrc_e(po + 2 * pb, nrec);
  }
  // IN1d partials of this tile, accumulated ABOUT A PIVOT: the row's own value at the tile's first frame (always inside the
  // utterance).  Uncentred float32 sums lose eps32 * mean^2 / var of the variance they are combined into (E[x^2] - mean^2):
  // nothing at T = 1001, but 6 x the float32 oracle's own error when a row has 2-3 frames (an instance norm over three
  // nearly equal values; tests/test_gpu_parity.py, shortest inputs).  The centred sums are exact to float32 round-off of
  // THEMSELVES and go back to plain (sum, sum of squares) in float64 below, so the consumer's format does not change.
  float s1[16], s2[16], piv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    s1[r] = 0.f; s2[r] = 0.f;
    piv[r] = __shfl(acc[0][r] + rv[0][r], lane & 32, 64);          // frame t0 of channel (r, half)
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int t = t0 + s * 32 + l31;
    const bool ok = t < T;
    const float m = ok ? 1.f : 0.f;
    const unsigned vo = ok ? (unsigned)((wave * 32 + 4 * half) * Tp + t) * 4u : 0x80000000u;
    float vrow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co_off = (r & 3) + 8 * (r >> 2);
      const float v = acc[s][r] + rv[s][r];
      vrow[r] = v;
      if (!Y_OCT3)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, vo + (unsigned)(co_off * Tp) * 4u, 0, 0);
      const float vm = (v - piv[r]) * m;
      s1[r] += vm;
      s2[r] = fmaf(vm, vm, s2[r]);
    }
    if (Y_OCT3)     // octets 4 * wave + {0, 2} + half of this kernel's 16 octets, frame t
      store_oct_row<3>(vrow, rs_o, (unsigned)t * 16u + (unsigned)(4 * wave + half) * OP16, OP16, ok, ok);
  }
  if (y_part) {              // this 128-frame tile's partial of every output channel (tcn_dw_k adds a row's tiles in order)
    const float x1 = reduce16_halfwave(s1, lane);
    const float x2 = reduce16_halfwave(s2, lane);
    if ((lane & 16) == 0) {
      const int q = lane & 15;
      const int co = wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
      float pq = piv[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) pq = (q == r) ? piv[r] : pq;
      // sum x = S1 + n c, sum x^2 = S2 + 2 c S1 + n c^2 with the tile's n valid frames, in float64
      const double c = (double)pq, S1 = (double)x1, S2 = (double)x2;
      const double nv = (double)((T - t0) < PW_TT ? (T - t0) : PW_TT);
      y_part[((long long)n * C + co) * nps + blockIdx.x] = make_double2(S1 + nv * c, S2 + 2.0 * c * S1 + nv * c * c);
    }
  }
}

int tcn_part_slots(int T) { return (T + PW_TT - 1) / PW_TT; }

hipError_t launch_tcn_prepare(const float* raw, long long raw_bstride, int raw_c0, const dstat_t* raw_stats,
                              int raw_sstride, float* x, double2* x_part, int C, int T, int Tp, int n_samples,
                              hipStream_t s, int raw_oct3) {
  hipLaunchKernelGGL(tcn_prepare_k, dim3(C, n_samples), dim3(256), 0, s, raw, raw_bstride, raw_c0, raw_stats,
                     raw_sstride, x, x_part, tcn_part_slots(T), C, T, Tp, raw_oct3);
  return hipGetLastError();
}

// ChannelwiseLayerNorm statistics (model.py:583-606): one thread per frame, two passes over the C channels (mean, then
// the biased variance about it, as torch.mean / torch.var(unbiased=False)); reads are coalesced along t.
__global__ __launch_bounds__(256) void tcn_cln_stats_k(const float* x, float2* fstat, int C, int T, int Tp) {
  const int t = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (t >= Tp) return;
  float2 o = make_float2(0.f, 0.f);
  if (t < T) {
    const float* p = x + (long long)n * C * Tp + t;
    float m = 0.f;
    for (int c = 0; c < C; ++c) m += p[(long long)c * Tp];
    m /= (float)C;
    float v = 0.f;
    for (int c = 0; c < C; ++c) { const float dd = p[(long long)c * Tp] - m; v = fmaf(dd, dd, v); }
    v /= (float)C;
    o = make_float2(m, 1.f / sqrtf(v + GLN_EPS));
  }
  fstat[(long long)n * Tp + t] = o;
}

hipError_t launch_tcn_cln_stats(const float* x, float2* fstat, int C, int T, int Tp, int n_samples, hipStream_t s) {
  hipLaunchKernelGGL(tcn_cln_stats_k, dim3((Tp + 255) / 256, n_samples), dim3(256), 0, s, x, fstat, C, T, Tp);
  return hipGetLastError();
}

hipError_t launch_tcn_dw(const float* x, const double2* x_part, int x_np, const float* wdw, const float* prelu, float* d,
                         double2* gln_part, int C, int T, int Tp, int dilation, int n_samples, hipStream_t s, int norm_kind,
                         const float* nsc, const float* nsh, const float2* fstat) {
  if (dilation < 1 || dilation > DW_HALO) return hipErrorInvalidValue;
  if (C % 4) return hipErrorInvalidValue;
  if (norm_kind < 0 || norm_kind > 3 || (norm_kind && (!nsc || !nsh)) || (norm_kind == 2 && !fstat)) return hipErrorInvalidValue;
  // LDS row = the frames of one segment + two halos: a 4-second utterance (T = 1001) takes 18 KB per workgroup instead
  // of the 32 KB of a full 1920-frame segment, i.e. 8 instead of 5 workgroups per CU of this latency-bound kernel
  const int tq = (T + 3) & ~3;
  const int seg_f = tq < DW_SEG ? tq : DW_SEG;
  const dim3 g(C / 4, n_samples);
  const size_t lds = (size_t)4 * (seg_f + 2 * DW_HALO) * sizeof(float);
#define MN_DW(K) hipLaunchKernelGGL(tcn_dw_k<K>, g, dim3(256), lds, s, x, x_part, x_np, tcn_part_slots(T), wdw, prelu, d, gln_part, \
                                    C, T, Tp, dilation, seg_f, nsc, nsh, fstat)
  if (norm_kind == 0) MN_DW(0); else if (norm_kind == 1) MN_DW(1); else if (norm_kind == 2) MN_DW(2); else MN_DW(3);
#undef MN_DW
  return hipGetLastError();
}

hipError_t launch_tcn_pw(const float* d, const double2* gln_part, const float* gamma, const float* beta,
                         const float* wpw, const float* residual, float* y, long long y_bstride, int y_c0,
                         double2* y_part, int C, int T, int Tp, int n_samples, hipStream_t s, int y_oct3_cbuf, int x6) {
  if (C != 128) return hipErrorInvalidValue;
  const dim3 g((T + PW_TT - 1) / PW_TT, n_samples);
  const int nps = tcn_part_slots(T);
  if (y_oct3_cbuf) {
    if ((y_c0 & 7) || (y_oct3_cbuf & 7)) return hipErrorInvalidValue;
    if (x6) hipLaunchKernelGGL((tcn_pw_k<true, true>), g, dim3(256), 0, s, d, gln_part, gamma, beta, wpw, residual, y,
                               y_bstride, y_c0, y_part, nps, T, Tp, y_oct3_cbuf);
    else hipLaunchKernelGGL((tcn_pw_k<true, false>), g, dim3(256), 0, s, d, gln_part, gamma, beta, wpw, residual, y,
                            y_bstride, y_c0, y_part, nps, T, Tp, y_oct3_cbuf);
  } else {
    if (x6) hipLaunchKernelGGL((tcn_pw_k<false, true>), g, dim3(256), 0, s, d, gln_part, gamma, beta, wpw, residual, y,
                               y_bstride, y_c0, y_part, nps, T, Tp, 0);
    else hipLaunchKernelGGL((tcn_pw_k<false, false>), g, dim3(256), 0, s, d, gln_part, gamma, beta, wpw, residual, y,
                            y_bstride, y_c0, y_part, nps, T, Tp, 0);
  }
  return hipGetLastError();
}

}  // namespace mn
