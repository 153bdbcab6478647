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
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace mn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

int conv_cop(int Cout) { return Cout <= 32 ? 32 : 64; }
int conv_rows(int sf, int tr2) { return tr2 ? 3 : sf * (FT - 1) + 3; }

// MODE 0: forward / stride-1-transposed conv (sf = 1, NR = 6 staged rows); MODE 1: stride-2 conv (NR = 9);
// MODE 2: stride-2 transposed conv (NR = 3).
//
// Software pipeline per K-chunk (guide T14, "issue early / write late"): the global loads of chunk k+1 (NR float4 of
// the input patch + 1 halo scalar + the weight slab share per thread) are issued into registers BEFORE the MFMA loop
// of chunk k and are normalised and written to LDS after it, so HBM/L2 latency hides under the matrix work.
// Staging roles are division-free: thread (q = tid & 31, ci = tid >> 5) owns frames t0+4q..t0+4q+3 of channel ci
// for every staged row; threads < 16*NR own the two halo frames t0-1 / t0+128 of one (row, channel).
template <int NCO, int MODE>
__global__ __launch_bounds__(256, (NCO == 1 ? 3 : 2)) void conv3x3_mfma(const ConvArgs a) {
  constexpr int COP = NCO * 32;
  constexpr int NR = MODE == 0 ? 6 : (MODE == 1 ? 9 : 3);
  constexpr int SF = MODE == 1 ? 2 : 1;
  constexpr bool TR2 = MODE == 2;
  constexpr int NW4 = 9 * CK * COP / 4;          // float4 per weight slab
  constexpr int NWI = (NW4 + 255) / 256;
  extern __shared__ __align__(16) float smem[];
  float* s_in = smem;                         // [CK][NR][TW]: col 3 = frame t0-1, cols 4..131 = t0..t0+127, col 132 = t0+128
  float* s_w = s_in + CK * NR * TW;           // [9][CK][COP]
  float2* s_nrm = reinterpret_cast<float2*>(s_w + 9 * CK * COP);   // [CinP] (mean, rstd)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t0 = blockIdx.x * TT;
  const int f0 = blockIdx.y * FT;
  const int n = blockIdx.z / a.ncg;
  const int cg = blockIdx.z - n * a.ncg;
  const int T = a.T, Tp = a.Tp, Fin = a.Fin, Cin = a.Cin;
  const int nchunk = (Cin + CK - 1) / CK;
  const int fin0 = TR2 ? (f0 >> 1) - 1 : SF * f0 - a.padf;

  // instance-norm parameters of the input channels (normalise-on-load)
  for (int c = tid; c < nchunk * CK; c += 256) {
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
  const float4* w_g = reinterpret_cast<const float4*>(a.w + (long long)cg * nchunk * (9 * CK * COP));

  // staging roles
  const int sq = tid & 31, sci = tid >> 5;
  const int tg = t0 + 4 * sq;
  const bool tok = tg < Tp;
  const int hr = tid >> 4, hci = (tid >> 1) & 7, hside = tid & 1;
  const int htg = hside ? t0 + TT : t0 - 1;
  const bool hok = (hr < NR) && htg >= 0 && htg < T && (fin0 + hr) >= 0 && (fin0 + hr) < Fin;

  float4 pin[NR];
  float ph = 0.f;
  float4 pw[NWI];

  auto issue = [&](int kc) {
    const int c = kc * CK + sci;
    const bool cok = (c < Cin) && tok;
    const float* base = in_n + (long long)c * Fin * Tp + tg;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int fin = fin0 + r;
      pin[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cok && fin >= 0 && fin < Fin) pin[r] = *reinterpret_cast<const float4*>(base + (long long)fin * Tp);
    }
    ph = 0.f;
    const int c2 = kc * CK + hci;
    if (hok && c2 < Cin) ph = in_n[((long long)c2 * Fin + (fin0 + hr)) * Tp + htg];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int idx = tid + 256 * i;
      pw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (NW4 % 256 == 0 || idx < NW4) pw[i] = w_g[(long long)kc * NW4 + idx];
    }
  };
  auto commit = [&](int kc) {
    const int c = kc * CK + sci;
    const bool cok = (c < Cin) && tok;
    const float2 m = s_nrm[c];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int fin = fin0 + r;
      const bool ok = cok && fin >= 0 && fin < Fin;
      float4 v = pin[r];
      v.x = (ok && tg + 0 < T) ? (v.x - m.x) * m.y : 0.f;
      v.y = (ok && tg + 1 < T) ? (v.y - m.x) * m.y : 0.f;
      v.z = (ok && tg + 2 < T) ? (v.z - m.x) * m.y : 0.f;
      v.w = (ok && tg + 3 < T) ? (v.w - m.x) * m.y : 0.f;
      *reinterpret_cast<float4*>(s_in + (sci * NR + r) * TW + 4 + 4 * sq) = v;
    }
    if (hr < NR) {
      const int c2 = kc * CK + hci;
      const float2 m2 = s_nrm[c2];
      s_in[(hci * NR + hr) * TW + (hside ? TT + 4 : 3)] = (hok && c2 < Cin) ? (ph - m2.x) * m2.y : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int idx = tid + 256 * i;
      if (NW4 % 256 == 0 || idx < NW4) reinterpret_cast<float4*>(s_w)[idx] = pw[i];
    }
  };

  const int f = f0 + wave;
  const bool row_ok = f < a.Fout;                       // wave-uniform
  int nseg = (T - t0 + 31) >> 5;
  nseg = nseg > 4 ? 4 : nseg;

  f32x16 acc[NCO][4];
#pragma unroll
  for (int j = 0; j < NCO; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][s][r] = 0.f;

  const int half = lane >> 5, l31 = lane & 31;

  issue(0);
  __syncthreads();          // s_nrm visible
  commit(0);
  __syncthreads();

  for (int kc = 0; kc < nchunk; ++kc) {
    const bool more = (kc + 1 < nchunk);
    if (more) issue(kc + 1);
    if (row_ok) {
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
        for (int kf = 0; kf < 3; ++kf) {
          int rl;
          if (TR2) {
            const int v = (f - f0) + kf;      // f0 is a multiple of 4: parity of (f + kf - 2)
            if (v & 1) continue;
            rl = v >> 1;
          } else {
            rl = SF * (f - f0) + kf;
          }
          const float* wrow = s_w + ((kt * 3 + kf) * CK + half) * COP + l31;
          const float* irow = s_in + (half * NR + rl) * TW + l31 + kt + 3;
#pragma unroll
          for (int cp = 0; cp < CK / 2; ++cp) {
            float av[NCO], bv[4];
#pragma unroll
            for (int j = 0; j < NCO; ++j) av[j] = wrow[cp * 2 * COP + j * 32];
#pragma unroll
            for (int s = 0; s < 4; ++s) bv[s] = (s < nseg) ? irow[cp * 2 * NR * TW + s * 32] : 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              if (s < nseg) {
#pragma unroll
                for (int j = 0; j < NCO; ++j)
                  acc[j][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[s], acc[j][s], 0, 0, 0);
              }
            }
          }
        }
      }
    }
    __syncthreads();        // every wave is done reading this chunk
    if (more) {
      commit(kc + 1);
      __syncthreads();
    }
  }

  float* s_red = smem;   // [FT waves][COP][2]  (safe: the loop ends with a barrier after the last reads)
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

static size_t conv_lds_bytes(int NR, int cop, int Cin) {
  const int nchunk = (Cin + CK - 1) / CK;
  return (size_t)(CK * NR * TW + 9 * CK * cop) * sizeof(float) + (size_t)nchunk * CK * sizeof(float2);
}

template <int NCO, int MODE>
static hipError_t set_lds_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma<NCO, MODE>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}

hipError_t conv_init() {
  hipError_t e;
  if ((e = set_lds_attr<1, 0>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 1>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1, 2>()) != hipSuccess) return e;
  if ((e = set_lds_attr<2, 0>()) != hipSuccess) return e;
  if ((e = set_lds_attr<2, 1>()) != hipSuccess) return e;
  return set_lds_attr<2, 2>();
}

hipError_t launch_conv(const ConvArgs& a, int n_samples, hipStream_t s) {
  dim3 grid((a.T + TT - 1) / TT, (a.Fout + FT - 1) / FT, n_samples * a.ncg);
  const size_t lds = conv_lds_bytes(a.NR, a.cop, a.Cin);
  const int mode = a.tr2 ? 2 : (a.sf == 2 ? 1 : 0);
  if (a.NR != conv_rows(a.sf, a.tr2)) return hipErrorInvalidValue;
#define MN_LAUNCH(NCO, MODE) hipLaunchKernelGGL((conv3x3_mfma<NCO, MODE>), grid, dim3(256), lds, s, a)
  if (a.cop == 32) {
    if (mode == 0) MN_LAUNCH(1, 0); else if (mode == 1) MN_LAUNCH(1, 1); else MN_LAUNCH(1, 2);
  } else {
    if (mode == 0) MN_LAUNCH(2, 0); else if (mode == 1) MN_LAUNCH(2, 1); else MN_LAUNCH(2, 2);
  }
#undef MN_LAUNCH
  return hipGetLastError();
}

}  // namespace mn
