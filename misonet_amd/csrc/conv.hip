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

template <int NCO>
__global__ __launch_bounds__(256) void conv3x3_mfma(const ConvArgs a) {
  constexpr int COP = NCO * 32;
  extern __shared__ __align__(16) float smem[];
  const int NR = a.NR;
  float* s_in = smem;                         // [CK][NR][TW]
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
  const int fin0 = a.tr2 ? (f0 >> 1) - 1 : a.sf * f0 - a.padf;

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
  const float* w_g = a.w + (long long)cg * nchunk * (9 * CK * COP);

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

  for (int kc = 0; kc < nchunk; ++kc) {
    __syncthreads();   // previous chunk fully consumed (and s_nrm visible on the first pass)
    // ---- stage the normalised input patch: rows fin0..fin0+NR-1, frames t0-4..t0+131 ----
    const int n4 = CK * NR * (TW / 4);
    for (int i = tid; i < n4; i += 256) {
      const int q = i % (TW / 4);
      const int rr = i / (TW / 4);
      const int r = rr % NR;
      const int ci = rr / NR;
      const int c = kc * CK + ci;
      const int fin = fin0 + r;
      const int tg = t0 - 4 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < Cin && fin >= 0 && fin < Fin && tg >= 0 && tg < Tp) {
        v = *reinterpret_cast<const float4*>(in_n + ((long long)c * Fin + fin) * Tp + tg);
        const float2 m = s_nrm[c];
        v.x = (tg + 0 < T) ? (v.x - m.x) * m.y : 0.f;
        v.y = (tg + 1 < T) ? (v.y - m.x) * m.y : 0.f;
        v.z = (tg + 2 < T) ? (v.z - m.x) * m.y : 0.f;
        v.w = (tg + 3 < T) ? (v.w - m.x) * m.y : 0.f;
      }
      *reinterpret_cast<float4*>(s_in + (ci * NR + r) * TW + 4 * q) = v;
    }
    // ---- stage the weight slab of this chunk (already in LDS order) ----
    {
      const float4* wsrc = reinterpret_cast<const float4*>(w_g + (long long)kc * (9 * CK * COP));
      float4* wdst = reinterpret_cast<float4*>(s_w);
      for (int i = tid; i < 9 * CK * COP / 4; i += 256) wdst[i] = wsrc[i];
    }
    __syncthreads();

    if (row_ok) {
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
        for (int kf = 0; kf < 3; ++kf) {
          int rl;
          if (a.tr2) {
            const int v = (f - f0) + kf;      // f0 is a multiple of 4: parity of (f + kf - 2)
            if (v & 1) continue;
            rl = v >> 1;
          } else {
            rl = a.sf * (f - f0) + kf;
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
  }
  __syncthreads();   // all waves done with s_in / s_w: reuse the front of LDS for the statistics

  float* s_red = smem;   // [FT waves][COP][2]
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

hipError_t conv_init() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma<1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma<2>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}

hipError_t launch_conv(const ConvArgs& a, int n_samples, hipStream_t s) {
  dim3 grid((a.T + TT - 1) / TT, (a.Fout + FT - 1) / FT, n_samples * a.ncg);
  const size_t lds = conv_lds_bytes(a.NR, a.cop, a.Cin);
  if (a.cop == 32)
    hipLaunchKernelGGL(conv3x3_mfma<1>, grid, dim3(256), lds, s, a);
  else
    hipLaunchKernelGGL(conv3x3_mfma<2>, grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

}  // namespace mn
