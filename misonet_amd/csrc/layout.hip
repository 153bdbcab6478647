// Layout conversion between the reference's spectrogram layout (complex64 [B, ch, T, F], F innermost;
// reference model.py:77-80,103-111) and the kernels' planar layout (float32 [n][c][F][Tp], frames innermost).
// All three kernels are LDS tile transposes of a [32 frames] x [F bins] tile: global reads and writes are both
// contiguous runs.  pack also materialises the circular microphone shifts of MISO1_Inference
// (torch.roll(mix, -k, dims=1), reference tester.py:1034,1050) so the 6 shifted forwards run as one batch.
#include "kernels.hpp"

namespace mn {

constexpr int LT = 32;        // frames per tile
constexpr int LFMAX = 132;    // >= F (129)

__global__ __launch_bounds__(256) void pack_k(const float2* src, int Mseg, int T, int F, float* dst,
                                              long long dst_bstride, int Tp, int c_re, int c_im, int nshift) {
  __shared__ float s_re[LFMAX][LT + 1];
  __shared__ float s_im[LFMAX][LT + 1];
  const int t0 = blockIdx.x * LT, m = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  const float2* sp = src + ((long long)(b * Mseg + m) * T + t0) * F;
  const int nt = min(LT, T - t0);
  for (int i = tid; i < nt * F; i += 256) {
    const int tl = i / F, f = i - tl * F;
    const float2 v = sp[i];
    s_re[f][tl] = v.x;
    s_im[f][tl] = v.y;
  }
  __syncthreads();
  const int tl = tid & 31, fr = tid >> 5;   // 8 rows per pass
  for (int k = 0; k < nshift; ++k) {
    // destination sample b*nshift + k holds roll(src, -k): dst channel md = (m - k) mod Mseg
    int md = m - k;
    if (md < 0) md += Mseg;
    float* dn = dst + (long long)(b * nshift + k) * dst_bstride;
    float* dre = dn + (long long)(c_re + md) * F * Tp + t0;
    float* dim_ = dn + (long long)(c_im + md) * F * Tp + t0;
    for (int f = fr; f < F; f += 8) {
      if (tl < nt) {
        dre[(long long)f * Tp + tl] = s_re[f][tl];
        dim_[(long long)f * Tp + tl] = s_im[f][tl];
      }
    }
  }
}

// bstride: floats between consecutive samples of the source buffer (the activation arena is sample-major: net.hip)
// mode 0: output o = n*S + s reads planes (c_re0 + s, c_im0 + s) of sample n
// mode 1: aligned MISO1 estimates: o = (b*S + j)*M + m reads sample b*M + m, speaker q = sel[(b*M + m)*S + j]
__global__ __launch_bounds__(256) void unpack_k(const float* src, long long bstride, int Tp, int S, int T, int F,
                                                int c_re0, int c_im0, int mode, int M, const int* sel, float2* dst,
                                                int* nan_flag) {
  __shared__ float s_re[LFMAX][LT + 1];
  __shared__ float s_im[LFMAX][LT + 1];
  const int t0 = blockIdx.x * LT, o = blockIdx.y;
  const int tid = threadIdx.x;
  long long pre, pim;
  if (mode == 0) {
    const int n = o / S, s = o - n * S;
    pre = (long long)n * bstride + (long long)(c_re0 + s) * F * Tp;
    pim = (long long)n * bstride + (long long)(c_im0 + s) * F * Tp;
  } else {
    const int m = o % M, bj = o / M, j = bj % S, b = bj / S;
    const int n = b * M + m;
    const int q = sel[n * S + j];
    pre = (long long)n * bstride + (long long)(c_re0 + q) * F * Tp;
    pim = (long long)n * bstride + (long long)(c_im0 + q) * F * Tp;
  }
  const float* sre = src + pre + t0;
  const float* sim = src + pim + t0;
  const int nt = min(LT, T - t0);
  const int tl = tid & 31, fr = tid >> 5;
  for (int f = fr; f < F; f += 8) {
    if (tl < nt) {
      s_re[f][tl] = sre[(long long)f * Tp + tl];
      s_im[f][tl] = sim[(long long)f * Tp + tl];
    }
  }
  __syncthreads();
  float2* dp = dst + ((long long)o * T + t0) * F;
  bool bad = false;
  for (int i = tid; i < nt * F; i += 256) {
    const int t = i / F, f = i - t * F;
    const float2 v = make_float2(s_re[f][t], s_im[f][t]);
    bad |= (v.x != v.x) || (v.y != v.y);
    dp[i] = v;
  }
  if (nan_flag && bad) atomicOr(nan_flag, 1);
}

// oct = 2 / 3 / 4: the source buffer is in the oct layout of the bf16x3 / bf16x6 / f16x3 DMA dataflow (kernels.hpp), value =
// the sum of its pieces (exact for three bf16 parts)
__global__ __launch_bounds__(256) void export_k(const float* src, long long src_bstride, int c0, int C, int Fq, int T,
                                                int Tp, const dstat_t* stats, int sstride, int ident_c, float* dst,
                                                int oct) {
  __shared__ float s_v[LFMAX][LT + 1];
  const int t0 = blockIdx.x * LT, c = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x;
  float mean = 0.f, rstd = 1.f;
  if (stats && c >= ident_c) {
    const dstat_t* st = stats + ((long long)n * sstride + c0 + c) * (2 * DS_NL);
    const double cnt = (double)Fq * (double)T;
    const double m = dstat_read(st) / cnt;
    double var = dstat_read(st + DS_NL) / cnt - m * m;
    var = var > 0.0 ? var : 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));
  }
  const float* sp = src + (long long)n * src_bstride + (long long)(c0 + c) * Fq * Tp + t0;
  const int nt = min(LT, T - t0);
  const int tl = tid & 31, fr = tid >> 5;
  if (oct) {
    const unsigned short* hb = reinterpret_cast<const unsigned short*>(src + (long long)n * src_bstride);
    const long long half_e = (long long)(sstride >> 3) * Fq * Tp * 8;       // bf16 elements per half
    const int ch = c0 + c;
    for (int f = fr; f < Fq; f += 8)
      if (tl < nt) {
        const long long e = (((long long)(ch >> 3) * Fq + f) * Tp + t0 + tl) * 8 + (ch & 7);
        float v;
        if (oct == 4) {                                                       // fp16 pieces (f16x3)
          const _Float16* hh = reinterpret_cast<const _Float16*>(hb);
          v = (float)hh[e] + (float)hh[half_e + e];
        } else {
          v = __uint_as_float((unsigned)hb[e] << 16) + __uint_as_float((unsigned)hb[half_e + e] << 16);
          if (oct == 3) v += __uint_as_float((unsigned)hb[2 * half_e + e] << 16);
        }
        s_v[f][tl] = (v - mean) * rstd;
      }
  } else {
    for (int f = fr; f < Fq; f += 8)
      if (tl < nt) s_v[f][tl] = (sp[(long long)f * Tp + tl] - mean) * rstd;
  }
  __syncthreads();
  float* dp = dst + (((long long)n * C + c) * T + t0) * Fq;
  for (int i = tid; i < nt * Fq; i += 256) {
    const int t = i / Fq, f = i - t * Fq;
    dp[i] = s_v[f][t];
  }
}

hipError_t launch_pack(const float2* src, int B, int Mseg, int T, int F, float* dst, long long dst_bstride, int Tp,
                       int c_re, int c_im, int nshift, hipStream_t s) {
  if (F > LFMAX) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_k, dim3((T + LT - 1) / LT, Mseg, B), dim3(256), 0, s, src, Mseg, T, F, dst, dst_bstride, Tp,
                     c_re, c_im, nshift);
  return hipGetLastError();
}

hipError_t launch_unpack_ex(const float* src, long long src_bstride, int Tp, int S, int T, int F, int c_re0, int c_im0, int mode,
                            int M, const int* sel, float2* dst, int n_out, int* nan_flag, hipStream_t s) {
  if (F > LFMAX) return hipErrorInvalidValue;
  hipLaunchKernelGGL(unpack_k, dim3((T + LT - 1) / LT, n_out), dim3(256), 0, s, src, src_bstride, Tp, S, T, F, c_re0, c_im0,
                     mode, M, sel, dst, nan_flag);
  return hipGetLastError();
}

hipError_t launch_unpack(const float* src, long long src_bstride, int Tp, int S, int T, int F, float2* dst,
                         int n_samples, int* nan_flag, hipStream_t s) {
  return launch_unpack_ex(src, src_bstride, Tp, S, T, F, 0, S, 0, 1, nullptr, dst, n_samples * S, nan_flag, s);
}

hipError_t launch_export(const float* src, long long src_bstride, int c0, int C, int Fq, int T, int Tp,
                         const dstat_t* stats, int sstride, int ident_c, float* dst, int n_samples, hipStream_t s,
                         int oct) {
  if (Fq > LFMAX) return hipErrorInvalidValue;
  hipLaunchKernelGGL(export_k, dim3((T + LT - 1) / LT, C, n_samples), dim3(256), 0, s, src, src_bstride, c0, C, Fq, T,
                     Tp, stats, sstride, ident_c, dst, oct);
  return hipGetLastError();
}

}  // namespace mn
