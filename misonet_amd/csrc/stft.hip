// On-device STFT front-end: replaces AudioDataset_Test's per-channel scipy.signal.stft + "/scale" +
// permute (reference dataloader/data.py:505-522, 540-544) AND the pack step: waveform in, planar network input out.
//
//   X[m][t][f] = sum_{j<256} x[m][64 t - 128 + j] * hann[j] * exp(-2 pi i f j / 256),   x = 0 outside [0, L)
//   (hann-256, hop 64, zero 'boundary' padding of 128 samples, un-normalised: SciPy's 'spectrum' scaling is undone by
//    the reference's "/scale", data.py:497-498,542).
//
// The DFT is a dense [258 x 256] x [256 x frames] product, so it runs on the fp32 matrix cores: A = windowed twiddles
// (rows 0..128 = cos, rows 144..272 = -sin, from a table built in float64 on the host; read straight from L2,
// 128-byte coalesced because the table is stored [j][row]), B = the frames of one microphone, expanded in LDS as
// Bm[frame][sample] with row pitch 257 floats so the 32 lanes of a B fragment (one per frame) hit 32 different banks.
// Output goes directly to the planar layout [n][c][f][Tp] with the circular microphone shifts of MISO1_Inference
// (tester.py:1034,1050) materialised, exactly like pack_k.
#include "kernels.hpp"

namespace mn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ST_N = 256;         // nperseg
constexpr int ST_HOP = 64;
constexpr int ST_ROWS = 288;      // 9 MFMA row tiles: cos rows 0..128, -sin rows 144..272
constexpr int ST_IM0 = 144;
constexpr int ST_FR = 64;         // frames per workgroup
constexpr int ST_PITCH = 257;

// wav: [B][L][Mw] float32 (time-major, microphones interleaved: librosa.load(...).T, data.py:605-616);
// mic m of utterance b = wav[(b*L + l)*Mw + m].
__global__ __launch_bounds__(256) void stft_pack_k(const float* wav, int L, int Mw, int T, const float* twid,
                                                   float* dst, long long dst_bstride, int Tp, int F, int c_re,
                                                   int c_im, int nshift) {
  extern __shared__ __align__(16) float st_smem[];
  float* s_seg = st_smem;                                   // [ST_FR*ST_HOP + 192]
  float* s_bm = s_seg + (ST_FR * ST_HOP + 192 + 63) / 64 * 64;   // [ST_FR][ST_PITCH]
  const int t0 = blockIdx.x * ST_FR, m = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int nseg = ST_FR * ST_HOP + 192;
  const long long base = (long long)b * L;
  const int s0 = t0 * ST_HOP - 128;
  for (int i = tid; i < nseg; i += 256) {
    const int l = s0 + i;
    s_seg[i] = (l >= 0 && l < L) ? wav[(base + l) * Mw + m] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < ST_FR * ST_N; i += 256) {
    const int n = i >> 8, j = i & 255;
    s_bm[n * ST_PITCH + j] = s_seg[n * ST_HOP + j];
  }
  __syncthreads();

  // wave w owns row tiles w, w+4 (and 8 for wave 0) x two 32-frame column tiles
  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;
  const int ntile = (wave == 0) ? 3 : 2;
  for (int k = 0; k < ST_N; k += 2) {
    const int j = k + half;
    float bv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) bv[q] = s_bm[(q * 32 + l31) * ST_PITCH + j];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < ntile) {
        const float av = twid[j * ST_ROWS + (wave + 4 * i) * 32 + l31];
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[q], acc[i][q], 0, 0, 0);
      }
    }
  }
  // store: row -> (part, f); column -> frame
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i < ntile) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = t0 + q * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wave + 4 * i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          int f, cbase;
          if (row < F) { f = row; cbase = c_re; }
          else if (row >= ST_IM0 && row < ST_IM0 + F) { f = row - ST_IM0; cbase = c_im; }
          else continue;
          if (t >= T) continue;
          const float v = acc[i][q][r];
          for (int ks = 0; ks < nshift; ++ks) {
            int md = m - ks;
            if (md < 0) md += Mw;
            dst[(long long)(b * nshift + ks) * dst_bstride + ((long long)(cbase + md) * F + f) * Tp + t] = v;
          }
        }
      }
    }
  }
}

static size_t stft_lds_bytes() {
  return (size_t)((ST_FR * ST_HOP + 192 + 63) / 64 * 64 + ST_FR * ST_PITCH) * sizeof(float);
}

hipError_t stft_init() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_pack_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                             128 * 1024);
}

// twiddle table [256][288] (host, float64 math): windowed cos / -sin
void stft_build_twiddles(float* tw) {
  const double pi = 3.14159265358979323846;
  for (int j = 0; j < ST_N; ++j) {
    const double w = 0.5 - 0.5 * cos(2.0 * pi * j / ST_N);          // periodic hann (scipy get_window('hann', 256))
    for (int r = 0; r < ST_ROWS; ++r) {
      double v = 0.0;
      if (r < 129) v = w * cos(2.0 * pi * (double)((r * j) % ST_N) / ST_N);
      else if (r >= ST_IM0 && r < ST_IM0 + 129) v = -w * sin(2.0 * pi * (double)(((r - ST_IM0) * j) % ST_N) / ST_N);
      tw[j * ST_ROWS + r] = (float)v;
    }
  }
}
int stft_twiddle_count() { return ST_N * ST_ROWS; }

hipError_t launch_stft_pack(const float* wav, int B, int L, int Mw, int T, const float* twid, float* dst,
                            long long dst_bstride, int Tp, int F, int c_re, int c_im, int nshift, hipStream_t s) {
  if (F != 129) return hipErrorInvalidValue;
  hipLaunchKernelGGL(stft_pack_k, dim3((T + ST_FR - 1) / ST_FR, Mw, B), dim3(256), stft_lds_bytes(), s, wav, L, Mw, T,
                     twid, dst, dst_bstride, Tp, F, c_re, c_im, nshift);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// On-device iSTFT + int16 (reference tester.py:949-952, 979-990: "x scale" -> scipy.signal.istft(hann, 256, 192) -> x 32767
// -> astype(int16)), the inverse of the front-end above:
//
//   y[n] = sum_t w[k] z_t[k] / sum_t w[k]^2,   k = n - 64 t + 128,   z_t = irfft_256(X_t)   (frames t with 0 <= k < 256)
//   z_t[k] = 1/256 sum_f c_f (Re X_t[f] cos(2 pi f k / 256) - Im X_t[f] sin(2 pi f k / 256)),   c_0 = c_128 = 1, else 2
//
// for n in [0, 64 (T - 1)): exactly torch.istft(center=True, length=64 (T - 1)) = SciPy's istft of the reference on the
// samples the reference keeps.  The windowed inverse DFT of 64 frames is a dense [256 x 258] x [258 x 64] product on the fp32
// matrix cores (A = the windowed inverse twiddles, float64-built, read from L2; B = the spectrogram tile in LDS with an odd
// row pitch); the overlap-add of the four frames that cover an output hop, the division by the window envelope (1.5 inside,
// the true partial sums at both ends), x 32767 and the truncating cast follow from LDS.  A workgroup owns 61 output hops
// (frames j0 - 1 .. j0 + 62); spec complex64 [N][T][129] (F innermost, the boundary layout) -> int16 [N][64 (T - 1)] and / or
// float32 of the same shape.
constexpr int IS_K = 260;          // 129 re + 129 im, padded to a multiple of 2 (x2 MFMA)... rows 258, 259 are zeros
constexpr int IS_FR = 64;          // frames per workgroup
constexpr int IS_HOPS = IS_FR - 3; // output hops per workgroup
constexpr int IS_BP = 261;         // LDS row pitch of the spectrogram tile (floats, odd)
constexpr int IS_ZP = 257;         // LDS row pitch of the time-segment tile

__global__ __launch_bounds__(256) void istft_k(const float2* spec, int T, const float* itw /*[IS_K][256]*/, short* out_i16,
                                               float* out_f32) {
  // itw: [IS_K][256] inverse twiddles, then [256] squared window values (float64-built)
  extern __shared__ __align__(16) float is_smem[];
  float* s_b = is_smem;                                     // [IS_FR][IS_BP]: (re f = 0..128 | im f = 0..128 | 0 0)
  float* s_z = is_smem;                                     // [IS_FR][IS_ZP] (re-uses s_b after the product)
  const int n = blockIdx.y, j0 = blockIdx.x * IS_HOPS;      // output hops j0 .. j0 + 60; frames j0 - 1 .. j0 + 62
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const float2* sp = spec + (long long)n * T * 129;
  for (int i = tid; i < IS_FR * 129; i += 256) {
    const int fr = i / 129, f = i - fr * 129;
    const int t = j0 - 1 + fr;
    float2 v = make_float2(0.f, 0.f);
    if (t >= 0 && t < T) v = sp[(long long)t * 129 + f];
    s_b[fr * IS_BP + f] = v.x;
    s_b[fr * IS_BP + 129 + f] = v.y;
  }
  if (tid < IS_FR) { s_b[tid * IS_BP + 258] = 0.f; s_b[tid * IS_BP + 259] = 0.f; }
  __syncthreads();
  // wave w owns sample rows 64 w .. 64 w + 63 (two 32-row tiles) x two 32-frame column tiles
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;
  for (int k = 0; k < IS_K; k += 2) {
    const int kk = k + half;
    float bv[2], av[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) bv[q] = s_b[(q * 32 + l31) * IS_BP + kk];
#pragma unroll
    for (int i = 0; i < 2; ++i) av[i] = itw[kk * 256 + (2 * wave + i) * 32 + l31];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[q], acc[i][q], 0, 0, 0);
  }
  __syncthreads();                                          // every wave is done reading s_b
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (2 * wave + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;       // sample k inside the frame
        s_z[(q * 32 + l31) * IS_ZP + row] = acc[i][q][r];                              // already x window / 256
      }
  __syncthreads();
  // overlap-add: output sample 64 j + m  <-  frames j - 1 (k = 192 + m), j (128 + m), j + 1 (64 + m), j + 2 (m)
  const long long Ls = 64LL * (T - 1);
  for (int i = tid; i < IS_HOPS * 64; i += 256) {
    const int jl = i >> 6, m = i & 63;
    const int j = j0 + jl;
    if (j >= T - 1) continue;
    float sum = 0.f, env = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = j - 1 + q, k = 192 - 64 * q + m;
      if (t >= 0 && t < T) {
        sum += s_z[(jl + q) * IS_ZP + k];
        env += itw[IS_K * 256 + k];                              // w[k]^2 (the envelope: 1.5 inside the signal)
      }
    }
    const float y = sum / env;
    const long long o = (long long)n * Ls + 64LL * j + m;
    if (out_f32) out_f32[o] = y;
    if (out_i16) out_i16[o] = (short)(int)(y * 32767.0f);                              // C-style truncation, as astype(int16)
  }
}

// inverse twiddle table [IS_K][256] (host, float64 math): row kk < 129: w[k] c_f / 256 cos, 129 <= kk < 258: -w[k] c_f / 256 sin
void istft_build_twiddles(float* tw) {
  const double pi = 3.14159265358979323846;
  for (int kk = 0; kk < IS_K; ++kk)
    for (int k = 0; k < 256; ++k) {
      const double w = 0.5 - 0.5 * cos(2.0 * pi * k / 256.0);
      double v = 0.0;
      if (kk < 129) {
        const int f = kk;
        v = w * ((f == 0 || f == 128) ? 1.0 : 2.0) / 256.0 * cos(2.0 * pi * (double)((f * k) % 256) / 256.0);
      } else if (kk < 258) {
        const int f = kk - 129;
        v = -w * ((f == 0 || f == 128) ? 1.0 : 2.0) / 256.0 * sin(2.0 * pi * (double)((f * k) % 256) / 256.0);
      }
      tw[kk * 256 + k] = (float)v;
    }
  for (int k = 0; k < 256; ++k) {
    const double w = 0.5 - 0.5 * cos(2.0 * pi * k / 256.0);
    tw[IS_K * 256 + k] = (float)(w * w);
  }
}
int istft_twiddle_count() { return IS_K * 256 + 256; }

static size_t istft_lds_bytes() {
  const size_t a = (size_t)IS_FR * IS_BP, b = (size_t)IS_FR * IS_ZP;
  return (a > b ? a : b) * sizeof(float);
}

hipError_t istft_init() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&istft_k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
}

hipError_t launch_istft(const void* spec, int N, int T, const float* itw, short* out_i16, float* out_f32, hipStream_t s) {
  if (T < 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(istft_k, dim3((T - 1 + IS_HOPS - 1) / IS_HOPS, N), dim3(256), istft_lds_bytes(), s,
                     reinterpret_cast<const float2*>(spec), T, itw, out_i16, out_f32);
  return hipGetLastError();
}

}  // namespace mn
