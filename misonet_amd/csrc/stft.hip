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

}  // namespace mn
