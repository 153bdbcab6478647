// Internal declarations shared by the HIP translation units of libmisonet_hip.so (gfx950 only).
//
// Activation layout in HBM ("planar"): float32 [n][c][f][Tp], frames (t) innermost, Tp = T rounded up to 32 so
// every row starts 128-byte aligned.  Conv outputs are stored RAW (bias + ELU applied, instance norm NOT applied);
// each producer accumulates per-(n,c) sum / sum-of-squares exactly (det_stats.hpp) next to the buffer and every consumer
// normalises while it stages its input tile into LDS ("normalise on load").  Dense-block concatenation is free:
// a block's tensors are channel slices of one buffer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "det_stats.hpp"

namespace mn {

constexpr int CK = 8;       // input channels per K-chunk of the implicit GEMM
constexpr int TT = 128;     // output frames per workgroup (4 MFMA column tiles of 32)
constexpr int TW = 136;     // staged frames per input row: [t0-4, t0+132) -> 34 aligned float4
constexpr int FT = 4;       // output rows (frequency bins) per workgroup = waves per workgroup
constexpr float IN_EPS = 1e-5f;    // nn.InstanceNorm{1,2}d default eps (reference model.py:413,579)
constexpr float GLN_EPS = 1e-8f;   // reference model.py:6

// Experiment switches (kernel variants for A/B runs, timelines, parts switched off): they exist only in the experiment build
// (`make exp` -> libmisonet_hip_exp.so, -DMISONET_EXPERIMENTS; tools/gpu_*.sh select it through MISONET_LIB_PATH).  In the
// product library every switch is its default, a compile-time constant: no environment variable changes what it runs.
#ifdef MISONET_EXPERIMENTS
inline int exp_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
constexpr int exp_env(const char*, int dflt) { return dflt; }
#endif

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int frames_pitch(int T) { return round_up(T, 32); }

// ---- 3x3 convolution family (reference model.py:401-482), one launch per layer ------------------------------
struct ConvArgs {
  const float* in;          // planar buffer base
  const dstat_t* in_stats;  // [n][in_sstride][2][DS_NL] (sum, sumsq as det_stats.hpp limbs) or nullptr when every input channel is identity
  float* out;
  dstat_t* out_stats;       // [n][out_sstride][2][DS_NL], accumulated with integer atomics (ds_add) when act != 0
  const float* w;           // packed [ncg][nchunk][9][CK][COP]
  const float* bias;        // [ncg*COP]
  const unsigned short* w16; // bf16x3 path: packed [ncg][nchunk16][hi|lo][9][2][COP][8] bf16, or nullptr
  const void* ww6;          // bf16x6w path (conv_wino6.hip): the same weights as three bf16 pieces, [cg32][K-step of 16][xi][nu][piece][lane][8], or nullptr
  const float* wsm;         // conv_few.hip (<= 4 output channels): [Cin][9 = kt * 3 + kf][4 co] conv-form taps, or nullptr
  const float* ww16;        // f32w path: the LAST 16 output channels of a layer with Cout % 32 == 16 as their own Winograd image for the 16-row body
                            // (conv_wino.hip G16): [chunk of 8][K-step of 4 ci][pos / 4][ci % 4][16 co][pos % 4], or nullptr
  const float* w1d;         // f32w path, frequency-strided layers (conv.hip W1D): 1-D Winograd weights along T, [cg32][chunk of 8][nu * 3 + kf][ci][32 co], or nullptr
  const float* ww;          // f32w path (conv_wino.hip): Winograd-domain weights U = G g G^T, [cg32][chunk of 8][pos / 4][ci][32 co][pos % 4], or nullptr
  long long in_bstride;     // floats per sample of the input buffer
  long long out_bstride;
  int in_sstride, out_sstride;   // channels per sample in the stats arrays (= channels of the whole buffer)
  int in_c0, Cin, Fin;
  int ident_c;              // input channels [0, ident_c) are consumed as they are (no instance norm)
  int out_c0, Cout, Fout;
  int T, Tp;
  int sf;                   // frequency stride of a forward conv (1 or 2)
  int padf;                 // frequency zero padding of the conv form (0, 1, or 2 for a stride-1 transposed conv)
  int tr2;                  // 1: stride-2 transposed conv (fin = (f + kf - 2) / 2 when even)
  int act;                  // 1: ELU + statistics for the following instance norm; 0: raw output
  int NR;                   // staged input rows per workgroup
  int ncg;                  // output-channel groups (grid.z = n_samples * ncg)
  int cop;                  // 32 or 64 output channels per group
  // "oct" activation layout of the bf16x3 DMA dataflow (conv_bf16_dma.hip): per sample [hi | lo] halves, each
  // [c/8][f][Tp][8] bf16 (8 channels of one frame = one 16-byte unit), values RAW (bias + ELU, no instance norm).
  int in_oct, out_oct;      // layout of the input / output buffer: 0 planar float32, 1 oct (bf16x3: hi | lo halves),
                            // 3 oct3 (bf16x6, conv_bf16x6.hip: hi | mid | lo parts, 6 bytes per element),
                            // 4 oct with fp16 pieces (f16x3)
  float wscale, descale;    // f16x3: power-of-two scale 2^k carried by the folded weights and its inverse (1, 1 otherwise)
  const void* wps;          // per-sample weights with the instance norm of the input folded in (conv_wprep), LDS image order
  long long wps_nstride;    // bytes between samples (0: one image shared by all samples)
  const float* btab;        // [n][ncg*32][9] border-aware shift table: sum_ci W[co][ci][tap] * shift[ci], or nullptr
  long long btab_nstride;   // floats between samples
  int xcd;                  // 1: 1-D grid with the XCD-aware tile order of conv_tile() (ntx, nty, nsamp valid)
  int ntx, nty, nsamp;      // frame tiles, row tiles, samples of this launch
  unsigned long long* dbg_buf;   // timeline stamps of one workgroup (MISONET_TIMELINE=1, experiments only)
  int dbg;                  // timing experiments only (MISONET_WS_DEBUG bits): 1 skip MFMAs, 4 skip epilogue, 8 skip stores,
                            // 16 skip statistics reductions, 32 no deferred epilogue
};
// Workgroup -> tile.  Hardware hands consecutive workgroup ids to the 8 XCDs round-robin and every XCD has its own
// L2, so with the natural (t, f, n) order the 8 frame tiles of a row sit on 8 different XCDs and every halo line is
// fetched twice.  XCD order: id & 7 picks the XCD, id >> 3 walks that XCD's own samples (n % 8 == xcd) tile by
// tile (frame tile fastest, then output-channel group, then row tile): all tiles that share halos or re-read the same
// input with another channel group are co-resident on ONE XCD.
struct ConvTile { int t_tile, f_tile, n, cg; bool valid; };
__device__ __forceinline__ ConvTile conv_tile(const ConvArgs& a) {
  ConvTile r;
  if (a.xcd) {
    const unsigned id = blockIdx.x;
    const unsigned xcd = id & 7u, k = id >> 3;
    const unsigned per = (unsigned)(a.ntx * a.nty * a.ncg);
    const unsigned grp = k / per;
    unsigned tile = k - grp * per;
    r.n = (int)(grp * 8u + xcd);
    r.t_tile = (int)(tile % (unsigned)a.ntx);
    tile /= (unsigned)a.ntx;
    r.cg = (int)(tile % (unsigned)a.ncg);
    r.f_tile = (int)(tile / (unsigned)a.ncg);
    r.valid = r.n < a.nsamp;
  } else {
    r.t_tile = blockIdx.x; r.f_tile = blockIdx.y;
    r.n = blockIdx.z / a.ncg; r.cg = blockIdx.z - r.n * a.ncg;
    r.valid = true;
  }
  return r;
}
// fills xcd/ntx/nty/nsamp and returns the launch grid
inline dim3 conv_grid(ConvArgs& a, int n_samples, int tt, int ft, int xcd) {
  a.ntx = (a.T + tt - 1) / tt; a.nty = (a.Fout + ft - 1) / ft; a.nsamp = n_samples; a.xcd = xcd;
  if (xcd) return dim3((unsigned)(8 * ((n_samples + 7) / 8) * a.ntx * a.nty * a.ncg), 1, 1);
  return dim3(a.ntx, a.nty, n_samples * a.ncg);
}
int conv_xcd_env();                          // MISONET_XCD (default 1)
int device_cus();                            // compute units of the CURRENT device (cached per device, thread-safe); <= 0 on error
int conv_cop(int Cout);                      // 32 (Cout <= 32) or 64
int conv_rows(int sf, int tr2);              // NR for the mode
hipError_t launch_conv(const ConvArgs& a, int n_samples, hipStream_t s);
hipError_t conv_init();                      // dynamic-LDS attributes
// Winograd F(2x2, 3x3) on the fp32 matrix cores for the stride-1 same-padded layers (conv_wino.hip; precision mode "f32w")
bool conv_wino_ok(const ConvArgs& a);
hipError_t launch_conv_wino(const ConvArgs& a, int n_samples, hipStream_t s);
hipError_t conv_wino_init();
// <= 4 output channels, stride 1, no activation (the network's last layer) on the vector ALU (conv_few.hip; needs a.wsm)
bool conv_few_ok(const ConvArgs& a);
hipError_t launch_conv_few(const ConvArgs& a, int n_samples, hipStream_t s);
hipError_t conv_few_init();
bool conv_wino6_ok(const ConvArgs& a);
hipError_t launch_conv_wino6(const ConvArgs& a, int n_samples, hipStream_t s);
hipError_t conv_wino6_init();
hipError_t launch_conv_bf16(const ConvArgs& a, int n_samples, hipStream_t s);   // conv_bf16.hip (needs a.w16)
hipError_t conv_bf16_init();
hipError_t launch_conv_bf16_dma(const ConvArgs& a, int n_samples, hipStream_t s);   // conv_bf16_dma.hip (oct input)
hipError_t launch_conv_wprep(const ConvArgs& a, const float* wf, int n_samples, hipStream_t s);   // a.in_oct == 4: fp16 pieces
hipError_t conv_bf16_dma_init();
// bf16x6 (fp32-faithful) DMA dataflow, conv_bf16x6.hip: oct3 input, oct3 or planar output
hipError_t launch_conv_bf16x6(const ConvArgs& a, int n_samples, hipStream_t s);
hipError_t launch_conv_wprep6(const ConvArgs& a, const float* wf6, int n_samples, hipStream_t s);
hipError_t conv_bf16x6_init();
long long conv_bf16x6_wps_bytes(int Cin, int Cout);   // per-sample folded-weight bytes of one layer
// the network's first layer in the bf16x6 arithmetic: planar float32 in (consumed as it is), oct3 / planar out; wimg = the
// layer's weights as ONE 3-part image per 8-channel chunk (864 16-byte units each, conv_wprep6_k's order; net.hip packs it)
hipError_t launch_conv_x6_first(const ConvArgs& a, const void* wimg, int n_samples, hipStream_t s);

// ---- TCN (reference model.py:486-632) -----------------------------------------------------------------------------
// Statistics travel as float64 (sum, sum of squares) PARTIALS, one per producing workgroup, added by the consumer in index
// order (no atomics, bit-reproducible): IN1d [n][C][tcn_part_slots(T)] (one per 128-frame tile; tcn_prepare writes slot 0
// only), gLN [n][C / 4] (one per 4-channel group).
int tcn_part_slots(int T);
// x0 = IN2d(raw) materialised as the residual stream + its per-(n,c) statistics
hipError_t launch_tcn_prepare(const float* raw, long long raw_bstride, int raw_c0, const dstat_t* raw_stats, int raw_sstride,
                              float* x, double2* x_part, int C, int T, int Tp, int n_samples, hipStream_t s,
                              int raw_oct3 = 0);   // raw_oct3: source in the bf16x6 oct3 layout
// d = PReLU(dwconv_dilated(ELU(norm(x)))) ; gLN partials of d.  x_np: partials per row of x_part (1 or tcn_part_slots(T)).
// norm = the TemporalBlock's outer norm (model.py:530,535; chose_norm model.py:570-581), norm_kind: 0 InstanceNorm1d;
// 1 gLN over (C, T) of a sample with (nsc, nsh) = (gamma, beta) per channel; 2 cLN over the channels of every frame with
// (gamma, beta) and fstat [n][Tp] (mean, rstd) from launch_tcn_cln_stats; 3 BatchNorm1d in eval mode with (nsc, nsh) = the
// folded (weight / sqrt(running_var + eps), bias - running_mean * that) per channel.
hipError_t launch_tcn_dw(const float* x, const double2* x_part, int x_np, const float* wdw /*[C][3]*/, const float* prelu /*[1]*/,
                         float* d, double2* gln_part /*[n][C/4]*/, int C, int T, int Tp, int dilation, int n_samples, hipStream_t s,
                         int norm_kind = 0, const float* nsc = nullptr, const float* nsh = nullptr, const float2* fstat = nullptr);
// per-frame mean / rstd over the C channels of x [n][C][Tp] (ChannelwiseLayerNorm, model.py:583-606): fstat [n][Tp]
hipError_t launch_tcn_cln_stats(const float* x, float2* fstat, int C, int T, int Tp, int n_samples, hipStream_t s);
// y = pwconv(gLN(d)) (+ residual) ; IN partials of y
hipError_t launch_tcn_pw(const float* d, const double2* gln_part, const float* gamma, const float* beta,
                         const float* wpw /*packed [C/CK... see tcn.hip]*/, const float* residual /*or nullptr*/,
                         float* y, long long y_bstride, int y_c0, double2* y_part /*[n][C][slots]*/, int C, int T, int Tp,
                         int n_samples, hipStream_t s,
                         int y_oct3_cbuf = 0, int x6 = 0);   // != 0: y is an oct3 buffer with that many channels (bf16x6 mode)

// ---- layout conversion ----------------------------------------------------------------------------------------
// complex64 [B][Mseg][T][F] -> planar real/imag channel planes; optional circular mic shifts (tester.py:1034,1050):
// destination sample n = b*nshift + k receives source channel (m + k) % Mseg at destination channel m.
hipError_t launch_pack(const float2* src, int B, int Mseg, int T, int F, float* dst, long long dst_bstride, int Tp,
                       int c_re, int c_im, int nshift, hipStream_t s);
// planar [n][2S][F][Tp] -> complex64 [n][S][T][F]; sets *nan_flag when a NaN is seen (model.py:109-110)
hipError_t launch_unpack(const float* src, long long src_bstride, int Tp, int S, int T, int F, float2* dst, int n_samples,
                         int* nan_flag, hipStream_t s);
// planar view (+ optional instance norm) -> float32 [n][C][T][F]  (diagnostic taps)
hipError_t launch_export(const float* src, long long src_bstride, int c0, int C, int Fq, int T, int Tp,
                         const dstat_t* stats, int sstride, int ident_c, float* dst, int n_samples, hipStream_t s,
                         int oct = 0);   // oct: bf16 parts of a source in the oct layout (0 planar, 2, 3; sstride = channels
                                         // of the whole buffer)

// ---- MVDR + PIT -------------------------------------------------------------------------------------------------
// Accessor for a multichannel complex STFT with frames contiguous: element (b, f, m, t) =
//   re[b*sb + f*sf + m*sm + t*st], im likewise.  Interleaved complex64 [B,F,M,T]: re=base, im=base+1, st=2.
struct CView {
  const float* re; const float* im;
  long long sb, sf, sm; int st;
};
struct MvdrArgs {
  CView mix;              // observation Y
  const float* est;       // planar MISO1 output buffer [B*M][2S][F][Tp] (pipeline mode) or nullptr
  long long est_bstride;  // floats per sample
  const int* sel;         // [B][M][S]: estimated-speaker index to use for (b, mic m, aligned speaker j), or nullptr
  CView src;              // source estimate S when est == nullptr (drop-in mode)
  int S;                  // speakers handled per utterance (grid.z)
  int B, F, M, T, Tp;
  float epsi;
};
long long mvdr_ws_bytes(int B, int S, int F, int M);
// out: complex, element (b, spk, t, f) at out_re[b*ob + spk*os + t*ot + f*of]
struct COut { float* re; float* im; long long ob, os, ot, of; };
hipError_t launch_mvdr(const MvdrArgs& a, const COut& out, void* ws, hipStream_t s);
hipError_t launch_mvdr_debug(const void* ws, int B, int S, int F, int M, double* steer, double* w, hipStream_t s);

// dist[b][i][j] = sum_{t,f} | |A_i| - |B_j| | for S = 1..4 speakers: per-bin float64 partials, added in bin order (bit-
// reproducible); then sel = the cheapest of the S! permutations (itertools order, first minimum).
struct PitArgs {
  CView a, b;             // sm = speaker stride here; (b, f, spk, t) addressing
  int B, F, T;
};
// K candidates per anchor: grid row bk = b*K + k uses anchor b and candidate bk
hipError_t launch_pit_dist_k(const PitArgs& p, int S, int K, double* part /*[B*K][F][S][S]*/, hipStream_t s);
hipError_t launch_pit_pick(const double* part, int F, int S, int n, double* dist /*[n][S][S]*/, int* sel /*[n][S]*/, hipStream_t s);

}  // namespace mn
