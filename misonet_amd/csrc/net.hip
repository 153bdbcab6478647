// Host runtime of libmisonet_hip.so: layer plan of a MISO trunk (reference model.py:8-111 / 282-395), weight
// repacking into the kernels' layouts, workspace layout, forward scheduling, and the C ABI of include/misonet.h.
#include "kernels.hpp"
#include "../../include/misonet.h"

#include <math.h>
#include <cmath>
#include <algorithm>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

namespace mn {
hipError_t launch_unpack_ex(const float* src, long long src_bstride, int Tp, int S, int T, int F, int c_re0, int c_im0, int mode,
                            int M, const int* sel, float2* dst, int n_out, int* nan_flag, hipStream_t s);
hipError_t launch_compose_sel(const int* shift_sel, const int* clean_sel, int B, int M, int S, int* out, hipStream_t s);
hipError_t launch_assemble3(const float* in1, long long in1_bstride, const float* out1, long long out1_bstride,
                            const int* sel, int B, int M, int S, int ref_ch, int F, int Tp, float* in3,
                            long long in3_bstride, hipStream_t s);
hipError_t launch_stft_pack(const float* wav, int B, int L, int Mw, int T, const float* twid, float* dst,
                            long long dst_bstride, int Tp, int F, int c_re, int c_im, int nshift, hipStream_t s);
hipError_t stft_init();
hipError_t launch_istft(const void* spec, int N, int T, const float* itw, short* out_i16, float* out_f32, hipStream_t s);
hipError_t istft_init();
void istft_build_twiddles(float* tw);
int istft_twiddle_count();
void stft_build_twiddles(float* tw);
int stft_twiddle_count();
}  // namespace mn

using namespace mn;

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIPCHK(expr)                                                                                    \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return fail(MISONET_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// ---- optional per-launch timing with HIP events on the caller's stream (bench.py roofline leg) ------------------
enum { PK_CONV = 0, PK_TCN, PK_MVDR, PK_OTHER, PK_N };
struct ProfRec { int kind; hipEvent_t e0, e1; };
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  std::vector<ProfRec> recs;
  bool overflow = false;
};
// one state per device (events belong to the device that was current when they were created); a process that drives
// several GPUs profiles each of them independently
constexpr int MAX_DEV = 64;
static Prof g_profs[MAX_DEV];
static std::atomic<int> g_prof_any{0};          // fast path: no hipGetDevice per launch while nobody profiles
static int cur_dev() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEV) d = 0;
  return d;
}
struct ProfScope {
  hipStream_t s; int kind; hipEvent_t e0 = nullptr, e1 = nullptr; bool active = false; Prof* pr = nullptr;
  ProfScope(hipStream_t s_, int kind_) : s(s_), kind(kind_) {
    if (!g_prof_any.load(std::memory_order_relaxed)) return;
    pr = &g_profs[cur_dev()];
    if (!pr->on) return;
    if (pr->used + 2 > pr->pool.size()) { pr->overflow = true; return; }
    e0 = pr->pool[pr->used++];
    e1 = pr->pool[pr->used++];
    active = (hipEventRecord(e0, s) == hipSuccess);
  }
  ~ProfScope() {
    if (!active) return;
    if (hipEventRecord(e1, s) == hipSuccess) pr->recs.push_back({kind, e0, e1});
  }
};

static int frontend_init();   // STFT / iSTFT tables of the current device (below, with the front-end entry points)

// ---------------------------------------------------------------------------------------------------------------
struct Tensor {
  std::string name;
  long long numel;
  std::vector<float> host;
  bool set = false;
};

enum { B_IN = 0, B_E0, B_E1, B_E2, B_E3, B_E4, B_D0, B_D1, B_D2, B_D3, B_D4, B_D5, B_D6, B_X2, B_X3, B_X4, B_X5, B_X6,
       B_OUT, B_TXA, B_TXB, B_TD, B_TP, NBUF };

struct BufSpec { int C = 0, F = 0; };

struct ConvL {
  int in_buf, in_c0, Cin, ident_c;
  int out_buf, out_c0, Cout;
  int sf, padf, tr2, act;
  bool transposed;
  int wt, bt;                   // tensor indices
  int cop, ncg;
  long long w_off = 0, b_off = 0;   // offsets (floats) into the device weight arena
  long long w16_off = 0;            // offset (floats) of the bf16 hi/lo packed weights
  long long wf_off = 0;             // offset (floats) of the float32 weights in bf16-image order (conv_wprep_k source)
  long long wf6_off = 0;            // offset (floats) of the float32 weights [cg][chunk of 8][tap][32][8] (conv_wprep6_k source)
  long long w6s_off = -1;           // first layer only: offset (floats) of the SHARED 3-part bf16 image [chunk][864 units]
  long long ww_off = -1;            // stride-1 same-padded layers: offset (floats) of the Winograd-domain weights (conv_wino.hip)
  long long ww6_off = -1;           // the same in three bf16 pieces (conv_wino6.hip)
  long long ww16_off = -1;          // Cout % 32 == 16: Winograd image of the last 16 channels for the 16-row body (conv_wino.hip G16)
  long long w1d_off = -1;           // stride-2 convs / transposed convs: 1-D Winograd image along T (conv.hip W1D; f32w mode)
  long long wsm_off = -1;           // <= 4 output channels, no activation (the last layer): [Cin][9][4] image of conv_few.hip
  float wscale = 1.f;               // f16x3: power of two that brings max |W| of the layer to [32, 64)
};

struct TcnHalf {
  int dw, prelu, gamma, beta, pw; long long o_dw, o_prelu, o_gamma, o_beta, o_pw;
  // the OUTER norm in front of this half (model.py:530,535; cfg.tcn_norm): tensors (gLN / cLN: gamma, beta; BatchNorm1d:
  // weight, bias, running_mean, running_var) and the per-channel (scale, shift) pair the kernels read
  int on[4] = {-1, -1, -1, -1};
  long long o_nsc = 0, o_nsh = 0;
};
struct TcnBlock { int dilation; TcnHalf h[2]; };

struct Tap { std::string name; int buf, c0, C; bool normalised; };

struct Layout {
  int N, T, Tp;
  long long data_off[NBUF];      // floats: activation buffers (b < B_TXA): offset INSIDE a sample's block of the arena (sample n
                                 // at + n * sample_stride); TCN buffers: offset of the whole [N][128][Tp] block
  long long sample_stride;       // floats per sample of the activation arena
  long long in_ext_off = -1;     // >= 0: the network input lives OUTSIDE this workspace, at ws + in_ext_off bytes, with
  long long in_ext_bstride = 0;  // in_ext_bstride floats between samples (the pipeline's MISO3 input, see pipe_layout)
  long long stats_off[NBUF];     // 8-byte words (dstat_t): [N][C][2][DS_NL] per buffer
  long long tcn_xs, tcn_ps, tcn_gln;   // words (2 per double2 partial): [15][N*128*slots], [14][N*128*slots], [28][N*32]
  long long stats_doubles;       // words in all
  long long data_base;           // bytes from ws start to the float arena
  long long wps_base, wps_nstride;   // bytes: per-sample folded weights of the layer in flight (DMA dataflow)
  long long btab_base, btab_nstride; // bytes / floats: per-sample border-aware shift table
  long long fstat_base;              // bytes: [N][Tp] float2 per-frame (mean, rstd) of the cLN outer norm (cfg.tcn_norm == 2)
  long long total_bytes;
};

struct misonet_net {
  misonet_cfg cfg;
  int S;                         // speakers out = out_ch / 2
  BufSpec bufs[NBUF];
  std::vector<Tensor> tensors;
  std::vector<ConvL> enc, dec;
  std::vector<TcnBlock> tcn;
  std::vector<Tap> taps;
  float* w_dev = nullptr;
  bool committed = false;
  bool keep_taps = false;        // true: no buffer shares memory with another (every tap stays readable after a forward)
  int precision = 3;             // 0: exact f32 MFMA, 1: bf16x3 planar, 2: bf16x3 DMA dataflow, 3: bf16x6 DMA dataflow (the
                                 // default: fp32-faithful, what bench.py reports), 4: f16x3 DMA dataflow, 5: f32 MFMA with the
                                 // dense-block convs in Winograd F(2x2, 3x3) form ("f32w": planar float32 layout like mode 0)
};
// Product arithmetic modes: 0 "f32", 3 "bf16x6", 5 "f32w".  The measured alternatives that earn nothing (1 / 2 "bf16x3": 16-bit
// operands, 4 "f16x3": 22-bit operands, 6 "bf16x6w": correct but slower than mode 3 -- DESIGN 3.4) exist only in the experiment
// build (`make exp`): their kernels (conv_bf16.hip, conv_bf16_dma.hip, conv_wino6.hip), weight images and dispatch are compiled
// out of the product library, which answers MISONET_EINVAL to them.
#ifdef MISONET_EXPERIMENTS
#define MN_ALT_MODES 1
#else
#define MN_ALT_MODES 0
#endif
// the planar-float32 modes: every activation buffer is float32 [c][f][Tp], instance norm applied while staging
static inline bool planar_f32(const misonet_net* n) { return n->precision == 0 || n->precision == 5 || n->precision == 6; }

static int find_tensor(const misonet_net* n, const std::string& name) {
  for (size_t i = 0; i < n->tensors.size(); ++i)
    if (n->tensors[i].name == name) return (int)i;
  return -1;
}
static int add_tensor(misonet_net* n, const std::string& name, long long numel) {
  Tensor t;
  t.name = name;
  t.numel = numel;
  n->tensors.push_back(t);
  return (int)n->tensors.size() - 1;
}

static ConvL make_conv(misonet_net* n, const std::string& prefix, int in_buf, int in_c0, int Cin, int ident_c, int out_buf,
                       int out_c0, int Cout, int sf, int padf, bool transposed, int act) {
  ConvL L;
  L.in_buf = in_buf; L.in_c0 = in_c0; L.Cin = Cin; L.ident_c = ident_c;
  L.out_buf = out_buf; L.out_c0 = out_c0; L.Cout = Cout;
  L.transposed = transposed;
  L.act = act;
  if (transposed) {
    L.tr2 = (sf == 2);
    L.sf = 1;
    L.padf = 2;          // stride-1 transposed conv == conv with flipped taps and full padding
  } else {
    L.tr2 = 0; L.sf = sf; L.padf = padf;
  }
  L.wt = add_tensor(n, prefix + ".weight", (long long)Cin * Cout * 9);
  L.bt = add_tensor(n, prefix + ".bias", Cout);
  L.cop = L.tr2 ? 32 : conv_cop(Cout);       // (conv3x3_mfma runs two output rows per wave on the stride-2 transposed layers)
  L.ncg = (Cout + L.cop - 1) / L.cop;
  return L;
}

// DenseBlock(init_ch, g1, g2) on buffer `buf` ([x | y0..y3]); y4 -> (out_buf, out_c0)   (model.py:437-482)
static void add_dense(misonet_net* n, std::vector<ConvL>& v, const std::string& prefix, int buf, int init_ch, int g1, int g2,
                      int ident_c, int out_buf, int out_c0) {
  for (int i = 0; i < 5; ++i) {
    const int cin = init_ch + i * g1;
    char nm[64];
    snprintf(nm, sizeof(nm), ".conv%d.0", i + 1);
    if (i < 4)
      v.push_back(make_conv(n, prefix + nm, buf, 0, cin, ident_c, buf, cin, g1, 1, 1, false, 1));
    else
      v.push_back(make_conv(n, prefix + nm, buf, 0, cin, ident_c, out_buf, out_c0, g2, 1, 1, false, 1));
  }
}

static int freq_after_conv(int F, int sf) { return (F - 3) / sf + 1; }

static int build_plan(misonet_net* n) {
  const misonet_cfg& c = n->cfg;
  if (c.n_freq != 129) return fail(MISONET_EINVAL, "n_freq must be 129 (got %d): the encoder must reduce F to one bin", c.n_freq);
  if (c.tcn_norm < 0 || c.tcn_norm > 3)
    return fail(MISONET_EINVAL, "tcn_norm must be 0 (IN), 1 (gLN), 2 (cLN) or 3 (BatchNorm1d, eval) (got %d)", c.tcn_norm);
  if (c.in_ch < 2 || c.in_ch % 2 || c.out_ch < 2 || c.out_ch % 2 || c.in_ch > 64 || c.out_ch > 32)
    return fail(MISONET_EINVAL, "in_ch/out_ch must be even and small (got %d/%d)", c.in_ch, c.out_ch);
  for (int i = 0; i < 7; ++i) {
    if (c.en_ch[i] <= 0 || c.en_ch[i] % 8 || c.de_ch[i] <= 0 || c.de_ch[i] % 8)
      return fail(MISONET_EINVAL, "bottleneck channels must be positive multiples of 8");
    if (c.en_ch[6 - i] != c.de_ch[i])
      return fail(MISONET_EINVAL, "U-Net skip concat needs en_ch[6-i] == de_ch[i] (model.py:35,99)");
  }
  if (c.en_ch[6] != 128) return fail(MISONET_EINVAL, "the TCN is TemporalConvNet(2,7,128,128,128) (model.py:31): en_ch[6] must be 128");
  for (int i = 0; i < 7; ++i)
    if (c.en_ch[i] > 128) return fail(MISONET_EINVAL, "channel counts above 128 are not supported");
  n->S = c.out_ch / 2;
  const int* en = c.en_ch;
  const int* de = c.de_ch;
  // frequency sizes per encoder level (model.py:40-54): F_e[b] = output bins of encoder b
  int Fe[7];
  Fe[0] = freq_after_conv(c.n_freq, 1);
  for (int b = 1; b < 7; ++b) Fe[b] = freq_after_conv(Fe[b - 1], b == 6 ? 1 : 2);
  if (Fe[6] != 1) return fail(MISONET_EINVAL, "encoder does not reduce to one frequency bin");
  // buffers
  n->bufs[B_IN] = {c.in_ch, c.n_freq};
  for (int b = 0; b < 5; ++b) n->bufs[B_E0 + b] = {5 * en[b], Fe[b]};
  n->bufs[B_D0] = {2 * de[0], Fe[6]};
  n->bufs[B_D1] = {2 * de[1], Fe[5]};
  for (int i = 2; i < 7; ++i) {
    n->bufs[B_D0 + i] = {6 * de[i], Fe[6 - i]};
    n->bufs[B_X2 + (i - 2)] = {2 * de[i], Fe[6 - i]};
  }
  n->bufs[B_OUT] = {c.out_ch, c.n_freq};
  for (int b = B_TXA; b <= B_TP; ++b) n->bufs[b] = {128, 1};

  // ---- encoders (model.py:40-54); xs[b] is written straight into decoder buffer D[6-b] at channel de[6-b] ----
  char nm[96];
  for (int b = 0; b < 7; ++b) {
    const int skip_buf = B_D0 + (6 - b), skip_c0 = de[6 - b];
    const int src_buf = (b == 0) ? B_IN : B_D0 + (7 - b);
    const int src_c0 = (b == 0) ? 0 : de[7 - b];
    const int cin = (b == 0) ? c.in_ch : en[b - 1];
    if (b == 0) {
      n->enc.push_back(make_conv(n, "encoders.0.0.conv2d", B_IN, 0, cin, cin, B_E0, 0, en[0], 1, 0, false, 0));
      add_dense(n, n->enc, "encoders.0.1", B_E0, en[0], en[0], en[0], en[0], skip_buf, skip_c0);
    } else if (b < 5) {
      snprintf(nm, sizeof(nm), "encoders.%d.0.net.0", b);
      n->enc.push_back(make_conv(n, nm, src_buf, src_c0, cin, 0, B_E0 + b, 0, en[b], 2, 0, false, 1));
      snprintf(nm, sizeof(nm), "encoders.%d.1", b);
      add_dense(n, n->enc, nm, B_E0 + b, en[b], en[b], en[b], 0, skip_buf, skip_c0);
    } else {
      snprintf(nm, sizeof(nm), "encoders.%d.0.net.0", b);
      n->enc.push_back(make_conv(n, nm, src_buf, src_c0, cin, 0, skip_buf, skip_c0, en[b], b == 6 ? 1 : 2, 0, false, 1));
    }
  }
  // ---- decoders (model.py:56-73) ----
  for (int i = 0; i < 7; ++i) {
    const int buf = B_D0 + i;
    const int cin = 2 * de[i];
    const int cout = (i == 6) ? c.out_ch : de[i + 1];
    const int obuf = (i == 6) ? B_OUT : B_D0 + i + 1;
    if (i >= 2) {
      snprintf(nm, sizeof(nm), "decoders.%d.0", i);
      add_dense(n, n->dec, nm, buf, cin, cin / 2, cin, 0, B_X2 + (i - 2), 0);
      snprintf(nm, sizeof(nm), i == 6 ? "decoders.%d.1.deconv2d" : "decoders.%d.1.net.0", i);
      n->dec.push_back(make_conv(n, nm, B_X2 + (i - 2), 0, cin, 0, obuf, 0, cout, i == 6 ? 1 : 2, 0, true, i == 6 ? 0 : 1));
    } else {
      snprintf(nm, sizeof(nm), "decoders.%d.0.net.0", i);
      // decoder 0 consumes [TCN output (raw, identity) | xs[6] (instance-normalised)]
      n->dec.push_back(make_conv(n, nm, buf, 0, cin, i == 0 ? de[0] : 0, obuf, 0, cout, i == 0 ? 1 : 2, 0, true, 1));
    }
  }
  // ---- TCN (model.py:486-567) ----
  for (int r = 0; r < 2; ++r)
    for (int x = 0; x < 7; ++x) {
      TcnBlock tb;
      tb.dilation = 1 << x;
      for (int h = 0; h < 2; ++h) {
        if (c.tcn_norm) {                          // state_dict order: the norm module precedes its DepthwiseSeparableConv
          snprintf(nm, sizeof(nm), "TCN.temporal_conv_net.%d.%d.net.%d", r, x, h == 0 ? 0 : 3);
          const std::string q(nm);
          if (c.tcn_norm == 3) {
            tb.h[h].on[0] = add_tensor(n, q + ".weight", 128);
            tb.h[h].on[1] = add_tensor(n, q + ".bias", 128);
            tb.h[h].on[2] = add_tensor(n, q + ".running_mean", 128);
            tb.h[h].on[3] = add_tensor(n, q + ".running_var", 128);
          } else {
            tb.h[h].on[0] = add_tensor(n, q + ".gamma", 128);
            tb.h[h].on[1] = add_tensor(n, q + ".beta", 128);
          }
        }
        snprintf(nm, sizeof(nm), "TCN.temporal_conv_net.%d.%d.net.%d.net", r, x, h == 0 ? 2 : 5);
        const std::string p(nm);
        tb.h[h].dw = add_tensor(n, p + ".0.weight", 128 * 3);
        tb.h[h].prelu = add_tensor(n, p + ".1.weight", 1);
        tb.h[h].gamma = add_tensor(n, p + ".2.gamma", 128);
        tb.h[h].beta = add_tensor(n, p + ".2.beta", 128);
        tb.h[h].pw = add_tensor(n, p + ".3.weight", 128 * 128);
      }
      n->tcn.push_back(tb);
    }
  // ---- taps ----
  n->taps.push_back({"enc0_conv", B_E0, 0, en[0], false});
  for (int b = 0; b < 7; ++b) {
    snprintf(nm, sizeof(nm), "enc%d", b);
    n->taps.push_back({nm, B_D0 + (6 - b), de[6 - b], en[b], true});
  }
  n->taps.push_back({"tcn_out", B_D0, 0, 128, false});
  for (int i = 0; i < 6; ++i) {
    snprintf(nm, sizeof(nm), "dec%d", i);
    n->taps.push_back({nm, B_D0 + i + 1, 0, de[i + 1], true});
  }
  n->taps.push_back({"dec6", B_OUT, 0, c.out_ch, false});
  return MISONET_OK;
}

static long long align_up(long long x, long long a) { return (x + a - 1) / a * a; }

// precision 2 ("bf16x3") and 3 ("bf16x6"), the DMA dataflows: the dense-block buffers and everything between them travel
// in the oct layout (2 bf16 parts = the bytes of float32, or 3 parts = 6 bytes per element); the network input/output
// and the TCN stay planar float32, and so do the F <= 3 bottleneck buffers except in bf16x6.  Returns the ConvArgs::in_oct /
// out_oct code.
static inline int buf_oct(const misonet_net* n, int b) {
  if (n->precision < 2 || n->precision >= 5) return 0;
  const bool o = (b >= B_E0 && b <= B_E4) || (b >= B_D2 && b <= B_D6) || (b >= B_X2 && b <= B_X6);
  // bf16x6 also keeps the F <= 3 bottleneck buffers D0 / D1 in its layout (the TCN reads / writes it at its two ends), so
  // that encoder 6 and decoders 0-1 run on the persistent kernel instead of the one-row-per-wave f32 kernel
  if (n->precision == 3 && (b == B_D0 || b == B_D1)) return 3;
  // f16x3: the first dense-block buffer holds the RAW first-layer output (un-normalised, its scale follows the input), which
  // two fp16 pieces cannot carry for every input scale nor to 24 bits: it stays in the exact three-bf16 layout and the
  // layers that read it run on the bf16x6 kernel (its last conv writes the fp16 layout of the next buffer)
  if (n->precision == 4 && b == B_E0) return 3;
  return o ? (n->precision == 3 ? 3 : (n->precision == 4 ? 4 : 1)) : 0;
}
// floats one sample occupies in buffer b (an oct3 buffer holds 1.5 floats per element; C is a multiple of 8 there)
static inline long long buf_floats(const misonet_net* n, int Tp, int b) {
  const long long e = (long long)n->bufs[b].C * n->bufs[b].F * Tp;
  return buf_oct(n, b) == 3 ? e + e / 2 : e;
}
// floats between consecutive samples of buffer b: the activation arena is SAMPLE-major (all buffers of a sample in one
// block, so that buffers whose lifetimes do not overlap can share memory: make_layout), the TCN buffers are buffer-major
static inline long long bstride(const misonet_net* n, const Layout& L, int b) {
  if (b == B_IN && L.in_ext_off >= 0) return L.in_ext_bstride;
  return b >= B_TXA ? buf_floats(n, L.Tp, b) : L.sample_stride;
}

// Lifetime of an activation buffer in steps of a forward: 0 = input written (pack / MISO3 input assembly), 1 + b =
// encoder b, 8 = TCN, 9 + i = decoder i, 16 = results read (unpack; the pipeline also reads MISO1's input and output
// planes after the forward: PIT, MVDR, MISO3 input assembly).  The only cross-level lifetime of the reference is the
// skip list xs (model.py:84-99): decoder buffer D[i] receives xs[6 - i] at encoder 6 - i and is consumed by decoder i;
// an encoder's dense-block buffer E[b] is dead once its last conv has written xs[b], X[i] lives inside decoder i.
static void buf_lifetime(int b, int& t0, int& t1) {
  if (b == B_IN) { t0 = 0; t1 = 16; }
  else if (b >= B_E0 && b <= B_E4) { t0 = t1 = 1 + (b - B_E0); }
  else if (b >= B_D0 && b <= B_D6) { const int i = b - B_D0; t0 = 7 - i; t1 = 9 + i; }
  else if (b >= B_X2 && b <= B_X6) { t0 = t1 = 9 + 2 + (b - B_X2); }
  else { t0 = 15; t1 = 16; }                                     // B_OUT
}

static Layout make_layout(const misonet_net* n, int N, int T, bool ext_in = false) {
  Layout L;
  L.N = N; L.T = T; L.Tp = frames_pitch(T);
  long long so = 0;
  for (int b = 0; b < NBUF; ++b) {
    L.stats_off[b] = so;
    so += (long long)N * n->bufs[b].C * 2 * DS_NL;
  }
  const long long tslots = tcn_part_slots(T);          // TCN statistics: plain double2 partials (tcn.hip), never zeroed
  so = (so + 1) & ~1LL;                                // 16-byte alignment of the double2 arrays
  L.tcn_xs = so;  so += 15LL * N * 128 * tslots * 2;
  L.tcn_ps = so;  so += 14LL * N * 128 * tslots * 2;
  L.tcn_gln = so; so += 28LL * N * 32 * 2;
  L.stats_doubles = so;
  L.data_base = align_up(256 + so * 8, 256);
  // ---- activation arena: per-sample offsets by first fit over (lifetime x address) rectangles, largest buffer first ----
  {
    struct Rect { int b, t0, t1; long long off, size; };
    std::vector<Rect> placed;
    std::vector<int> order;
    for (int b = 0; b < B_TXA; ++b)
      if (!(ext_in && b == B_IN)) order.push_back(b);          // an external input takes no room in the arena
    L.data_off[B_IN] = 0;
    std::sort(order.begin(), order.end(), [&](int x, int y) {
      const long long sx = buf_floats(n, L.Tp, x), sy = buf_floats(n, L.Tp, y);
      return sx != sy ? sx > sy : x < y;
    });
    long long top = 0;
    for (int b : order) {
      Rect r;
      r.b = b;
      r.size = align_up(buf_floats(n, L.Tp, b), 64);
      buf_lifetime(b, r.t0, r.t1);
      if (n->keep_taps) { r.t0 = 0; r.t1 = 16; }
      // candidate offsets: 0 and the end of every placed rectangle that is alive at the same time
      std::vector<long long> cand(1, 0);
      for (const Rect& q : placed)
        if (q.t0 <= r.t1 && r.t0 <= q.t1) cand.push_back(q.off + q.size);
      std::sort(cand.begin(), cand.end());
      for (long long o : cand) {
        bool ok = true;
        for (const Rect& q : placed)
          if (q.t0 <= r.t1 && r.t0 <= q.t1 && o < q.off + q.size && q.off < o + r.size) { ok = false; break; }
        if (ok) { r.off = o; break; }
      }
      placed.push_back(r);
      L.data_off[b] = r.off;
      top = std::max(top, r.off + r.size);
    }
    L.sample_stride = top;
  }
  long long d = (long long)N * L.sample_stride;
  for (int b = B_TXA; b < NBUF; ++b) {
    L.data_off[b] = d;
    d += align_up((long long)N * buf_floats(n, L.Tp, b), 64);
  }
  long long wmax = 0, cmax = 0;
  auto scan = [&](const std::vector<ConvL>& v) {
    for (const ConvL& c : v) {
      const long long g = (c.Cout + 31) / 32, k = (c.Cin + 15) / 16;
      wmax = std::max(wmax, g * k * (2LL * 9 * 2 * 32 * 16));
      wmax = std::max(wmax, conv_bf16x6_wps_bytes((c.Cin + 7) / 8 * 8, c.Cout));
      cmax = std::max(cmax, g * 32);
    }
  };
  scan(n->enc); scan(n->dec);
  L.wps_base = align_up(L.data_base + d * 4, 256);
  L.wps_nstride = wmax;
  L.btab_base = align_up(L.wps_base + wmax * N, 256);
  L.btab_nstride = cmax * 9 * 4;                    // up to 4 shares of the shift table (conv_wprep6_k)
  L.fstat_base = align_up(L.btab_base + L.btab_nstride * 4 * N, 256);
  L.total_bytes = L.fstat_base + (n->cfg.tcn_norm == 2 ? (long long)N * L.Tp * 8 : 0);
  return L;
}

static inline float* buf_ptr(const Layout& L, void* ws, int b) {
  if (b == B_IN && L.in_ext_off >= 0) return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + L.in_ext_off);
  return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + L.data_base) + L.data_off[b];
}
static inline dstat_t* stats_base(void* ws) { return reinterpret_cast<dstat_t*>(reinterpret_cast<char*>(ws) + 256); }
static inline dstat_t* stats_ptr(const Layout& L, void* ws, int b) { return stats_base(ws) + L.stats_off[b]; }

static int run_conv(const misonet_net* n, const Layout& L, void* ws, const ConvL& c, hipStream_t s, int n0 = 0, int nb = -1) {
  ConvArgs a;
  a.in = buf_ptr(L, ws, c.in_buf);
  a.in_stats = stats_ptr(L, ws, c.in_buf);
  a.out = buf_ptr(L, ws, c.out_buf);
  a.out_stats = stats_ptr(L, ws, c.out_buf);
  a.w = n->w_dev + c.w_off;
  a.bias = n->w_dev + c.b_off;
  a.w16 = (MN_ALT_MODES && !planar_f32(n)) ? reinterpret_cast<const unsigned short*>(n->w_dev + c.w16_off) : nullptr;
  a.ww = (n->precision == 5 && c.ww_off >= 0) ? n->w_dev + c.ww_off : nullptr;
  a.ww6 = (MN_ALT_MODES && n->precision == 6 && c.ww6_off >= 0) ? n->w_dev + c.ww6_off : nullptr;
  a.wsm = (planar_f32(n) && c.wsm_off >= 0) ? n->w_dev + c.wsm_off : nullptr;
  a.w1d = (n->precision == 5 && c.w1d_off >= 0) ? n->w_dev + c.w1d_off : nullptr;
  a.ww16 = (n->precision == 5 && c.ww16_off >= 0) ? n->w_dev + c.ww16_off : nullptr;
  a.in_bstride = bstride(n, L, c.in_buf);
  a.out_bstride = bstride(n, L, c.out_buf);
  a.in_sstride = n->bufs[c.in_buf].C;
  a.out_sstride = n->bufs[c.out_buf].C;
  a.in_c0 = c.in_c0; a.Cin = c.Cin; a.Fin = n->bufs[c.in_buf].F; a.ident_c = c.ident_c;
  a.out_c0 = c.out_c0; a.Cout = c.Cout; a.Fout = n->bufs[c.out_buf].F;
  a.T = L.T; a.Tp = L.Tp;
  a.sf = c.sf; a.padf = c.padf; a.tr2 = c.tr2; a.act = c.act;
  a.NR = conv_rows(c.sf, c.tr2);
  a.ncg = c.ncg; a.cop = c.cop;
  a.dbg = 0;
  a.dbg_buf = nullptr;
  a.xcd = 0; a.ntx = a.nty = a.nsamp = 0;
  a.in_oct = a.out_oct = 0; a.wps = nullptr; a.wps_nstride = 0; a.btab = nullptr; a.btab_nstride = 0;
  if (nb < 0) nb = L.N;
  a.in += (long long)n0 * a.in_bstride;
  a.out += (long long)n0 * a.out_bstride;
  a.in_stats += (long long)n0 * a.in_sstride * 2 * DS_NL;
  a.out_stats += (long long)n0 * a.out_sstride * 2 * DS_NL;
  a.in_oct = buf_oct(n, c.in_buf);
  a.out_oct = buf_oct(n, c.out_buf);
  // bf16x6 / f16x3: the planar-input layers (network input, F <= 3 bottleneck) run on the exact f32 kernel
  if (n->precision >= 3 && !a.in_oct) a.w16 = nullptr;
  a.wscale = a.descale = 1.f;
  if (a.in_oct == 4) { a.wscale = c.wscale; a.descale = 1.f / c.wscale; }
  if (a.w16 || a.in_oct) { a.cop = 32; a.ncg = (c.Cout + 31) / 32; }
  // MISONET_SYNC_DEBUG=1: name every conv launch and wait for it (fault hunting)
  static const int sync_dbg = exp_env("MISONET_SYNC_DEBUG", 0);
  struct SyncDbg {
    hipStream_t s; const ConvArgs& a; int on;
    ~SyncDbg() {
      if (!on) return;
      fprintf(stderr, "[conv] Cin=%d Cout=%d Fin=%d Fout=%d T=%d in_oct=%d out_oct=%d mode=%d ... ", a.Cin, a.Cout, a.Fin, a.Fout,
              a.T, a.in_oct, a.out_oct, a.tr2 ? 2 : (a.sf == 2 ? 1 : 0));
      const hipError_t e = hipStreamSynchronize(s);
      fprintf(stderr, "%s\n", hipGetErrorString(e));
    }
  } sync_guard{s, a, sync_dbg};
  // bf16x6 / f16x3 networks: the planar-input first layer in the bf16x6 arithmetic (MISONET_X6_FIRST=0: the exact-f32 kernel
  // of rounds 1-3, for A/B runs)
  static const int x6first_env = exp_env("MISONET_X6_FIRST", 1);
  if (x6first_env && n->precision >= 3 && !a.in_oct && a.out_oct == 3 && c.w6s_off >= 0 && !a.act) {
    ProfScope ps(s, PK_CONV);
    HIPCHK(launch_conv_x6_first(a, n->w_dev + c.w6s_off, nb, s));
    return MISONET_OK;
  }
  if (a.in_oct == 3) {
    a.wps = reinterpret_cast<char*>(ws) + L.wps_base + (long long)n0 * L.wps_nstride;
    a.wps_nstride = L.wps_nstride;
    a.btab = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + L.btab_base) + (long long)n0 * L.btab_nstride;
    a.btab_nstride = L.btab_nstride;
    {
      ProfScope pw(s, PK_OTHER);
      HIPCHK(launch_conv_wprep6(a, n->w_dev + c.wf6_off, nb, s));
    }
    ProfScope ps(s, PK_CONV);
    HIPCHK(launch_conv_bf16x6(a, nb, s));
    return MISONET_OK;
  }
#if MN_ALT_MODES
  if (a.in_oct) {
    a.wps = reinterpret_cast<char*>(ws) + L.wps_base + (long long)n0 * L.wps_nstride;
    a.wps_nstride = L.wps_nstride;
    a.btab = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + L.btab_base) + (long long)n0 * L.btab_nstride;
    a.btab_nstride = L.btab_nstride;
    {
      ProfScope pw(s, PK_OTHER);                                   // its own kind: part of the step, not of the conv kernel
      HIPCHK(launch_conv_wprep(a, n->w_dev + c.wf_off, nb, s));
    }
    ProfScope ps(s, PK_CONV);
    HIPCHK(launch_conv_bf16_dma(a, nb, s));
    return MISONET_OK;
  }
#else
  if (a.in_oct) return fail(MISONET_EINVAL, "activation layout %d exists only in the experiment build", a.in_oct);
#endif
  {
    ProfScope ps(s, PK_CONV);
#if MN_ALT_MODES
    if (a.w16) HIPCHK(launch_conv_bf16(a, nb, s));
    else if (a.ww6 && conv_wino6_ok(a)) HIPCHK(launch_conv_wino6(a, nb, s));
    else
#endif
    if (a.wsm && conv_few_ok(a)) HIPCHK(launch_conv_few(a, nb, s));
    else if (a.ww && conv_wino_ok(a)) HIPCHK(launch_conv_wino(a, nb, s));
    else HIPCHK(launch_conv(a, nb, s));
  }
  return MISONET_OK;
}

// The workspace header (256 bytes): word 0 = NaN flag; word 2 = LAYOUT STAMP of the forward that last ran in it -- which
// buffer plan (shared arena or one memory per buffer), arithmetic mode and geometry the offsets inside belong to.
// misonet_net_tap compares it before it reads a buffer (a tap after keep_activations(net, 1) on a workspace whose
// forward ran with the SHARED plan passed the old host-side check and read past the smaller workspace).
static unsigned layout_stamp(const misonet_net* n, const Layout& L) {
  unsigned h = 2166136261u;
  auto mix = [&](unsigned long long v) { for (int i = 0; i < 8; ++i) { h ^= (unsigned)(v & 0xff); h *= 16777619u; v >>= 8; } };
  mix(n->keep_taps ? 1 : 0); mix((unsigned)n->precision); mix((unsigned)L.N); mix((unsigned)L.T); mix((unsigned long long)L.total_bytes);
  return h | 1u;                                  // never 0 (= "no forward has run here")
}

// IN buffer already filled (planar).  Leaves the result (raw) in B_OUT.
static int forward_planar(misonet_net* n, const Layout& L, void* ws, hipStream_t s) {
  // the nan flag + the conv statistics (integer limbs are ACCUMULATED); the TCN partial arrays behind them are plainly written
  HIPCHK(hipMemsetAsync(ws, 0, (size_t)(256 + L.tcn_xs * 8), s));
  HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(reinterpret_cast<char*>(ws) + 8), (int)layout_stamp(n, L), 1, s));
  // Optional sample sub-batching of the conv stacks (MISONET_SUBBATCH = samples per pass at F = 127; deeper levels take
  // proportionally more): keeps a level's producer->consumer traffic inside the 256 MiB Infinity Cache.
  static const int sub_env = exp_env("MISONET_SUBBATCH", 0);
  auto run_stack = [&](const std::vector<ConvL>& v) -> int {
    if (sub_env <= 0) {
      for (const ConvL& c : v) { int r = run_conv(n, L, ws, c, s); if (r) return r; }
      return MISONET_OK;
    }
    // group consecutive layers by output frequency size; sub-batch each group
    size_t i = 0;
    while (i < v.size()) {
      const int Fg = n->bufs[v[i].out_buf].F;
      size_t j = i;
      while (j < v.size() && n->bufs[v[j].out_buf].F == Fg) ++j;
      int sb = sub_env * (127 / (Fg > 0 ? Fg : 1));
      if (sb < 1) sb = 1;
      for (int n0 = 0; n0 < L.N; n0 += sb) {
        const int nb = (L.N - n0) < sb ? (L.N - n0) : sb;
        for (size_t k = i; k < j; ++k) { int r = run_conv(n, L, ws, v[k], s, n0, nb); if (r) return r; }
      }
      i = j;
    }
    return MISONET_OK;
  };
  { int r = run_stack(n->enc); if (r) return r; }
  // ---- TCN ----
  {
    ProfScope ps_tcn(s, PK_TCN);
    const int N = L.N, T = L.T, Tp = L.Tp;
    double2* xs = reinterpret_cast<double2*>(stats_base(ws) + L.tcn_xs);
    double2* ps = reinterpret_cast<double2*>(stats_base(ws) + L.tcn_ps);
    double2* gl = reinterpret_cast<double2*>(stats_base(ws) + L.tcn_gln);
    const int tslots = tcn_part_slots(T);
    const long long per = (long long)N * 128 * tslots;
    const long long gper = (long long)N * 32;
    float* xa = buf_ptr(L, ws, B_TXA);
    float* xb = buf_ptr(L, ws, B_TXB);
    float* td = buf_ptr(L, ws, B_TD);
    float* tp = buf_ptr(L, ws, B_TP);
    const int skip_c0 = n->cfg.de_ch[0];
    HIPCHK(launch_tcn_prepare(buf_ptr(L, ws, B_D0), bstride(n, L, B_D0), skip_c0, stats_ptr(L, ws, B_D0),
                              n->bufs[B_D0].C, xa, xs, 128, T, Tp, N, s, buf_oct(n, B_D0) == 3));
    float* cur = xa;
    float* nxt = xb;
    // bf16x6 mode: the point-wise convs in the same arithmetic as the 3x3 convs (MISONET_TCN_X6=0: fp32 MFMA, A/B runs)
    static const int tcn_x6_env = exp_env("MISONET_TCN_X6", 1);
    const int tcn_x6 = (n->precision == 3 && tcn_x6_env) ? 1 : 0;
    for (int k = 0; k < 14; ++k) {
      const TcnBlock& tb = n->tcn[k];
      const float* W = n->w_dev;
      const int nk = n->cfg.tcn_norm;
      float2* fstat = reinterpret_cast<float2*>(reinterpret_cast<char*>(ws) + L.fstat_base);
      if (nk == 2) HIPCHK(launch_tcn_cln_stats(cur, fstat, 128, T, Tp, N, s));
      HIPCHK(launch_tcn_dw(cur, xs + k * per, k == 0 ? 1 : tslots, W + tb.h[0].o_dw, W + tb.h[0].o_prelu, td, gl + (2 * k) * gper,
                           128, T, Tp, tb.dilation, N, s, nk, W + tb.h[0].o_nsc, W + tb.h[0].o_nsh, fstat));
      HIPCHK(launch_tcn_pw(td, gl + (2 * k) * gper, W + tb.h[0].o_gamma, W + tb.h[0].o_beta, W + tb.h[0].o_pw,
                           nullptr, tp, 128LL * Tp, 0, ps + k * per, 128, T, Tp, N, s, 0, tcn_x6));
      if (nk == 2) HIPCHK(launch_tcn_cln_stats(tp, fstat, 128, T, Tp, N, s));
      HIPCHK(launch_tcn_dw(tp, ps + k * per, tslots, W + tb.h[1].o_dw, W + tb.h[1].o_prelu, td,
                           gl + (2 * k + 1) * gper, 128, T, Tp, tb.dilation, N, s, nk, W + tb.h[1].o_nsc, W + tb.h[1].o_nsh, fstat));
      const bool last = (k == 13);
      float* y = last ? buf_ptr(L, ws, B_D0) : nxt;
      HIPCHK(launch_tcn_pw(td, gl + (2 * k + 1) * gper, W + tb.h[1].o_gamma, W + tb.h[1].o_beta,
                           W + tb.h[1].o_pw, cur, y, last ? bstride(n, L, B_D0) : 128LL * Tp, 0,
                           xs + (k + 1) * per, 128, T, Tp, N, s, (last && buf_oct(n, B_D0) == 3) ? n->bufs[B_D0].C : 0, tcn_x6));
      float* t = cur; cur = nxt; nxt = t;
    }
  }
  { int r = run_stack(n->dec); if (r) return r; }
  return MISONET_OK;
}

// ---------------------------------------------------------------------------------------------------------------
extern "C" {

const char* misonet_strerror(int code) {
  switch (code) {
    case MISONET_OK: return "ok";
    case MISONET_EINVAL: return "invalid argument / unsupported geometry";
    case MISONET_ESTATE: return "invalid call order or missing tensor";
    case MISONET_EHIP: return "HIP runtime error";
    case MISONET_ENOMEM: return "workspace too small";
    case MISONET_ENAN: return "NaN in network output";
    default: return "unknown error";
  }
}
const char* misonet_last_error(void) { return g_err; }
int misonet_version(void) { return 450; }   // 450: product modes 0 / 3 / 5 only (1, 2, 4, 6: experiment build); 410: misonet_pipeline_create accepts miso3 == NULL (separation-only pipeline); 420: misonet_istft; 430: misonet_frontend_init, precision mode 5 (f32w); 440: precision mode 6 (bf16x6w)

int misonet_net_create(const misonet_cfg* cfg, misonet_net** out) {
  if (!cfg || !out) return fail(MISONET_EINVAL, "null argument");
  misonet_net* n = new misonet_net();
  n->cfg = *cfg;
  int r = build_plan(n);
  if (r) { delete n; return r; }
  *out = n;
  return MISONET_OK;
}

int misonet_net_destroy(misonet_net* n) {
  if (!n) return MISONET_OK;
  if (n->w_dev) (void)hipFree(n->w_dev);
  delete n;
  return MISONET_OK;
}

int misonet_net_num_tensors(const misonet_net* n) { return n ? (int)n->tensors.size() : 0; }
const char* misonet_net_tensor_name(const misonet_net* n, int i) {
  if (!n || i < 0 || i >= (int)n->tensors.size()) return nullptr;
  return n->tensors[i].name.c_str();
}
long long misonet_net_tensor_numel(const misonet_net* n, int i) {
  if (!n || i < 0 || i >= (int)n->tensors.size()) return -1;
  return n->tensors[i].numel;
}

int misonet_net_set_tensor(misonet_net* n, const char* key, const float* host, long long numel) {
  if (!n || !key || !host) return fail(MISONET_EINVAL, "null argument");
  const int i = find_tensor(n, key);
  if (i < 0) return fail(MISONET_EINVAL, "unexpected state_dict key '%s'", key);
  if (n->tensors[i].numel != numel)
    return fail(MISONET_EINVAL, "size mismatch for '%s': expected %lld elements, got %lld", key, n->tensors[i].numel, numel);
  n->tensors[i].host.assign(host, host + numel);
  n->tensors[i].set = true;
  n->committed = false;
  return MISONET_OK;
}

// conv3x3_mfma's weight image [cg][chunk of CK ci][tap = kt * 3 + kf][ci][COP co] (conv-form taps, zero padded)
static void direct_image(const float* W, int Cin, int Cout, int COP, int ncg, bool transposed, float* w) {
  struct { int Cin, Cout, ncg; bool transposed; } c = {Cin, Cout, ncg, transposed};
  const int nchunk = (Cin + CK - 1) / CK;
  for (int cg = 0; cg < c.ncg; ++cg)
    for (int kc = 0; kc < nchunk; ++kc)
      for (int kt = 0; kt < 3; ++kt)
        for (int kf = 0; kf < 3; ++kf)
          for (int cil = 0; cil < CK; ++cil)
            for (int col = 0; col < COP; ++col) {
              const int ci = kc * CK + cil, co = cg * COP + col;
              float v = 0.f;
              if (ci < c.Cin && co < c.Cout) {
                if (c.transposed)   // ConvTranspose2d weight [Cin][Cout][3][3], taps flipped into conv form
                  v = W[(((long long)ci * c.Cout + co) * 3 + (2 - kt)) * 3 + (2 - kf)];
                else                // Conv2d weight [Cout][Cin][3][3]
                  v = W[(((long long)co * c.Cin + ci) * 3 + kt) * 3 + kf];
              }
              w[((((long long)cg * nchunk + kc) * 9 + (kt * 3 + kf)) * CK + cil) * COP + col] = v;
            }
}
static void pack_conv(const misonet_net* n, const ConvL& c, std::vector<float>& arena) {
  const std::vector<float>& Bv = n->tensors[c.bt].host;
  const int COP = c.cop;
  direct_image(n->tensors[c.wt].host.data(), c.Cin, c.Cout, c.cop, c.ncg, c.transposed, arena.data() + c.w_off);
  float* b = arena.data() + c.b_off;
  for (int co = 0; co < c.ncg * COP; ++co) b[co] = co < c.Cout ? Bv[co] : 0.f;
}

static inline unsigned short f32_to_bf16_rne(float f) {
  unsigned int u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static inline float bf16_to_f32(unsigned short h) {
  unsigned int u = (unsigned int)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// bf16x3 path: [cg][chunk of 16 ci][hi|lo][tap][octet h][COP][8] bf16 (conv_bf16.hip)
static void pack_conv_bf16(const misonet_net* n, ConvL& c, std::vector<float>& arena) {
  if (!MN_ALT_MODES) return;
  const std::vector<float>& W = n->tensors[c.wt].host;
  {
    // f16x3: a static power of two per layer brings max |W| to [32, 64): with rstd in [2^-6, 2^9] the folded weights
    // W * rstd stay below fp16's 65504 and the lo pieces of all but negligible weights stay normal
    float m = 0.f;
    for (float v : W) m = std::max(m, fabsf(v));
    c.wscale = (m > 0.f && std::isfinite(m)) ? exp2f(floorf(log2f(64.f / m))) : 1.f;
  }
  const int nchunk = (c.Cin + 15) / 16;
  const int COP = 32;                       // the bf16x3 kernels always work on 32-channel output groups
  const int ncg16 = (c.Cout + 31) / 32;
  unsigned short* w = reinterpret_cast<unsigned short*>(arena.data() + c.w16_off);
  float* wf = arena.data() + c.wf_off;
  const long long img = 9LL * 2 * COP * 8;
  for (int cg = 0; cg < ncg16; ++cg)
    for (int kc = 0; kc < nchunk; ++kc)
      for (int kt = 0; kt < 3; ++kt)
        for (int kf = 0; kf < 3; ++kf)
          for (int h = 0; h < 2; ++h)
            for (int col = 0; col < COP; ++col)
              for (int e = 0; e < 8; ++e) {
                const int ci = kc * 16 + 8 * h + e, co = cg * COP + col;
                float v = 0.f;
                if (ci < c.Cin && co < c.Cout) {
                  if (c.transposed) v = W[(((long long)ci * c.Cout + co) * 3 + (2 - kt)) * 3 + (2 - kf)];
                  else v = W[(((long long)co * c.Cin + ci) * 3 + kt) * 3 + kf];
                }
                const unsigned short hi = f32_to_bf16_rne(v);
                const unsigned short lo = f32_to_bf16_rne(v - bf16_to_f32(hi));
                const long long base = ((long long)cg * nchunk + kc) * 2 * img;
                const long long idx = ((((long long)(kt * 3 + kf)) * 2 + h) * COP + col) * 8 + e;
                w[base + idx] = hi;
                w[base + img + idx] = lo;
                wf[((long long)cg * nchunk + kc) * img + idx] = v;
              }
}

// bf16x6 path: float32 weights [cg][chunk of 8 ci][tap = kt*3 + kf][32 co][8 ci], zero padded (conv_wprep6_k source)
static void pack_conv_wf6(const misonet_net* n, const ConvL& c, std::vector<float>& arena) {
  const std::vector<float>& W = n->tensors[c.wt].host;
  const int nchunk = (c.Cin + 7) / 8;
  const int ncg = (c.Cout + 31) / 32;
  float* wf = arena.data() + c.wf6_off;
  for (int cg = 0; cg < ncg; ++cg)
    for (int kc = 0; kc < nchunk; ++kc)
      for (int kt = 0; kt < 3; ++kt)
        for (int kf = 0; kf < 3; ++kf)
          for (int col = 0; col < 32; ++col)
            for (int e = 0; e < 8; ++e) {
              const int ci = kc * 8 + e, co = cg * 32 + col;
              float v = 0.f;
              if (ci < c.Cin && co < c.Cout) {
                if (c.transposed) v = W[(((long long)ci * c.Cout + co) * 3 + (2 - kt)) * 3 + (2 - kf)];
                else v = W[(((long long)co * c.Cin + ci) * 3 + kt) * 3 + kf];
              }
              wf[((((long long)cg * nchunk + kc) * 9 + (kt * 3 + kf)) * 32 + col) * 8 + e] = v;
            }
}

// the first layer's weights as one exact 3-part bf16 image per 8-channel chunk, in conv_wprep6_k's unit order (unit
// ((kf*3 + p)*2 + kt)*32 + co for kt < 2, 576 + (kf*3 + p)*32 + co for kt = 2; 8 channels per unit): w = hi + mid + lo exactly
static void pack_conv_w6s(const misonet_net* n, const ConvL& c, std::vector<float>& arena) {
  if (c.w6s_off < 0) return;
  const std::vector<float>& W = n->tensors[c.wt].host;
  const int nchunk = (c.Cin + 7) / 8;
  unsigned short* img = reinterpret_cast<unsigned short*>(arena.data() + c.w6s_off);
  for (int kc = 0; kc < nchunk; ++kc)
    for (int kt = 0; kt < 3; ++kt)
      for (int kf = 0; kf < 3; ++kf)
        for (int co = 0; co < 32; ++co)
          for (int e = 0; e < 8; ++e) {
            const int ci = kc * 8 + e;
            const float v = (ci < c.Cin && co < c.Cout) ? W[(((long long)co * c.Cin + ci) * 3 + kt) * 3 + kf] : 0.f;
            const unsigned short h = f32_to_bf16_rne(v);
            const float r1 = v - bf16_to_f32(h);
            const unsigned short m = f32_to_bf16_rne(r1);
            const unsigned short l = f32_to_bf16_rne(r1 - bf16_to_f32(m));
            const unsigned short part[3] = {h, m, l};
            for (int p = 0; p < 3; ++p) {
              const long long unit = kt < 2 ? ((kf * 3 + p) * 2 + kt) * 32 + co : 576 + (kf * 3 + p) * 32 + co;
              img[((long long)kc * 864 + unit) * 8 + e] = part[p];
            }
          }
}

// conv_few.hip (the <= 4-channel last layer on the vector ALU): [ci][tap = kt * 3 + kf][4 co] conv-form taps, zero padded
static void pack_conv_few(const misonet_net* n, const ConvL& c, std::vector<float>& arena) {
  if (c.wsm_off < 0) return;
  const std::vector<float>& W = n->tensors[c.wt].host;
  float* img = arena.data() + c.wsm_off;
  for (int ci = 0; ci < c.Cin; ++ci)
    for (int kt = 0; kt < 3; ++kt)
      for (int kf = 0; kf < 3; ++kf)
        for (int co = 0; co < 4; ++co) {
          float v = 0.f;
          if (co < c.Cout) {
            if (c.transposed) v = W[(((long long)ci * c.Cout + co) * 3 + (2 - kt)) * 3 + (2 - kf)];
            else v = W[(((long long)co * c.Cin + ci) * 3 + kt) * 3 + kf];
          }
          img[((long long)ci * 9 + (kt * 3 + kf)) * 4 + co] = v;
        }
}

// f32w path: Winograd-domain weights U = G g G^T of a stride-1 same-padded conv (conv_wino.hip), G = [1 0 0; .5 .5 .5; .5 -.5 .5;
// 0 0 1]; position pos = xi * 4 + nu with xi along frequency (kf) and nu along time (kt).  Image order: [cg of 32 co][chunk of
// 8 ci][pos / 4][ci][co][pos % 4], zero padded past Cout.  Computed in double, rounded once.  Signs: positions with nu = 2
// carry a minus because the kernel's packed input transform produces -V there (conv_wino.hip, pk_t23); positions with nu = 3
// and positions with xi = 3 carry one each (both: none) so that the inverse transform A^T M A = sums with a single mixed-sign
// step per row (conv_wino.hip epilogue: the accumulators hold -M there).
// conv.hip W1D: U = G g along T per (co, ci, kf), conv-form taps (a transposed conv's are flipped), [cg32][chunk][nu * 3 + kf][ci][32]
static void w1d_image(const float* W, int Cin, int Cout, bool transposed, float* img) {
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int nchunk = (Cin + CK - 1) / CK, ncg = (Cout + 31) / 32;
  for (int cg = 0; cg < ncg; ++cg)
    for (int kc = 0; kc < nchunk; ++kc)
      for (int nu = 0; nu < 4; ++nu)
        for (int kf = 0; kf < 3; ++kf)
          for (int cil = 0; cil < CK; ++cil)
            for (int col = 0; col < 32; ++col) {
              const int ci = kc * CK + cil, co = cg * 32 + col;
              double u = 0.0;
              if (ci < Cin && co < Cout)
                for (int kt = 0; kt < 3; ++kt) {
                  const double g = transposed ? (double)W[(((long long)ci * Cout + co) * 3 + (2 - kt)) * 3 + (2 - kf)]
                                              : (double)W[(((long long)co * Cin + ci) * 3 + kt) * 3 + kf];
                  u += G[nu][kt] * g;
                }
              img[((((long long)cg * nchunk + kc) * 12 + (nu * 3 + kf)) * CK + cil) * 32 + col] = (float)u;
            }
}
static void wino_image(const float* W, int Cin, int Cout, float* img) {
  struct { int Cin, Cout; } c = {Cin, Cout};
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int nchunk = c.Cin / 8, ncg = (c.Cout + 31) / 32;
  for (int cg = 0; cg < ncg; ++cg)
    for (int kc = 0; kc < nchunk; ++kc)
      for (int cil = 0; cil < 8; ++cil)
        for (int col = 0; col < 32; ++col) {
          const int ci = kc * 8 + cil, co = cg * 32 + col;
          for (int pos = 0; pos < 16; ++pos) {
            const int xi = pos >> 2, nu = pos & 3;
            double u = 0.0;
            if (co < c.Cout)
              for (int kt = 0; kt < 3; ++kt)
                for (int kf = 0; kf < 3; ++kf)
                  u += G[xi][kf] * G[nu][kt] * (double)W[(((long long)co * c.Cin + ci) * 3 + kt) * 3 + kf];
            const bool minus = (nu == 2) != ((nu == 3) != (xi == 3));
            img[(((((long long)cg * nchunk + kc) * 4 + (pos >> 2)) * 8 + cil) * 32 + col) * 4 + (pos & 3)] = (float)(minus ? -u : u);
          }
        }
}
// ... and of output channels [co0, co0 + 16) for the 16-row body: [chunk of 8 ci][K-step s of 4 ci][pos / 4][ci % 4][16 co][pos % 4],
// the same signs
static void wino_image16(const float* W, int Cin, int co0, float* img) {
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int nchunk = Cin / 8;
  for (int kc = 0; kc < nchunk; ++kc)
    for (int cil = 0; cil < 8; ++cil)
      for (int col = 0; col < 16; ++col) {
        const int ci = kc * 8 + cil, co = co0 + col;
        for (int pos = 0; pos < 16; ++pos) {
          const int xi = pos >> 2, nu = pos & 3;
          double u = 0.0;
          for (int kt = 0; kt < 3; ++kt)
            for (int kf = 0; kf < 3; ++kf)
              u += G[xi][kf] * G[nu][kt] * (double)W[(((long long)co * Cin + ci) * 3 + kt) * 3 + kf];
          const bool minus = (nu == 2) != ((nu == 3) != (xi == 3));
          const int s = cil >> 2, kk = cil & 3;
          img[((((((long long)kc * 2 + s) * 4 + (pos >> 2)) * 4 + kk) * 16 + col) * 4) + (pos & 3)] = (float)(minus ? -u : u);
        }
      }
}
static void pack_conv_wino(const misonet_net* n, const ConvL& c, std::vector<float>& arena) {
  if (c.ww_off < 0) return;
  wino_image(n->tensors[c.wt].host.data(), c.Cin, c.Cout, arena.data() + c.ww_off);
  if (c.ww16_off >= 0) wino_image16(n->tensors[c.wt].host.data(), c.Cin, c.Cout - 16, arena.data() + c.ww16_off);
}

// Run-time self-check of the f32w kernel, once per device at the first commit: conv3x3_wino_f32 drives 256 fixed
// AGPRs and hand-placed wait states through inline asm -- the build guard (tools/check_wino_build.py) covers what the compiler
// may do to it, this covers the machine: one small layer (40 -> 48 channels, 11 x 70, two samples: ragged row and column tiles, a
// 32-channel group on the 32-row body and 16 channels on the 16-row body) through the Winograd kernels AND through conv3x3_mfma,
// compared element by element.  A mismatch
// disables mode 5 on this device: misonet_net_set_precision(5) then fails loudly instead of computing garbage.
static std::atomic<int> g_wino_state[MAX_DEV] = {};          // 0 not checked, 1 ok, 2 failed
static std::mutex g_wino_mu;
static int wino_selftest() {
  const int d = cur_dev();
  int st = g_wino_state[d].load(std::memory_order_acquire);
  if (st) return st;
  std::lock_guard<std::mutex> lk(g_wino_mu);
  st = g_wino_state[d].load(std::memory_order_acquire);
  if (st) return st;
  const int Cin = 40, Cout = 48, F = 11, T = 70, Tp = frames_pitch(T), N = 2;   // (48 = a 32-channel group on the 32-row body + 16 channels on the 16-row body)
  std::vector<float> W((size_t)Cout * Cin * 9), bias(Cout), x((size_t)N * Cin * F * Tp);
  unsigned rng = 12345u;
  auto rnd = [&rng]() { rng = rng * 1664525u + 1013904223u; return (float)((int)(rng >> 9) - (1 << 22)) * (1.f / (1 << 22)); };
  for (float& v : W) v = 0.1f * rnd();
  for (float& v : bias) v = 0.3f * rnd();
  for (float& v : x) v = rnd();
  const int nchunk = Cin / CK;
  std::vector<float> wd((size_t)nchunk * 9 * CK * 64), ww((size_t)2 * nchunk * 16 * 8 * 32), ww16((size_t)nchunk * 2048);
  direct_image(W.data(), Cin, Cout, 64, 1, false, wd.data());
  wino_image(W.data(), Cin, Cout, ww.data());
  wino_image16(W.data(), Cin, Cout - 16, ww16.data());
  const size_t out_n = (size_t)N * Cout * F * Tp, st_n = (size_t)N * Cout * 2 * DS_NL;
  float *dx = nullptr, *dwd = nullptr, *dww = nullptr, *dww16 = nullptr, *db = nullptr, *dy = nullptr;
  dstat_t* dst = nullptr;
  auto release = [&]() { (void)hipFree(dx); (void)hipFree(dwd); (void)hipFree(dww); (void)hipFree(dww16); (void)hipFree(db); (void)hipFree(dy); (void)hipFree(dst); };
  bool ok = hipMalloc(reinterpret_cast<void**>(&dx), x.size() * 4) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&dwd), wd.size() * 4) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&dww), ww.size() * 4) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&dww16), ww16.size() * 4) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&db), 128 * 4) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&dy), 2 * out_n * 4) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&dst), 2 * st_n * 8) == hipSuccess;
  std::vector<float> y(2 * out_n);
  std::vector<dstat_t> sv(2 * st_n);
  if (ok) {
    std::vector<float> b128(128, 0.f);
    std::copy(bias.begin(), bias.end(), b128.begin());
    ok = hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(dwd, wd.data(), wd.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(dww, ww.data(), ww.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(dww16, ww16.data(), ww16.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(db, b128.data(), 128 * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemset(dy, 0, 2 * out_n * 4) == hipSuccess && hipMemset(dst, 0, 2 * st_n * 8) == hipSuccess;
  }
  if (ok) {
    ConvArgs a = {};
    a.in = dx; a.in_stats = nullptr; a.w = dwd; a.bias = db; a.ww = dww; a.ww16 = dww16;
    a.in_bstride = (long long)Cin * F * Tp; a.out_bstride = (long long)Cout * F * Tp;
    a.in_sstride = Cin; a.out_sstride = Cout;
    a.in_c0 = 0; a.Cin = Cin; a.Fin = F; a.ident_c = Cin;              // input consumed as it is: no statistics to read
    a.out_c0 = 0; a.Cout = Cout; a.Fout = F; a.T = T; a.Tp = Tp;
    a.sf = 1; a.padf = 1; a.tr2 = 0; a.act = 1; a.NR = conv_rows(1, 0); a.ncg = 1; a.cop = 64;
    a.wscale = a.descale = 1.f;
    a.out = dy; a.out_stats = dst;
    ok = conv_wino_ok(a) && launch_conv_wino(a, N, nullptr) == hipSuccess;
    a.ww = nullptr; a.ww16 = nullptr; a.out = dy + out_n; a.out_stats = dst + st_n;
    ok = ok && launch_conv(a, N, nullptr) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
         hipMemcpy(y.data(), dy, 2 * out_n * 4, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(sv.data(), dst, 2 * st_n * 8, hipMemcpyDeviceToHost) == hipSuccess;
  }
  release();
  if (ok) {
    double worst = 0.0, scale = 0.0;
    for (int nn = 0; nn < N; ++nn)
      for (int c = 0; c < Cout; ++c)
        for (int f = 0; f < F; ++f)
          for (int t = 0; t < T; ++t) {
            const size_t i = (((size_t)nn * Cout + c) * F + f) * Tp + t;
            worst = std::max(worst, (double)fabsf(y[i] - y[out_n + i]));
            scale = std::max(scale, (double)fabsf(y[out_n + i]));
          }
    ok = std::isfinite(worst) && scale > 0.0 && worst <= 2e-5 * scale;
    for (size_t i = 0; ok && i < st_n / DS_NL; ++i) {                    // the statistics of both kernels (sum, sum of squares)
      auto rd = [](const dstat_t* p) { long long L[DS_NL]; for (int k = 0; k < DS_NL; ++k) L[k] = (long long)p[k]; return dstat_combine(L); };
      const double a1 = rd(sv.data() + i * DS_NL), a2 = rd(sv.data() + st_n + i * DS_NL);
      ok = std::isfinite(a1) && fabs(a1 - a2) <= 1e-4 * (fabs(a2) + 1.0);
    }
    if (!ok) fprintf(stderr, "[misonet] conv3x3_wino_f32 self-check FAILED on device %d (max |diff| %.3e of %.3e): mode f32w disabled\n", d, worst, scale);
  }
  st = ok ? 1 : 2;
  g_wino_state[d].store(st, std::memory_order_release);
  return st;
}

// ... and in the three-piece bf16 form of the bf16x6w mode (conv_wino6.hip): per (cg of 32 co, K-step of 16 ci) four QUARTERS
// (position rows xi), each [nu][piece h | m | l][lane = co + 32 * (ci / 8 % 2)][8 bf16 = channels 8 * (ci / 8 % 2) .. + 7]: one
// lane's MFMA A operand is 16 consecutive bytes.  U in double, rounded once to float32, then split exactly at fixed bit
// positions (top 8 / next 8 / last 8 significant bits).  No sign flips (the kernel's transform is the plain B^T d B).
static void pack_conv_wino6(const misonet_net* n, const ConvL& c, std::vector<float>& arena) {
  if (c.ww6_off < 0) return;
  const std::vector<float>& W = n->tensors[c.wt].host;
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int nk = (c.Cin + 15) / 16, ncg = (c.Cout + 31) / 32;
  unsigned short* img = reinterpret_cast<unsigned short*>(arena.data() + c.ww6_off);
  auto top16 = [](float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); };
  for (int cg = 0; cg < ncg; ++cg)
    for (int kk = 0; kk < nk; ++kk)
      for (int pos = 0; pos < 16; ++pos)
        for (int col = 0; col < 32; ++col)
          for (int cil = 0; cil < 16; ++cil) {
            const int ci = kk * 16 + cil, co = cg * 32 + col, xi = pos >> 2, nu = pos & 3;
            double u = 0.0;
            if (co < c.Cout && ci < c.Cin)
              for (int kt = 0; kt < 3; ++kt)
                for (int kf = 0; kf < 3; ++kf)
                  u += G[xi][kf] * G[nu][kt] * (double)W[(((long long)co * c.Cin + ci) * 3 + kt) * 3 + kf];
            const float v = (float)u;
            unsigned vb; memcpy(&vb, &v, 4);
            unsigned hb = vb & 0xffff0000u, wb = vb & 0xffffff00u;
            float h, w; memcpy(&h, &hb, 4); memcpy(&w, &wb, 4);
            const float m = w - h, l = v - w;
            const int lane = col + 32 * (cil >> 3), j = cil & 7;
            const long long q = (((long long)cg * nk + kk) * 4 + xi) * (3 * 4 * 64 * 8);
            img[q + ((long long)(nu * 3 + 0) * 64 + lane) * 8 + j] = top16(h);
            img[q + ((long long)(nu * 3 + 1) * 64 + lane) * 8 + j] = top16(m);
            img[q + ((long long)(nu * 3 + 2) * 64 + lane) * 8 + j] = top16(l);
          }
}

int misonet_net_commit(misonet_net* n) {
  if (!n) return fail(MISONET_EINVAL, "null argument");
  for (const Tensor& t : n->tensors)
    if (!t.set) return fail(MISONET_ESTATE, "missing state_dict key '%s'", t.name.c_str());
  long long off = 0;
  auto take = [&off](long long nfl) { long long o = off; off += (nfl + 63) / 64 * 64; return o; };
  auto place = [&](std::vector<ConvL>& v) {
    for (ConvL& c : v) {
      const int nchunk = (c.Cin + CK - 1) / CK;
      c.w_off = take((long long)c.ncg * nchunk * 9 * CK * c.cop);
      c.b_off = take((long long)c.ncg * c.cop);
      if (MN_ALT_MODES) {
        c.w16_off = take((long long)((c.Cout + 31) / 32) * ((c.Cin + 15) / 16) * 2 * 9 * 2 * 32 * 8 / 2);   // u16 -> floats
        c.wf_off = take((long long)((c.Cout + 31) / 32) * ((c.Cin + 15) / 16) * 9 * 2 * 32 * 8);
      }
      c.wf6_off = take((long long)((c.Cout + 31) / 32) * ((c.Cin + 7) / 8) * 9 * 32 * 8);
      // the first layer (planar network input, consumed un-normalised, <= 16 in / <= 32 out channels): shared 3-part image
      if (c.in_buf == B_IN && !c.transposed && c.sf == 1 && c.Cin <= 16 && c.Cout <= 32 && c.ident_c >= c.Cin)
        c.w6s_off = take((long long)((c.Cin + 7) / 8) * 864 * 4);          // 864 units x 16 bytes = x 4 floats
      // the DenseBlock convs (stride 1, same padding, Cin a multiple of 8): Winograd-domain image for the f32w mode
      if (!c.transposed && c.sf == 1 && c.padf == 1 && c.Cin % 8 == 0 && c.Cin <= 256)
      {
        c.ww_off = take((long long)((c.Cout + 31) / 32) * (c.Cin / 8) * 16 * 8 * 32);
        if ((c.Cout & 31) == 16) c.ww16_off = take((long long)(c.Cin / 8) * 2048);
      }
      // stride-2 convs / transposed convs, and a network's first layer (12 / 16 -> 24 channels): 1-D Winograd image along T
      if (((c.tr2 || c.sf == 2) && c.act) || (c.transposed && c.sf == 1 && !c.tr2 && c.padf == 2 && n->bufs[c.in_buf].F == 1 && c.act) || (!c.transposed && c.sf == 1 && c.padf == 0 && n->bufs[c.in_buf].F == 3 && c.Cout % 128 == 0 && c.act) || (!c.transposed && c.sf == 1 && !(c.padf == 2 && n->bufs[c.in_buf].F == 1) && c.Cin < 3 * CK && c.Cout <= 32))
        c.w1d_off = take((long long)((c.Cout + 31) / 32) * nchunk * 12 * CK * 32);
      if (c.Cout <= 4 && c.sf == 1 && !c.tr2 && !c.act && c.Cin % 4 == 0 && c.Cin <= 256) c.wsm_off = take((long long)c.Cin * 36);
      if (MN_ALT_MODES && !c.transposed && c.sf == 1 && c.padf == 1 && c.Cin % 8 == 0 && c.Cin >= 24 && c.Cin <= 256)
        c.ww6_off = take((long long)((c.Cout + 31) / 32) * ((c.Cin + 15) / 16) * (16 * 3 * 64 * 16 / 4));
    }
  };
  place(n->enc);
  place(n->dec);
  for (TcnBlock& tb : n->tcn)
    for (int h = 0; h < 2; ++h) {
      tb.h[h].o_dw = take(128 * 3);
      tb.h[h].o_prelu = take(1);
      tb.h[h].o_gamma = take(128);
      tb.h[h].o_beta = take(128);
      tb.h[h].o_pw = take(128 * 128);
      tb.h[h].o_nsc = take(128);
      tb.h[h].o_nsh = take(128);
    }
  std::vector<float> arena((size_t)off, 0.f);
  for (ConvL& c : n->enc) { pack_conv(n, c, arena); pack_conv_bf16(n, c, arena); pack_conv_wf6(n, c, arena); pack_conv_w6s(n, c, arena); pack_conv_wino(n, c, arena); pack_conv_wino6(n, c, arena); }
  for (ConvL& c : n->enc) if (c.w1d_off >= 0) w1d_image(n->tensors[c.wt].host.data(), c.Cin, c.Cout, c.transposed, arena.data() + c.w1d_off);
  for (ConvL& c : n->dec) if (c.w1d_off >= 0) w1d_image(n->tensors[c.wt].host.data(), c.Cin, c.Cout, c.transposed, arena.data() + c.w1d_off);
  for (ConvL& c : n->dec) { pack_conv_few(n, c, arena); pack_conv(n, c, arena); pack_conv_bf16(n, c, arena); pack_conv_wf6(n, c, arena); pack_conv_wino(n, c, arena); pack_conv_wino6(n, c, arena); }
  for (const TcnBlock& tb : n->tcn)
    for (int h = 0; h < 2; ++h) {
      const TcnHalf& H = tb.h[h];
      memcpy(arena.data() + H.o_dw, n->tensors[H.dw].host.data(), 128 * 3 * sizeof(float));
      arena[H.o_prelu] = n->tensors[H.prelu].host[0];
      memcpy(arena.data() + H.o_gamma, n->tensors[H.gamma].host.data(), 128 * sizeof(float));
      memcpy(arena.data() + H.o_beta, n->tensors[H.beta].host.data(), 128 * sizeof(float));
      if (n->cfg.tcn_norm == 3) {                              // BatchNorm1d in eval mode (run.py:79,106): affine, eps 1e-5
        const float *w = n->tensors[H.on[0]].host.data(), *b = n->tensors[H.on[1]].host.data(),
                    *rm = n->tensors[H.on[2]].host.data(), *rv = n->tensors[H.on[3]].host.data();
        for (int ch = 0; ch < 128; ++ch) {
          const float sc = w[ch] / sqrtf(rv[ch] + 1e-5f);
          arena[H.o_nsc + ch] = sc;
          arena[H.o_nsh + ch] = b[ch] - rm[ch] * sc;
        }
      } else if (n->cfg.tcn_norm) {                            // gLN / cLN: gamma, beta
        memcpy(arena.data() + H.o_nsc, n->tensors[H.on[0]].host.data(), 128 * sizeof(float));
        memcpy(arena.data() + H.o_nsh, n->tensors[H.on[1]].host.data(), 128 * sizeof(float));
      }
      const std::vector<float>& P = n->tensors[H.pw].host;     // [co][ci][1] -> [ci][co]
      for (int co = 0; co < 128; ++co)
        for (int ci = 0; ci < 128; ++ci) arena[H.o_pw + (long long)ci * 128 + co] = P[(long long)co * 128 + ci];
    }
  if (n->w_dev) { (void)hipFree(n->w_dev); n->w_dev = nullptr; }
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&n->w_dev), arena.size() * sizeof(float)));
  HIPCHK(hipMemcpy(n->w_dev, arena.data(), arena.size() * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(conv_init());
  HIPCHK(conv_wino_init());
  HIPCHK(conv_few_init());
#ifdef MISONET_EXPERIMENTS                  // (timing ablations compute garbage by design: MISONET_WINO_DBG skips the self-check)
  if (!exp_env("MISONET_WINO_DBG", 0))
#endif
  if (wino_selftest() != 1 && n->precision == 5)
    return fail(MISONET_ESTATE, "the f32w kernel failed its self-check on this device (see stderr): use mode 0 (f32) or 3 (bf16x6)");
#if MN_ALT_MODES
  HIPCHK(conv_wino6_init());
  HIPCHK(conv_bf16_init());
  HIPCHK(conv_bf16_dma_init());
#endif
  HIPCHK(conv_bf16x6_init());
  { int rf = frontend_init(); if (rf) return rf; }     // STFT / iSTFT tables: never allocated inside an asynchronous call
  n->committed = true;
  return MISONET_OK;
}

int misonet_net_set_precision(misonet_net* n, int mode) {
  if (!n) return fail(MISONET_EINVAL, "null argument");
  const bool product = (mode == 0 || mode == 3 || mode == 5);
  if (!product && !(MN_ALT_MODES && mode >= 1 && mode <= 6))
    return fail(MISONET_EINVAL, "precision mode must be 0 (f32), 3 (bf16x6) or 5 (f32w: f32 with the dense-block convs in Winograd form)%s",
                MN_ALT_MODES ? "; experiment build: also 1 / 2 (bf16x3), 4 (f16x3), 6 (bf16x6w)"
                             : "; modes 1, 2, 4, 6 exist only in the experiment build (make exp)");
  if (mode == 5 && g_wino_state[cur_dev()].load(std::memory_order_acquire) == 2)
    return fail(MISONET_ESTATE, "the f32w kernel failed its self-check on this device: mode 5 is disabled (use 0 or 3)");
  n->precision = mode;
  return MISONET_OK;
}
int misonet_net_get_precision(const misonet_net* n) { return n ? n->precision : -1; }

int misonet_net_buffer_plan(const misonet_net* n, int n_frames, int max_buffers, long long* offset_bytes,
                            long long* size_bytes, int* first_step, int* last_step) {
  if (!n || n_frames <= 0 || !offset_bytes || !size_bytes || !first_step || !last_step) return fail(MISONET_EINVAL, "null argument");
  const Layout L = make_layout(n, 1, n_frames);
  int k = 0;
  for (int b = 0; b < B_TXA && k < max_buffers; ++b, ++k) {
    offset_bytes[k] = L.data_off[b] * 4;
    size_bytes[k] = buf_floats(n, L.Tp, b) * 4;
    buf_lifetime(b, first_step[k], last_step[k]);
    if (n->keep_taps) { first_step[k] = 0; last_step[k] = 16; }
  }
  return k;
}

int misonet_net_keep_activations(misonet_net* n, int keep) {
  if (!n) return fail(MISONET_EINVAL, "null argument");
  n->keep_taps = keep != 0;
  return MISONET_OK;
}

long long misonet_net_workspace_bytes(const misonet_net* n, int n_samples, int n_frames) {
  if (!n || n_samples <= 0 || n_frames <= 0) return -1;
  return make_layout(n, n_samples, n_frames).total_bytes;
}

int misonet_net_forward(misonet_net* n, int n_seg, const void* const* seg_dev, const int* seg_ch, int B, int T,
                        void* out_dev, void* ws, long long ws_bytes, misonet_stream stream) {
  if (!n || !seg_dev || !seg_ch || !out_dev || !ws) return fail(MISONET_EINVAL, "null argument");
  if (!n->committed) return fail(MISONET_ESTATE, "misonet_net_commit has not been called");
  if (B <= 0 || T <= 0) return fail(MISONET_EINVAL, "B and T must be positive");
  int tot = 0;
  for (int i = 0; i < n_seg; ++i) tot += seg_ch[i];
  if (2 * tot != n->cfg.in_ch) return fail(MISONET_EINVAL, "input segments give %d complex channels, network expects %d", tot, n->cfg.in_ch / 2);
  const Layout L = make_layout(n, B, T);
  if (ws_bytes < L.total_bytes) return fail(MISONET_ENOMEM, "workspace %lld < %lld bytes", ws_bytes, L.total_bytes);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int c0 = 0;
  for (int i = 0; i < n_seg; ++i) {
    HIPCHK(launch_pack(reinterpret_cast<const float2*>(seg_dev[i]), B, seg_ch[i], T, n->cfg.n_freq, buf_ptr(L, ws, B_IN),
                       bstride(n, L, B_IN), L.Tp, c0, tot + c0, 1, s));
    c0 += seg_ch[i];
  }
  int r = forward_planar(n, L, ws, s);
  if (r) return r;
  HIPCHK(launch_unpack(buf_ptr(L, ws, B_OUT), bstride(n, L, B_OUT), L.Tp, n->S, T, n->cfg.n_freq,
                       reinterpret_cast<float2*>(out_dev), B, reinterpret_cast<int*>(ws), s));
  return MISONET_OK;
}

int misonet_net_check(misonet_net* n, const void* ws, misonet_stream stream) {
  if (!n || !ws) return fail(MISONET_EINVAL, "null argument");
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, ws, sizeof(int), hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream)));
  HIPCHK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
  if (flag) return fail(MISONET_ENAN, "NaN in network output");
  return MISONET_OK;
}

int misonet_net_tap_shape(const misonet_net* n, const char* name, int* C, int* F) {
  if (!n || !name) return fail(MISONET_EINVAL, "null argument");
  for (const Tap& t : n->taps)
    if (t.name == name) {
      if (C) *C = t.C;
      if (F) *F = n->bufs[t.buf].F;
      return MISONET_OK;
    }
  return fail(MISONET_EINVAL, "unknown tap '%s'", name);
}

int misonet_net_tap(misonet_net* n, const char* name, const void* ws, int B, int T, float* dst, misonet_stream stream) {
  if (!n || !name || !ws || !dst) return fail(MISONET_EINVAL, "null argument");
  const Layout L = make_layout(n, B, T);
  for (const Tap& t : n->taps)
    if (t.name == name) {
      if (!n->keep_taps && t.buf != B_OUT)
        return fail(MISONET_ESTATE, "tap '%s': activation buffers share memory; call misonet_net_keep_activations(net, 1) "
                                    "before the forward", name);
      {
        // the plan the workspace was WRITTEN with must be the one this call would read it with (a diagnostic entry point:
        // the 4-byte read-back synchronises the stream)
        unsigned stamp = 0;
        hipStream_t s_ = reinterpret_cast<hipStream_t>(stream);
        HIPCHK(hipMemcpyAsync(&stamp, reinterpret_cast<const char*>(ws) + 8, sizeof(stamp), hipMemcpyDeviceToHost, s_));
        HIPCHK(hipStreamSynchronize(s_));
        if (stamp != layout_stamp(n, L))
          return fail(MISONET_ESTATE, "tap '%s': the forward in this workspace ran with another buffer plan / mode / shape "
                                      "(keep_activations, precision, B or T changed since); run the forward again", name);
      }
      void* w = const_cast<void*>(ws);
      HIPCHK(launch_export(buf_ptr(L, w, t.buf), bstride(n, L, t.buf), t.c0, t.C, n->bufs[t.buf].F, T, L.Tp,
                           t.normalised ? stats_ptr(L, w, t.buf) : nullptr, n->bufs[t.buf].C, 0, dst, B,
                           reinterpret_cast<hipStream_t>(stream), buf_oct(n, t.buf) >= 3 ? buf_oct(n, t.buf) : (buf_oct(n, t.buf) ? 2 : 0)));
      return MISONET_OK;
    }
  return fail(MISONET_EINVAL, "unknown tap '%s'", name);
}

// ---- MVDR / PIT drop-in entry points ---------------------------------------------------------------------------
long long misonet_mvdr_workspace_bytes(int B, int F, int M) { return mvdr_ws_bytes(B, 1, F, M); }

int misonet_mvdr(const void* src, const void* mix, int B, int F, int M, int T, float epsi, void* out, void* ws,
                 long long ws_bytes, misonet_stream stream) {
  if (!src || !mix || !out || !ws) return fail(MISONET_EINVAL, "null argument");
  if (M < 2 || M > 8) return fail(MISONET_EINVAL, "M must be in [2, 8] (got %d)", M);
  if (B <= 0 || F <= 0 || T <= 0) return fail(MISONET_EINVAL, "B, F, T must be positive");
  if (ws_bytes < mvdr_ws_bytes(B, 1, F, M)) return fail(MISONET_ENOMEM, "workspace too small");
  MvdrArgs a;
  const float* y = reinterpret_cast<const float*>(mix);
  const float* x = reinterpret_cast<const float*>(src);
  a.mix = {y, y + 1, 2LL * F * M * T, 2LL * M * T, 2LL * T, 2};
  a.src = {x, x + 1, 2LL * F * M * T, 2LL * M * T, 2LL * T, 2};
  a.est = nullptr; a.est_bstride = 0; a.sel = nullptr;
  a.S = 1; a.B = B; a.F = F; a.M = M; a.T = T; a.Tp = T; a.epsi = epsi;
  float* o = reinterpret_cast<float*>(out);
  COut co = {o, o + 1, 2LL * T * F, 0, 2LL * F, 2};      // [B,T,F] complex64 (tester.py:1134)
  HIPCHK(launch_mvdr(a, co, ws, reinterpret_cast<hipStream_t>(stream)));
  return MISONET_OK;
}

int misonet_mvdr_debug(const void* ws, int B, int F, int M, void* steer, void* w, misonet_stream stream) {
  if (!ws) return fail(MISONET_EINVAL, "null argument");
  HIPCHK(launch_mvdr_debug(ws, B, 1, F, M, reinterpret_cast<double*>(steer), reinterpret_cast<double*>(w),
                           reinterpret_cast<hipStream_t>(stream)));
  return MISONET_OK;
}

long long misonet_pit_scratch_bytes(int B, int S, int F) {
  if (B <= 0 || S <= 0 || F <= 0) return -1;
  return (long long)B * S * S * (F + 1) * (long long)sizeof(double);
}

int misonet_pit_select(const void* anchor, const void* cand, int B, int S, int T, int F, int* sel, double* dist,
                       long long dist_bytes, misonet_stream stream) {
  if (!anchor || !cand || !sel || !dist) return fail(MISONET_EINVAL, "null argument (dist is required: B*S*S*(F+1) doubles)");
  if (S < 1 || S > 4) return fail(MISONET_EINVAL, "PIT alignment enumerates S! permutations: 1 <= num_spks <= 4 (got %d)", S);
  if (B <= 0 || T <= 0 || F <= 0) return fail(MISONET_EINVAL, "B, T, F must be positive");
  if (dist_bytes < misonet_pit_scratch_bytes(B, S, F))
    return fail(MISONET_ENOMEM, "dist scratch %lld < %lld bytes (B*S*S*(F+1) doubles)", dist_bytes, misonet_pit_scratch_bytes(B, S, F));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const float* a = reinterpret_cast<const float*>(anchor);
  const float* c = reinterpret_cast<const float*>(cand);
  PitArgs p;
  // [B,S,T,F] complex64: element (b, f, spk, t) at ((b*S + spk)*T + t)*F + f
  p.a = {a, a + 1, 2LL * S * T * F, 2, 2LL * T * F, 2 * F};
  p.b = {c, c + 1, 2LL * S * T * F, 2, 2LL * T * F, 2 * F};
  p.B = B; p.F = F; p.T = T;
  double* part = dist + (long long)B * S * S;          // per-bin partials [B][F][S][S] behind the result
  HIPCHK(launch_pit_dist_k(p, S, 1, part, s));
  HIPCHK(launch_pit_pick(part, F, S, B, dist, sel, s));
  return MISONET_OK;
}

// ---- STFT front-end ------------------------------------------------------------------------------------------------
// twiddle table + the > 64 KB dynamic-LDS attribute of stft_pack_k, per device (the table lives in the memory of the
// device that was current when it was first needed)
// (atomic: the getters read them outside the mutex -- acquire / release, a reader sees the table fully built or not at all)
static std::atomic<float*> g_twid[MAX_DEV] = {};
static std::atomic<float*> g_itwid[MAX_DEV] = {};
static std::mutex g_front_mu;
// Builds both tables on the CURRENT device (hipMalloc + synchronous copy + kernel attributes).  misonet_net_commit and
// misonet_pipeline_create call it, so every path that runs a network has them before its first asynchronous call -- a HIP
// graph may capture misonet_pipeline_run_wav / misonet_istft as the first call of a process.  Idempotent, thread-safe.
static int frontend_init() {
  const int d = cur_dev();
  std::lock_guard<std::mutex> lk(g_front_mu);
  if (!g_twid[d].load(std::memory_order_acquire)) {
    std::vector<float> tw((size_t)stft_twiddle_count());
    stft_build_twiddles(tw.data());
    float* p = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&p), tw.size() * sizeof(float)));
    HIPCHK(hipMemcpy(p, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(stft_init());
    g_twid[d].store(p, std::memory_order_release);
  }
  if (!g_itwid[d].load(std::memory_order_acquire)) {
    std::vector<float> tw((size_t)istft_twiddle_count());
    istft_build_twiddles(tw.data());
    float* p = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&p), tw.size() * sizeof(float)));
    HIPCHK(hipMemcpy(p, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(istft_init());
    g_itwid[d].store(p, std::memory_order_release);
  }
  return MISONET_OK;
}
int misonet_frontend_init(void) { return frontend_init(); }

// the table of the current device; a stand-alone misonet_stft / misonet_istft without any committed network on this device
// builds it on first use (that one call allocates and synchronises: not inside a stream capture; include/misonet.h)
static int get_twiddles(const float** out) {
  const int d = cur_dev();
  if (!g_twid[d].load(std::memory_order_acquire)) { int r = frontend_init(); if (r) return r; }
  *out = g_twid[d].load(std::memory_order_acquire);
  return MISONET_OK;
}
static int get_itwiddles(const float** out) {
  const int d = cur_dev();
  if (!g_itwid[d].load(std::memory_order_acquire)) { int r = frontend_init(); if (r) return r; }
  *out = g_itwid[d].load(std::memory_order_acquire);
  return MISONET_OK;
}

int misonet_istft(const void* spec_dev, int N, int T, void* out_i16_dev, float* out_f32_dev, misonet_stream stream) {
  if (!spec_dev || (!out_i16_dev && !out_f32_dev)) return fail(MISONET_EINVAL, "null argument");
  if (N <= 0 || T < 2) return fail(MISONET_EINVAL, "N must be positive and T >= 2 (got %d, %d)", N, T);
  const float* tw;
  int r = get_itwiddles(&tw);
  if (r) return r;
  HIPCHK(launch_istft(spec_dev, N, T, tw, reinterpret_cast<short*>(out_i16_dev), out_f32_dev,
                      reinterpret_cast<hipStream_t>(stream)));
  return MISONET_OK;
}

int misonet_stft_frames(int n_samples) { return n_samples > 0 ? n_samples / 64 + 1 : -1; }

long long misonet_stft_workspace_bytes(int B, int M, int n_samples) {
  const int T = misonet_stft_frames(n_samples);
  if (B <= 0 || M <= 0 || T <= 0) return -1;
  return (long long)B * 2 * M * 129 * frames_pitch(T) * 4;
}

int misonet_stft(const float* wav_dev, int B, int n_samples, int M, void* out_c64, void* ws, long long ws_bytes,
                 misonet_stream stream) {
  if (!wav_dev || !out_c64 || !ws) return fail(MISONET_EINVAL, "null argument");
  if (B <= 0 || M <= 0 || M > 64 || n_samples <= 0) return fail(MISONET_EINVAL, "bad B / M / n_samples");
  if (ws_bytes < misonet_stft_workspace_bytes(B, M, n_samples)) return fail(MISONET_ENOMEM, "workspace too small");
  const int T = misonet_stft_frames(n_samples), Tp = frames_pitch(T), F = 129;
  const float* tw;
  int r = get_twiddles(&tw);
  if (r) return r;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* planar = reinterpret_cast<float*>(ws);
  const long long bs = 2LL * M * F * Tp;
  HIPCHK(launch_stft_pack(wav_dev, B, n_samples, M, T, tw, planar, bs, Tp, F, 0, M, 1, s));
  HIPCHK(launch_unpack(planar, bs, Tp, M, T, F, reinterpret_cast<float2*>(out_c64), B, nullptr, s));
  return MISONET_OK;
}

// ---- fused pipeline ----------------------------------------------------------------------------------------------
struct misonet_pipeline {
  misonet_net* n1;
  misonet_net* n3;
  int M, S, ref_ch;
  float epsi;
};

struct PipeLayout {
  Layout L1, L3;
  long long off_ws1, off_ws3, off_clean, off_dist, off_sel, off_mvdr, total;
  long long clean_bstride;
};

static PipeLayout pipe_layout(const misonet_pipeline* p, int B, int T) {
  PipeLayout P;
  // MISO3 runs in MISO1's workspace (MISO1 is finished when MISO3 starts; what the steps in between read of it -- its input
  // and output planes -- is consumed before the MISO3 forward writes anything): only the MISO3 INPUT, which those steps
  // build while MISO1's planes are still being read, has its own memory
  P.L1 = make_layout(p->n1, B * p->M, T);
  if (p->n3) P.L3 = make_layout(p->n3, B * p->S, T, true);
  else { P.L3 = Layout(); P.L3.total_bytes = 0; }       // separation-only pipeline: no MISO3 workspace, no MISO3 input
  const int F = p->n1->cfg.n_freq, Tp = P.L1.Tp;
  long long o = 256;                                   // [0]: nan flag
  // PIT distances [B*M + B][S][S] followed by their per-bin partials [B*M + B][F][S][S] (mvdr.hip pit_dist_k)
  P.off_dist = o;  o += align_up((long long)(B * p->M + B) * p->S * p->S * (F + 1) * 8, 256);
  P.off_sel = o;   o += align_up((long long)(B * p->M * p->S * 2 + B * p->S) * 4, 256);
  P.off_mvdr = o;  o += align_up(mvdr_ws_bytes(B, p->S, F, p->M), 256);
  P.clean_bstride = (long long)2 * p->S * F * Tp;
  P.off_clean = o; o += align_up(P.clean_bstride * B * 4, 256);
  P.L3.in_ext_bstride = p->n3 ? (long long)p->n3->cfg.in_ch * F * Tp : 0;
  const long long in3_bytes = align_up(P.L3.in_ext_bstride * B * p->S * 4, 256);
  P.off_ws1 = o;   o += align_up(std::max(P.L1.total_bytes, P.L3.total_bytes), 256);
  P.off_ws3 = P.off_ws1;
  P.L3.in_ext_off = o - P.off_ws3;                     // relative to the (shared) workspace base
  o += in3_bytes;
  P.total = o;
  return P;
}

int misonet_pipeline_create(misonet_net* n1, misonet_net* n3, int num_mic, int num_spk, int ref_ch, float epsi,
                            misonet_pipeline** out) {
  // n3 == NULL: a separation-only pipeline (MISO1_Inference + alignments: the body shared by the reference's
  // Tester_Beamforming, tester.py:340-449) -- misonet_pipeline_run then only accepts out == NULL, bf_out == NULL
  if (!n1 || !out) return fail(MISONET_EINVAL, "null argument");
  if (num_spk < 1 || num_spk > 4) return fail(MISONET_EINVAL, "num_spk must be in [1, 4] (PIT enumerates num_spk! permutations)");
  if (num_mic < 2 || num_mic > 8) return fail(MISONET_EINVAL, "num_mic must be in [2, 8]");
  if (ref_ch < 0 || ref_ch >= num_mic) return fail(MISONET_EINVAL, "ref_ch out of range");
  if (n1->cfg.in_ch != 2 * num_mic || n1->cfg.out_ch != 2 * num_spk)
    return fail(MISONET_EINVAL, "MISO_1 geometry does not match num_mic/num_spk");
  if (n3 && (n3->cfg.in_ch != 2 * (num_mic + 2) || n3->cfg.out_ch != 2))
    return fail(MISONET_EINVAL, "MISO_3 geometry must be in_ch = 2*(num_mic+2), out_ch = 2");
  { int rf = frontend_init(); if (rf) return rf; }
  misonet_pipeline* p = new misonet_pipeline{n1, n3, num_mic, num_spk, ref_ch, epsi};
  *out = p;
  return MISONET_OK;
}
int misonet_pipeline_destroy(misonet_pipeline* p) { delete p; return MISONET_OK; }

long long misonet_pipeline_workspace_bytes(const misonet_pipeline* p, int B, int T) {
  if (!p || B <= 0 || T <= 0) return -1;
  return pipe_layout(p, B, T).total;
}

static int pipeline_run_impl(misonet_pipeline* p, const void* mix, const void* clean, const float* wav,
                             const float* clean_wav, int n_samples, int B, int T, void* out, void* bf_out,
                             void* miso1_out, void* ws, long long ws_bytes, misonet_stream stream) {
  if (!p || (!mix && !wav) || (!out && !miso1_out) || !ws) return fail(MISONET_EINVAL, "null argument");
  if (!p->n1->committed || (p->n3 && !p->n3->committed)) return fail(MISONET_ESTATE, "networks not committed");
  if (!p->n3 && (out || bf_out))
    return fail(MISONET_ESTATE, "this pipeline was created without MISO_3 (separation only): out and bf_out must be NULL");
  if (B <= 0 || T <= 0) return fail(MISONET_EINVAL, "B and T must be positive");
  const PipeLayout P = pipe_layout(p, B, T);
  if (ws_bytes < P.total) return fail(MISONET_ENOMEM, "workspace %lld < %lld bytes", ws_bytes, P.total);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  char* base = reinterpret_cast<char*>(ws);
  void* ws1 = base + P.off_ws1;
  void* ws3 = base + P.off_ws3;
  const int M = p->M, S = p->S, F = p->n1->cfg.n_freq, Tp = P.L1.Tp;
  misonet_net *n1 = p->n1, *n3 = p->n3;
  double* dist_shift = reinterpret_cast<double*>(base + P.off_dist);          // [B*M][S][S]
  double* dist_clean = dist_shift + (long long)B * M * S * S;                  // [B][S][S]
  double* part_shift = dist_clean + (long long)B * S * S;                      // [B*M][F][S][S]
  double* part_clean = part_shift + (long long)B * M * F * S * S;              // [B][F][S][S]
  int* sel_shift = reinterpret_cast<int*>(base + P.off_sel);                   // [B*M][S]
  int* sel_final = sel_shift + (long long)B * M * S;                           // [B*M][S]
  int* sel_clean = sel_final + (long long)B * M * S;                           // [B][S]
  HIPCHK(hipMemsetAsync(base, 0, 256, s));                                     // nan flag

  // 1. MISO1_Inference: the M circular shifts as one batch of B*M samples (tester.py:1033-1051)
  float* in1 = buf_ptr(P.L1, ws1, B_IN);
  const long long in1_bs = bstride(n1, P.L1, B_IN);
  const float* tw = nullptr;
  if (wav) { int rt = get_twiddles(&tw); if (rt) return rt; }
  if (wav) HIPCHK(launch_stft_pack(wav, B, n_samples, M, T, tw, in1, in1_bs, Tp, F, 0, M, M, s));
  else HIPCHK(launch_pack(reinterpret_cast<const float2*>(mix), B, M, T, F, in1, in1_bs, Tp, 0, M, M, s));
  int r = forward_planar(n1, P.L1, ws1, s);
  if (r) return r;
  float* out1 = buf_ptr(P.L1, ws1, B_OUT);
  const long long out1_bs = bstride(n1, P.L1, B_OUT);
  const long long plane = (long long)F * Tp;

  // 2. align the speakers of every shift to the reference-mic forward (tester.py:1043-1065)
  {
    PitArgs q;
    const float* anc = out1 + (long long)p->ref_ch * out1_bs;
    q.a = {anc, anc + S * plane, (long long)M * out1_bs, Tp, plane, 1};
    q.b = {out1, out1 + S * plane, out1_bs, Tp, plane, 1};
    q.B = B; q.F = F; q.T = T;
    HIPCHK(launch_pit_dist_k(q, S, M, part_shift, s));
    HIPCHK(launch_pit_pick(part_shift, F, S, B * M, dist_shift, sel_shift, s));
  }
  // 3. align to the clean references at ref_ch (tester.py:889-915), optional
  if (clean || clean_wav) {
    float* cl = reinterpret_cast<float*>(base + P.off_clean);
    if (clean_wav) HIPCHK(launch_stft_pack(clean_wav, B, n_samples, S, T, tw, cl, P.clean_bstride, Tp, F, 0, S, 1, s));
    else HIPCHK(launch_pack(reinterpret_cast<const float2*>(clean), B, S, T, F, cl, P.clean_bstride, Tp, 0, S, 1, s));
    // anchors = clean sources; candidates = shift-aligned ref-mic estimates.  The ref-mic forward is never
    // permuted by step 2 (its distance matrix has a zero diagonal), so the raw OUT1 planes are the candidates.
    PitArgs q;
    const float* cand = out1 + (long long)p->ref_ch * out1_bs;
    q.a = {cl, cl + S * plane, P.clean_bstride, Tp, plane, 1};
    q.b = {cand, cand + S * plane, (long long)M * out1_bs, Tp, plane, 1};
    q.B = B; q.F = F; q.T = T;
    HIPCHK(launch_pit_dist_k(q, S, 1, part_clean, s));
    HIPCHK(launch_pit_pick(part_clean, F, S, B, dist_clean, sel_clean, s));
  }
  HIPCHK(launch_compose_sel(sel_shift, (clean || clean_wav) ? sel_clean : nullptr, B, M, S, sel_final, s));

  if (out) {   // out == NULL: separation only (MISO1_Inference + alignments), e.g. for the utterance-wise beamformer
    // 4. MISO3 input = [mixture | beamformer | MISO1 estimate at ref_ch] (tester.py:936-939), B*S samples
    float* in3 = buf_ptr(P.L3, ws3, B_IN);
    const long long in3_bs = bstride(n3, P.L3, B_IN);
    HIPCHK(launch_assemble3(in1, in1_bs, out1, out1_bs, sel_final, B, M, S, p->ref_ch, F, Tp, in3, in3_bs, s));

    // 5. MVDR per aligned speaker (tester.py:917-924, 1071-1136); writes the beamformer planes of the MISO3 input
    {
      MvdrArgs a;
      a.mix = {in1, in1 + (long long)M * plane, (long long)M * in1_bs, Tp, plane, 1};   // shift-0 sample = un-rolled mixture
      a.est = out1; a.est_bstride = out1_bs; a.sel = sel_final;
      a.src = {nullptr, nullptr, 0, 0, 0, 1};
      a.S = S; a.B = B; a.F = F; a.M = M; a.T = T; a.Tp = Tp; a.epsi = p->epsi;
      COut co = {in3 + (long long)M * plane, in3 + (long long)(2 * M + 2) * plane, (long long)S * in3_bs, in3_bs, 1, Tp};
      ProfScope ps(s, PK_MVDR);
      HIPCHK(launch_mvdr(a, co, base + P.off_mvdr, s));
    }
    // (the aligned MISO1 estimates leave the shared workspace before MISO3 overwrites it)
    if (miso1_out)
      HIPCHK(launch_unpack_ex(out1, out1_bs, Tp, S, T, F, 0, S, 1, M, sel_final, reinterpret_cast<float2*>(miso1_out),
                              B * S * M, reinterpret_cast<int*>(base), s));
    // 6. MISO3 per speaker (tester.py:1231-1244), in MISO1's workspace
    r = forward_planar(n3, P.L3, ws3, s);
    if (r) return r;
    HIPCHK(launch_unpack(buf_ptr(P.L3, ws3, B_OUT), bstride(n3, P.L3, B_OUT), Tp, 1, T, F, reinterpret_cast<float2*>(out),
                         B * S, reinterpret_cast<int*>(base), s));
    if (bf_out)
      HIPCHK(launch_unpack_ex(in3, in3_bs, Tp, 1, T, F, M, 2 * M + 2, 0, 1, nullptr,
                              reinterpret_cast<float2*>(bf_out), B * S, reinterpret_cast<int*>(base), s));
  } else if (miso1_out) {
    HIPCHK(launch_unpack_ex(out1, out1_bs, Tp, S, T, F, 0, S, 1, M, sel_final, reinterpret_cast<float2*>(miso1_out),
                            B * S * M, reinterpret_cast<int*>(base), s));
  }
  return MISONET_OK;
}

int misonet_pipeline_run(misonet_pipeline* p, const void* mix, const void* clean, int B, int T, void* out, void* bf_out,
                         void* miso1_out, void* ws, long long ws_bytes, misonet_stream stream) {
  return pipeline_run_impl(p, mix, clean, nullptr, nullptr, 0, B, T, out, bf_out, miso1_out, ws, ws_bytes, stream);
}

int misonet_pipeline_run_wav(misonet_pipeline* p, const float* wav, const float* clean_wav, int B, int n_samples,
                             void* out, void* bf_out, void* miso1_out, void* ws, long long ws_bytes,
                             misonet_stream stream) {
  if (n_samples <= 0) return fail(MISONET_EINVAL, "n_samples must be positive");
  return pipeline_run_impl(p, nullptr, nullptr, wav, clean_wav, n_samples, B, misonet_stft_frames(n_samples), out, bf_out,
                           miso1_out, ws, ws_bytes, stream);
}

int misonet_pipeline_check(misonet_pipeline* p, const void* ws, misonet_stream stream) {
  if (!p || !ws) return fail(MISONET_EINVAL, "null argument");
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, ws, sizeof(int), hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream)));
  HIPCHK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
  if (flag) return fail(MISONET_ENAN, "NaN in pipeline output");
  return MISONET_OK;
}

// ---- per-launch profiling -----------------------------------------------------------------------------------------
int misonet_profile_begin(int max_launches) {
  if (max_launches <= 0) return fail(MISONET_EINVAL, "max_launches must be positive");
  Prof& pr = g_profs[cur_dev()];
  while (pr.pool.size() < (size_t)max_launches * 2) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    pr.pool.push_back(e);
  }
  pr.used = 0;
  pr.recs.clear();
  pr.overflow = false;
  if (!pr.on) g_prof_any.fetch_add(1);
  pr.on = true;
  return MISONET_OK;
}
int misonet_profile_end(double* ms_by_kind, long long* launches_by_kind) {
  Prof& pr = g_profs[cur_dev()];
  if (pr.on) g_prof_any.fetch_sub(1);
  pr.on = false;
  if (!ms_by_kind || !launches_by_kind) return fail(MISONET_EINVAL, "null argument");
  for (int k = 0; k < PK_N; ++k) { ms_by_kind[k] = 0.0; launches_by_kind[k] = 0; }
  for (const ProfRec& r : pr.recs) {
    HIPCHK(hipEventSynchronize(r.e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, r.e0, r.e1));
    ms_by_kind[r.kind] += ms;
    launches_by_kind[r.kind] += 1;
  }
  if (pr.overflow) return fail(MISONET_ENOMEM, "profile event pool exhausted");
  return MISONET_OK;
}

// ---- events --------------------------------------------------------------------------------------------------------
int misonet_event_create(void** ev) {
  hipEvent_t e;
  HIPCHK(hipEventCreate(&e));
  *ev = e;
  return MISONET_OK;
}
int misonet_event_record(void* ev, misonet_stream stream) {
  HIPCHK(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), reinterpret_cast<hipStream_t>(stream)));
  return MISONET_OK;
}
int misonet_event_elapsed_ms(void* start, void* stop, float* ms) {
  HIPCHK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
  HIPCHK(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
  return MISONET_OK;
}
int misonet_event_destroy(void* ev) {
  HIPCHK(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
  return MISONET_OK;
}

}  // extern "C"
