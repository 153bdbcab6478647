// conv3x3_bf16x3_dma: the bf16x3 3x3 convolution (conv_bf16.hip; reference model.py:401-482) fed by LDS-DMA.
//
// conv3x3_bf16x3 is bound by the per-lane work of staging its input -- 24 buffer loads, 48 FMAs, 24 bf16 splits and 12
// LDS writes per lane and K-chunk -- not by the matrix pipe (48 % busy) or HBM (profiles/r01_final_bf16x3_*).  This
// kernel removes that work from the consumer altogether:
//   * activations travel in the "oct" layout (kernels.hpp): the PRODUCER's epilogue stores bias + ELU output already
//     split into bf16 hi / lo, 8 channels of one frame per 16-byte unit -- exactly one lane's MFMA B operand;
//   * the instance norm of the input, x_n = x * scale[n][ci] + shift[n][ci], is folded into the weights instead of the
//     data: conv_wprep_k scales the weights per sample (W'[n] = W * scale[n]), splits them into bf16 hi / lo in LDS
//     image order, and reduces the shift term to a 9-entry border-aware bias table per (sample, output channel) --
//     zero padding is applied AFTER normalisation in the reference, so a tap that falls into the padding must not
//     contribute its W * shift (conv_epilogue.hpp picks the taps that are inside);
//   * staging is then a pure copy: `buffer_load_dwordx4 ... lds` moves 1 KB per wave instruction from HBM/L2 straight
//     into the LDS operand images, no VGPRs, no VALU.  Rows / frames / channel octets outside the image get an
//     out-of-range offset, for which the hardware writes zeros.
// Same GEMM roles, tile (4 rows x 128 frames x 32 output channels), LDS images and MFMA loop as conv3x3_bf16x3.
#include "kernels.hpp"
#include "conv_epilogue.hpp"
#include "conv_bf16_core.hpp"
#include <stdio.h>
#include <stdlib.h>

namespace mn {

#define MN_LDS(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_u(unsigned long long pa, unsigned bytes) {
  const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pa);
  const unsigned phi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)phi << 32) | plo), 0,
                                           __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// MFMA loop of one K-chunk (16 input channels) with ROW REUSE: wave w owns the 32 frames [t0 + 32w, t0 + 32w + 32) of
// all four output rows of the tile.  An input fragment B(R, kt) = 16 channels x 32 frames of staged row R, shifted by
// tap kt, serves every output row f' with f' + kf = R (up to three), and the 18 weight fragments of the chunk stay in
// registers for the whole chunk.  LDS reads per chunk and wave: 18 (A) + 2 * 3 * NR (B) x 1 KB instead of
// 18 + 72 with one row per wave -- the one-row mapping ran the LDS at ~100 B/clk/CU of its 128 B/clk, which is what
// capped the matrix pipe at ~50 % busy.
__device__ __forceinline__ void chunk_load_a(bf16x8 (&ah)[9], bf16x8 (&al)[9], const bf16x8* s_whi, const bf16x8* s_wlo,
                                             int half, int l31) {
  const int wb = half * 32 + l31;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    ah[tap] = s_whi[wb + tap * 64];
    al[tap] = s_wlo[wb + tap * 64];
  }
}

struct NoIssue { __device__ __forceinline__ void operator()(int) const {} };

// one 32x32x16 MFMA on 16-bit pieces: bf16 (bf16x3 mode) or fp16 (f16x3 mode); the LDS images are the same 16-byte units
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// `issue(st)` is called once per step, between the step's operand reads and its MFMAs: the software-pipelined kernel
// uses it to spread the LDS-DMA instructions of the NEXT chunk over the MFMA loop (a burst of 19 DMA instructions
// costs a wave ~3k cycles of issue time; in the shadow of the matrix pipe it costs nothing).
template <int NR, int SF, bool TR2, bool F16 = false, class Issue = NoIssue>
__device__ __forceinline__ void chunk_mfma_rows_a(f32x16 (&acc)[4], const bf16x8 (&ah)[9], const bf16x8 (&al)[9],
                                                  const bf16x8* s_xhi, const bf16x8* s_xlo, int wave, int half,
                                                  int l31, const Issue& issue = Issue()) {
  const int xb = half * TW + 3 + 32 * wave + l31;        // + R * 2 * TW + kt
  constexpr int NSTEP = 3 * NR;
  bf16x8 bh[2], bl[2];
#pragma unroll
  for (int st = -1; st < NSTEP; ++st) {
    if (st + 1 < NSTEP) {
      const int kt_ = (st + 1) / NR, R_ = (st + 1) % NR;
      bh[(st + 1) & 1] = s_xhi[xb + R_ * 2 * TW + kt_];
      bl[(st + 1) & 1] = s_xlo[xb + R_ * 2 * TW + kt_];
    }
    __builtin_amdgcn_sched_barrier(0);     // keep the next step's ds_reads AHEAD of this step's MFMAs
    if (st >= 0) {
      const int kt = st / NR, R = st % NR;
      const int cur = st & 1;
      // three passes (lo x hi, hi x lo, hi x hi) over the output rows fed by this fragment, so consecutive MFMAs hit
      // different accumulators
#pragma unroll
      for (int term = 0; term < 3; ++term) {
#pragma unroll
        for (int fr = 0; fr < 4; ++fr) {
#pragma unroll
          for (int kf = 0; kf < 3; ++kf) {
            const bool use = TR2 ? ((((fr + kf) & 1) == 0) && (((fr + kf) >> 1) == R)) : (SF * fr + kf == R);
            if (use) {
              const int tap = kt * 3 + kf;
              if (term == 0) acc[fr] = mfma16<F16>(al[tap], bh[cur], acc[fr]);
              else if (term == 1) acc[fr] = mfma16<F16>(ah[tap], bl[cur], acc[fr]);
              else acc[fr] = mfma16<F16>(ah[tap], bh[cur], acc[fr]);
            }
          }
        }
        if (term == 0) issue(st);          // behind the first MFMAs of the step: the matrix pipe is busy while these issue
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NR, int SF, bool TR2, bool F16 = false>
__device__ __forceinline__ void chunk_mfma_rows(f32x16 (&acc)[4], const bf16x8* s_xhi, const bf16x8* s_xlo,
                                                const bf16x8* s_whi, const bf16x8* s_wlo, int wave, int half,
                                                int l31) {
  bf16x8 ah[9], al[9];
  chunk_load_a(ah, al, s_whi, s_wlo, half, l31);
  chunk_mfma_rows_a<NR, SF, TR2, F16>(acc, ah, al, s_xhi, s_xlo, wave, half, l31);
}

template <int MODE, bool F16 = false>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16x3_dma(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (the LDS-DMA builtin has no host form)
  constexpr int COP = 32;
  constexpr int NR = MODE == 0 ? 6 : (MODE == 1 ? 9 : 3);
  constexpr int SF = MODE == 1 ? 2 : 1;
  constexpr bool TR2 = MODE == 2;
  constexpr int NPAIR = 2 * NR;                      // (row, channel-octet) pairs per chunk
  constexpr int XN = NPAIR * TW;                     // 16-byte units per input image (hi or lo)
  constexpr int WN = 9 * 2 * COP;                    // 16-byte units per weight image (hi or lo)
  constexpr int NXI = (XN + 255) / 256;              // DMA instructions per wave and input image
  constexpr int NWI = (2 * WN + 255) / 256;          // DMA instructions per wave for the weight images
  extern __shared__ __align__(16) unsigned char smem_b[];
  bf16x8* s_xhi = reinterpret_cast<bf16x8*>(smem_b);
  bf16x8* s_xlo = s_xhi + XN;
  bf16x8* s_whi = s_xlo + XN;                        // hi image followed by lo image (as packed in HBM)
  bf16x8* s_wlo = s_whi + WN;
  float* s_bs = reinterpret_cast<float*>(s_wlo + WN);          // [FT][2][16]: bias + shift, accumulator order
  float* s_bl = s_bs + FT * COP;                               // left-tap share
  float* s_br = s_bl + FT * COP;                               // right-tap share
  float* s_bt = s_br + FT * COP;                               // [COP][9]

  const ConvTile ct = conv_tile(a);
  if (!ct.valid) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t0 = ct.t_tile * TT;
  const int f0 = ct.f_tile * FT;
  const int n = ct.n;
  const int cg = ct.cg;
  const int T = a.T, Tp = a.Tp, Fin = a.Fin, Cin = a.Cin;
  const int nchunk = (Cin + CKB - 1) / CKB;
  const int fin0 = TR2 ? (f0 >> 1) - 1 : SF * f0 - a.padf;

  // ---- descriptors: hi and lo halves of this sample's input (octets beyond the slice are out of range -> zeros) ----
  const unsigned P16 = (unsigned)Fin * (unsigned)Tp * 16u;                   // bytes per octet plane
  const unsigned long long in_b = reinterpret_cast<unsigned long long>(a.in) + (unsigned long long)n * a.in_bstride * 4ull;
  const unsigned in_rec = (unsigned)((a.in_c0 + Cin) >> 3) * P16;
  const __amdgpu_buffer_rsrc_t rs_hi = make_rsrc_u(in_b, in_rec);
  const __amdgpu_buffer_rsrc_t rs_lo = make_rsrc_u(in_b + (unsigned long long)(a.in_sstride >> 3) * P16, in_rec);
  const unsigned wbytes = (unsigned)nchunk * (unsigned)(2 * WN) * 16u;       // one (sample, group) weight image set
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc_u(
      reinterpret_cast<unsigned long long>(a.wps) + (unsigned long long)n * a.wps_nstride + (unsigned long long)cg * wbytes, wbytes);

  // ---- per-lane source offsets of the units this lane copies (unit u -> pair p = u / TW, frame slot j = u % TW) ----
  unsigned xo[NXI];
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    const int u = (i * 4 + wave) * 64 + lane;
    const int p = u / TW, j = u - p * TW;
    const int fin = fin0 + (p >> 1);
    const int t = t0 - 4 + j;
    const bool ok = u < XN && fin >= 0 && fin < Fin && t >= 0 && t < T && j >= 3 && j <= TT + 4;
    xo[i] = ok ? ((unsigned)(((a.in_c0 >> 3) + (p & 1)) * Fin + fin) * (unsigned)Tp + (unsigned)t) * 16u : 0x80000000u;
  }
  const unsigned xstep = 2u * P16;                                           // two octets per K-chunk
  const unsigned wo = (unsigned)tid * 16u;

  // stage chunk KC: HBM/L2 -> LDS, no registers
#define DMA_STAGE(KC)                                                                                           \
  {                                                                                                             \
    _Pragma("unroll") for (int i = 0; i < NXI; ++i) {                                                           \
      const int ub = (i * 4 + wave) * 64;                                                                       \
      if (ub < XN) {                                                                                            \
        if (ub + 64 <= XN || ub + lane < XN) {                                                                  \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_hi, MN_LDS(s_xhi + ub), 16, xo[i], 0, 0, 0);              \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_lo, MN_LDS(s_xlo + ub), 16, xo[i], 0, 0, 0);              \
        }                                                                                                       \
      }                                                                                                         \
      xo[i] += xstep;                                                                                           \
    }                                                                                                           \
    const unsigned wsoff_ = (unsigned)(KC) * (unsigned)(2 * WN) * 16u;                                          \
    _Pragma("unroll") for (int i = 0; i < NWI; ++i) {                                                           \
      const int ub = (i * 4 + wave) * 64;                                                                       \
      if (ub < 2 * WN)                                                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, MN_LDS(s_whi + ub), 16, wo + (unsigned)i * 4096u, wsoff_, 0, 0); \
    }                                                                                                           \
  }

  const unsigned long long ts0 = clock64();
  const bool stamp = a.dbg_buf && tid == 0 && ct.t_tile == 3 && ct.f_tile == 5 && n == 7 && cg == 0;
  int si = 0;
#define STAMP() do { if (stamp && si < 60) a.dbg_buf[si++] = clock64() - ts0; } while (0)
  DMA_STAGE(0)
  STAMP();

  // ---- bias + folded instance-norm shift per output row (tables of conv_epilogue_rows) ----
  if (a.btab) {
    const float* bt = a.btab + (long long)n * a.btab_nstride + (long long)cg * COP * 9;
    for (int i = tid; i < COP * 9; i += 256) s_bt[i] = bt[i];
  }
  __syncthreads();
  if (lane < COP) {
    const int f = f0 + wave;
    float b3[3] = {0.f, 0.f, 0.f};
    if (a.btab) {
#pragma unroll
      for (int kf = 0; kf < 3; ++kf) {
        bool ok;
        if (TR2) {
          const int q = f + kf - 2;
          ok = (q >= 0) && !(q & 1) && (q >> 1) < Fin;
        } else {
          const int fi = SF * f + kf - a.padf;
          ok = fi >= 0 && fi < Fin;
        }
        if (ok) {
#pragma unroll
          for (int kt = 0; kt < 3; ++kt) b3[kt] += s_bt[lane * 9 + kt * 3 + kf];     // table entries are indexed like the weight images: kt * 3 + kf
        }
      }
    }
    b3[1] += a.bias[cg * COP + lane];
    // accumulator order: channel co = (i&3) + 8*(i>>2) + 4*h  ->  h = (co>>2)&1, i = (co&3) + 4*(co>>3)
    const int slot = (wave * 2 + ((lane >> 2) & 1)) * 16 + (lane & 3) + 4 * (lane >> 3);
    s_bs[slot] = b3[0] + b3[1] + b3[2];
    s_bl[slot] = b3[0];
    s_br[slot] = b3[2];
  }

  f32x16 acc[4];
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r4][r] = 0.f;
  const int half = lane >> 5, l31 = lane & 31;
  const bool wave_live = (t0 + 32 * wave < T);       // this wave's frames exist (ragged last tile)

  for (int kc = 0; kc < nchunk; ++kc) {
    // chunk kc complete (hipcc does not count LDS-DMA loads in its s_waitcnt bookkeeping: wait explicitly)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP();
    __syncthreads();
    STAMP();
    if (wave_live && !(a.dbg & 1)) {
      __builtin_amdgcn_s_setprio(1);
      chunk_mfma_rows<NR, SF, TR2, F16>(acc, s_xhi, s_xlo, s_whi, s_wlo, wave, half, l31);
      __builtin_amdgcn_s_setprio(0);
    }
    STAMP();
    __syncthreads();                                  // every wave is done reading the images
    STAMP();
    if (kc + 1 < nchunk) DMA_STAGE(kc + 1)
  }
#undef DMA_STAGE
  STAMP();

  // ---- epilogue (conv_epilogue.hpp) ----
  float* s_red = reinterpret_cast<float*>(smem_b);   // [4 waves][COP][2]
  if (!(a.dbg & 4)) conv_epilogue_rows<F16>(a, acc, n, cg, f0, t0 + 32 * wave, lane, s_red + wave * (COP * 2), s_bs, s_bl, s_br);
  STAMP();
  if (a.act) {
    __syncthreads();
    if (tid < COP * 2) {
      const int co_l = tid >> 1, which = tid & 1;
      const int co = cg * COP + co_l;
      if (co < a.Cout) {
        float tot = 0.f;
        for (int w = 0; w < 4; ++w) tot += s_red[(w * COP + co_l) * 2 + which];
        dstat_add(a.out_stats + (((long long)n * a.out_sstride + a.out_c0 + co) * 2 + which) * DS_NL, (double)tot);
      }
    }
  }
  STAMP();
  if (stamp) a.dbg_buf[63] = si;
#undef STAMP
#endif
}

// MFMA loop of one K-chunk with row reuse; the weight image stays valid for the whole chunk (fully double-buffered
// kernel), so the weight fragments are read from LDS on the fly, single-buffered: fragment (kt, kf) is reloaded with the
// next time tap's weights in the first step after its last use, which is at least one full step before its next use.
//   use(fr, kf, R): output row fr takes tap kf from staged input row R.
// The tile has FTR output rows (4; 2 for the stride-2 layers, whose 4-row tile would need 9 staged input rows).
template <int SF, bool TR2>
__device__ __forceinline__ constexpr bool mfma_use(int fr, int kf, int R) {
  constexpr int FTR = (SF == 2) ? 2 : 4;
  return fr < FTR && (TR2 ? ((((fr + kf) & 1) == 0) && (((fr + kf) >> 1) == R)) : (SF * fr + kf == R));
}
template <int NR, int SF, bool TR2>
__device__ __forceinline__ constexpr int mfma_last_r(int kf) {        // last staged row that uses tap kf
  int last = 0;
  for (int R = 0; R < NR; ++R)
    for (int fr = 0; fr < 4; ++fr)
      if (mfma_use<SF, TR2>(fr, kf, R)) last = R;
  return last;
}

// `issue(st)` runs once per step behind the first MFMAs of the step: the deferred epilogue of the PREVIOUS tile is
// spread over these slots, so its VALU work and stores execute in the shadow of the matrix pipe.
template <int NR, int SF, bool TR2, bool F16 = false, class Issue = NoIssue>
__device__ __forceinline__ void chunk_mfma_rows_w(f32x16 (&acc)[4], const bf16x8* s_xhi, const bf16x8* s_xlo,
                                                  const bf16x8* s_whi, const bf16x8* s_wlo, int wave, int half,
                                                  int l31, const Issue& issue = Issue()) {
  const int wb = half * 32 + l31;                        // + tap * 64
  const int xb = half * TW + 3 + 32 * wave + l31;        // + R * 2 * TW + kt
  constexpr int NSTEP = 3 * NR;
  bf16x8 ah[3], al[3], bh[2], bl[2];
#pragma unroll
  for (int kf = 0; kf < 3; ++kf) { ah[kf] = s_whi[wb + kf * 64]; al[kf] = s_wlo[wb + kf * 64]; }
#pragma unroll
  for (int st = -1; st < NSTEP; ++st) {
    if (st + 1 < NSTEP) {
      const int kt_ = (st + 1) / NR, R_ = (st + 1) % NR;
      bh[(st + 1) & 1] = s_xhi[xb + R_ * 2 * TW + kt_];
      bl[(st + 1) & 1] = s_xlo[xb + R_ * 2 * TW + kt_];
    }
    if (st >= 0) {
      // weights of the next time tap into the fragments whose last use was the PREVIOUS step
      const int kt = st / NR, R = st % NR;
#pragma unroll
      for (int kf = 0; kf < 3; ++kf) {
        const int lr = mfma_last_r<NR, SF, TR2>(kf);
        const bool now = (lr + 1 < NR) ? (R == lr + 1 && kt < 2) : (R == 0 && kt > 0);
        const int ktn = (lr + 1 < NR) ? kt + 1 : kt;
        if (now) {
          ah[kf] = s_whi[wb + (ktn * 3 + kf) * 64];
          al[kf] = s_wlo[wb + (ktn * 3 + kf) * 64];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // keep the next step's ds_reads AHEAD of this step's MFMAs
    if (st >= 0) {
      const int R = st % NR;
      const int cur = st & 1;
#pragma unroll
      for (int term = 0; term < 3; ++term) {
#pragma unroll
        for (int fr = 0; fr < 4; ++fr) {
#pragma unroll
          for (int kf = 0; kf < 3; ++kf) {
            if (mfma_use<SF, TR2>(fr, kf, R)) {
              if (term == 0) acc[fr] = mfma16<F16>(al[kf], bh[cur], acc[fr]);
              else if (term == 1) acc[fr] = mfma16<F16>(ah[kf], bl[cur], acc[fr]);
              else acc[fr] = mfma16<F16>(ah[kf], bh[cur], acc[fr]);
            }
          }
        }
        if (term == 0) issue(st);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// conv3x3_bf16x3_dma2: the software-pipelined form of the kernel above (stride-1 and transposed layers).
//   * ONE persistent workgroup per CU, 8 waves, 148 KB of LDS = two complete stages (input images + weight images of one
//     K-chunk each) + two sets of epilogue tables.  Workgroup b belongs to XCD b & 7 and walks that XCD's tile list
//     (conv_tile order) with stride = workgroups per XCD.
//   * waves 0-3 = CONSUMERS (one per SIMD): MFMAs and the tile epilogue, nothing else.
//     waves 4-7 = PRODUCERS (one per SIMD): LDS-DMA of the next chunk, the epilogue tables of the next tile, the float64
//     statistics atomics of the previous tile.  A DMA instruction holds its wave until the memory pipeline accepts it
//     (measured ~130 cycles each in a burst), so it must not come from a wave that feeds the matrix pipe.
//   * ONE workgroup barrier per K-chunk g: producers arrive when the DMA of chunk g has landed, consumers when the MFMAs
//     of chunk g - 1 are done (stage (g+1)&1 free).  Behind it producers put chunk g + 1 in flight while consumers run
//     the MFMAs of chunk g.  Chunks are numbered across tiles: the last chunk of a tile stages chunk 0 of the next one,
//     so tile set-up and the first load latency hide behind the last MFMAs and the epilogue.
template <int MODE, bool F16 = false>
__global__ __launch_bounds__(512, 1) void conv3x3_bf16x3_dma2(const ConvArgs a, int nslots) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int COP = 32;
  constexpr int SF = MODE == 1 ? 2 : 1;
  constexpr bool TR2 = MODE == 2;
  constexpr int FTR = MODE == 1 ? 2 : 4;             // output rows per tile
  constexpr int NR = MODE == 0 ? 6 : (MODE == 1 ? 5 : 3);   // staged input rows: SF * (FTR - 1) + 3, or 3 (transposed)
  constexpr int NPAIR = 2 * NR;
  constexpr int XN = NPAIR * TW;                     // 16-byte units per input image (hi or lo)
  constexpr int WN = 9 * 2 * COP;                    // 16-byte units per weight image (hi or lo)
  constexpr int SN = 2 * XN + 2 * WN;                // units per stage: [x hi | x lo | w hi | w lo]
  constexpr int NXI = (XN + 255) / 256;
  constexpr int NWI = (2 * WN + 255) / 256;
  extern __shared__ __align__(16) unsigned char smem_b[];
  bf16x8* s_stage = reinterpret_cast<bf16x8*>(smem_b);         // [2][SN]
  float* s_tab = reinterpret_cast<float*>(s_stage + 2 * SN);   // [2 sets][bs | bl | br][FT][2][16]
  float* s_red = s_tab + 2 * 3 * FT * COP;                     // [2 sets][4 waves][COP][2]
  float* s_ctr = s_red + 2 * 4 * COP * 2;                      // [4 sets][2 half-waves][16]: ELU(bias), the centre the
                                                               // activations are stored about (conv_epilogue.hpp)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int rw = wave & 3;                             // index inside the role
  const int half = lane >> 5, l31 = lane & 31;
  const int T = a.T, Tp = a.Tp, Fin = a.Fin, Cin = a.Cin;
  const int nchunk = (Cin + CKB - 1) / CKB;
  // deferred epilogue: rows 0-1 of a finished tile are post-processed during K-chunks 0-1 of the next tile
  const bool deferred = nchunk >= 2 && a.act && a.out_oct && !(a.dbg & 32);

  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const unsigned per = (unsigned)(a.ntx * a.nty * a.ncg);                    // tiles per sample
  const unsigned nk = (unsigned)((a.nsamp + 7 - (int)xcd) / 8) * per;        // tiles of this XCD's samples (n % 8 == xcd)
  unsigned k = slot;
  if (k >= nk) return;

  int t0, f0, n, cg;
#define TILE_COORDS(K)                                                                                          \
  {                                                                                                             \
    const unsigned grp_ = (K) / per;                                                                            \
    unsigned tile_ = (K) - grp_ * per;                                                                          \
    n = (int)(grp_ * 8u + xcd);                                                                                 \
    /* row tile fastest: the 32 workgroups of an XCD work on neighbouring rows of one frame tile (+1 % over frame */ \
    /* tile fastest; the fetch traffic is the same, the halo rows are simply still warm in the L2) */              \
    f0 = (int)(tile_ % (unsigned)a.nty) * FTR;                                                                  \
    tile_ /= (unsigned)a.nty;                                                                                   \
    cg = (int)(tile_ % (unsigned)a.ncg);                                                                        \
    t0 = (int)(tile_ / (unsigned)a.ncg) * TT;                                                                   \
  }

  if (producer) {
    // =============================================== producers ===============================================
    const unsigned P16 = (unsigned)Fin * (unsigned)Tp * 16u;
    const unsigned in_rec = (unsigned)((a.in_c0 + Cin) >> 3) * P16;
    const unsigned wbytes = (unsigned)nchunk * (unsigned)(2 * WN) * 16u;
    const unsigned xstep = 2u * P16;
    const unsigned wo = (unsigned)(tid & 255) * 16u;
    __amdgpu_buffer_rsrc_t rs_hi, rs_lo, rs_w;
    unsigned xo[NXI];

#define TILE_SETUP()                                                                                            \
  {                                                                                                             \
    const int fin0_ = TR2 ? (f0 >> 1) - 1 : SF * f0 - a.padf;                                                   \
    const unsigned long long in_b_ =                                                                            \
        reinterpret_cast<unsigned long long>(a.in) + (unsigned long long)n * a.in_bstride * 4ull;               \
    rs_hi = make_rsrc_u(in_b_, in_rec);                                                                         \
    rs_lo = make_rsrc_u(in_b_ + (unsigned long long)(a.in_sstride >> 3) * P16, in_rec);                         \
    rs_w = make_rsrc_u(reinterpret_cast<unsigned long long>(a.wps) + (unsigned long long)n * a.wps_nstride +    \
                           (unsigned long long)cg * wbytes, wbytes);                                            \
    _Pragma("unroll") for (int i = 0; i < NXI; ++i) {                                                           \
      const int u = (i * 4 + rw) * 64 + lane;                                                                   \
      const int p = u / TW, j = u - p * TW;                                                                     \
      const int fin = fin0_ + (p >> 1);                                                                         \
      const int t = t0 - 4 + j;                                                                                 \
      const bool ok = u < XN && fin >= 0 && fin < Fin && t >= 0 && t < T && j >= 3 && j <= TT + 4;              \
      xo[i] = ok ? ((unsigned)(((a.in_c0 >> 3) + (p & 1)) * Fin + fin) * (unsigned)Tp + (unsigned)t) * 16u      \
                 : 0x80000000u;                                                                                 \
    }                                                                                                           \
  }

    // chunk KC of the current tile -> stage SB
#define DMA_STAGE(KC, SB)                                                                                       \
  {                                                                                                             \
    bf16x8* st_ = s_stage + (SB) * SN;                                                                          \
    _Pragma("unroll") for (int i = 0; i < NXI; ++i) {                                                           \
      const int ub = (i * 4 + rw) * 64;                                                                         \
      if (ub < XN) {                                                                                            \
        if (ub + 64 <= XN || ub + lane < XN) {                                                                  \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_hi, MN_LDS(st_ + ub), 16, xo[i], 0, 0, 0);                \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_lo, MN_LDS(st_ + XN + ub), 16, xo[i], 0, 0, 0);           \
        }                                                                                                       \
      }                                                                                                         \
      xo[i] += xstep;                                                                                           \
    }                                                                                                           \
    const unsigned wsoff_ = (unsigned)(KC) * (unsigned)(2 * WN) * 16u;                                          \
    _Pragma("unroll") for (int i = 0; i < NWI; ++i) {                                                           \
      const int ub = (i * 4 + rw) * 64;                                                                         \
      if (ub < 2 * WN)                                                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, MN_LDS(st_ + 2 * XN + ub), 16, wo + (unsigned)i * 4096u, \
                                                 wsoff_, 0, 0);                                                 \
    }                                                                                                           \
  }

    // epilogue tables of the current tile, set TS: producer wave rw builds output row f0 + rw, lane = output channel
#define TILE_TABLES(TS, CS)                                                                                     \
  {                                                                                                             \
    if (lane < COP) {                                                                                           \
      const int f_ = f0 + rw;                                                                                   \
      float b3[3] = {0.f, 0.f, 0.f};                                                                            \
      if (a.btab) {                                                                                             \
        const float* bt_ = a.btab + (long long)n * a.btab_nstride + (long long)(cg * COP + lane) * 9;           \
        _Pragma("unroll") for (int kf = 0; kf < 3; ++kf) {                                                      \
          bool ok_;                                                                                             \
          if (TR2) {                                                                                            \
            const int q_ = f_ + kf - 2;                                                                         \
            ok_ = (q_ >= 0) && !(q_ & 1) && (q_ >> 1) < Fin;                                                    \
          } else {                                                                                              \
            const int fi_ = SF * f_ + kf - a.padf;                                                              \
            ok_ = fi_ >= 0 && fi_ < Fin;                                                                        \
          }                                                                                                     \
          _Pragma("unroll") for (int kt = 0; kt < 3; ++kt) {                                                    \
            const float v_ = bt_[kt * 3 + kf];                                                                  \
            b3[kt] += ok_ ? v_ : 0.f;                                                                           \
          }                                                                                                     \
        }                                                                                                       \
      }                                                                                                         \
      b3[1] += a.bias[cg * COP + lane];                                                                         \
      /* accumulator order: channel co = (i&3) + 8*(i>>2) + 4*h  ->  h = (co>>2)&1, i = (co&3) + 4*(co>>3) */    \
      const int slot_ = (rw * 2 + ((lane >> 2) & 1)) * 16 + (lane & 3) + 4 * (lane >> 3);                       \
      float* tb_ = s_tab + (TS) * (3 * FT * COP);                                                               \
      const float wsc_ = F16 ? a.wscale : 1.f;   /* the accumulators START at these values: same scale as W' */     \
      tb_[slot_] = (b3[0] + b3[1] + b3[2]) * wsc_;                                                              \
      tb_[FT * COP + slot_] = b3[0] * wsc_;                                                                     \
      tb_[2 * FT * COP + slot_] = b3[2] * wsc_;                                                                 \
      if (rw == 0) s_ctr[(CS) * COP + slot_] = a.act ? elu_fast(a.bias[cg * COP + lane]) : 0.f;                 \
    }                                                                                                           \
  }

    // float64 statistics of a finished tile (producer wave 0): sum of the four consumer partials per channel
#define TILE_STATS(PN, PCG, RS)                                                                                 \
  {                                                                                                             \
    if (a.act && rw == 0) {                                                                                     \
      const float* sr_ = s_red + (RS) * (4 * COP * 2);                                                          \
      const int co_l = lane >> 1, which = lane & 1;                                                             \
      const int co = (PCG) * COP + co_l;                                                                        \
      if (co < a.Cout) {                                                                                        \
        float tot = 0.f;                                                                                        \
        for (int w = 0; w < 4; ++w) tot += sr_[(w * COP + co_l) * 2 + which];                                   \
        dstat_add(a.out_stats + (((long long)(PN) * a.out_sstride + a.out_c0 + co) * 2 + which) * DS_NL, (double)tot); \
      }                                                                                                         \
    }                                                                                                           \
  }

    TILE_COORDS(k)
    TILE_SETUP()
    DMA_STAGE(0, 0)
    TILE_TABLES(0, 0)
    unsigned g = 0, ti = 0;
    // finished tiles whose statistics are still to be flushed: p1 = previous tile, p2 = the one before.  With the
    // deferred epilogue the partials of tile j are complete only after chunk 1 of tile j + 1.
    int p1_n = 0, p1_cg = 0, p2_n = 0, p2_cg = 0;
    for (;;) {
      bool more = false;
      const int c_n = n, c_cg = cg;
      for (int kc = 0; kc < nchunk; ++kc, ++g) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // chunk g has landed (hipcc does not count LDS-DMA loads)
        __syncthreads();                                       // barrier g
        if (kc == 0) {
          if (!deferred && ti >= 1) TILE_STATS(p1_n, p1_cg, (ti + 1) & 1)
          if (deferred && ti >= 2) TILE_STATS(p2_n, p2_cg, ti & 1)
        }
        if (kc + 1 < nchunk) {
          DMA_STAGE(kc + 1, (g + 1) & 1)
        } else {
          k += (unsigned)nslots;
          more = k < nk;
          if (more) {
            TILE_COORDS(k)
            TILE_SETUP()
            DMA_STAGE(0, (g + 1) & 1)
            TILE_TABLES((ti + 1) & 1, (ti + 1) & 3)
          }
        }
      }
      p2_n = p1_n; p2_cg = p1_cg;
      p1_n = c_n; p1_cg = c_cg;
      ++ti;
      if (!more) break;
    }
    __syncthreads();                                           // final barrier: the last epilogue is done
    if (deferred && ti >= 2) TILE_STATS(p2_n, p2_cg, ti & 1)
    TILE_STATS(p1_n, p1_cg, (ti + 1) & 1)
#undef TILE_SETUP
#undef DMA_STAGE
#undef TILE_TABLES
#undef TILE_STATS
  } else {
    // =============================================== consumers ===============================================
    TILE_COORDS(k)
    if (deferred) {
      const unsigned OP16 = (unsigned)a.Fout * (unsigned)Tp * 16u;           // bytes per output octet plane
      // rows 0 .. NDEF-1 of a tile are post-processed during the next tile, the others right after its last MFMA.
      // Register budget: 16 VGPRs per deferred row on top of acc 64 + statistics 32 + operands 40; 3 rows already
      // spill inside the MFMA loops and are no faster than 2.
      constexpr int NDEF = 2;
      f32x16 prev[4];
      EpiState es;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[r4][r] = 0.f;
      es.pmt = 0.f; es.pvo0 = 0u; es.prows = 0; es.prow_b = (unsigned)Tp * 16u;
#pragma unroll
      for (int i = 0; i < 8; ++i) { es.s1[i] = f32x2{0.f, 0.f}; es.s2[i] = f32x2{0.f, 0.f}; }
      es.okk0 = es.okk1 = false;
      es.dsc = F16 ? a.descale : 1.f;
      es.ctr = s_ctr;
#pragma unroll
      for (int i = 0; i < 2; ++i) { es.PH[i][0] = es.PH[i][1] = es.PL[i][0] = es.PL[i][1] = 0u; }
      __amdgpu_buffer_rsrc_t prs_h = make_rsrc_u(reinterpret_cast<unsigned long long>(a.out), 0u);
      __amdgpu_buffer_rsrc_t prs_l = prs_h;
      unsigned g = 0, ti = 0;
      int tcount = 0;
      for (;;) {
        if (a.dbg_buf && tid == 0 && blockIdx.x == 8 && tcount < 10) {
          a.dbg_buf[40 + tcount] = wall_clock64(); a.dbg_buf[50 + tcount] = clock64(); ++tcount;
        }
        f32x16 acc[4];
        const bool wave_live = (t0 + 32 * wave < T);
        float* sred_prev = s_red + ((ti + 1) & 1) * (4 * COP * 2) + wave * (COP * 2);   // partials of tile ti - 1
        // one K-chunk; ROW = row of the previous tile post-processed in the shadow of its MFMAs (-1: none)
#define RUN_CHUNK(ROW)                                                                                          \
        {                                                                                                       \
          __syncthreads();                                     /* barrier g: stage g & 1 holds chunk g */         \
          if ((ROW) == 0) {                                                                                     \
            const float* tb = s_tab + (ti & 1) * (3 * FT * COP);                                                \
            conv_acc_init_rows(acc, t0 + 32 * wave, T, lane, tb, tb + FT * COP, tb + 2 * FT * COP);             \
          }                                                                                                     \
          const bf16x8* st_ = s_stage + (g & 1) * SN;                                                           \
          auto hook = [&](int stp) {                                                                            \
            if ((ROW) >= 0)                                                                                     \
              conv_epi_step<((ROW) >= 0 ? (ROW) : 0), 3 * NR, F16>(stp, prev, es, prs_h, prs_l, OP16, lane); \
          };                                                                                                    \
          if (wave_live && !(a.dbg & 1)) {                                                                      \
            __builtin_amdgcn_s_setprio(1);                                                                      \
            chunk_mfma_rows_w<NR, SF, TR2, F16>(acc, st_, st_ + XN, st_ + 2 * XN, st_ + 2 * XN + WN, wave, half, l31, hook); \
            __builtin_amdgcn_s_setprio(0);                                                                      \
          } else {                                                                                              \
            _Pragma("unroll") for (int stp = 0; stp < 3 * NR; ++stp) hook(stp);                                 \
          }                                                                                                     \
          ++g;                                                                                                  \
        }
        static_assert(NDEF == 2, "chunk schedule below is written for two deferred rows");
        RUN_CHUNK(0)
        RUN_CHUNK(1)
        conv_epi_reduce(es, sred_prev, lane);
        for (int kc = 2; kc < nchunk; ++kc) RUN_CHUNK(-1)
        // ---- the finished tile: rows NDEF..3 now, rows 0..NDEF-1 become the "previous tile" ----
        {
          const int tw = t0 + 32 * wave;
          const int t = tw + l31;
          const int cbase = cg * COP;
          es.pmt = (t < T) ? 1.f : 0.f;
          es.prows = (a.Fout - f0) < FTR ? (a.Fout - f0) : FTR;
          es.pvo0 = (unsigned)(f0 * Tp + t) * 16u + (unsigned)(half + (cbase >> 3)) * OP16;
          es.okk0 = cbase + (0 + half) * 8 < a.Cout;
          es.okk1 = cbase + (2 + half) * 8 < a.Cout;
          es.ctr = s_ctr + (ti & 3) * COP;
          const unsigned long long pa = reinterpret_cast<unsigned long long>(a.out) +
                                        (unsigned long long)n * a.out_bstride * 4ull + (unsigned long long)(a.out_c0 >> 3) * OP16;
          const unsigned nrec = (unsigned)(a.Cout >> 3) * OP16;
          prs_h = make_rsrc_u(pa, nrec);
          prs_l = make_rsrc_u(pa + (unsigned long long)(a.out_sstride >> 3) * OP16, nrec);
          if (NDEF <= 2 && FTR > 2) {
#pragma unroll
            for (int stp = 0; stp < 3 * NR; ++stp) conv_epi_step<2, 3 * NR, F16>(stp, acc, es, prs_h, prs_l, OP16, lane);
          }
          if (NDEF <= 3 && FTR > 3) {
#pragma unroll
            for (int stp = 0; stp < 3 * NR; ++stp) conv_epi_step<3, 3 * NR, F16>(stp, acc, es, prs_h, prs_l, OP16, lane);
          }
#pragma unroll
          for (int r4 = 0; r4 < NDEF; ++r4) prev[r4] = acc[r4];
        }
        ++ti;
        k += (unsigned)nslots;
        if (k >= nk) break;
        TILE_COORDS(k)
      }
      // ---- flush: the deferred rows of the last tile, not overlapped ----
      {
        float* sred_prev = s_red + ((ti + 1) & 1) * (4 * COP * 2) + wave * (COP * 2);
#pragma unroll
        for (int stp = 0; stp < 3 * NR; ++stp) conv_epi_step<0, 3 * NR, F16>(stp, prev, es, prs_h, prs_l, OP16, lane);
#pragma unroll
        for (int stp = 0; stp < 3 * NR; ++stp) conv_epi_step<1, 3 * NR, F16>(stp, prev, es, prs_h, prs_l, OP16, lane);
        conv_epi_reduce(es, sred_prev, lane);
      }
#undef RUN_CHUNK
    } else {
    unsigned g = 0, ti = 0;
      int tcount = 0;
      for (;;) {
        if (a.dbg_buf && tid == 0 && blockIdx.x == 8 && tcount < 10) {
          a.dbg_buf[40 + tcount] = wall_clock64(); a.dbg_buf[50 + tcount] = clock64(); ++tcount;
        }
        f32x16 acc[4];
        const bool wave_live = (t0 + 32 * wave < T);             // this consumer's frames exist (ragged last tile)
        const bool stamp = a.dbg_buf && tid == 0 && t0 == 3 * TT && f0 == 5 * FT && n == 7 && cg == 0;
        const unsigned long long ts0 = clock64();
        int si = 0;
#define STAMP() do { if (stamp && si < 36) a.dbg_buf[si++] = clock64() - ts0; } while (0)
        for (int kc = 0; kc < nchunk; ++kc, ++g) {
          STAMP();
          __syncthreads();                                       // barrier g: stage g & 1 holds chunk g
          STAMP();
          if (kc == 0) {                                         // accumulators start at bias + folded shift (tables of this
            const float* tb = s_tab + (ti & 1) * (3 * FT * COP); // tile: written by the producers before barrier g)
            conv_acc_init_rows(acc, t0 + 32 * wave, T, lane, tb, tb + FT * COP, tb + 2 * FT * COP);
          }
          if (wave_live && !(a.dbg & 1)) {
            const bf16x8* st = s_stage + (g & 1) * SN;
            __builtin_amdgcn_s_setprio(1);
            chunk_mfma_rows_w<NR, SF, TR2, F16>(acc, st, st + XN, st + 2 * XN, st + 2 * XN + WN, wave, half, l31);
            __builtin_amdgcn_s_setprio(0);
          }
        }
        STAMP();
        if (!(a.dbg & 4))
          conv_epilogue_rows_nb<2, F16>(a, acc, n, cg, f0, t0 + 32 * wave, lane, s_red + (ti & 1) * (4 * COP * 2) + wave * (COP * 2), FTR,
                                        a.act ? s_ctr + (ti & 3) * COP : nullptr);
        STAMP();
        if (stamp) a.dbg_buf[63] = si;
#undef STAMP
        ++ti;
        k += (unsigned)nslots;
        if (k >= nk) break;
        TILE_COORDS(k)
      }
    }
    __syncthreads();                                           // final barrier
  }
#undef TILE_COORDS
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_wprep_k: fold the instance norm of a layer's input into per-sample weights.
//   wf   : [ncg][nchunk][9][2][32][8] float32 (LDS image order, zero padded), shared by all samples
//   wps  : [n][ncg][nchunk][hi|lo][9][2][32][8] bf16   = split(wf * scale[n][ci])
//   btab : [n][ncg*32][9] float32                     = sum_ci wf[..ci..] * shift[n][ci]   (float64 accumulation)
// One workgroup of 288 threads per (sample, group); thread p owns (tap, output channel) = (p / 32, p % 32) and walks the
// input channels in a fixed order, so the table is deterministic.
// F16 (f16x3 mode): the pieces are fp16 and carry the layer's power-of-two scale `wscale` (so that the lo pieces of small
// weights stay in fp16's normal range); btab stays unscaled (the consumer scales the accumulator start values).
template <bool F16>
__global__ __launch_bounds__(288) void conv_wprep_k(const float* wf, const dstat_t* in_stats, int in_sstride, int in_c0,
                                                    int Cin, int ident_c, int Fin, int T, int nchunk, int ncg,
                                                    unsigned short* wps, long long wps_nstride_b, float* btab,
                                                    long long btab_nstride, float wscale) {
  extern __shared__ float2 s_nrm[];                  // [nchunk*16] (scale, shift)
  const int n = blockIdx.x / ncg, cg = blockIdx.x - n * ncg;
  const int tid = threadIdx.x;
  for (int c = tid; c < nchunk * CKB; c += 288) {
    float mean = 0.f, rstd = (c < Cin) ? 1.f : 0.f;
    if (c >= ident_c && c < Cin) {
      const dstat_t* st = in_stats + ((long long)n * in_sstride + in_c0 + c) * (2 * DS_NL);
      const double cnt = (double)Fin * (double)T;
      const double m = dstat_read(st) / cnt;
      double var = dstat_read(st + DS_NL) / cnt - m * m;
      var = var > 0.0 ? var : 0.0;
      mean = (float)m;
      rstd = (float)(1.0 / sqrt(var + (double)IN_EPS));
    }
    s_nrm[c] = make_float2(rstd, -mean * rstd);
  }
  __syncthreads();
  const int tap = tid >> 5, co = tid & 31;
  const float* wsrc = wf + (long long)cg * nchunk * (9 * 2 * 32 * 8);
  unsigned short* wdst = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(wps) + (long long)n * wps_nstride_b) +
                         (long long)cg * nchunk * (2 * 9 * 2 * 32 * 8);
  double bsum = 0.0;
  constexpr int U = 4;                               // chunks per batch: all 16 loads of a batch are issued together
  for (int kc0 = 0; kc0 < nchunk; kc0 += U) {
    float4 wl[U][2][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kc = (kc0 + u < nchunk) ? kc0 + u : nchunk - 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int unit = (tap * 2 + h) * 32 + co;
        const float4* src = reinterpret_cast<const float4*>(wsrc + ((long long)kc * (9 * 2 * 32) + unit) * 8);
        wl[u][h][0] = src[0];
        wl[u][h][1] = src[1];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kc = kc0 + u;
      if (kc < nchunk) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int unit = (tap * 2 + h) * 32 + co;
          const float4 w0 = wl[u][h][0], w1 = wl[u][h][1];
          const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
          float ws[8];
          float bs = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float2 m = s_nrm[kc * CKB + h * 8 + e];
            ws[e] = F16 ? wv[e] * m.x * wscale : wv[e] * m.x;
            bs = fmaf(wv[e], m.y, bs);
          }
          bsum += (double)bs;
          u32x4_t hi, lo;
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            unsigned a_, b_;
            split_pair_x<F16>(ws[2 * e2], ws[2 * e2 + 1], a_, b_);
            hi[e2] = a_; lo[e2] = b_;
          }
          u32x4_t* d = reinterpret_cast<u32x4_t*>(wdst + (long long)kc * (2 * 9 * 2 * 32 * 8));
          d[unit] = hi;
          d[9 * 2 * 32 + unit] = lo;
        }
      }
    }
  }
  btab[(long long)n * btab_nstride + (long long)(cg * 32 + co) * 9 + tap] = (float)bsum;
}

static size_t dma2_lds_bytes(int NR) {
  return (size_t)(2 * (2 * NR * 2 * TW + 2 * 9 * 2 * 32)) * 16 + (size_t)(2 * 3 * FT * 32 + 2 * 4 * 32 * 2 + 4 * 32) * sizeof(float);
}

static size_t dma_lds_bytes(int NR) {
  return (size_t)(2 * NR * 2 * TW + 2 * 9 * 2 * 32) * 16 + (size_t)(3 * FT * 32 + 32 * 9) * sizeof(float);
}

template <int MODE, bool F16>
static hipError_t dma_set_attr() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16x3_dma<MODE, F16>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16x3_dma2<MODE, F16>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t conv_bf16_dma_init() {
  hipError_t e;
  if ((e = dma_set_attr<0, false>()) != hipSuccess) return e;
  if ((e = dma_set_attr<1, false>()) != hipSuccess) return e;
  if ((e = dma_set_attr<2, false>()) != hipSuccess) return e;
  if ((e = dma_set_attr<0, true>()) != hipSuccess) return e;
  if ((e = dma_set_attr<1, true>()) != hipSuccess) return e;
  return dma_set_attr<2, true>();
}

hipError_t launch_conv_wprep(const ConvArgs& a, const float* wf, int n_samples, hipStream_t s) {
  const int nchunk = (a.Cin + CKB - 1) / CKB;
  if (a.in_oct == 4)
    hipLaunchKernelGGL(conv_wprep_k<true>, dim3(n_samples * a.ncg), dim3(288), (size_t)nchunk * CKB * sizeof(float2), s, wf,
                       a.in_stats, a.in_sstride, a.in_c0, a.Cin, a.ident_c, a.Fin, a.T, nchunk, a.ncg,
                       reinterpret_cast<unsigned short*>(const_cast<void*>(a.wps)), a.wps_nstride,
                       const_cast<float*>(a.btab), a.btab_nstride, a.wscale);
  else
    hipLaunchKernelGGL(conv_wprep_k<false>, dim3(n_samples * a.ncg), dim3(288), (size_t)nchunk * CKB * sizeof(float2), s, wf,
                       a.in_stats, a.in_sstride, a.in_c0, a.Cin, a.ident_c, a.Fin, a.T, nchunk, a.ncg,
                       reinterpret_cast<unsigned short*>(const_cast<void*>(a.wps)), a.wps_nstride,
                       const_cast<float*>(a.btab), a.btab_nstride, 1.f);
  return hipGetLastError();
}

hipError_t launch_conv_bf16_dma(const ConvArgs& a_in, int n_samples, hipStream_t s) {
  ConvArgs a = a_in;
  if (!a.in_oct || !a.wps || a.cop != 32 || (a.Cin & 7) || (a.in_c0 & 7) || (a.in_sstride & 7)) return hipErrorInvalidValue;
  if (a.out_oct && ((a.Cout & 7) || (a.out_c0 & 7) || (a.out_sstride & 7))) return hipErrorInvalidValue;
  if (a.in_oct != 1 && a.in_oct != 4) return hipErrorInvalidValue;
  if (a.out_oct && a.out_oct != a.in_oct) return hipErrorInvalidValue;       // same piece format on both sides
  const bool f16 = a.in_oct == 4;
  if (a.NR != conv_rows(a.sf, a.tr2)) return hipErrorInvalidValue;
  {
    static int dbg = -1;
    if (dbg < 0) dbg = exp_env("MISONET_WS_DEBUG", 0);
    a.dbg = dbg;
  }
  const dim3 grid = conv_grid(a, n_samples, TT, FT, conv_xcd_env());
  size_t lds = dma_lds_bytes(a.NR);
  {
    static int pad = -1;                          // MISONET_DMA_ONE=1: pad the LDS request so only ONE workgroup fits a CU (experiments)
    if (pad < 0) pad = exp_env("MISONET_DMA_ONE", 0);
    if (pad && lds < 100 * 1024) lds = 100 * 1024;
  }
  const int mode = a.tr2 ? 2 : (a.sf == 2 ? 1 : 0);
  // MISONET_TIMELINE=1: clock64() stamps of one workgroup of the first few (Cin = 96, F = 63) launches (experiments only)
  static int tl_env = -1;
  static int tl_done = 0;
  static unsigned long long* tl_buf = nullptr;
  if (tl_env < 0) tl_env = exp_env("MISONET_TIMELINE", 0);
  const bool do_tl = tl_env && tl_done < 3 && mode == 0 && a.Cin == 96 && a.Fout == 63 && n_samples >= 8;
  a.dbg_buf = nullptr;
  if (do_tl) {
    if (!tl_buf && hipMalloc(reinterpret_cast<void**>(&tl_buf), 64 * 8) != hipSuccess) tl_buf = nullptr;
    if (tl_buf) { (void)hipMemsetAsync(tl_buf, 0, 64 * 8, s); a.dbg_buf = tl_buf; }
  }
  static int dma2_env = -1;
  if (dma2_env < 0) dma2_env = exp_env("MISONET_DMA2", 1);
  const int nchunk_l = (a.Cin + CKB - 1) / CKB;
  // stride-2 layers run the pipelined kernel on 2-row tiles (5 staged rows; 9 rows x two stages do not fit the LDS)
  const bool dma2_ok = mode != 1 || (nchunk_l >= 2 && a.act && a.out_oct);
  if (dma2_env && dma2_ok) {
    if (mode == 1) (void)conv_grid(a, n_samples, TT, 2, 1);          // tile geometry with 2 output rows
    // persistent launch: one workgroup per CU, capped by the largest per-XCD tile list
    const int g_cus = device_cus();
    if (g_cus <= 0) return hipErrorUnknown;
    const long long nk_max = (long long)((n_samples + 7) / 8) * a.ntx * a.nty * a.ncg;
    int nslots = g_cus / 8;
    if (nslots < 1) nslots = 1;
    if (nslots > nk_max) nslots = (int)nk_max;
    const dim3 pgrid((unsigned)(8 * nslots), 1, 1);
    const size_t lds2 = dma2_lds_bytes(mode == 1 ? 5 : a.NR);
    if (f16) {
      if (mode == 0) hipLaunchKernelGGL((conv3x3_bf16x3_dma2<0, true>), pgrid, dim3(512), lds2, s, a, nslots);
      else if (mode == 1) hipLaunchKernelGGL((conv3x3_bf16x3_dma2<1, true>), pgrid, dim3(512), lds2, s, a, nslots);
      else hipLaunchKernelGGL((conv3x3_bf16x3_dma2<2, true>), pgrid, dim3(512), lds2, s, a, nslots);
    } else {
      if (mode == 0) hipLaunchKernelGGL((conv3x3_bf16x3_dma2<0, false>), pgrid, dim3(512), lds2, s, a, nslots);
      else if (mode == 1) hipLaunchKernelGGL((conv3x3_bf16x3_dma2<1, false>), pgrid, dim3(512), lds2, s, a, nslots);
      else hipLaunchKernelGGL((conv3x3_bf16x3_dma2<2, false>), pgrid, dim3(512), lds2, s, a, nslots);
    }
  } else if (f16) {
    if (mode == 0) hipLaunchKernelGGL((conv3x3_bf16x3_dma<0, true>), grid, dim3(256), lds, s, a);
    else if (mode == 1) hipLaunchKernelGGL((conv3x3_bf16x3_dma<1, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv3x3_bf16x3_dma<2, true>), grid, dim3(256), lds, s, a);
  } else if (mode == 0) hipLaunchKernelGGL((conv3x3_bf16x3_dma<0, false>), grid, dim3(256), lds, s, a);
  else if (mode == 1) hipLaunchKernelGGL((conv3x3_bf16x3_dma<1, false>), grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL((conv3x3_bf16x3_dma<2, false>), grid, dim3(256), lds, s, a);
  if (do_tl && tl_buf) {
    unsigned long long h[64];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, tl_buf, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[timeline-dma] Cin=%d Fout=%d n=%d stamps=%llu:", a.Cin, a.Fout, n_samples, h[63]);
    for (unsigned long long i = 0; i < h[63] && i < 40; ++i) fprintf(stderr, " %llu", h[i]);
    fprintf(stderr, " | tile starts (100 MHz wall clock, deltas):");
    for (int i = 41; i < 50 && h[i]; ++i) fprintf(stderr, " %llu/%llu", h[i] - h[i - 1], h[i + 10] - h[i + 9]);
    fprintf(stderr, "\n");
    ++tl_done;
  }
  return hipGetLastError();
}

}  // namespace mn
