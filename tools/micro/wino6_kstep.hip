// Prototype of the main loop of a Winograd F(2x2,3x3) conv in bf16x6 arithmetic (the 2-D form): how many cycles
// does ONE K-step (16 input channels x 32 output channels x 32 tiles per wave = 96 v_mfma_f32_32x32x16_bf16) take when the input
// transform, the instance norm and the exact three-piece split of the transformed operand are VALU work of the same wave?
//
//   per lane (tile n = lane % 32, K-half kh = lane / 32): 8 channels x 4 x 4 raw patch from LDS -> t = (a * ra + sa) +- (b * rb + sb)
//   (two staged rows of one position row xi) -> V[xi][nu] (4 adds) -> hi / mid / lo bf16 pieces (11 VALU per pair of values)
//   -> B operands of the 4 positions of xi; A operands = U pieces [pos][piece][co][kh] from LDS; 6 MFMAs per position into the
//   16 fixed accumulators a[16 p : 16 p + 15].  Position row xi + 1 is prepared while the 24 MFMAs of xi run (one sub-step of
//   11-13 VALU behind every MFMA).  All operands are LDS-resident (no DMA, no epilogue): this is the CEILING of such a kernel.
// Checked against a float64 host evaluation of the same K-step (layouts, split, term set); prints cycles per K-step.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <utility>
#include <type_traits>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define LP(T, off) (reinterpret_cast<const __attribute__((address_space(3))) T*>(off))

constexpr int NCH = 16, NROWS = 10, RW = 68;
constexpr unsigned RAW_B = 0;                                   // [16][10][68] float
constexpr unsigned U_B = NCH * NROWS * RW * 4;                  // [16 pos][3][64 lanes] x 16 bytes
constexpr unsigned NRM_B = U_B + 16 * 3 * 64 * 16;              // [16][10] float2
constexpr unsigned LDS_BYTES = NRM_B + NCH * NROWS * 8;

#define ACL1(b) "a" #b
#define ACLOB \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
  "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63", \
  "a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95", \
  "a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127", \
  "a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159", \
  "a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191", \
  "a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223", \
  "a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"

template <int P>
__device__ __forceinline__ void mfma6(u32x4 a, u32x4 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(16 * P), "n"(16 * P + 15), "v"(a), "v"(b) : ACLOB);
}
template <int I> __device__ __forceinline__ float agpr_get() { float x; asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(I)); return x; }
template <int I> __device__ __forceinline__ void agpr_zero() { asm volatile("v_accvgpr_write_b32 a[%0], 0" ::"n"(I)); }
template <int I, int N> struct Rep { template <class F> __device__ __forceinline__ static void run(F&& f) { f(std::integral_constant<int, I>{}); if constexpr (I + 1 < N) Rep<I + 1, N>::run(f); } };

__device__ __forceinline__ unsigned fu(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float uf(unsigned x) { return __uint_as_float(x); }

struct Pipe {
  u32x4 B[2][4][3];        // [buffer][nu][piece]: component q = channels 2 q | 2 q + 1 (bf16 pair)
  u32x4 U[2][3];           // [buffer][piece] of one position
  float ra[2][4], rb[2][4];   // raw words of the two channels of a pair unit: staged rows (a, b) of the position row
  f2 na[2], nb[2];
  float v[4][2];           // V[nu][channel of the pair]
};

// position row xi: staged rows (a, b) and the sign of b
template <int XI> struct RowsOf { static constexpr int a = XI == 0 ? 0 : (XI == 2 ? 2 : 1), b = XI == 0 ? 2 : (XI == 1 ? 2 : (XI == 2 ? 1 : 3)); static constexpr bool plus = XI == 1; };

template <int XI, int UNIT>
__device__ __forceinline__ void fetch_raw(Pipe& p, const unsigned (&rawc)[8], unsigned nrmb) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int j = 2 * UNIT + c;
    const unsigned oa = rawc[j] + (unsigned)(RowsOf<XI>::a * RW * 4), ob = rawc[j] + (unsigned)(RowsOf<XI>::b * RW * 4);
    const f2 a0 = *LP(f2, oa), a1 = *LP(f2, oa + 8), b0 = *LP(f2, ob), b1 = *LP(f2, ob + 8);
    p.ra[c][0] = a0.x; p.ra[c][1] = a0.y; p.ra[c][2] = a1.x; p.ra[c][3] = a1.y;
    p.rb[c][0] = b0.x; p.rb[c][1] = b0.y; p.rb[c][2] = b1.x; p.rb[c][3] = b1.y;
    p.na[c] = *LP(f2, nrmb + (unsigned)((j * NROWS + RowsOf<XI>::a) * 8));
    p.nb[c] = *LP(f2, nrmb + (unsigned)((j * NROWS + RowsOf<XI>::b) * 8));
  }
}
// T: t[k] = (a[k] * ra + sa) +- (b[k] * rb + sb); V[nu] = (t0 - t2, t1 + t2, t2 - t1, t1 - t3)
template <int XI>
__device__ __forceinline__ void transform(Pipe& p, int c) {
  const float s = RowsOf<XI>::plus ? p.na[c].y + p.nb[c].y : p.na[c].y - p.nb[c].y;
  const float rb = RowsOf<XI>::plus ? p.nb[c].x : -p.nb[c].x;
  float t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) t[k] = fmaf(p.rb[c][k], rb, fmaf(p.ra[c][k], p.na[c].x, s));
  p.v[0][c] = t[0] - t[2]; p.v[1][c] = t[1] + t[2]; p.v[2][c] = t[2] - t[1]; p.v[3][c] = t[1] - t[3];
}
// S: exact split of the pair (v0, v1) into three bf16 pairs (truncation split: hi = top 16 bits, the rest is exact in fp32)
__device__ __forceinline__ void split(float v0, float v1, unsigned& H, unsigned& M, unsigned& L) {
  // fixed-position pieces: h = top 8 significant bits, w = top 16, mid = w - h, lo = v - w (both exact, <= 8 significant bits each);
  // every piece depends on v only: dependency depth 3 (and -> sub -> perm) instead of 5
  const float h0 = uf(fu(v0) & 0xffff0000u), h1 = uf(fu(v1) & 0xffff0000u);
  const float w0 = uf(fu(v0) & 0xffffff00u), w1 = uf(fu(v1) & 0xffffff00u);
  const float m0 = w0 - h0, m1 = w1 - h1, q0 = v0 - w0, q1 = v1 - w1;
  H = __builtin_amdgcn_perm(fu(v1), fu(v0), 0x07060302u);
  M = __builtin_amdgcn_perm(fu(m1), fu(m0), 0x07060302u);
  L = __builtin_amdgcn_perm(fu(q1), fu(q0), 0x07060302u);
}
template <int POS>
__device__ __forceinline__ void fetch_u(Pipe& p, int buf, unsigned ub) {
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) p.U[buf][pc] = *LP(u32x4, ub + (unsigned)((POS * 3 + pc) * 1024));
}

// the 24 slots of position row XI (operands in buffer XI & 1); behind slot s: sub-step s % 6 of pair unit s / 6 of row XN = XI + 1 mod 4
template <int XI, int S>
__device__ __forceinline__ void slot(Pipe& p, const unsigned (&rawb)[8], unsigned nrmb, unsigned ub, bool work, bool mm = true, bool ld = true) {
  constexpr int nu = S / 6, term = S % 6, P = XI * 4 + nu, cb = XI & 1, nb = cb ^ 1, XN = (XI + 1) & 3;
  constexpr int ub_ = P & 1;
  // small terms first: (Ul Vh) (Uh Vl) (Um Vm) (Um Vh) (Uh Vm) (Uh Vh)
  constexpr int ap = term == 0 ? 2 : ((term == 2 || term == 3) ? 1 : 0);
  constexpr int bp = term == 1 ? 2 : ((term == 2 || term == 4) ? 1 : 0);
  if (mm) mfma6<P>(p.U[ub_][ap], p.B[cb][nu][bp]);
  else asm volatile("; operands %0 %1" ::"v"(p.U[ub_][ap]), "v"(p.B[cb][nu][bp]) : "memory");
  if (term == 1) fetch_u<(P + 1) & 15>(p, ub_ ^ 1, ub);          // the next position's U pieces (its first use is 5 slots away)
  if (work) {
    constexpr int unit = S / 6, k = S % 6;
    if (k == 0) transform<XN>(p, 0);
    if (k == 1) transform<XN>(p, 1);
    if (k >= 2) {
      unsigned H, M, L;
      split(p.v[k - 2][0], p.v[k - 2][1], H, M, L);
      p.B[nb][k - 2][0][unit] = H; p.B[nb][k - 2][1][unit] = M; p.B[nb][k - 2][2][unit] = L;
    }
    if (k == 2 && !ld) { asm volatile("" : "+v"(p.ra[0][0]), "+v"(p.ra[1][0]), "+v"(p.rb[0][0]), "+v"(p.rb[1][0])); }
    if (k == 2 && ld) {                                           // both channels transformed: the raw registers are free
      if (unit < 3) fetch_raw<XN, (unit + 1) & 3>(p, rawb, nrmb);
      else fetch_raw<(XN + 1) & 3, 0>(p, rawb, nrmb);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int XI>
__device__ __forceinline__ void row(Pipe& p, const unsigned (&rawb)[8], unsigned nrmb, unsigned ub, bool work, bool mm = true, bool ld = true) {
  Rep<0, 24>::run([&](auto s) __attribute__((always_inline)) { slot<XI, decltype(s)::value>(p, rawb, nrmb, ub, work, mm, ld); });
}

// MODE 0: full K-step; 1: MFMAs + U fetch only (no VALU work, no raw fetch); 2: no MFMAs; 3: no MFMAs, no raw / norm fetches
template <int MODE>
__global__ __launch_bounds__(256, 1) void kstep_k(const unsigned* img, float* out, unsigned long long* cyc, int nsteps, int dump) {
  extern __shared__ __align__(16) unsigned smem[];
  for (unsigned i = threadIdx.x; i < LDS_BYTES / 4; i += 256) smem[i] = img[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tc = lane & 31, kh = lane >> 5;
  const unsigned lds0 = (unsigned)(unsigned long long)((__attribute__((address_space(3))) void*)(smem));
  unsigned rawb[8];                             // one base register per channel (ds_read2_b64 offsets reach 2 KB)
#pragma unroll
  for (int j = 0; j < 8; ++j) { rawb[j] = lds0 + RAW_B + (unsigned)(((kh * 8 + j) * NROWS + 2 * wave) * RW + 2 * tc) * 4u; asm volatile("" : "+v"(rawb[j])); }
  unsigned nrmb = lds0 + NRM_B + (unsigned)((kh * 8) * NROWS + 2 * wave) * 8u;
  unsigned ub = lds0 + U_B + (unsigned)lane * 16u;
  asm volatile("" : "+v"(nrmb)); asm volatile("" : "+v"(ub));
  Rep<0, 256>::run([&](auto i) __attribute__((always_inline)) { agpr_zero<decltype(i)::value>(); });
  Pipe p;
  // prologue: the operands of position row 0 (buffer 0), the U pieces of position 0, the raw words of (row 1, unit 0)
  Rep<0, 4>::run([&](auto u) __attribute__((always_inline)) {
    constexpr int unit = decltype(u)::value;
    fetch_raw<0, unit>(p, rawb, nrmb);
    transform<0>(p, 0); transform<0>(p, 1);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      unsigned H, M, L;
      split(p.v[nu][0], p.v[nu][1], H, M, L);
      p.B[0][nu][0][unit] = H; p.B[0][nu][1][unit] = M; p.B[0][nu][2][unit] = L;
    }
  });
  fetch_u<0>(p, 0, ub);
  fetch_raw<1, 0>(p, rawb, nrmb);
  asm volatile("s_nop 4");
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < nsteps; ++it) {
    row<0>(p, rawb, nrmb, ub, MODE != 1, MODE < 2, MODE != 3);
    row<1>(p, rawb, nrmb, ub, MODE != 1, MODE < 2, MODE != 3);
    row<2>(p, rawb, nrmb, ub, MODE != 1, MODE < 2, MODE != 3);
    row<3>(p, rawb, nrmb, ub, MODE != 1, MODE < 2, MODE != 3);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_nop 15\n\ts_nop 7");
  if (dump && blockIdx.x == 0) {
    float* o = out + (size_t)threadIdx.x * 256;
    Rep<0, 256>::run([&](auto i) __attribute__((always_inline)) { o[decltype(i)::value] = agpr_get<decltype(i)::value>(); });
  }
  if (threadIdx.x == 0 && blockIdx.x == 1) cyc[0] = t1 - t0;
  if (MODE >= 1) { unsigned x = 0; for (int b = 0; b < 2; ++b) for (int n = 0; n < 4; ++n) for (int q = 0; q < 3; ++q) x ^= p.B[b][n][q][0] ^ p.B[b][n][q][1] ^ p.B[b][n][q][2] ^ p.B[b][n][q][3]; if (x == 0x12345678u) out[0] = 1.f; }     // (keeps the operand registers alive in MODE 1)
}

static unsigned short bf16_trunc(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }
static float bf16_val(unsigned short h) { unsigned u = (unsigned)h << 16; float x; memcpy(&x, &u, 4); return x; }

int main() {
  std::vector<unsigned> img(LDS_BYTES / 4);
  float* raw = reinterpret_cast<float*>(img.data() + RAW_B / 4);
  unsigned short* U = reinterpret_cast<unsigned short*>(img.data() + U_B / 4);
  float* nrm = reinterpret_cast<float*>(img.data() + NRM_B / 4);
  srand(7);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (int i = 0; i < NCH * NROWS * RW; ++i) raw[i] = 3.f * rnd() + 0.5f;
  for (int c = 0; c < NCH; ++c)
    for (int r = 0; r < NROWS; ++r) { nrm[(c * NROWS + r) * 2] = 0.5f + 0.5f * fabsf(rnd()); nrm[(c * NROWS + r) * 2 + 1] = 0.3f * rnd(); }
  std::vector<float> Uf(16 * 32 * NCH);                                 // [pos][co][c]
  for (auto& x : Uf) x = 0.2f * rnd();
  for (int pos = 0; pos < 16; ++pos)
    for (int co = 0; co < 32; ++co)
      for (int c = 0; c < NCH; ++c) {
        const float w = Uf[(pos * 32 + co) * NCH + c];
        const unsigned short h = bf16_trunc(w); const float r1 = w - bf16_val(h);
        const unsigned short m = bf16_trunc(r1); const float r2 = r1 - bf16_val(m);
        const unsigned short l = bf16_trunc(r2);
        const int lane = co + 32 * (c / 8), j = c % 8;
        U[(((pos * 3 + 0) * 64 + lane) * 8) + j] = h; U[(((pos * 3 + 1) * 64 + lane) * 8) + j] = m; U[(((pos * 3 + 2) * 64 + lane) * 8) + j] = l;
      }
  unsigned* d_img; float* d_out; unsigned long long* d_cyc;
  hipMalloc(&d_img, LDS_BYTES); hipMalloc(&d_out, 256 * 256 * 4); hipMalloc(&d_cyc, 8);
  hipMemcpy(d_img, img.data(), LDS_BYTES, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)kstep_k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipFuncSetAttribute((const void*)kstep_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipFuncSetAttribute((const void*)kstep_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipFuncSetAttribute((const void*)kstep_k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  // ---- correctness: one K-step of workgroup 0 against float64 ----
  hipLaunchKernelGGL(kstep_k<0>, dim3(2), dim3(256), LDS_BYTES, 0, d_img, d_out, d_cyc, 1, 1);
  std::vector<float> got(256 * 256);
  if (hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 1; }
  double maxerr = 0, maxref = 0;
  const int BT[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
  for (int wave = 0; wave < 4; ++wave)
    for (int tcn = 0; tcn < 32; ++tcn) {
      double V[16][NCH];
      for (int c = 0; c < NCH; ++c) {
        double d[4][4];
        for (int i = 0; i < 4; ++i)
          for (int k = 0; k < 4; ++k) {
            const int r = 2 * wave + i;
            d[i][k] = (double)raw[(c * NROWS + r) * RW + 2 * tcn + k] * nrm[(c * NROWS + r) * 2] + nrm[(c * NROWS + r) * 2 + 1];
          }
        for (int xi = 0; xi < 4; ++xi)
          for (int nu = 0; nu < 4; ++nu) {
            double s = 0;
            for (int i = 0; i < 4; ++i)
              for (int k = 0; k < 4; ++k) s += BT[xi][i] * d[i][k] * BT[nu][k];
            V[xi * 4 + nu][c] = s;
          }
      }
      for (int pos = 0; pos < 16; ++pos)
        for (int co = 0; co < 32; ++co) {
          double s = 0;
          for (int c = 0; c < NCH; ++c) s += (double)Uf[(pos * 32 + co) * NCH + c] * V[pos][c];
          // D layout: lane = tile + 32 * ((co / 4) & 1), register i = (co % 4) + 4 * (co / 8)
          const int lane = tcn + 32 * ((co >> 2) & 1), i = (co & 3) + 4 * (co >> 3);
          const double g = got[(size_t)(wave * 64 + lane) * 256 + pos * 16 + i];
          maxerr = fmax(maxerr, fabs(g - s)); maxref = fmax(maxref, fabs(s));
        }
    }
  printf("one K-step vs float64: max |err| = %.3e, max |ref| = %.3e  (rel %.2e)\n", maxerr, maxref, maxerr / maxref);
  // ---- timing ----
  const int N = 200;
  unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(kstep_k<0>, dim3(256), dim3(256), LDS_BYTES, 0, d_img, d_out, d_cyc, N, 0);
    hipMemcpy(&c0, d_cyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(kstep_k<1>, dim3(256), dim3(256), LDS_BYTES, 0, d_img, d_out, d_cyc, N, 0);
    hipMemcpy(&c1, d_cyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(kstep_k<2>, dim3(256), dim3(256), LDS_BYTES, 0, d_img, d_out, d_cyc, N, 0);
    hipMemcpy(&c2, d_cyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(kstep_k<3>, dim3(256), dim3(256), LDS_BYTES, 0, d_img, d_out, d_cyc, N, 0);
    hipMemcpy(&c3, d_cyc, 8, hipMemcpyDeviceToHost);
  }
  printf("no MFMAs: %8.1f cycles; no MFMAs, no raw / norm fetches: %8.1f\n", (double)c2 / N, (double)c3 / N);
  printf("full K-step (96 MFMAs + transform + norm + split): %8.1f cycles   (MFMAs + U fetch only: %8.1f; 96 x 32 = 3072)\n", (double)c0 / N, (double)c1 / N);
  printf("the direct bf16x6 kernel issues 216 MFMAs = 6912 matrix cycles for the same 128 outputs x 16 channels x 32 output channels\n");
  return 0;
}
