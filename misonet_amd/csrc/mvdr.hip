// MVDR beamformer (reference tester.py:1071-1136, 1138-1167, 1211-1228) and PIT speaker alignment
// (tester.py:1043-1065, 889-915) as an HBM-bound batched small-matrix path: one workgroup per (utterance, bin,
// speaker) streams the M x T slabs once, accumulates both spatial covariance matrices in registers, reduces them
// across the wavefronts, and solves the M x M Hermitian eigenproblem in float64 (cyclic complex Jacobi).
//
//   k1 mvdr_scm_eig : Phi_s = S S^H / T, Phi_n = (Y-S)(Y-S)^H / T  (tester.py:1091-1100, 1138-1152)
//                     principal eigenvector of Phi_s                 (tester.py:1107-1115)
//                     d <- d / d[0];  d <- d * sqrt(M / ||d||_2)      (tester.py:1119-1123; norm, not norm^2)
//   k2 mvdr_solve   : sequential-in-f phase correction               (tester.py:1154-1167)
//                     w = (Phi_n + eps I)^-1 d / (d^H (Phi_n + eps I)^-1 d)   (tester.py:1211-1225)
//   k3 mvdr_apply   : out[t] = sum_m conj(w_m) Y_m[t]                (tester.py:1227-1228)
#include "kernels.hpp"

namespace mn {


struct cd { double re, im; };
__device__ inline cd cmul(cd a, cd b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ inline cd cmulc(cd a, cd b) { return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; }  // a*conj(b)
__device__ inline cd cadd(cd a, cd b) { return {a.re + b.re, a.im + b.im}; }
__device__ inline cd csub(cd a, cd b) { return {a.re - b.re, a.im - b.im}; }
__device__ inline cd cconj(cd a) { return {a.re, -a.im}; }
__device__ inline double cabs2(cd a) { return a.re * a.re + a.im * a.im; }
__device__ inline cd cdiv(cd a, cd b) {
  const double d = cabs2(b);
  return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}

// workspace layout (doubles): steer0 [B][S][F][M][2] | phin [B][S][F][M][M][2] | steer1 [B][S][F][M][2] | w [B][S][F][M][2]
__host__ __device__ inline long long ws_steer0(int B, int S, int F, int M) { return 0; }
__host__ __device__ inline long long ws_phin(int B, int S, int F, int M) { return (long long)B * S * F * M * 2; }
__host__ __device__ inline long long ws_steer1(int B, int S, int F, int M) {
  return ws_phin(B, S, F, M) + (long long)B * S * F * M * M * 2;
}
__host__ __device__ inline long long ws_w(int B, int S, int F, int M) {
  return ws_steer1(B, S, F, M) + (long long)B * S * F * M * 2;
}
long long mvdr_ws_bytes(int B, int S, int F, int M) {
  return (ws_w(B, S, F, M) + (long long)B * S * F * M * 2) * (long long)sizeof(double);
}

// source estimate of aligned speaker `spk` at microphone m: pointers to its frame row for bin f
__device__ inline void src_row(const MvdrArgs& a, int b, int f, int m, int spk, const float*& re, const float*& im,
                               int& st) {
  if (a.est) {
    const int n = b * a.M + m;
    const int q = a.sel ? a.sel[n * a.S + spk] : spk;
    const long long plane = (long long)a.F * a.Tp;
    const float* base = a.est + (long long)n * a.est_bstride + (long long)f * a.Tp;
    re = base + (long long)q * plane;
    im = base + (long long)(a.S + q) * plane;
    st = 1;
  } else {
    const long long off = (long long)b * a.src.sb + (long long)f * a.src.sf + (long long)m * a.src.sm;
    re = a.src.re + off;
    im = a.src.im + off;
    st = a.src.st;
  }
}

template <int M>
__global__ __launch_bounds__(256) void mvdr_scm_eig(const MvdrArgs a, double* ws) {
  constexpr int NT = M * (M + 1) / 2;
  __shared__ double s_part[4][2 * NT * 2];
  __shared__ double s_A[M][M][2];
  __shared__ double s_V[M][M][2];
  const int f = blockIdx.x, b = blockIdx.y, spk = blockIdx.z;
  const int tid = threadIdx.x;
  const float *sre[M], *sim[M], *yre[M], *yim[M];
  int sst = 1;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    src_row(a, b, f, m, spk, sre[m], sim[m], sst);
    const long long off = (long long)b * a.mix.sb + (long long)f * a.mix.sf + (long long)m * a.mix.sm;
    yre[m] = a.mix.re + off;
    yim[m] = a.mix.im + off;
  }
  const int yst = a.mix.st;
  float ps[NT][2], pn[NT][2];
#pragma unroll
  for (int i = 0; i < NT; ++i) { ps[i][0] = ps[i][1] = pn[i][0] = pn[i][1] = 0.f; }
  const int nthr = blockDim.x, nw = nthr >> 6;
  for (int t = tid; t < a.T; t += nthr) {
    float xr[M], xi[M], nr[M], ni[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      xr[m] = sre[m][(long long)t * sst];
      xi[m] = sim[m][(long long)t * sst];
      nr[m] = yre[m][(long long)t * yst] - xr[m];      // noise = mix - source (tester.py:1095)
      ni[m] = yim[m][(long long)t * yst] - xi[m];
    }
    int k = 0;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j, ++k) {
        ps[k][0] += xr[i] * xr[j] + xi[i] * xi[j];
        ps[k][1] += xi[i] * xr[j] - xr[i] * xi[j];
        pn[k][0] += nr[i] * nr[j] + ni[i] * ni[j];
        pn[k][1] += ni[i] * nr[j] - nr[i] * ni[j];
      }
  }
  // block reduction in float64
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double v1 = ps[k][c], v2 = pn[k][c];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        v1 += __shfl_xor(v1, m, 64);
        v2 += __shfl_xor(v2, m, 64);
      }
      if (lane == 0) {
        s_part[wave][(k * 2 + c)] = v1;
        s_part[wave][2 * NT + (k * 2 + c)] = v2;
      }
    }
  __syncthreads();
  const double invT = 1.0 / (double)a.T;
  const long long idx = ((long long)(b * a.S + spk) * a.F + f);
  for (int e = tid; e < 2 * NT * 2; e += nthr) {
    double v = 0.0;
    for (int w = 0; w < nw; ++w) v += s_part[w][e];
    v *= invT;
    const int which = e / (2 * NT);            // 0: Phi_s, 1: Phi_n
    const int kc = e - which * 2 * NT;
    const int k = kc >> 1, c = kc & 1;
    // k -> (i, j), i >= j
    int i = 0, rem = k;
    while (rem > i) { rem -= (i + 1); ++i; }
    const int j = rem;
    if (which == 0) {
      s_A[i][j][c] = v;
      s_A[j][i][c] = c ? -v : v;
      if (i == j && c) s_A[i][i][1] = 0.0;
    } else {
      double* pn_o = ws + ws_phin(a.B, a.S, a.F, M) + idx * (M * M * 2);
      pn_o[(i * M + j) * 2 + c] = (i == j && c) ? 0.0 : v;
      if (i != j) pn_o[(j * M + i) * 2 + c] = c ? -v : v;
    }
  }
  for (int e = tid; e < M * M; e += nthr) {
    const int i = e / M, j = e - i * M;
    s_V[i][j][0] = (i == j) ? 1.0 : 0.0;
    s_V[i][j][1] = 0.0;
  }
  __syncthreads();
  // Complex Jacobi on the Hermitian s_A with a PARALLEL (round-robin) ordering; eigenvectors accumulate in the columns of
  // s_V.  A sweep is ME - 1 rounds of M / 2 DISJOINT pivot pairs; disjoint rotations commute, so a round applies
  // A <- R^H A R with R = R_1 R_2 .. in three phases separated by barriers: (0) every lane of pair j computes that
  // pair's rotation from the untouched matrix, (1) lane (j, k) updates row k of the columns (p_j, q_j) of A and V,
  // (2) lane (j, k) updates column k of the rows (p_j, q_j) of A.  M (M - 1) / 2 rotations of a sweep cost ME - 1 round
  // latencies instead of M (M - 1) / 2: the 6 x 6 problem was ~280 of the kernel's 320 us of dependent float64 work.
  {
    constexpr int ME = (M + 1) & ~1;                       // players of the tournament (a dummy when M is odd)
    constexpr int NPR = ME / 2;                            // pairs per round
    const int pj = tid / M, pk = tid - pj * M;             // this lane's pair slot and index (valid when pj < NPR)
    double scale = 0.0;
    for (int i = 0; i < M; ++i) scale += fabs(s_A[i][i][0]);
    for (int sweep = 0; sweep < 16; ++sweep) {
      double off = 0.0;
      for (int p = 0; p < M; ++p)
        for (int q = p + 1; q < M; ++q) off += s_A[p][q][0] * s_A[p][q][0] + s_A[p][q][1] * s_A[p][q][1];
      if (off <= 1e-28 * scale * scale || off == 0.0) break;    // uniform: every lane reads the same elements
      for (int rd = 0; rd < ME - 1; ++rd) {
        // circle method: slot 0 pairs the fixed player ME - 1 with rd, slot i pairs (rd + i) with (rd - i) modulo ME - 1
        int p = 0, q = 0;
        bool act = pj < NPR;
        if (act) {
          if (pj == 0) { p = rd; q = ME - 1; }
          else { p = (rd + pj) % (ME - 1); q = (rd - pj + (ME - 1)) % (ME - 1); }
          if (p > q) { const int t = p; p = q; q = t; }
          act = q < M;                                     // the dummy player sits out
        }
        cd Rpp = {1.0, 0.0}, Rpq = {0.0, 0.0}, Rqp = {0.0, 0.0}, Rqq = {1.0, 0.0};
        if (act) {
          const cd apq = {s_A[p][q][0], s_A[p][q][1]};
          const double g = sqrt(cabs2(apq));
          if (g > 1e-300) {
            const double app = s_A[p][p][0], aqq = s_A[q][q][0];
            const double tau = (aqq - app) / (2.0 * g);
            const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = tt * cs;
            const cd ph = {apq.re / g, apq.im / g};        // e^{i phi}
            // R restricted to (p,q): Rpp = c, Rpq = s, Rqp = -s e^{-i phi}, Rqq = c e^{-i phi}
            Rpp = {cs, 0.0}; Rpq = {sn, 0.0};
            Rqp = {-sn * ph.re, sn * ph.im}; Rqq = {cs * ph.re, -cs * ph.im};
          } else {
            act = false;
          }
        }
        __syncthreads();                                   // every lane has read its pivot before anyone overwrites it
        if (act) {                                         // A <- A R (columns p, q), V <- V R: row k = pk
          const int k = pk;
          const cd akp = {s_A[k][p][0], s_A[k][p][1]}, akq = {s_A[k][q][0], s_A[k][q][1]};
          const cd np_ = cadd(cmul(akp, Rpp), cmul(akq, Rqp));
          const cd nq_ = cadd(cmul(akp, Rpq), cmul(akq, Rqq));
          s_A[k][p][0] = np_.re; s_A[k][p][1] = np_.im;
          s_A[k][q][0] = nq_.re; s_A[k][q][1] = nq_.im;
          const cd vkp = {s_V[k][p][0], s_V[k][p][1]}, vkq = {s_V[k][q][0], s_V[k][q][1]};
          const cd vp_ = cadd(cmul(vkp, Rpp), cmul(vkq, Rqp));
          const cd vq_ = cadd(cmul(vkp, Rpq), cmul(vkq, Rqq));
          s_V[k][p][0] = vp_.re; s_V[k][p][1] = vp_.im;
          s_V[k][q][0] = vq_.re; s_V[k][q][1] = vq_.im;
        }
        __syncthreads();
        if (act) {                                         // A <- R^H A (rows p, q): column k = pk
          const int k = pk;
          const cd apk = {s_A[p][k][0], s_A[p][k][1]}, aqk = {s_A[q][k][0], s_A[q][k][1]};
          const cd np_ = cadd(cmul(cconj(Rpp), apk), cmul(cconj(Rqp), aqk));
          const cd nq_ = cadd(cmul(cconj(Rpq), apk), cmul(cconj(Rqq), aqk));
          s_A[p][k][0] = np_.re; s_A[p][k][1] = np_.im;
          s_A[q][k][0] = nq_.re; s_A[q][k][1] = nq_.im;
        }
        __syncthreads();
        if (act && pk == 0) {
          s_A[p][q][0] = s_A[p][q][1] = 0.0;
          s_A[q][p][0] = s_A[q][p][1] = 0.0;
          s_A[p][p][1] = 0.0;
          s_A[q][q][1] = 0.0;
        }
        __syncthreads();
      }
    }
  }
  if (tid == 0) {
    int best = 0;                                         // argmax eigenvalue, first on ties (tester.py:1110)
    for (int i = 1; i < M; ++i)
      if (s_A[i][i][0] > s_A[best][best][0]) best = i;
    cd v[M];
    for (int i = 0; i < M; ++i) v[i] = {s_V[i][best][0], s_V[i][best][1]};
    const cd v0 = v[0];
    double nrm = 0.0;
    for (int i = 0; i < M; ++i) {
      v[i] = cdiv(v[i], v0);                              // tester.py:1119
      nrm += cabs2(v[i]);
    }
    const double sc = sqrt((double)M / sqrt(nrm));        // tester.py:1123: sqrt(M / ||d||)
    double* o = ws + ws_steer0(a.B, a.S, a.F, M) + idx * (M * 2);
    for (int i = 0; i < M; ++i) {
      o[i * 2 + 0] = v[i].re * sc;
      o[i * 2 + 1] = v[i].im * sc;
    }
  }
}

// one workgroup per (b, spk): thread 0 runs the sequential phase correction over f, then thread f solves bin f
template <int M>
__global__ __launch_bounds__(256) void mvdr_solve(int B, int S, int F, double epsi, double* ws) {
  extern __shared__ double s_d[];     // [F][M][2]
  const int b = blockIdx.x, spk = blockIdx.y;
  const int tid = threadIdx.x;
  const long long base = (long long)(b * S + spk) * F;
  const double* d0 = ws + ws_steer0(B, S, F, M) + base * (M * 2);
  for (int i = tid; i < F * M * 2; i += blockDim.x) s_d[i] = d0[i];
  __syncthreads();
  // sequential-in-f phase correction (tester.py:1161-1167): bin f is rotated by the phase of <d[f], d_corrected[f-1]>.
  // The dependence from bin to bin is kept exactly; inside a bin lane m < M owns microphone m (the M products and the M
  // rotations run on M lanes, the sum over m is a butterfly over 8 lanes with zeros in the unused ones).
  if (tid < 8) {
    const int m = tid;
    const bool live = m < M;
    cd prv = live ? cd{s_d[m * 2], s_d[m * 2 + 1]} : cd{0.0, 0.0};
    for (int f = 1; f < F; ++f) {
      const cd cur = live ? cd{s_d[(f * M + m) * 2], s_d[(f * M + m) * 2 + 1]} : cd{0.0, 0.0};
      cd z = cmulc(cur, prv);
#pragma unroll
      for (int w = 4; w >= 1; w >>= 1) {
        z.re += __shfl_xor(z.re, w, 8);
        z.im += __shfl_xor(z.im, w, 8);
      }
      const double az = sqrt(cabs2(z));
      cd rot = {1.0, 0.0};                                // exp(-j angle(z)); angle(0) = 0
      if (az > 0.0) rot = {z.re / az, -z.im / az};
      prv = cmul(cur, rot);
      if (live) {
        s_d[(f * M + m) * 2] = prv.re;
        s_d[(f * M + m) * 2 + 1] = prv.im;
      }
    }
  }
  __syncthreads();
  double* d1 = ws + ws_steer1(B, S, F, M) + base * (M * 2);
  for (int i = tid; i < F * M * 2; i += blockDim.x) d1[i] = s_d[i];
  for (int f = tid; f < F; f += blockDim.x) {
    const double* pn = ws + ws_phin(B, S, F, M) + (base + f) * (M * M * 2);
    cd A[M][M + 1];
#pragma unroll
    for (int i = 0; i < M; ++i) {
#pragma unroll
      for (int j = 0; j < M; ++j) A[i][j] = {pn[(i * M + j) * 2], pn[(i * M + j) * 2 + 1]};
      A[i][i].re += epsi;                                 // tester.py:1086-1088,1221
      A[i][M] = {s_d[(f * M + i) * 2], s_d[(f * M + i) * 2 + 1]};
    }
    // Gaussian elimination with partial pivoting (fully unrolled so A stays in registers)
#pragma unroll
    for (int k = 0; k < M; ++k) {
      int piv = k;
      double best = cabs2(A[k][k]);
#pragma unroll
      for (int i = k + 1; i < M; ++i) {
        const double v = cabs2(A[i][k]);
        if (v > best) { best = v; piv = i; }
      }
#pragma unroll
      for (int i = k + 1; i < M; ++i) {
        if (piv == i) {
#pragma unroll
          for (int j = 0; j <= M; ++j) { const cd tmp = A[k][j]; A[k][j] = A[i][j]; A[i][j] = tmp; }
        }
      }
      const cd pk = A[k][k];
#pragma unroll
      for (int i = k + 1; i < M; ++i) {
        const cd fac = cdiv(A[i][k], pk);
#pragma unroll
        for (int j = k; j <= M; ++j) A[i][j] = csub(A[i][j], cmul(fac, A[k][j]));
      }
    }
    cd x[M];
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
      cd acc = A[i][M];
#pragma unroll
      for (int j = i + 1; j < M; ++j) acc = csub(acc, cmul(A[i][j], x[j]));
      x[i] = cdiv(acc, A[i][i]);
    }
    cd den = {0.0, 0.0};                                  // d^H x (tester.py:1223)
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const cd di = {s_d[(f * M + i) * 2], s_d[(f * M + i) * 2 + 1]};
      den = cadd(den, cmul(cconj(di), x[i]));
    }
    double* wo = ws + ws_w(B, S, F, M) + (base + f) * (M * 2);
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const cd wi = cdiv(x[i], den);
      wo[i * 2] = wi.re;
      wo[i * 2 + 1] = wi.im;
    }
  }
}

template <int M>
__global__ __launch_bounds__(256) void mvdr_apply(const MvdrArgs a, const COut out, const double* ws) {
  const int f = blockIdx.x, b = blockIdx.y, spk = blockIdx.z;
  const double* wp = ws + ws_w(a.B, a.S, a.F, M) + ((long long)(b * a.S + spk) * a.F + f) * (M * 2);
  float wr[M], wi[M];
  const float *yre[M], *yim[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    wr[m] = (float)wp[m * 2];
    wi[m] = (float)wp[m * 2 + 1];
    const long long off = (long long)b * a.mix.sb + (long long)f * a.mix.sf + (long long)m * a.mix.sm;
    yre[m] = a.mix.re + off;
    yim[m] = a.mix.im + off;
  }
  const int yst = a.mix.st;
  float* ore = out.re + (long long)b * out.ob + (long long)spk * out.os + (long long)f * out.of;
  float* oim = out.im + (long long)b * out.ob + (long long)spk * out.os + (long long)f * out.of;
  for (int t = threadIdx.x; t < a.T; t += 256) {
    float re = 0.f, im = 0.f;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const float yr = yre[m][(long long)t * yst], yi = yim[m][(long long)t * yst];
      re += wr[m] * yr + wi[m] * yi;                      // conj(w) * y
      im += wr[m] * yi - wi[m] * yr;
    }
    ore[(long long)t * out.ot] = re;
    oim[(long long)t * out.ot] = im;
  }
}

template <int M>
static hipError_t mvdr_run(const MvdrArgs& a, const COut& out, double* ws, hipStream_t s) {
  dim3 g(a.F, a.B, a.S);
  // one wave per (b, f, spk): the 6x6 Jacobi runs on one lane (~100 us of dependent float64 work), so what matters is
  // how many of them are in flight per CU -- 8 single-wave workgroups instead of 2 four-wave ones
  hipLaunchKernelGGL(mvdr_scm_eig<M>, g, dim3(64), 0, s, a, ws);
  hipLaunchKernelGGL(mvdr_solve<M>, dim3(a.B, a.S), dim3(256), (size_t)a.F * M * 2 * sizeof(double), s, a.B, a.S, a.F,
                     (double)a.epsi, ws);
  hipLaunchKernelGGL(mvdr_apply<M>, g, dim3(256), 0, s, a, out, (const double*)ws);
  return hipGetLastError();
}

hipError_t launch_mvdr(const MvdrArgs& a, const COut& out, void* ws, hipStream_t s) {
  double* w = reinterpret_cast<double*>(ws);
  switch (a.M) {
    case 2: return mvdr_run<2>(a, out, w, s);
    case 3: return mvdr_run<3>(a, out, w, s);
    case 4: return mvdr_run<4>(a, out, w, s);
    case 5: return mvdr_run<5>(a, out, w, s);
    case 6: return mvdr_run<6>(a, out, w, s);
    case 7: return mvdr_run<7>(a, out, w, s);
    case 8: return mvdr_run<8>(a, out, w, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_mvdr_debug(const void* ws, int B, int S, int F, int M, double* steer, double* w, hipStream_t s) {
  const double* p = reinterpret_cast<const double*>(ws);
  const size_t n = (size_t)B * S * F * M * 2 * sizeof(double);
  hipError_t e = hipSuccess;
  if (steer) e = hipMemcpyAsync(steer, p + ws_steer1(B, S, F, M), n, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess && w) e = hipMemcpyAsync(w, p + ws_w(B, S, F, M), n, hipMemcpyDeviceToDevice, s);
  return e;
}

// ---- PIT distances, S speakers (1..4) -----------------------------------------------------------------------------
// part[bk][f][i][j] = sum_t | |A_i| - |B_j| | of frequency bin f   (tester.py:1047-1052, 906-908), float64.
// grid (F, B*K): blockIdx.y = b*K + k; anchors are per b, candidates per (b, k).  Every reduction runs in a fixed
// order (no atomics): pit_pick_k adds the F partials of an item in bin order, so the distances -- and with them the
// argmin over the permutations -- are bit-reproducible.
template <int S>
__global__ __launch_bounds__(256) void pit_dist_k(const PitArgs p, int K, double* part) {
  __shared__ double s_tmp[4][S * S];
  const int f = blockIdx.x, bk = blockIdx.y;
  const int b = bk / K;
  const long long oa = (long long)b * p.a.sb + (long long)f * p.a.sf;
  const long long ob = (long long)bk * p.b.sb + (long long)f * p.b.sf;
  double d[S * S];
#pragma unroll
  for (int i = 0; i < S * S; ++i) d[i] = 0.0;
  for (int t = threadIdx.x; t < p.T; t += 256) {
    const long long ia = oa + (long long)t * p.a.st, ib = ob + (long long)t * p.b.st;
    float am[S], bm[S];
#pragma unroll
    for (int i = 0; i < S; ++i) {
      const float ar = p.a.re[ia + i * p.a.sm], ai = p.a.im[ia + i * p.a.sm];
      const float br = p.b.re[ib + i * p.b.sm], bi = p.b.im[ib + i * p.b.sm];
      am[i] = sqrtf(ar * ar + ai * ai);
      bm[i] = sqrtf(br * br + bi * bi);
    }
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
      for (int j = 0; j < S; ++j) d[i * S + j] += fabsf(am[i] - bm[j]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < S * S; ++i) {
    double v = d[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0) s_tmp[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < S * S) {
    const int i = threadIdx.x;
    part[((long long)bk * gridDim.x + f) * (S * S) + i] = (s_tmp[0][i] + s_tmp[1][i]) + (s_tmp[2][i] + s_tmp[3][i]);
  }
}

// sel[bk][i] = perm[i] of the cheapest permutation, cost(perm) = sum_i dist[i][perm[i]]; permutations in the order of
// itertools.permutations (lexicographic), first minimum on ties (torch.argmin; tester.py:1053-1064, 909-915)
template <int S>
__global__ __launch_bounds__(64) void pit_pick_k(const double* part, int F, double* dist, int* sel) {
  __shared__ double s_d[S * S];
  const int i = blockIdx.x;                          // item (b, k)
  if (threadIdx.x < S * S) {
    const double* q = part + (long long)i * F * (S * S) + threadIdx.x;
    double acc = 0.0;
    for (int f = 0; f < F; ++f) acc += q[(long long)f * (S * S)];      // fixed order: bin 0, 1, ...
    s_d[threadIdx.x] = acc;
    dist[(long long)i * (S * S) + threadIdx.x] = acc;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double* d = s_d;
  int perm[S], best[S];
#pragma unroll
  for (int k = 0; k < S; ++k) { perm[k] = k; best[k] = k; }
  double cbest = 0.0;
  bool first = true;
  for (;;) {
    double c = 0.0;
#pragma unroll
    for (int k = 0; k < S; ++k) c += d[k * S + perm[k]];
    if (first || c < cbest) {
      cbest = c; first = false;
#pragma unroll
      for (int k = 0; k < S; ++k) best[k] = perm[k];
    }
    // next lexicographic permutation
    int a = S - 2;
    while (a >= 0 && perm[a] > perm[a + 1]) --a;
    if (a < 0) break;
    int b2 = S - 1;
    while (perm[b2] < perm[a]) --b2;
    { const int t = perm[a]; perm[a] = perm[b2]; perm[b2] = t; }
    for (int lo = a + 1, hi = S - 1; lo < hi; ++lo, --hi) { const int t = perm[lo]; perm[lo] = perm[hi]; perm[hi] = t; }
  }
  for (int k = 0; k < S; ++k) sel[i * S + k] = best[k];
}

hipError_t launch_pit_dist_k(const PitArgs& p, int S, int K, double* part, hipStream_t s) {
  const dim3 g(p.F, p.B * K);
  switch (S) {
    case 1: hipLaunchKernelGGL(pit_dist_k<1>, g, dim3(256), 0, s, p, K, part); break;
    case 2: hipLaunchKernelGGL(pit_dist_k<2>, g, dim3(256), 0, s, p, K, part); break;
    case 3: hipLaunchKernelGGL(pit_dist_k<3>, g, dim3(256), 0, s, p, K, part); break;
    case 4: hipLaunchKernelGGL(pit_dist_k<4>, g, dim3(256), 0, s, p, K, part); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_pit_pick(const double* part, int F, int S, int n, double* dist, int* sel, hipStream_t s) {
  const dim3 g(n);
  switch (S) {
    case 1: hipLaunchKernelGGL(pit_pick_k<1>, g, dim3(64), 0, s, part, F, dist, sel); break;
    case 2: hipLaunchKernelGGL(pit_pick_k<2>, g, dim3(64), 0, s, part, F, dist, sel); break;
    case 3: hipLaunchKernelGGL(pit_pick_k<3>, g, dim3(64), 0, s, part, F, dist, sel); break;
    case 4: hipLaunchKernelGGL(pit_pick_k<4>, g, dim3(64), 0, s, part, F, dist, sel); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// final[b][m][j] = shift_sel[b][m][clean_sel[b][j]]   (tester.py:1065 then 915)
__global__ void compose_sel_k(const int* shift_sel, const int* clean_sel, int B, int M, int S, int* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * M * S) return;
  const int j = i % S, bm = i / S, b = bm / M;
  const int cj = clean_sel ? clean_sel[b * S + j] : j;
  out[i] = shift_sel[bm * S + cj];
}
hipError_t launch_compose_sel(const int* shift_sel, const int* clean_sel, int B, int M, int S, int* out,
                              hipStream_t s) {
  const int n = B * M * S;
  hipLaunchKernelGGL(compose_sel_k, dim3((n + 63) / 64), dim3(64), 0, s, shift_sel, clean_sel, B, M, S, out);
  return hipGetLastError();
}

// MISO3 input assembly (tester.py:936-939, model.py:360-364): sample n = b*S + j gets
//   channels [0,M) / [M+2, 2M+2): mixture re / im ; M / 2M+2: beamformer (written by mvdr_apply) ;
//   M+1 / 2M+3: MISO1 estimate of aligned speaker j at ref_ch.
__global__ __launch_bounds__(256) void assemble3_k(const float* in1, long long in1_bstride, const float* out1,
                                                   long long out1_bstride, const int* sel, int M, int S, int ref_ch,
                                                   int F, int Tp, float* in3, long long in3_bstride) {
  const int cch = blockIdx.x;          // 0..2M-1 mixture planes, 2M: est re, 2M+1: est im
  const int n3 = blockIdx.y;           // b*S + j
  const int b = n3 / S, j = n3 - b * S;
  const long long plane = (long long)F * Tp;
  const float* src;
  int cd3;
  if (cch < 2 * M) {
    src = in1 + (long long)(b * M) * in1_bstride + (long long)cch * plane;      // shift-0 sample = un-rolled mixture
    cd3 = cch < M ? cch : cch + 2;
  } else {
    const int n1 = b * M + ref_ch;
    const int q = sel[n1 * S + j];
    const int im = cch - 2 * M;
    src = out1 + (long long)n1 * out1_bstride + (long long)(im * S + q) * plane;
    cd3 = im ? 2 * M + 3 : M + 1;
  }
  float4* d = reinterpret_cast<float4*>(in3 + (long long)n3 * in3_bstride + (long long)cd3 * plane);
  const float4* s4 = reinterpret_cast<const float4*>(src);
  for (long long i = threadIdx.x + (long long)blockIdx.z * 256; i < plane / 4; i += 256LL * gridDim.z) d[i] = s4[i];
}
hipError_t launch_assemble3(const float* in1, long long in1_bstride, const float* out1, long long out1_bstride,
                            const int* sel, int B, int M, int S, int ref_ch, int F, int Tp, float* in3,
                            long long in3_bstride, hipStream_t s) {
  hipLaunchKernelGGL(assemble3_k, dim3(2 * M + 2, B * S, 16), dim3(256), 0, s, in1, in1_bstride, out1, out1_bstride,
                     sel, M, S, ref_ch, F, Tp, in3, in3_bstride);
  return hipGetLastError();
}

}  // namespace mn
