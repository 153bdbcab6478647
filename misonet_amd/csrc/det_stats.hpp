// Order-independent (bit-reproducible) accumulation of the instance-norm / gLN statistics.
//
// The reference's nn.InstanceNorm2d / GlobalLayerNorm (model.py:413,430,445,609-632) are deterministic; float64 atomics
// are not: their additions arrive in a different order on every run, so two runs differed in the last bit of a few
// statistics.  Here a statistic is a FIXED-POINT number of DS_NL signed 64-bit limbs (limb i carries 40 payload bits of
// weight 2^(40 i - 80); the other 24 bits of the word are headroom for 2^23 un-normalised additions).  A tile's partial
// sum (a float32 or float64 value) is cut into its limbs exactly and each non-zero limb is added with an INTEGER atomic.
// Integer addition is associative, so the accumulated value does not depend on the order in which tiles finish, on the
// tile -> CU schedule, on the batch a sample runs in or on its position in it.  Range: |partial| < 2^118 with a
// resolution of 2^-80 (anything a float32 partial of finite data holds; what is below 2^-80 is far below the eps of the
// norms and is truncated toward zero); a non-finite or larger partial poisons the statistic, which then reads as NaN.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MN_HD __host__ __device__ __forceinline__
#else                                        // plain C++ (tests/test_det_stats.py builds the limb arithmetic with g++)
#include <math.h>
#define MN_HD inline
#endif

namespace mn {

typedef unsigned long long dstat_t;
constexpr int DS_NL = 5;                     // limbs per statistic
// Poison: a non-finite / out-of-range partial RAISES the top limb to DS_POISON with an atomic MAX.  That is idempotent --
// any number of bad partials leaves it at >= 2^62 - 2^61 (legitimate additions move the limb by < 2^61 in all: |q| < 2^38,
// 2^23 additions), so dstat_combine's test `top limb >= 2^61` holds whatever the order and the count.  (Round 3 ADDED
// 2^62 per bad partial: two made the limb negative, four wrapped it to zero -- and a NaN activation poisons many tiles.)
constexpr long long DS_POISON = 1ll << 62;

// the poison step on plain memory (the kernels use atomicMax; tests/helpers/det_stats_host.cpp uses this)
MN_HD void dstat_poison_limb(long long& top) { top = top > DS_POISON ? top : DS_POISON; }

// v -> q[0 .. DS_NL-1] with v = sum q[i] 2^(40 i - 80) + (a remainder below 2^-80, truncated toward zero); |q[i]| < 2^40.
// Returns false (q untouched) when v is inf, NaN or >= 2^118 in magnitude.
MN_HD bool dstat_split(double v, long long (&q)[DS_NL]) {
  if (!(fabs(v) < 0x1p118)) return false;
  v *= 0x1p80;                               // exact: |v| < 2^198
  const double up[DS_NL] = {1.0, 0x1p40, 0x1p80, 0x1p120, 0x1p160};
  const double dn[DS_NL] = {1.0, 0x1p-40, 0x1p-80, 0x1p-120, 0x1p-160};
#pragma unroll
  for (int i = DS_NL - 1; i >= 0; --i) {
    const double t = trunc(v * dn[i]);       // |t| < 2^40 (top limb: < 2^38); exact
    v = fma(-t, up[i], v);                   // exact: removes the leading bits
    // t is an integer below 2^40 in magnitude: its two's-complement value sits in the low mantissa bits of t + 1.5 * 2^52
    // (2 instructions; a double -> int64 conversion is emulated with ~20 on the GPU)
    q[i] = __builtin_bit_cast(long long, t + 0x1.8p52) - 0x4338000000000000ll;
  }
  return true;
}

// the value of the limbs L (two's complement, un-normalised), NaN when poisoned
MN_HD double dstat_combine(const long long (&Lin)[DS_NL]) {
  long long L[DS_NL];
#pragma unroll
  for (int i = 0; i < DS_NL; ++i) L[i] = Lin[i];
  // (no early return for the poisoned case: a branch on the top limb made the compiler load that limb first, wait for it, and
  // only then load the other four -- two memory round trips per statistic in front of every conv tile, round 6)
  const bool poisoned = L[DS_NL - 1] >= (1ll << 61);
  // carry-normalise limbs 0 .. DS_NL-2 into [0, 2^40): the value is then  L[4] 2^160 + ... + L[0]  with one sign
#pragma unroll
  for (int i = 0; i < DS_NL - 1; ++i) {
    const long long c = L[i] >> 40;          // arithmetic shift = floor
    L[i] -= c << 40;
    L[i + 1] += c;
  }
  double s = (double)L[DS_NL - 1];
#pragma unroll
  for (int i = DS_NL - 2; i >= 0; --i) s = fma(s, 0x1p40, (double)L[i]);
  return poisoned ? __builtin_bit_cast(double, 0x7ff8000000000000ll) : s * 0x1p-80;
}

#if defined(__HIPCC__)
// p: the DS_NL limbs of one statistic (zeroed at the start of a forward)
__device__ __forceinline__ void dstat_add(dstat_t* p, double v) {
  long long q[DS_NL];
  if (!dstat_split(v, q)) {                  // inf, NaN or out of range: poison (dstat_read returns NaN), idempotent
    atomicMax(reinterpret_cast<long long*>(p + (DS_NL - 1)), DS_POISON);
    return;
  }
#pragma unroll
  for (int i = 0; i < DS_NL; ++i)
    if (q[i]) atomicAdd(p + i, (unsigned long long)q[i]);
}

__device__ __forceinline__ double dstat_read(const dstat_t* p) {
  long long L[DS_NL];
#pragma unroll
  for (int i = 0; i < DS_NL; ++i) L[i] = (long long)p[i];
  return dstat_combine(L);
}
#endif

}  // namespace mn
