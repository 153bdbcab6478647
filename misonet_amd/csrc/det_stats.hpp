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
#include <hip/hip_runtime.h>

namespace mn {

typedef unsigned long long dstat_t;
constexpr int DS_NL = 5;                     // limbs per statistic

// p: the DS_NL limbs of one statistic (zeroed at the start of a forward)
__device__ __forceinline__ void dstat_add(dstat_t* p, double v) {
  if (!(fabs(v) < 0x1p118)) {                // inf, NaN or out of range: poison (ds_read returns NaN)
    atomicAdd(p + (DS_NL - 1), 1ull << 62);
    return;
  }
  v *= 0x1p80;                               // exact: |v| < 2^198
  const double up[DS_NL] = {1.0, 0x1p40, 0x1p80, 0x1p120, 0x1p160};
  const double dn[DS_NL] = {1.0, 0x1p-40, 0x1p-80, 0x1p-120, 0x1p-160};
#pragma unroll
  for (int i = DS_NL - 1; i >= 0; --i) {
    const double q = trunc(v * dn[i]);       // |q| < 2^40 (top limb: < 2^38); exact
    v = fma(-q, up[i], v);                   // exact: removes the leading bits
    // q is an integer below 2^40 in magnitude: its two's-complement value sits in the low mantissa bits of q + 1.5 * 2^52
    // (2 instructions; a double -> int64 conversion is emulated with ~20)
    const long long qi = __double_as_longlong(q + 0x1.8p52) - 0x4338000000000000ll;
    if (qi) atomicAdd(p + i, (unsigned long long)qi);
  }
}

__device__ __forceinline__ double dstat_read(const dstat_t* p) {
  long long L[DS_NL];
#pragma unroll
  for (int i = 0; i < DS_NL; ++i) L[i] = (long long)p[i];
  if (L[DS_NL - 1] >= (1ll << 61)) return __longlong_as_double(0x7ff8000000000000ll);
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
  return s * 0x1p-80;
}

// statistic `which` (0: sum, 1: sum of squares) of entity e in an array [entities][2][DS_NL]
__device__ __forceinline__ dstat_t* dstat_at(dstat_t* base, long long e, int which) { return base + (e * 2 + which) * DS_NL; }
__device__ __forceinline__ const dstat_t* dstat_at(const dstat_t* base, long long e, int which) {
  return base + (e * 2 + which) * DS_NL;
}

}  // namespace mn
