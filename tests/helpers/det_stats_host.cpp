// Host build of the limb arithmetic of misonet_amd/csrc/det_stats.hpp for tests/test_det_stats.py (g++, no HIP): an
// accumulator with plain integer adds stands in for the integer atomics of the kernels.
#include "../../misonet_amd/csrc/det_stats.hpp"

extern "C" {
int ds_nl(void) { return mn::DS_NL; }
// adds v to the limbs acc[DS_NL]; returns 0 when v poisons the statistic
int ds_accumulate(long long* acc, double v) {
  long long q[mn::DS_NL];
  if (!mn::dstat_split(v, q)) { mn::dstat_poison_limb(acc[mn::DS_NL - 1]); return 0; }
  for (int i = 0; i < mn::DS_NL; ++i) acc[i] += q[i];
  return 1;
}
void ds_split(double v, long long* q) {
  long long t[mn::DS_NL] = {0, 0, 0, 0, 0};
  (void)mn::dstat_split(v, t);
  for (int i = 0; i < mn::DS_NL; ++i) q[i] = t[i];
}
double ds_value(const long long* acc) {
  long long L[mn::DS_NL];
  for (int i = 0; i < mn::DS_NL; ++i) L[i] = acc[i];
  return mn::dstat_combine(L);
}
}
