// Device helpers shared by tsem.hip and the translation units that instantiate the fused kernel (tsem_fz_*.hip).
#pragma once
#include <hip/hip_runtime.h>

// The column map: colmap[j] = part << CM_PS | log2(copies) << CM_LS | first accumulator slot of column j in its part.
// (14 slot bits since round 4: a part of the SPLIT layout holds up to 15 424 columns — one table per pass, tsem_fused.h modes 5 / 7.)
constexpr int CM_LS = 14, CM_PS = 17;
constexpr unsigned CM_SM = (1u << CM_LS) - 1u;

// log1p(x) for finite x >= 0 (the lnl passes evaluate it once per stored entry: 2*10^9 times per
// pass at config 4, where the library routine's generality made the pass compute-bound).  The
// classic argument reduction 1+x = 2^k (1+f), sqrt(2)/2 < 1+f < sqrt(2), log(1+f) = f - f^2/2 +
// s (f^2/2 + R(s^2)), s = f/(2+f), with the rounding of 1+x corrected by c/u (W. Kahan / fdlibm's
// published log1p; minimax coefficients Lp1..Lp7 from there).  Error < 1 ulp on the range used.
__device__ __forceinline__ double ts_log1p_pos(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
               Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  if (x < 3.725290298461914e-09) return fma(-0.5 * x, x, x);         // x < 2^-28: x - x^2/2 is exact to rounding
  const double u = 1.0 + x;
  int hu = __double2hiint(u);
  int k = (hu >> 20) - 1023;
  // rounding error of 1 + x, relative to u (only matters while k is small; it underflows harmlessly later)
  const double c = (k > 0 ? 1.0 - (u - x) : x - (u - 1.0)) * __builtin_amdgcn_rcp(u);
  hu &= 0x000fffff;
  if (hu < 0x6a09e) { hu |= 0x3ff00000; } else { k += 1; hu |= 0x3fe00000; }   // 1+f in [sqrt(2)/2, sqrt(2))
  const double f = __hiloint2double(hu, __double2loint(u)) - 1.0;
  const double hfsq = 0.5 * f * f;
  const double d = 2.0 + f;                                            // in (1.7, 2.42): plain Newton reciprocal
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  double s = f * r;
  s = fma(fma(-d, s, f), r, s);
  const double z = s * s;
  const double R = z * fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, Lp7, Lp6), Lp5), Lp4), Lp3), Lp2), Lp1);
  const double dk = (double)k;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + (dk * ln2_lo + c))) - f);
}

__device__ __forceinline__ double recip0(double v) {
  // sparse_plus.py:16-22 — 1/v with inf -> 0
  double r = 1.0 / v;
  return isinf(r) ? 0.0 : r;
}

__device__ __forceinline__ void lds_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
