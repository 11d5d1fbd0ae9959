/*
 * gvt_math.h -- the three transcendental functions of the bounce path, written out.
 *
 * CosWeightedRandomHemisphereDirection2 (adapter/embree/EmbreeMeshAdapter.cpp:289-318) calls
 *     theta = acos(sqrt(1.0 - Xi1))   (double, then stored to float)
 *     sinf(theta), cosf(theta), sinf(phi), cosf(phi)
 * through the host's libm.  libm and the device's ocml differ in the last bits, and even glibc's own sinf has
 * CPU-dependent (FMA / non-FMA ifunc) variants, so neither is a definition two machines can share.  This header IS
 * the definition used by both the gfx950 adapter (csrc/trace.hip) and the CPU checker (oracle/gvt_oracle.c): IEEE
 * double add / multiply / sqrt in a fixed order, no fused multiply-add (both sides compile with -ffp-contract=off),
 * no library call.  Each function evaluates its result in double to < 1e-16 relative and rounds once to float, i.e.
 * it returns the correctly rounded float value except within ~1e-9 ulp of a tie.  Measured over all 2^24 values the
 * reference's RNG can produce (tests/test_oracle_pinning.py repeats it on a sample): theta is bit-identical to glibc's
 * (float)acos(sqrt(1.0 - Xi1)) for every Xi1; gvt_sinf / gvt_cosf equal the correctly rounded (float)sin((double)x) for
 * every phi, and glibc 2.35's sinf / cosf (0.56 ulp, not correctly rounded) in 98.7 % of them, never more than 1 ulp apart.
 *
 * Plain C99 / C++; GVT_MATH_FN is the function prefix (the HIP side adds __host__ __device__).
 */
#ifndef GVT_MATH_H
#define GVT_MATH_H

#ifndef GVT_MATH_FN
#define GVT_MATH_FN static inline
#endif

/* sin r and cos r for |r| <= pi/4 (+ reduction slack): Taylor series in z = r*r, Horner from the highest term.
 * Coefficients are (-1)^k / (2k+1)! and (-1)^k / (2k)!, correctly rounded; the first omitted term is < 1e-19. */
GVT_MATH_FN double gvt_sin_kernel(double r) {
  const double z = r * r;
  double p = -0x1.2f49b46814157p-57;      /* -1/19! */
  p = p * z + 0x1.952c77030ad4ap-49;      /*  1/17! */
  p = p * z + -0x1.ae7f3e733b81fp-41;     /* -1/15! */
  p = p * z + 0x1.6124613a86d09p-33;      /*  1/13! */
  p = p * z + -0x1.ae64567f544e4p-26;     /* -1/11! */
  p = p * z + 0x1.71de3a556c734p-19;      /*  1/9!  */
  p = p * z + -0x1.a01a01a01a01ap-13;     /* -1/7!  */
  p = p * z + 0x1.1111111111111p-7;       /*  1/5!  */
  p = p * z + -0x1.5555555555555p-3;      /* -1/3!  */
  return r + r * (z * p);
}
GVT_MATH_FN double gvt_cos_kernel(double r) {
  const double z = r * r;
  double p = -0x1.6827863b97d97p-53;      /* -1/18! */
  p = p * z + 0x1.ae7f3e733b81fp-45;      /*  1/16! */
  p = p * z + -0x1.93974a8c07c9dp-37;     /* -1/14! */
  p = p * z + 0x1.1eed8eff8d898p-29;      /*  1/12! */
  p = p * z + -0x1.27e4fb7789f5cp-22;     /* -1/10! */
  p = p * z + 0x1.a01a01a01a01ap-16;      /*  1/8!  */
  p = p * z + -0x1.6c16c16c16c17p-10;     /* -1/6!  */
  p = p * z + 0x1.5555555555555p-5;       /*  1/4!  */
  p = p * z + -0x1.0000000000000p-1;      /* -1/2!  */
  return 1.0 + z * p;
}

/* x = k*(pi/2) + r, |r| <= pi/4.  pi/2 = hi + lo with a 32-bit hi, so k*hi is exact for |k| < 2^20 (|x| < 1.6e6;
 * the bounce path passes theta in [0, pi/2] and phi in [0, 2 pi)). */
GVT_MATH_FN double gvt_reduce_pio2(double x, int *quadrant) {
  const double t = x * 0x1.45f306dc9c883p-1; /* 2/pi */
  const int k = (int)(t + (t >= 0.0 ? 0.5 : -0.5));
  const double kd = (double)k;
  *quadrant = k & 3;
  return (x - kd * 0x1.921fb54400000p+0) - kd * 0x1.0b4611a626331p-34;
}

GVT_MATH_FN float gvt_sinf(float x) {
  int q;
  const double r = gvt_reduce_pio2((double)x, &q);
  const double v = (q & 1) ? gvt_cos_kernel(r) : gvt_sin_kernel(r);
  return (float)((q & 2) ? -v : v);
}
GVT_MATH_FN float gvt_cosf(float x) {
  int q;
  const double r = gvt_reduce_pio2((double)x, &q);
  const double v = (q & 1) ? gvt_sin_kernel(r) : gvt_cos_kernel(r);
  return (float)(((q + 1) & 2) ? -v : v);
}

/* asin y for |y| <= 1/2: y + y*z*(c1 + c2 z + ...), z = y*y, c_n = (2n)! / (4^n (n!)^2 (2n+1)) correctly rounded;
 * 30 terms, the first omitted term is < 2e-21 at |y| = 1/2. */
GVT_MATH_FN double gvt_asin_half(double y) {
  const double z = y * y;
  double p = 0x1.b8d2e5667ce6cp-10;
  p = p * z + 0x1.cf7dea5b6e830p-10;
  p = p * z + 0x1.e82be60d9127ep-10;
  p = p * z + 0x1.018f963c229bfp-9;
  p = p * z + 0x1.1052bc5fa960ap-9;
  p = p * z + 0x1.208d3570ae5a6p-9;
  p = p * z + 0x1.3275586c5f2f0p-9;
  p = p * z + 0x1.464c0950f7d47p-9;
  p = p * z + 0x1.5c5f56efaaaabp-9;
  p = p * z + 0x1.750de64d7d05fp-9;
  p = p * z + 0x1.90cb77f60c7cep-9;
  p = p * z + 0x1.b026f57b13b14p-9;
  p = p * z + 0x1.d3d2a8e0dd67dp-9;
  p = p * z + 0x1.fcaf8fb6db6dbp-9;
  p = p * z + 0x1.15ee9d45d1746p-8;
  p = p * z + 0x1.31683bdef7bdfp-8;
  p = p * z + 0x1.51ba308d3dcb1p-8;
  p = p * z + 0x1.782dda12f684cp-8;
  p = p * z + 0x1.a6863d70a3d71p-8;
  p = p * z + 0x1.df3bd37a6f4dfp-8;
  p = p * z + 0x1.12ef3cf3cf3cfp-7;
  p = p * z + 0x1.3fde50d79435ep-7;
  p = p * z + 0x1.7a87878787878p-7;
  p = p * z + 0x1.c99999999999ap-7;
  p = p * z + 0x1.1c4ec4ec4ec4fp-6;
  p = p * z + 0x1.6e8ba2e8ba2e9p-6;
  p = p * z + 0x1.f1c71c71c71c7p-6;
  p = p * z + 0x1.6db6db6db6db7p-5;
  p = p * z + 0x1.3333333333333p-4;
  p = p * z + 0x1.5555555555555p-3;
  return y + y * (z * p);
}

/* acos x, x in [-1, 1] (values outside are clamped): pi/2 - asin x for |x| <= 1/2, 2 asin sqrt((1-|x|)/2) beyond (1-|x| is exact
 * there), reflected for x < 0.  sqrt is the IEEE correctly rounded one on both sides. */
GVT_MATH_FN double gvt_acos(double x) {
  const double pio2_hi = 0x1.921fb54442d18p+0, pio2_lo = 0x1.1a62633145c07p-54;
  const double ax = x < 0.0 ? -x : x;
  if (ax <= 0.5) return (pio2_hi - gvt_asin_half(x)) + pio2_lo;
  const double h = ax >= 1.0 ? 0.0 : (1.0 - ax) * 0.5;
  const double a = 2.0 * gvt_asin_half(__builtin_sqrt(h));
  return x < 0.0 ? (2.0 * pio2_hi - a) + 2.0 * pio2_lo : a;
}

#endif
