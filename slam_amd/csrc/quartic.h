// Real roots of a quartic in closed form (Ferrari's factorisation through the
// resolvent cubic) followed by Newton polishing on the original polynomial.
// Straight-line fp64 code: no data-dependent loops, so the 64 problems of a
// wavefront stay converged.  Usable from host code as well, which is how it is
// checked against numpy.roots without a GPU (tools/quartic_check.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

// Floating-point contraction in this header is confined to what the source spells out (a product
// and the sum it feeds in one expression, or an explicit fma): the functions here are inlined into
// several kernels whose results are compared bit for bit, and the compiler's default for device
// code (fuse whatever it finds once everything is inlined) makes that depend on the surroundings.
#pragma clang fp contract(on)
namespace slamhip_dev {

#define SLAMHIP_HD __host__ __device__ __forceinline__

SLAMHIP_HD double quartic_eval(double a, double b, double c, double d, double x) {
  return fma(fma(fma(x + a, x, b), x, c), x, d);
}

SLAMHIP_HD double quartic_polish(double a, double b, double c, double d, double x) {
  // Newton on x^4 + a x^3 + b x^2 + c x + d; a step is taken only while it is finite
  // and does not increase |f|, so a (near-)double root is left where it is.  Two steps: on
  // the 1e6 polynomials of tools/quartic_check.hip one already reaches the worst error of
  // three (1.35 x the root's own rounding sensitivity), and every step is a division in a
  // chain that every wavefront of the minimal solver walks four times.
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double f = quartic_eval(a, b, c, d, x);
    const double df = fma(fma(fma(4.0, x, 3.0 * a), x, 2.0 * b), x, c);
    const double xn = x - f / df;
    const bool ok = (df != 0.0) && (fabs(xn) <= 1.79e308) &&
                    (fabs(quartic_eval(a, b, c, d, xn)) <= fabs(f));
    x = ok ? xn : x;
  }
  return x;
}

// Largest real root of z^3 + A z^2 + B z + C.
SLAMHIP_HD double cubic_largest_root(double A, double B, double C) {
  const double third = 1.0 / 3.0;
  const double sh = A * third;
  const double P = fma(-A, sh, B);                                  // B - A^2/3
  const double Q = fma(A * sh, 2.0 * sh * third, fma(-sh, B, C));   // 2A^3/27 - AB/3 + C
  const double hq = 0.5 * Q, tp = P * third;
  const double disc = fma(hq, hq, tp * tp * tp);
  double t;
  if (disc > 0.0) {
    const double sq = sqrt(disc);
    const double u = cbrt(-hq - copysign(sq, hq));                   // the larger-magnitude cube root
    t = (u != 0.0) ? u - tp / u : 0.0;
  } else {
    // three real roots; k = 0 branch of the trigonometric form is the largest
    const double m = sqrt(fmax(-tp, 0.0));
    double arg = (m > 0.0) ? -hq / (m * m * m) : 0.0;
    arg = fmin(1.0, fmax(-1.0, arg));
    t = 2.0 * m * cos(acos(arg) * third);
  }
  double z = t - sh;
  // one Newton step on the cubic itself (the same check shows no difference between none and three)
#pragma unroll
  for (int it = 0; it < 1; ++it) {
    const double f = fma(fma(z + A, z, B), z, C);
    const double df = fma(fma(3.0, z, 2.0 * A), z, B);
    const double zn = z - f / df;
    const bool ok = (df != 0.0) && (fabs(zn) <= 1.79e308) &&
                    (fabs(fma(fma(zn + A, zn, B), zn, C)) <= fabs(f));
    z = ok ? zn : z;
  }
  return z;
}

// Roots of y^2 + s y + m (s may be negative) appended to out[n...]; a slightly
// negative discriminant (rounding at a double root) is read as zero.
SLAMHIP_HD int quadratic_roots(double s, double m, double tol, double (&out)[4], int n) {
  double D = fma(s, s, -4.0 * m);
  if (D < 0.0 && D >= -tol) D = 0.0;
  if (D >= 0.0) {
    const double sq = sqrt(D);
    const double qq = -0.5 * (s + copysign(sq, s));
    const double x1 = qq;
    const double x2 = (qq != 0.0) ? m / qq : -s - qq;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j == n) out[j] = x1;
      if (j == n + 1) out[j] = x2;
    }
    n += 2;
  }
  return n;
}

// Real roots of q[0] + q[1] x + ... + q[4] x^4 (q[4] != 0), ascending.
// Complex pairs are dropped; a double root is reported twice.
SLAMHIP_HD int quartic_real_roots_closed(const double (&q)[5], double (&roots)[4]) {
  const double inv = 1.0 / q[4];
  const double a = q[3] * inv, b = q[2] * inv, c = q[1] * inv, d = q[0] * inv;
  const double a4 = 0.25 * a;
  // depressed quartic y^4 + p y^2 + g y + r with x = y - a/4
  const double a2 = a * a;
  const double p = fma(-0.375, a2, b);
  const double g = fma(a, fma(0.125, a2, -0.5 * b), c);
  const double r = fma(a4, fma(a4, fma(-0.1875, a2, b), -c), d);   // d - ac/4 + a^2 b/16 - 3a^4/256
  double y[4] = {0.0, 0.0, 0.0, 0.0};
  int n = 0;
  // resolvent: z^3 + 2p z^2 + (p^2 - 4r) z - g^2 = 0 has a root z0 >= 0
  const double z0 = cubic_largest_root(2.0 * p, fma(p, p, -4.0 * r), -(g * g));
  const double scale = fmax(fmax(fabs(p), fabs(z0)), fmax(sqrt(fabs(r)), 1e-300));
  const double tol = 1e-13 * scale;   // on discriminants, which scale like p and z0
  if (z0 > 1e-14 * scale && z0 > 0.0) {
    const double s = sqrt(z0);
    const double h = g / s;
    n = quadratic_roots(s, 0.5 * (p + z0 - h), tol, y, n);
    n = quadratic_roots(-s, 0.5 * (p + z0 + h), tol, y, n);
  } else {
    // g == 0 (to rounding): biquadratic in y^2
    double D = fma(p, p, -4.0 * r);
    if (D < 0.0 && D >= -tol * scale) D = 0.0;
    if (D >= 0.0) {
      const double sq = sqrt(D);
      const double qq = -0.5 * (p + copysign(sq, p));
      const double w1 = qq;
      const double w2 = (qq != 0.0) ? r / qq : -p - qq;
      if (w1 >= 0.0) { const double t = sqrt(w1); y[0] = t; y[1] = -t; n = 2; }
      if (w2 >= 0.0) {
        const double t = sqrt(w2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j == n) y[j] = t;
          if (j == n + 1) y[j] = -t;
        }
        n += 2;
      }
    }
  }
  // back to x, polish on the monic quartic, sort ascending (unused slots last)
  const double big = 1.79e308;
#pragma unroll
  for (int j = 0; j < 4; ++j) roots[j] = (j < n) ? quartic_polish(a, b, c, d, y[j] - a4) : big;
#define SLAMHIP_CSWAP(i, j)                                   \
  {                                                           \
    const double lo = fmin(roots[i], roots[j]);               \
    const double hi = fmax(roots[i], roots[j]);               \
    roots[i] = lo;                                            \
    roots[j] = hi;                                            \
  }
  SLAMHIP_CSWAP(0, 1) SLAMHIP_CSWAP(2, 3) SLAMHIP_CSWAP(0, 2) SLAMHIP_CSWAP(1, 3) SLAMHIP_CSWAP(1, 2)
#undef SLAMHIP_CSWAP
  return n;
}

}  // namespace slamhip_dev
#pragma clang fp contract(fast)   // the compiler's default for device code, for whatever follows

