// Device code of the minimal pose solver (P3P + fourth-point selection), one
// problem per lane, fp64.  Method: see p3p.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "quartic.h"

// Floating-point contraction in this header is confined to what the source spells out (a product
// and the sum it feeds in one expression, or an explicit fma): the functions here are inlined into
// several kernels whose results are compared bit for bit, and the compiler's default for device
// code (fuse whatever it finds once everything is inlined) makes that depend on the surroundings.
#pragma clang fp contract(on)
namespace slamhip_dev {

struct Kinv { double m[9]; };

__device__ __forceinline__ void cross3(const double *a, const double *b, double *o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3(const double *a, const double *b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
// (a reciprocal square root in place of a root and a division: the minimal solver is a chain of
// some 2300 fp64 instructions per wavefront, a fifth of which were the divisions and roots of the
// normalisations, the first depth and the fourth point's projection)
__device__ __forceinline__ bool normalize3(double *a) {
  const double n2 = dot3(a, a);
  if (!(n2 > 0.0)) return false;
  const double inv = rsqrt(n2);
  a[0] *= inv; a[1] *= inv; a[2] *= inv;
  return true;
}

// Orthonormal frame (columns e1,e2,e3) of the triangle p1,p2,p3.
__device__ bool triangle_frame(const double *p1, const double *p2, const double *p3, double (&F)[3][3]) {
  double a[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  double b[3] = {p3[0] - p1[0], p3[1] - p1[1], p3[2] - p1[2]};
  if (!normalize3(a)) return false;
  double n[3];
  cross3(a, b, n);
  if (!normalize3(n)) return false;
  double m[3];
  cross3(n, a, m);
  for (int i = 0; i < 3; ++i) { F[i][0] = a[i]; F[i][1] = m[i]; F[i][2] = n[i]; }
  return true;
}

// One minimal problem: X[4][3] world points, f[4][3] unit bearings, nrm[3] the
// normalised image coordinates of the fourth point.  Returns false when no real
// solution with positive depths exists; otherwise Rt[12] (row-major [R|t]).
#ifndef SLAMHIP_P3P_WAVES
#define SLAMHIP_P3P_WAVES 3
#endif
__device__ __forceinline__ bool p3p_lane(const double (&X)[4][3], const double (&f)[4][3],
                                      const double (&nrm4)[2], double *bestRt) {
  double nrm[4][2];
  nrm[3][0] = nrm4[0];
  nrm[3][1] = nrm4[1];
  bool found = false;
  double best_err = 1.79e308;

  double d12[3], d13[3], d23[3];
  for (int c = 0; c < 3; ++c) {
    d12[c] = X[0][c] - X[1][c];
    d13[c] = X[0][c] - X[2][c];
    d23[c] = X[1][c] - X[2][c];
  }
  const double a12 = dot3(d12, d12), a13 = dot3(d13, d13), a23 = dot3(d23, d23);
  const double sqrt_a12 = sqrt(a12);
  const double c12 = dot3(f[0], f[1]), c13 = dot3(f[0], f[2]), c23 = dot3(f[1], f[2]);
  double Fw[3][3];
  const bool world_ok = a12 > 0.0 && a13 > 0.0 && a23 > 0.0 && triangle_frame(X[0], X[1], X[2], Fw);

  if (world_ok) {
    // E1: A1 u^2 + B1 u + C1(v) = 0,  E2: A2 u^2 + B2(v) u + C2(v) = 0
    const double A1 = a13, B1 = -2.0 * a13 * c12;
    const double C1[3] = {a13 - a12, 2.0 * a12 * c13, -a12};
    const double A2 = a23 - a12;
    const double B2[2] = {-2.0 * a23 * c12, 2.0 * a12 * c23};
    const double C2[3] = {a23, 0.0, -a12};
    double p[3], q[2], r[4];
    for (int i = 0; i < 3; ++i) p[i] = A1 * C2[i] - A2 * C1[i];           // A1 C2 - A2 C1
    q[0] = A1 * B2[0] - A2 * B1;                                            // A1 B2 - A2 B1
    q[1] = A1 * B2[1];
    r[0] = B1 * C2[0] - B2[0] * C1[0];                                      // B1 C2 - B2 C1
    r[1] = B1 * C2[1] - (B2[0] * C1[1] + B2[1] * C1[0]);
    r[2] = B1 * C2[2] - (B2[0] * C1[2] + B2[1] * C1[1]);
    r[3] = -(B2[1] * C1[2]);
    double quart[5] = {0, 0, 0, 0, 0};                                      // p^2 - q r
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) quart[i + j] += p[i] * p[j];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 4; ++j) quart[i + j] -= q[i] * r[j];
    double scale = 0.0;
    for (int i = 0; i < 5; ++i) scale = fmax(scale, fabs(quart[i]));
    double vr[4] = {0.0, 0.0, 0.0, 0.0};
    int nroots = 0;
    if (scale > 0.0 && fabs(quart[4]) > 1e-14 * scale) nroots = quartic_real_roots_closed(quart, vr);

#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s >= nroots) continue;
      const double v = vr[s];
      if (!(v > 0.0)) continue;
      const double qv = q[0] + q[1] * v;
      if (qv == 0.0) continue;
      const double u = -(p[0] + v * (p[1] + v * p[2])) / qv;
      if (!(u > 0.0)) continue;
      const double den = 1.0 + u * u - 2.0 * u * c12;
      if (!(den > 0.0)) continue;
      double d[3];
      d[0] = sqrt_a12 * rsqrt(den);
      d[1] = u * d[0];
      d[2] = v * d[0];
      // two Gauss-Newton steps on the three distance equations
      for (int it = 0; it < 2; ++it) {
        const double r0 = d[0] * d[0] + d[1] * d[1] - 2.0 * d[0] * d[1] * c12 - a12;
        const double r1 = d[0] * d[0] + d[2] * d[2] - 2.0 * d[0] * d[2] * c13 - a13;
        const double r2 = d[1] * d[1] + d[2] * d[2] - 2.0 * d[1] * d[2] * c23 - a23;
        const double j00 = 2.0 * (d[0] - d[1] * c12), j01 = 2.0 * (d[1] - d[0] * c12);
        const double j10 = 2.0 * (d[0] - d[2] * c13), j12 = 2.0 * (d[2] - d[0] * c13);
        const double j21 = 2.0 * (d[1] - d[2] * c23), j22 = 2.0 * (d[2] - d[1] * c23);
        // J = [[j00 j01 0],[j10 0 j12],[0 j21 j22]]
        const double det = -j00 * j12 * j21 - j01 * j10 * j22;
        if (fabs(det) < 1e-300) break;
        const double inv = 1.0 / det;
        const double dx0 = inv * (r0 * (-j12 * j21) - j01 * (r1 * j22 - j12 * r2));
        const double dx1 = inv * (j00 * (r1 * j22 - j12 * r2) - r0 * (j10 * j22));
        const double dx2 = inv * (j00 * (-j21 * r1) - j01 * (j10 * r2) + r0 * (j10 * j21));
        d[0] -= dx0; d[1] -= dx1; d[2] -= dx2;
      }
      if (!(d[0] > 0.0 && d[1] > 0.0 && d[2] > 0.0)) continue;
      double Y[3][3];
      for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) Y[k][c] = d[k] * f[k][c];
      double Fc[3][3];
      if (!triangle_frame(Y[0], Y[1], Y[2], Fc)) continue;
      double Rt[12];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          Rt[4 * i + j] = Fc[i][0] * Fw[j][0] + Fc[i][1] * Fw[j][1] + Fc[i][2] * Fw[j][2];
      for (int i = 0; i < 3; ++i)
        Rt[4 * i + 3] = Y[0][i] - (Rt[4 * i] * X[0][0] + Rt[4 * i + 1] * X[0][1] + Rt[4 * i + 2] * X[0][2]);
      // fourth point, error in normalised image coordinates
      double c4[3];
      for (int i = 0; i < 3; ++i)
        c4[i] = Rt[4 * i] * X[3][0] + Rt[4 * i + 1] * X[3][1] + Rt[4 * i + 2] * X[3][2] + Rt[4 * i + 3];
      const double iz = 1.0 / c4[2];
      const double ex = c4[0] * iz - nrm[3][0], ey = c4[1] * iz - nrm[3][1];
      const double err = ex * ex + ey * ey;
      if (err < best_err) {
        best_err = err;
        found = true;
        for (int i = 0; i < 12; ++i) bestRt[i] = Rt[i];
      }
    }
  }
  return found;
}

// Bearing and normalised coordinates of pixel (u, v).
__device__ __forceinline__ void pixel_to_bearing(const Kinv &ki, double u, double v, double (&f)[3],
                                                 double (&nrm)[2]) {
  double b[3];
  for (int c = 0; c < 3; ++c) b[c] = ki.m[3 * c + 0] * u + ki.m[3 * c + 1] * v + ki.m[3 * c + 2];
  const double ib = 1.0 / b[2];
  nrm[0] = b[0] * ib;
  nrm[1] = b[1] * ib;
  normalize3(b);
  for (int c = 0; c < 3; ++c) f[c] = b[c];
}

}  // namespace slamhip_dev
#pragma clang fp contract(fast)   // the compiler's default for device code, for whatever follows

