// Device code shared by the triangulation kernels: the null vector of the 4x4 DLT system, one
// problem per lane, fp64 -- by inverse iteration where that settles (every match that has
// passed the stereo filter), by one-sided (Hestenes) Jacobi sweeps otherwise.
#pragma once
#include <hip/hip_runtime.h>

// Floating-point contraction in this header is confined to what the source spells out (a product
// and the sum it feeds in one expression, or an explicit fma): the functions here are inlined into
// several kernels whose results are compared bit for bit, and the compiler's default for device
// code (fuse whatever it finds once everything is inlined) makes that depend on the surroundings.
#pragma clang fp contract(on)
namespace slamhip_dev {

struct CamPair {
  double PL[12];
  double PR[12];
};

constexpr int kMaxSweeps = 12;

// Rotate columns p and q of the 4x4 A (a[row][col]) and of V so that the two
// columns of A become orthogonal.  Returns true when a rotation was applied.
template <int P, int Qc>
__device__ __forceinline__ bool jacobi_pair(double (&a)[4][4], double (&v)[4][4]) {
  double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    alpha = fma(a[i][P], a[i][P], alpha);
    beta = fma(a[i][Qc], a[i][Qc], beta);
    gamma = fma(a[i][P], a[i][Qc], gamma);
  }
  // Columns already orthogonal to working precision: |gamma| <= eps sqrt(alpha beta), squared.
  if (gamma * gamma <= 4.930380657631324e-32 * (alpha * beta)) return false;
  // The rotation with tan(theta) = t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (beta - alpha) /
  // (2 gamma), written without a division: with d = beta - alpha, g = 2 gamma, h = hypot(d, g) and
  // D = |d| + h it is t = sign(d) g / D, so cos = D / sqrt(D^2 + g^2), sin = sign(d) g / sqrt(D^2 + g^2).
  // (Two reciprocal square roots; the quotient-and-root form costs three divisions and two roots,
  // which was most of the kernel's time.)
  const double d = beta - alpha, g = gamma + gamma;
  const double hh = fma(d, d, g * g);
  const double D = fabs(d) + hh * rsqrt(hh);
  const double r = rsqrt(fma(D, D, g * g));
  const double c = D * r;
  const double s = (d < 0.0 ? -g : g) * r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double ap = a[i][P], aq = a[i][Qc];
    a[i][P] = c * ap - s * aq;
    a[i][Qc] = s * ap + c * aq;
    const double vp = v[i][P], vq = v[i][Qc];
    v[i][P] = c * vp - s * vq;
    v[i][Qc] = s * vp + c * vq;
  }
  return true;
}

// The same vector by inverse iteration on B = A^T A (x <- B^-1 x, normalised): the error shrinks by
// (s4 / s3)^2 per step, and a match that has passed the row filter has s3 / s4 of some hundred (its two
// rays nearly meet), so three or four steps of two triangular solves reach what a step can no longer
// change -- a few hundred instructions where the Jacobi sweeps below take some two thousand.  B is
// factored once (Cholesky with the pivots' reciprocal roots kept; a pivot that rounding has pushed to
// zero or below, as for rays that meet exactly, is lifted to 2^-60 trace(B): a perturbation of B below
// its own rounding, and inverse iteration does not mind a nearly singular factor -- it is what singles
// the vector out).  Forming B squares the condition number, but the vector's error is eps s1^2 / s3^2
// (s1 / s3 is about 5 for stereo geometry), not eps s1^2 / s4^2.  Returns false when the steps have not
// settled (s4 close to s3: no well-defined answer, or a matrix of zeros); the caller then takes the sweeps.
constexpr int kInverseIterations = 16;
__device__ __forceinline__ bool null_vector_by_inverse_iteration(const double (&a)[4][4], double (&x)[4]) {
  double b00 = 0.0, b01 = 0.0, b02 = 0.0, b03 = 0.0, b11 = 0.0, b12 = 0.0, b13 = 0.0, b22 = 0.0, b23 = 0.0,
         b33 = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b00 = fma(a[i][0], a[i][0], b00); b01 = fma(a[i][0], a[i][1], b01); b02 = fma(a[i][0], a[i][2], b02);
    b03 = fma(a[i][0], a[i][3], b03); b11 = fma(a[i][1], a[i][1], b11); b12 = fma(a[i][1], a[i][2], b12);
    b13 = fma(a[i][1], a[i][3], b13); b22 = fma(a[i][2], a[i][2], b22); b23 = fma(a[i][2], a[i][3], b23);
    b33 = fma(a[i][3], a[i][3], b33);
  }
  const double lift = ((b00 + b11) + (b22 + b33)) * 8.673617379884035e-19;   // 2^-60 trace
  const double i0 = rsqrt(fmax(b00, lift));
  const double l10 = b01 * i0, l20 = b02 * i0, l30 = b03 * i0;
  const double i1 = rsqrt(fmax(fma(-l10, l10, b11), lift));
  const double l21 = fma(-l20, l10, b12) * i1, l31 = fma(-l30, l10, b13) * i1;
  const double i2 = rsqrt(fmax(fma(-l21, l21, fma(-l20, l20, b22)), lift));
  const double l32 = fma(-l31, l21, fma(-l30, l20, b23)) * i2;
  const double i3 = rsqrt(fmax(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, b33))), lift));
  // start: a point on the optical axis in front of the cameras, (0, 0, 3/4, 1) normalised
  double x0 = 0.0, x1 = 0.0, x2 = 0.6, x3 = 0.8;
  bool settled = false;
  for (int it = 0; it < kInverseIterations && !settled; ++it) {
    const double z0 = x0 * i0;
    const double z1 = fma(-l10, z0, x1) * i1;
    const double z2 = fma(-l21, z1, fma(-l20, z0, x2)) * i2;
    const double z3 = fma(-l32, z2, fma(-l31, z1, fma(-l30, z0, x3))) * i3;
    double y3 = z3 * i3;
    double y2 = fma(-l32, y3, z2) * i2;
    double y1 = fma(-l31, y3, fma(-l21, y2, z1)) * i1;
    double y0 = fma(-l30, y3, fma(-l20, y2, fma(-l10, y1, z0))) * i0;
    const double s = rsqrt(fma(y0, y0, fma(y1, y1, fma(y2, y2, y3 * y3))));
    y0 *= s; y1 *= s; y2 *= s; y3 *= s;
    // B^-1 is positive definite, so the step keeps the vector's sign: compare as they are
    const double change = fmax(fmax(fabs(y0 - x0), fabs(y1 - x1)), fmax(fabs(y2 - x2), fabs(y3 - x3)));
    settled = change <= 1e-13;      // false for NaN
    x0 = y0; x1 = y1; x2 = y2; x3 = y3;
  }
  x[0] = x0; x[1] = x1; x[2] = x2; x[3] = x3;
  return settled;
}

// Unit right singular vector of the smallest singular value of A.
__device__ __forceinline__ void smallest_right_singular_vector(double (&a)[4][4], double (&x)[4]) {
  if (null_vector_by_inverse_iteration(a, x)) return;
  double v[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;

  for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
    bool any = false;
    any |= jacobi_pair<0, 1>(a, v);
    any |= jacobi_pair<0, 2>(a, v);
    any |= jacobi_pair<0, 3>(a, v);
    any |= jacobi_pair<1, 2>(a, v);
    any |= jacobi_pair<1, 3>(a, v);
    any |= jacobi_pair<2, 3>(a, v);
    if (!any) break;
  }
  double n[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    n[j] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) n[j] = fma(a[i][j], a[i][j], n[j]);
  }
  // Column with the smallest norm (lowest index on exact ties).
  int k = 0;
  double best = n[0];
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (n[j] < best) { best = n[j]; k = j; }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    x[i] = k == 0 ? v[i][0] : (k == 1 ? v[i][1] : (k == 2 ? v[i][2] : v[i][3]));
  // V is a product of rotations, so the column is unit length up to rounding;
  // renormalise so the fp32 rounding below sees a unit vector like an SVD's.
  const double inv = rsqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] *= inv;
}


// Stereo filter decision (frontend/geometry/triangulation.py:104).
__device__ __forceinline__ bool stereo_keep(double uL, double vL, double uR, double vR,
                                            double pix_thresh, double disp_min) {
  return (fabs(vR - vL) <= pix_thresh) && ((uL - uR) >= disp_min);
}

// DLT system in cv2.triangulatePoints row order -> unit null vector x[4].
__device__ __forceinline__ void dlt_null_vector(const CamPair &cam, double uL, double vL, double uR,
                                                double vR, double (&x)[4]) {
  double a[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    a[0][c] = uL * cam.PL[8 + c] - cam.PL[0 + c];
    a[1][c] = vL * cam.PL[8 + c] - cam.PL[4 + c];
    a[2][c] = uR * cam.PR[8 + c] - cam.PR[0 + c];
    a[3][c] = vR * cam.PR[8 + c] - cam.PR[4 + c];
  }
  smallest_right_singular_vector(a, x);
}

}  // namespace slamhip_dev
#pragma clang fp contract(fast)   // the compiler's default for device code, for whatever follows

