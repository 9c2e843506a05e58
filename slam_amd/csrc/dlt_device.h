// Device code shared by the triangulation kernels: one-sided (Hestenes) Jacobi
// solve of the 4x4 DLT system, one problem per lane, fp64.
#pragma once
#include <hip/hip_runtime.h>

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

// Unit right singular vector of the smallest singular value of A.
__device__ __forceinline__ void smallest_right_singular_vector(double (&a)[4][4], double (&x)[4]) {
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
