// Device code shared by the consensus-scoring kernels: projection matrices of
// a hypothesis and the per-point, per-camera inlier decision (fp64).
#pragma once
#include <hip/hip_runtime.h>

namespace slamhip_dev {

constexpr int kPtsPerChunk = 128;  // points staged in LDS per block
constexpr int kPtStride = 8;       // doubles per staged point (7 used)

struct RansacCam {
  double K[9];
  double MR[12];
};

__device__ __forceinline__ void make_projections(const double *__restrict__ Rt,
                                                 const RansacCam &cam, double (&PL)[12],
                                                 double (&PR)[12]) {
  double rt[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) rt[i] = Rt[i];
  double rr[12];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double s = cam.MR[4 * i + 0] * rt[0 + j];
      s = fma(cam.MR[4 * i + 1], rt[4 + j], s);
      s = fma(cam.MR[4 * i + 2], rt[8 + j], s);
      if (j == 3) s += cam.MR[4 * i + 3];
      rr[4 * i + j] = s;
    }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double a = cam.K[3 * i + 0] * rt[0 + j];
      a = fma(cam.K[3 * i + 1], rt[4 + j], a);
      a = fma(cam.K[3 * i + 2], rt[8 + j], a);
      PL[4 * i + j] = a;
      double b = cam.K[3 * i + 0] * rr[0 + j];
      b = fma(cam.K[3 * i + 1], rr[4 + j], b);
      b = fma(cam.K[3 * i + 2], rr[8 + j], b);
      PR[4 * i + j] = b;
    }
}

// Literal reference sequence for one camera (used for borderline points only).
__device__ __forceinline__ bool inlier_exact(double x, double y, double w, double mu, double mv,
                                          double thr) {
  const double u = x / w, v = y / w;
  const double dx = __dsub_rn(mu, u), dy = __dsub_rn(mv, v);
  const double e = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
  return e <= thr;  // NaN (w == 0) compares false, as in NumPy
}

__device__ __forceinline__ bool inlier_one_camera(const double (&P)[12], double X, double Y,
                                                  double Z, double mu, double mv, double thr,
                                                  double thr2) {
  const double x = fma(P[0], X, fma(P[1], Y, fma(P[2], Z, P[3])));
  const double y = fma(P[4], X, fma(P[5], Y, fma(P[6], Z, P[7])));
  const double w = fma(P[8], X, fma(P[9], Y, fma(P[10], Z, P[11])));
  const double dx = fma(mu, w, -x);
  const double dy = fma(mv, w, -y);
  const double lhs = fma(dx, dx, dy * dy);
  const double rhs = thr2 * (w * w);
  bool in = lhs <= rhs;
  if (!(fabs(lhs - rhs) > 1e-10 * rhs)) in = inlier_exact(x, y, w, mu, mv, thr);
  return in;
}

// Inliers (both cameras) among `cnt` points staged in LDS as kPtStride doubles
// each (X, Y, Z, uL, vL, uR, vR, pad).
__device__ __forceinline__ int count_chunk_inliers(const double *__restrict__ lds, int cnt,
                                                   const double (&PL)[12], const double (&PR)[12],
                                                   double thr, double thr2) {
  int c = 0;
  for (int i = 0; i < cnt; ++i) {
    const double *q = lds + i * kPtStride;
    const bool inL = inlier_one_camera(PL, q[0], q[1], q[2], q[3], q[4], thr, thr2);
    const bool inR = inlier_one_camera(PR, q[0], q[1], q[2], q[5], q[6], thr, thr2);
    c += (inL && inR) ? 1 : 0;
  }
  return c;
}

}  // namespace slamhip_dev
