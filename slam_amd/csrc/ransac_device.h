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

// ---------------------------------------------------------------------------
// Packed-fp32 screening of the same test.
//
// Two points per instruction (v_pk_fma_f32 and friends) decide every
// (hypothesis, point, camera) whose fp32 evaluation of
//     diff = (u*w - x)^2 + (v*w - y)^2 - thr^2 * w^2
// is farther from zero than a rigorous bound E on |diff_fp32 - diff|; anything
// closer (or not finite) is re-evaluated by the fp64 path above, so the
// decisions, and therefore the counts, are exactly those of the fp64 path.
//
// Bound (eps = 2^-24; points and pixels are exact fp32 inputs; every fp32
// operation is rounded once or, if a multiply-add is not fused, twice):
//   P32 = fl(P):                     |P32 - P| <= eps |P|
//   x32 = P32 row . [X Y Z 1]:       |x32 - x| <= 8 eps A s,  s = |X|+|Y|+|Z|+1,
//                                    A = max |entry| of the row (Axy rows 0-1, Aw row 2)
//   d32 = u*w32 - x32:               |d32 - d| <= D + 2 eps |d32|,
//                                    D = 8 eps (umax s Aw + s Axy),  umax = max |pixel coordinate|
//   lhs = dx^2 + dy^2:               |lhs32 - lhs| <= 2 a D + 4 D^2 + 8 eps lhs32,  a = |dx32|+|dy32|
//   rhs = thr^2 w^2:                 |rhs32 - rhs| <= thr^2 (2 |w32| dw + dw^2) + 8 eps rhs32,
//                                    dw = 8 eps Aw s
// The cross terms are split with 2xy <= k x^2 + y^2/k (k = 2^-10; a <= sqrt(2 lhs)):
//   2 a D <= 2k lhs + D^2/k,   2 |w| dw <= k w^2 + dw^2/k,   hence
//   E = (2k + 8 eps)(lhs32 + rhs32) + (1/k + 4) D^2 + thr^2 (1 + 1/k) dw^2,
// i.e. one multiply-add per camera on top of a per-(hypothesis, point) constant,
// with no absolute values.  D and dw are evaluated with their constants doubled,
// which also covers the rounding of E itself.  The relative part keeps E above
// 2e-3 of either side, far outside the 1e-10 band in which the fp64 path falls
// back to the literal formula; in pixels the undecided band is about a thousandth
// of the threshold.
// ---------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kPairStride = 18;  // floats per staged pair: X Y Z uL vL uR vR s m, two lanes each
constexpr float kEps32 = 5.9604645e-8f;
constexpr float kAbsErr = 2.0f * 8.0f * kEps32 * 1.01f;   // doubled 8 eps
constexpr float kSplit = 1.0f / 1024.0f;                  // k of the cross-term split
constexpr float kRelErr = 2.0f * kSplit + 2.0f * 8.0f * kEps32 * 1.01f;

struct Hyp32 {
  float PL[12], PR[12];
  float Axy, Aw;
};

__device__ __forceinline__ void make_hyp32(const double (&PL)[12], const double (&PR)[12], Hyp32 &h) {
  double axy = 0.0, aw = 0.0;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    h.PL[i] = (float)PL[i];
    h.PR[i] = (float)PR[i];
    const double m = fmax(fabs(PL[i]), fabs(PR[i]));
    if (i < 8) axy = fmax(axy, m); else aw = fmax(aw, m);
  }
  h.Axy = (float)(axy * 1.000001);   // rounded up
  h.Aw = (float)(aw * 1.000001);
}

// Writes the staged record of the point pair (a, b) (b = a when the chunk has an
// odd point count; the caller ignores that second lane).
__device__ __forceinline__ void stage_pair(float *__restrict__ o, const float *__restrict__ X,
                                           const float *__restrict__ mL,
                                           const float *__restrict__ mR, size_t a, size_t b) {
  const size_t idx[2] = {a, b};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const size_t g = idx[k];
    const float x = X[3 * g], y = X[3 * g + 1], z = X[3 * g + 2];
    const float ul = mL[2 * g], vl = mL[2 * g + 1], ur = mR[2 * g], vr = mR[2 * g + 1];
    const float s = (fabsf(x) + fabsf(y) + fabsf(z) + 1.0f) * 1.000001f;
    const float um = fmaxf(fmaxf(fabsf(ul), fabsf(vl)), fmaxf(fabsf(ur), fabsf(vr)));
    o[0 + k] = x; o[2 + k] = y; o[4 + k] = z;
    o[6 + k] = ul; o[8 + k] = vl; o[10 + k] = ur; o[12 + k] = vr;
    o[14 + k] = s; o[16 + k] = um * s * 1.000001f;
  }
}

// diff = lhs - rhs and sum = lhs + rhs for one camera and a pair of points
__device__ __forceinline__ void screen_camera(const float (&P)[12], f2 X, f2 Y, f2 Z, f2 u, f2 v,
                                              float thr2f, f2 &diff, f2 &sum) {
  const f2 x = P[0] * X + (P[1] * Y + (P[2] * Z + P[3]));
  const f2 y = P[4] * X + (P[5] * Y + (P[6] * Z + P[7]));
  const f2 w = P[8] * X + (P[9] * Y + (P[10] * Z + P[11]));
  const f2 dx = u * w - x;
  const f2 dy = v * w - y;
  const f2 lhs = dx * dx + dy * dy;
  const f2 rhs = thr2f * (w * w);
  diff = lhs - rhs;
  sum = lhs + rhs;
}

// Inliers (both cameras) among `cnt` points staged as pairs (stage_pair).
__device__ __forceinline__ int count_chunk_inliers_screened(const float *__restrict__ lds, int cnt,
                                                            const double (&PL)[12],
                                                            const double (&PR)[12], const Hyp32 &h,
                                                            double thr, double thr2) {
  const float thr2f = (float)thr2;
  const float thr2up = thr2f * 1.01f;
  int c = 0;
  const int pairs = (cnt + 1) >> 1;
  for (int j = 0; j < pairs; ++j) {
    const f2 *q = reinterpret_cast<const f2 *>(lds + j * kPairStride);
    const f2 X = q[0], Y = q[1], Z = q[2], s = q[7], m = q[8];
    const f2 dw = (kAbsErr * h.Aw) * s;
    const f2 D = kAbsErr * (m * h.Aw + s * h.Axy);
    const f2 C = (1.0f / kSplit + 4.0f) * (D * D) + (thr2up * (1.0f + 1.0f / kSplit)) * (dw * dw);
    f2 dL, sL, dR, sR;
    screen_camera(h.PL, X, Y, Z, q[3], q[4], thr2f, dL, sL);
    screen_camera(h.PR, X, Y, Z, q[5], q[6], thr2f, dR, sR);
    const f2 eL = kRelErr * sL + C, eR = kRelErr * sR + C;
    const bool has1 = 2 * j + 1 < cnt;
    // NaN or infinity on either side makes `sure` false: the fp64 path decides
    const bool outL0 = dL.x > eL.x, outR0 = dR.x > eR.x, outL1 = dL.y > eL.y, outR1 = dR.y > eR.y;
    bool in0 = dL.x < -eL.x && dR.x < -eR.x, in1 = dL.y < -eL.y && dR.y < -eR.y;
    // decided: surely inside both cameras, or surely outside one of them
    const bool ok0 = in0 || outL0 || outR0;
    const bool ok1 = !has1 || in1 || outL1 || outR1;
    if (!(ok0 && ok1)) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (k == 0 ? ok0 : ok1) continue;
        const float *f = lds + j * kPairStride + k;
        const double Xd = (double)f[0], Yd = (double)f[2], Zd = (double)f[4];
        const bool in = inlier_one_camera(PL, Xd, Yd, Zd, (double)f[6], (double)f[8], thr, thr2) &&
                        inlier_one_camera(PR, Xd, Yd, Zd, (double)f[10], (double)f[12], thr, thr2);
        if (k == 0) in0 = in; else in1 = in;
      }
    }
    c += (in0 ? 1 : 0) + ((has1 && in1) ? 1 : 0);
  }
  return c;
}

}  // namespace slamhip_dev
