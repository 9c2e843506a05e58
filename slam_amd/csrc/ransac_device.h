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

// thr^2 for the squared form of `error <= thr`; a negative or NaN threshold accepts
// nothing in the reference (pnp.py:86-91), and so does lhs <= -w^2.
__device__ __forceinline__ double threshold_squared(double thr) { return thr >= 0.0 ? thr * thr : -1.0; }

// Literal reference sequence for one camera (used for borderline points only).
__device__ __forceinline__ bool inlier_exact(double x, double y, double w, double mu, double mv,
                                          double thr) {
  // every operation rounded on its own, as NumPy does it: hipcc would otherwise contract
  // dx*dx + dy*dy into a fused multiply-add (it does so through __dmul_rn / __dadd_rn too,
  // which the pragma does not reach: plain operators)
#pragma clang fp contract(off)
  const double u = x / w, v = y / w;
  const double dx = mu - u, dy = mv - v;
  const double e = sqrt(dx * dx + dy * dy);
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
// Bound (eps = 2^-24; points and pixels are exact fp32 inputs; every product-sum
// below is an explicit fused multiply-add, rounded once).  With s = |X|+|Y|+|Z| and,
// per group of rows (xy = rows 0-1, w = row 2, both cameras), A3 = max |entry| of the
// 3x3 part and At = max |entry| of the last column:
//   P32 = fl(P):                     |P32 - P| <= eps |P|
//   x32 = fma(P0,X, fma(P1,Y, fma(P2,Z, P3))):
//                                    |x32 - x| <= 4 eps (A3 s + At) (1 + 2 eps)
//   rectified pair (below), right x: x32' = fl(x32 - fl(c)) adds eps |c| + eps |x32'|
//                                    <= eps (A3 s + 3 At)  (|c| <= 2 At), so
//                                    8 eps (A3 s + At) holds for every projected
//                                    coordinate; the code uses 8 eps (1.01)
//   d32 = fma(u, w32, -x32):         |d32 - d| <= D + 2 eps |d32|,   D = s Da + Db,
//                                    Da = 8 eps (umax A3w + A3xy), Db = 8 eps (umax Atw + Atxy),
//                                    umax = max |pixel coordinate|
//   lhs = dx^2 + dy^2:               |lhs32 - lhs| <= 2 a D + 4 D^2 + 8 eps lhs32,  a = |dx32|+|dy32|
//   rhs = thr^2 w^2:                 |rhs32 - rhs| <= thr^2 (2 |w32| dw + dw^2) + 8 eps rhs32,
//                                    dw = s dwa + dwb,  dwa = 8 eps A3w,  dwb = 8 eps Atw
// The cross terms are split with 2xy <= k x^2 + y^2/k for any k > 0 (a <= sqrt(2 lhs)),
//   2 a D <= 2k lhs + D^2/k,   2 |w| dw <= k w^2 + dw^2/k,
// and the squares with (s p + q)^2 <= 2 s^2 p^2 + 2 q^2, hence
//   E = r (lhs32 + rhs32) + s^2 Ca + Cb,   r = 2k + 16 eps,
//   Ca = 2 (1/k + 4) Da^2 + 2 thr^2 (1 + 1/k) dwa^2,  Cb = the same of Db and dwb.
// With lhs32 + rhs32 = 2 rhs32 + diff32, diff32 > E is diff32 (1 - r) > 2 r rhs32 + C
// and diff32 < -E is diff32 (1 + r) < -(2 r rhs32 + C), so one threshold serves both:
//   T = (2 r thr^2 w32^2 + s^2 Ca + Cb) / (1 - r):   diff32 > T  => outside for sure,
//                                                    diff32 < -T => inside for sure.
// Ca, Cb and r belong to the hypothesis (umax is the maximum over the staged chunk)
// and s^2 to the point, staged rounded up next to it: T costs two multiply-adds per
// camera, and only two in all when the cameras share w (below), with no absolute
// values.  Keeping the last column apart matters for hypotheses far from the scene
// (|t| >> s), which random minimal samples produce all the time: there x is about P3
// and its rounding error about eps |P3|, not eps |P3| s.  k is chosen per hypothesis,
// k ~ (Da + Db) / (thr (A3w + Atw)), which balances the two parts of E for a point at
// the threshold; any k keeps the bound valid.  With KITTI numbers the undecided band
// is about 0.2 % of the threshold, far outside the 1e-10 band in which the fp64 path
// falls back to the literal formula.
//
// Domain.  The bound assumes that nothing overflows.  The matrices are screened as P times the
// power of two that brings their largest entry into [0.5, 1) (make_hyp32; the test is
// homogeneous in P, the scaling is exact), and screening is confined to s <= 1e4, umax <= 1e5
// and thr <= 1e3: then |x|, |y|, |w| <= 1e4 + 1, |d| <= 1e9 + 1e4 and lhs, rhs < 1e19.  Outside,
// or with anything not finite, T is made NaN or infinite (Ca for a hypothesis or a chunk, s^2
// for a point), no comparison with it holds and the fp64 path decides.  An entry of the scaled
// P below 2^-126 loses at most 2^-126 absolutely when it is rounded to fp32 (or flushed), which
// moves x, y, w by at most 1.2e-34 (s <= 1e4), d by 1.2e-29 and lhs by 4 |d| 1.2e-29 <= 5e-20,
// rhs by less; results that underflow later carry less still.  Cb has a floor of 1e-18 for
// these, and strict comparisons leave diff = T = 0 undecided.
//
// Rectified pair: when the right camera is the left one shifted along x
// (M_right = [I | (tx, 0, 0)]) and K is upper triangular, make_projections gives
// P_right equal to P_left in every entry but [3] (products with the exact 0 and 1
// entries are exact), so y and w are shared by both cameras and the right x is the
// left x minus c = P_left[3] - P_right[3]: 7 packed operations for the second camera
// instead of 16, and one threshold T for both.  The host decides this from K and
// M_right (rectified_stereo()).  The fp64 path uses the real last column of P_right.
// ---------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
#ifdef SLAMHIP_DEBUG_SPARSE
static __device__ unsigned long long g_dbg[2];
#endif

constexpr int kPairStride = 16;  // floats per staged pair: X Y Z uL vL uR vR s^2 (s = |X|+|Y|+|Z|), two lanes each
constexpr float kEps32 = 5.9604645e-8f;
constexpr float kAbsErr = 8.0f * kEps32 * 1.01f;
constexpr float kRelEps = 16.0f * kEps32 * 1.01f;
constexpr double kScreenMaxThr = 1e3;   // screening domain (see above)
constexpr double kScreenMaxP = 24.0;    // sum |entries| of the scaled matrices (finite <=> below it)
constexpr float kScreenMaxS = 1e4f;
constexpr float kScreenMaxPix = 1e5f;

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }

struct Hyp32 {
  float PL[12], PR[12];
  float A3xy, Atxy, A3w, Atw;  // max |entry|: 3x3 part / last column of rows 0-1 / row 2
  float cR;                    // P_left[3] - P_right[3] (rectified pair)
};

// Per hypothesis and staged chunk: T = aT w^2 + s^2 ca + cb.
struct Screen32 {
  float aT, ca, cb;
};

// True when P_right differs from P_left in entry [3] only (see above).
static inline bool rectified_stereo(const double *K, const double *MR) {
  static const double I34[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  for (int i = 0; i < 12; ++i)
    if (i != 3 && MR[i] != I34[i]) return false;
  return K[3] == 0.0 && K[6] == 0.0 && K[7] == 0.0;
}

// 0 = fp64 only, 1 = packed-fp32 screen, 2 = the same for a rectified pair.
static inline int screening_mode(int enabled, const double *K, const double *MR, double thr) {
  if (!enabled || !(thr >= 0.0 && thr <= kScreenMaxThr)) return 0;
  return rectified_stereo(K, MR) ? 2 : 1;
}

__device__ __forceinline__ void make_hyp32(const double (&PL)[12], const double (&PR)[12], Hyp32 &h) {
  // The test is homogeneous in the projection matrices (x, y and w scale together, lhs and rhs
  // with the square), so the fp32 image is taken of both matrices times one power of two that
  // brings their largest entry into [0.5, 1): exact, and no hypothesis is outside the screen's
  // range for its size alone.  That matters: a minimal sample with a repeated pixel (two
  // matches of one keypoint) yields finite poses with |t| of 1e7 ... 1e31, about one hypothesis
  // in two thousand, and each of them would otherwise send its whole wavefront through the
  // fp64 path for every point.
  double big = 0.0;
#pragma unroll
  for (int i = 0; i < 12; ++i) big = fmax(big, fmax(fabs(PL[i]), fabs(PR[i])));   // fmax drops NaN: see `total`
  int e = 0;
  if (big > 0.0 && big <= 1.7976931348623157e308) (void)frexp(big, &e);
  const double scale = ldexp(1.0, -e);
  double a3xy = 0.0, atxy = 0.0, a3w = 0.0, atw = 0.0, total = 0.0;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const double l = PL[i] * scale, r = PR[i] * scale;
    h.PL[i] = (float)l;
    h.PR[i] = (float)r;
    const double m = fmax(fabs(l), fabs(r));
    total += fabs(l) + fabs(r);   // NaN or infinite if any entry is
    if (i < 8) {
      if ((i & 3) == 3) atxy = fmax(atxy, m); else a3xy = fmax(a3xy, m);
    } else {
      if (i == 11) atw = m; else a3w = fmax(a3w, m);
    }
  }
  h.A3xy = (total <= kScreenMaxP) ? (float)(a3xy * 1.000001) : __builtin_nanf("");   // rounded up
  h.Atxy = (float)(atxy * 1.000001);
  h.A3w = (float)(a3w * 1.000001);
  h.Atw = (float)(atw * 1.000001);
  h.cR = (float)((PL[3] - PR[3]) * scale);
}

// Writes the staged record of the point pair (a, b) (b = a when the chunk has an
// odd point count; the caller ignores that second lane) and raises umax_bits to the
// bit pattern of the largest |pixel coordinate| of the pair (a NaN coordinate has
// the largest pattern of all and stays) and s2max_bits to that of the largest staged s^2.
__device__ __forceinline__ void stage_pair(float *__restrict__ o, const float *__restrict__ X,
                                           const float *__restrict__ mL,
                                           const float *__restrict__ mR, size_t a, size_t b,
                                           unsigned &umax_bits, unsigned &s2max_bits) {
  const size_t idx[2] = {a, b};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const size_t g = idx[k];
    const float x = X[3 * g], y = X[3 * g + 1], z = X[3 * g + 2];
    const float ul = mL[2 * g], vl = mL[2 * g + 1], ur = mR[2 * g], vr = mR[2 * g + 1];
    const float s = (fabsf(x) + fabsf(y) + fabsf(z)) * 1.000001f;
    o[0 + k] = x; o[2 + k] = y; o[4 + k] = z;
    o[6 + k] = ul; o[8 + k] = vl; o[10 + k] = ur; o[12 + k] = vr;
    o[14 + k] = (s <= kScreenMaxS) ? (s * s) * 1.000001f : __builtin_inff();   // NaN lands here too
    s2max_bits = max(s2max_bits, __float_as_uint(o[14 + k]));                 // not negative: bits order like floats
    umax_bits = max(max(umax_bits, __float_as_uint(fabsf(ul))), __float_as_uint(fabsf(vl)));
    umax_bits = max(max(umax_bits, __float_as_uint(fabsf(ur))), __float_as_uint(fabsf(vr)));
  }
}

// Maximum over the 64 lanes of a one-wavefront block.  Rows of 16 lanes reduce
// through DPP, the four rows through scalar registers.  Bit patterns of floats
// that are not negative order like the floats, with NaN above infinity.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));   // lanes 1,0,3,2
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));   // lanes 2,3,0,1
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));  // half-row mirror
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));  // row mirror
  const unsigned r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
  return max(max(r0, r1), max(r2, r3));
}

// thr in [0, kScreenMaxThr]; umax_bits from stage_pair / wave_max_u32.
__device__ __forceinline__ Screen32 make_screen(const Hyp32 &h, unsigned umax_bits, float thr, float thr2f) {
  const float umax = __uint_as_float(umax_bits);
  const float Da = kAbsErr * (umax * h.A3w + h.A3xy) * 1.000001f;
  const float Db = kAbsErr * (umax * h.Atw + h.Atxy) * 1.000001f;
  const float dwa = kAbsErr * h.A3w, dwb = kAbsErr * h.Atw;
  // any positive k is valid; the clamp also keeps it positive when a factor is not finite
  const float k = fminf(fmaxf(0.75f * (Da + Db) / (thr * (h.A3w + h.Atw)), 9.5367432e-7f), 0.0625f);
  const float ik = 1.0f / k * 1.000001f;
  const float g1 = 2.0f * (ik + 4.0f), g2 = 2.0f * thr2f * 1.01f * (1.0f + ik);
  const float r = (2.0f * k + kRelEps) * 1.000001f;   // <= 0.126
  const float inv = 1.00001f / (1.0f - r);
  Screen32 sc;
  sc.aT = 2.0f * r * thr2f * 1.000001f * inv;
  sc.ca = (g1 * (Da * Da) + g2 * (dwa * dwa)) * inv;
  sc.cb = (g1 * (Db * Db) + g2 * (dwb * dwb)) * inv + 1e-18f;
  if (!(umax <= kScreenMaxPix)) sc.ca = __builtin_nanf("");
  return sc;
}

// The packed-fp32 screen of one staged pair of points: in* = inside both cameras for sure;
// open* = lanes (of those active) whose point 0 / 1 is not decided -- neither inside for sure
// nor outside one camera for sure.  The lane masks come from ballots of the bare comparisons
// and are combined as 64-bit integers, which stays in scalar registers.
template <bool RECT>
__device__ __forceinline__ void screen_pair(const float *__restrict__ rec, const Hyp32 &h, f2 nthr2f,
                                            f2 aT, f2 ca, f2 cb, bool &in0, bool &in1,
                                            unsigned long long &open0, unsigned long long &open1) {
  const f2 *q = reinterpret_cast<const f2 *>(rec);
  const f2 X = q[0], Y = q[1], Z = q[2];
  const f2 C = fma2(q[7], ca, cb);
  const f2 x = fma2(splat(h.PL[0]), X, fma2(splat(h.PL[1]), Y, fma2(splat(h.PL[2]), Z, splat(h.PL[3]))));
  const f2 y = fma2(splat(h.PL[4]), X, fma2(splat(h.PL[5]), Y, fma2(splat(h.PL[6]), Z, splat(h.PL[7]))));
  const f2 w = fma2(splat(h.PL[8]), X, fma2(splat(h.PL[9]), Y, fma2(splat(h.PL[10]), Z, splat(h.PL[11]))));
  const f2 dx = fma2(q[3], w, -x), dy = fma2(q[4], w, -y);
  const f2 w2 = w * w;
  const f2 dL = fma2(nthr2f, w2, fma2(dx, dx, dy * dy));   // lhs - rhs, left camera
  const f2 TL = fma2(aT, w2, C);
  const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
  if (RECT) {
    const f2 xr = x - h.cR;
    const f2 dxr = fma2(q[5], w, -xr), dyr = fma2(q[6], w, -y);
    const f2 dR = fma2(nthr2f, w2, fma2(dxr, dxr, dyr * dyr));
    // one threshold for both cameras: the larger diff decides.  fmaxf drops a NaN
    // operand, but whatever makes one diff NaN makes T NaN or infinite as well
    const float m0 = fmaxf(dL.x, dR.x), m1 = fmaxf(dL.y, dR.y);
    in0 = m0 < -TL.x; in1 = m1 < -TL.y;
    open0 = active & ~(__builtin_amdgcn_ballot_w64(in0) | __builtin_amdgcn_ballot_w64(m0 > TL.x));
    open1 = active & ~(__builtin_amdgcn_ballot_w64(in1) | __builtin_amdgcn_ballot_w64(m1 > TL.y));
  } else {
    const f2 xr = fma2(splat(h.PR[0]), X, fma2(splat(h.PR[1]), Y, fma2(splat(h.PR[2]), Z, splat(h.PR[3]))));
    const f2 yr = fma2(splat(h.PR[4]), X, fma2(splat(h.PR[5]), Y, fma2(splat(h.PR[6]), Z, splat(h.PR[7]))));
    const f2 wr = fma2(splat(h.PR[8]), X, fma2(splat(h.PR[9]), Y, fma2(splat(h.PR[10]), Z, splat(h.PR[11]))));
    const f2 dxr = fma2(q[5], wr, -xr), dyr = fma2(q[6], wr, -yr);
    const f2 wr2 = wr * wr;
    const f2 dR = fma2(nthr2f, wr2, fma2(dxr, dxr, dyr * dyr));
    const f2 TR = fma2(aT, wr2, C);
    in0 = dL.x < -TL.x && dR.x < -TR.x; in1 = dL.y < -TL.y && dR.y < -TR.y;
    // decided: surely inside both cameras, or surely outside one of them
    open0 = active & ~((__builtin_amdgcn_ballot_w64(dL.x < -TL.x) & __builtin_amdgcn_ballot_w64(dR.x < -TR.x)) |
                       __builtin_amdgcn_ballot_w64(dL.x > TL.x) | __builtin_amdgcn_ballot_w64(dR.x > TR.x));
    open1 = active & ~((__builtin_amdgcn_ballot_w64(dL.y < -TL.y) & __builtin_amdgcn_ballot_w64(dR.y < -TR.y)) |
                       __builtin_amdgcn_ballot_w64(dL.y > TL.y) | __builtin_amdgcn_ballot_w64(dR.y > TR.y));
  }
}

// The left camera alone (rectified pair), one comparison per point: the lanes (of those active)
// for which this cannot put both points of the staged pair outside for sure.
//
// screen_pair's verdict "outside for sure" is max(dL, dR) > T with, for the left camera,
//     dL = fl(n w2 + inner),   T = fl(aT w2 + C),   n = -thr^2,  C = fl(s^2 ca + cb)
// (w2 = fl(w w), inner = fl(dx dx + fl(dy dy)): the same fp32 numbers here and there).  This
// evaluates D = fl(m w2 + inner) > Cx instead, with two constants of the hypothesis and chunk
// (left_gate):  m <= n - aT (1 + 2.1 eps)  and  Cx >= Cmax (1 + 4 eps) + 4 eta,  Cmax >= every
// C of the chunk, eta = 2^-150.  Proof that D > Cx implies dL > T (finite operands; eps = 2^-24):
// with the real numbers E_L = n w2 + inner, E_T = aT w2 + C >= 0, E_D = m w2 + inner, rounding
// is monotonic and Cx is a float, so D > Cx gives E_D > Cx; then
//     E_L - E_T >= E_D + 2.1 eps aT w2 - C > 4 eps Cmax + 4 eta + 2.1 eps aT w2
//                                          >= (2 eps E_T + 2 eta) / (1 - eps),
// which is E_L (1 - eps) - eta > E_T (1 + eps) + eta, and dL >= E_L (1 - eps) - eta (E_L > 0),
// T <= E_T (1 + eps) + eta.  Infinite or NaN operands: w2 infinite makes D -inf or NaN (open);
// inner infinite with w2 finite makes D = dL = +inf > T; a NaN or infinite Cx (anything outside
// the screen's domain) compares false, so everything stays open and screen_pair decides.
// For a lane not in the result screen_pair counts nothing and leaves nothing undecided.
struct LeftGate {
  float m, cx;
};
__device__ __forceinline__ LeftGate left_gate(Screen32 sc, float thr2f, unsigned s2max_bits) {
  LeftGate g;
  // fl(thr^2 + aT) >= (thr^2 + aT)(1 - eps), times 1.000001 rounded: >= (thr^2 + aT)(1 + 8e-7)
  // >= thr^2 + aT (1 + 2.1 eps)
  g.m = -((thr2f + sc.aT) * 1.000001f);
  // cb >= 1e-18 (make_screen), so the factor also covers 4 eta
  g.cx = __builtin_fmaf(__uint_as_float(s2max_bits), sc.ca, sc.cb) * 1.000001f;
  return g;
}
__device__ __forceinline__ unsigned long long left_open(const float *__restrict__ rec, const Hyp32 &h, f2 m,
                                                        f2 cx) {
  const f2 *q = reinterpret_cast<const f2 *>(rec);
  const f2 X = q[0], Y = q[1], Z = q[2];
  const f2 x = fma2(splat(h.PL[0]), X, fma2(splat(h.PL[1]), Y, fma2(splat(h.PL[2]), Z, splat(h.PL[3]))));
  const f2 y = fma2(splat(h.PL[4]), X, fma2(splat(h.PL[5]), Y, fma2(splat(h.PL[6]), Z, splat(h.PL[7]))));
  const f2 w = fma2(splat(h.PL[8]), X, fma2(splat(h.PL[9]), Y, fma2(splat(h.PL[10]), Z, splat(h.PL[11]))));
  const f2 dx = fma2(q[3], w, -x), dy = fma2(q[4], w, -y);
  const f2 w2 = w * w;
  const f2 D = fma2(m, w2, fma2(dx, dx, dy * dy));
  const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
  return active & ~(__builtin_amdgcn_ballot_w64(D.x > cx.x) & __builtin_amdgcn_ballot_w64(D.y > cx.y));
}

// 1 when the lane mask is not empty, in a scalar register (written as a select the compiler
// takes the flag through a vector register and keeps everything derived from it there).
__device__ __forceinline__ unsigned any_lane(unsigned long long lanes) {
  unsigned r;
  asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(lanes) : "scc");
  return r;
}

// fp64 decision for the staged point at f (lane k of its pair: f = record + k).
template <bool RECT>
__device__ __forceinline__ bool inlier_exact_staged(const float *__restrict__ f, const double (&PL)[12],
                                                    const double (&PR)[12], double thr, double thr2) {
  const double Xd = (double)f[0], Yd = (double)f[2], Zd = (double)f[4];
  bool in = inlier_one_camera(PL, Xd, Yd, Zd, (double)f[6], (double)f[8], thr, thr2);
  if (RECT) {
    // P_right is P_left with another [3] (as numbers: [7] and [11] carry 0 * (t_x + tx)
    // on top, which is not 0 only when t_x + tx is not finite; then P_right[3] is not
    // finite either, the right x is infinite or NaN and the point is no inlier)
    const double Pr[12] = {PL[0], PL[1], PL[2],  PR[3], PL[4], PL[5],
                           PL[6], PL[7], PL[8], PL[9], PL[10], PL[11]};
    in = in && fabs(PR[3]) <= 1.7976931348623157e308 &&
         inlier_one_camera(Pr, Xd, Yd, Zd, (double)f[10], (double)f[12], thr, thr2);
  } else {
    in = in && inlier_one_camera(PR, Xd, Yd, Zd, (double)f[10], (double)f[12], thr, thr2);
  }
  return in;
}

// Inliers (both cameras) among `cnt` points staged as pairs (stage_pair), cnt <= 128.
// `exact(f)`: the fp64 decision for the staged point at f (inlier_exact_staged with the
// hypothesis' fp64 projections, which the caller may work out only when asked).
//
// The loop over the pairs is straight-line code -- screen, count the sure inliers, and note in
// a scalar bit mask which pairs left any lane undecided (about one in a hundred) -- so that the
// compiler can overlap consecutive pairs (LDS reads of the next with the arithmetic of this
// one); a branch to the fp64 path inside it kept every iteration waiting for its own reads.
// The noted pairs are screened again afterwards and their undecided lanes take the fp64 path.
template <bool RECT, class Exact>
__device__ __forceinline__ int count_chunk_inliers_screened(const float *__restrict__ lds, int cnt,
                                                            const Hyp32 &h, Screen32 sc, double thr2,
                                                            Exact &&exact) {
  // uniform: keep it in scalar registers
  const f2 nthr2f = splat(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
      __builtin_bit_cast(int, -(float)thr2))));
  const f2 aT = splat(sc.aT), ca = splat(sc.ca), cb = splat(sc.cb);
  int c = 0;
  const int full = cnt >> 1;          // pairs of two real points
  const int fast = full & ~1;         // pairs the two-at-a-time loop covers
  unsigned long long noted = 0ull;    // wave-uniform, scalar registers
  for (int j = 0; j < fast; j += 2) {
    bool a_in0, a_in1, b_in0, b_in1;
    unsigned long long a_open0, a_open1, b_open0, b_open1;
    screen_pair<RECT>(lds + j * kPairStride, h, nthr2f, aT, ca, cb, a_in0, a_in1, a_open0, a_open1);
    screen_pair<RECT>(lds + (j + 1) * kPairStride, h, nthr2f, aT, ca, cb, b_in0, b_in1, b_open0, b_open1);
    c += (a_in0 ? 1 : 0) + (a_in1 ? 1 : 0) + (b_in0 ? 1 : 0) + (b_in1 ? 1 : 0);
    noted |= ((unsigned long long)any_lane(a_open0 | a_open1) | ((unsigned long long)any_lane(b_open0 | b_open1) << 1)) << j;
  }
  // not counted yet: the last pair of an odd number of pairs, and the last, single point of an
  // odd number of points (the second lane of its record is a copy)
  unsigned long long late = 0ull;
  if (full & 1) late |= 1ull << fast;
  if (cnt & 1) late |= 1ull << full;
  noted |= late;
  while (noted) {
    const int j = __builtin_ctzll(noted);
    noted &= noted - 1ull;
    bool in0, in1;
    unsigned long long open0, open1;
    screen_pair<RECT>(lds + j * kPairStride, h, nthr2f, aT, ca, cb, in0, in1, open0, open1);
    const bool two = j < full;
    const unsigned long long me = 1ull << __lane_id();
    if ((late >> j) & 1ull) c += (in0 ? 1 : 0) + ((two && in1) ? 1 : 0);
    if (open0 & me) c += exact(lds + j * kPairStride) ? 1 : 0;
    if (two && (open1 & me)) c += exact(lds + j * kPairStride + 1) ? 1 : 0;
  }
  return c;
}

// The same count for launches whose pairs are mostly unrelated (loop-closure candidates): there
// a hypothesis' inliers are little more than its own sample points, so most (hypothesis, staged
// pair) combinations can be dismissed by left_open -- were it not that a wavefront's 64
// hypotheses x 4 sample points land in more than half of a pair's staged records.  Hence:
// `mine` has bit j set when staged pair j of this chunk holds one of the lane's own sample
// points; those (lane, pair) combinations are NOT counted here -- the launch that solved the
// hypotheses has counted them and started the hypothesis' count there -- and
//   1. every staged pair goes through left_open, lanes with the pair in `mine` left out, which
//      yields a scalar bit mask of the pairs that keep a lane (a few in a hundred);
//   2. those pairs go through screen_pair as usual, again without the `mine` lanes.
// Every other (lane, staged pair) is either counted in 2 with the decisions of screen_pair /
// the fp64 path or dismissed for sure in 1: the same count.
template <bool RECT, class Exact>
__device__ __forceinline__ int count_chunk_inliers_sparse(const float *__restrict__ lds, int cnt,
                                                          const Hyp32 &h, Screen32 sc, double thr2,
                                                          unsigned s2max_bits, unsigned long long mine,
                                                          Exact &&exact) {
  static_assert(RECT, "left_open relies on the one threshold both cameras of a rectified pair share");
  const f2 nthr2f = splat(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
      __builtin_bit_cast(int, -(float)thr2))));
  const f2 aT = splat(sc.aT), ca = splat(sc.ca), cb = splat(sc.cb);
  const LeftGate gate = left_gate(sc, (float)thr2, s2max_bits);
  const f2 gm = splat(gate.m), gc = splat(gate.cx);
  const int full = cnt >> 1;          // pairs of two real points
  const int recs = (cnt + 1) >> 1;    // staged records (the last may hold one point and its copy)
  int c = 0;
  unsigned long long need = 0ull;     // wave-uniform, scalar registers
  // 32 staged pairs per half of `mine`: the lane's bit of pair j is a 32-bit and + compare
  const unsigned mine_half[2] = {(unsigned)mine, (unsigned)(mine >> 32)};
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const unsigned word = mine_half[half];
    const int end = min(recs, 32 * half + 32);
    int j = 32 * half;
    for (; j + 1 < end; j += 2) {
      const unsigned bit = 1u << (j & 31);
      const unsigned long long a = left_open(lds + j * kPairStride, h, gm, gc) &
                                   ~__builtin_amdgcn_ballot_w64((word & bit) != 0u);
      const unsigned long long b = left_open(lds + (j + 1) * kPairStride, h, gm, gc) &
                                   ~__builtin_amdgcn_ballot_w64((word & (bit << 1)) != 0u);
      need |= ((unsigned long long)any_lane(a) | ((unsigned long long)any_lane(b) << 1)) << j;
    }
    if (j < end) {
      const unsigned long long a = left_open(lds + j * kPairStride, h, gm, gc) &
                                   ~__builtin_amdgcn_ballot_w64((word & (1u << (j & 31))) != 0u);
      need |= (unsigned long long)any_lane(a) << j;
    }
  }
  // undecided points are only noted (per lane, one bit per staged pair and point) and taken
  // through the fp64 path at the end, from one place
  const unsigned long long me = 1ull << __lane_id();
  unsigned long long und0 = 0ull, und1 = 0ull;
  while (need) {
    const int k = __builtin_ctzll(need);
    need &= need - 1ull;
    bool in0, in1;
    unsigned long long open0, open1;
    screen_pair<RECT>(lds + k * kPairStride, h, nthr2f, aT, ca, cb, in0, in1, open0, open1);
    const bool two = k < full, other = !((mine >> k) & 1ull);
    c += (other && in0 ? 1 : 0) + ((other && two && in1) ? 1 : 0);
    und0 |= (other && (open0 & me)) ? 1ull << k : 0ull;
    und1 |= (other && two && (open1 & me)) ? 1ull << k : 0ull;
  }
#pragma clang loop unroll(disable)
  for (int pt = 0; pt < 2; ++pt) {
    unsigned long long u = pt ? und1 : und0;
    while (u) {
      const int k = __builtin_ctzll(u);
      u &= u - 1ull;
      c += exact(lds + k * kPairStride + pt) ? 1 : 0;
    }
  }
  return c;
}

// The same with the fp64 projections at hand.
template <bool RECT>
__device__ __forceinline__ int count_chunk_inliers_screened(const float *__restrict__ lds, int cnt,
                                                            const double (&PL)[12],
                                                            const double (&PR)[12], const Hyp32 &h,
                                                            Screen32 sc, double thr, double thr2) {
  return count_chunk_inliers_screened<RECT>(lds, cnt, h, sc, thr2, [&](const float *f) {
    return inlier_exact_staged<RECT>(f, PL, PR, thr, thr2);
  });
}

}  // namespace slamhip_dev
