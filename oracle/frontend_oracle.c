/*
 * CPU restatement in C/OpenMP of the per-frame front-end: stereo match, row / disparity
 * filter, DLT triangulation, temporal match, P3P RANSAC with the adaptive stopping rule,
 * least-squares refinement.  TEST INFRASTRUCTURE: the "port" CPU baseline that bench.py
 * times for the front-end (one frame, or one frame pair, per OpenMP thread), checked in
 * tests/ against the NumPy restatement (oracle/pipeline_oracle.py, oracle/slam_oracle.py)
 * and against the reference's own loop (tests/golden/ref_ransac_loop.npz).  Never linked
 * into the product path.
 *
 * What it follows in the reference (GuyShmueli/SLAM):
 *   backend/tracking/fill_database.py:140-240   the per-frame data flow
 *   frontend/vision/descriptor_matcher.py:23-34 match + sorted(key=distance)
 *   frontend/geometry/triangulation.py:85-109   cv2.triangulatePoints row order, filter
 *   frontend/geometry/pnp.py:86-160             projections, errors, supporters, RANSAC loop
 * The minimal solver and the refinement stand where cv2.solvePnP stands (pnp.py:36-43,
 * fill_database.py:204-207): PARITY UNPINNED against OpenCV itself (not installed); they
 * restate oracle/slam_oracle.py's p3p_solve / refine_pose.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ small linear algebra */
static void mat3_mul(const double *a, const double *b, double *c) {   /* 3x3 * 3x3 */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
static int mat3_inv(const double *m, double *o) {
  const double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
  if (det == 0.0) return 0;
  const double s = 1.0 / det;
  o[0] = c0 * s; o[1] = (m[2] * m[7] - m[1] * m[8]) * s; o[2] = (m[1] * m[5] - m[2] * m[4]) * s;
  o[3] = c1 * s; o[4] = (m[0] * m[8] - m[2] * m[6]) * s; o[5] = (m[2] * m[3] - m[0] * m[5]) * s;
  o[6] = c2 * s; o[7] = (m[1] * m[6] - m[0] * m[7]) * s; o[8] = (m[0] * m[4] - m[1] * m[3]) * s;
  return 1;
}
/* P (3x4) = K (3x3) @ Rt (3x4) */
static void k_times_rt(const double *K, const double *Rt, double *P) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) P[4 * i + j] = K[3 * i] * Rt[j] + K[3 * i + 1] * Rt[4 + j] + K[3 * i + 2] * Rt[8 + j];
}
/* (A_h @ B_h)[:3] for two 3x4 rigid blocks (utils/graph_utils.py homogenous / non_homogenous) */
static void rt_compose(const double *A, const double *B, double *C) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j)
      C[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j] + (j == 3 ? A[4 * i + 3] : 0.0);
  }
}
static void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
static double norm3(const double *a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
/* columns [a, n x a, n] of the frame spanned by three points (row-major 3x3 out) */
static void triangle_frame(const double *p1, const double *p2, const double *p3, double *F) {
  double a[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, b[3] = {p3[0] - p1[0], p3[1] - p1[1], p3[2] - p1[2]};
  double n[3], m[3];
  double s = norm3(a);
  a[0] /= s; a[1] /= s; a[2] /= s;
  cross3(a, b, n);
  s = norm3(n);
  n[0] /= s; n[1] /= s; n[2] /= s;
  cross3(n, a, m);
  for (int i = 0; i < 3; ++i) { F[3 * i] = a[i]; F[3 * i + 1] = m[i]; F[3 * i + 2] = n[i]; }
}
/* Gaussian elimination with partial pivoting, n <= 6; returns 0 when singular */
static int solve_small(double *A, double *b, int n) {
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r) if (fabs(A[r * n + c]) > fabs(A[piv * n + c])) piv = r;
    if (A[piv * n + c] == 0.0) return 0;
    if (piv != c) {
      for (int k = 0; k < n; ++k) { double t = A[c * n + k]; A[c * n + k] = A[piv * n + k]; A[piv * n + k] = t; }
      double t = b[c]; b[c] = b[piv]; b[piv] = t;
    }
    for (int r = c + 1; r < n; ++r) {
      const double f = A[r * n + c] / A[c * n + c];
      for (int k = c; k < n; ++k) A[r * n + k] -= f * A[c * n + k];
      b[r] -= f * b[c];
    }
  }
  for (int r = n - 1; r >= 0; --r) {
    double s = b[r];
    for (int k = r + 1; k < n; ++k) s -= A[r * n + k] * b[k];
    b[r] = s / A[r * n + r];
  }
  return 1;
}

/* ------------------------------------------------------------------ quartic roots */
/* All complex roots of c[0] + c[1] x + ... + c[deg] x^deg (deg <= 4, c[deg] != 0) by the
 * Weierstrass (Durand-Kerner) iteration; re/im out. */
static void poly_roots(const double *c, int deg, double *re, double *im) {
  double a[5];
  for (int i = 0; i <= deg; ++i) a[i] = c[i] / c[deg];
  double bound = 0.0;
  for (int i = 0; i < deg; ++i) if (fabs(a[i]) > bound) bound = fabs(a[i]);
  bound = 1.0 + bound;
  /* start on a circle well inside the root bound, at angles that are no symmetry of the roots */
  double rad = 0.0;
  for (int i = 0; i < deg; ++i) {
    const double t = pow(fabs(a[i]), 1.0 / (deg - i));
    if (t > rad) rad = t;
  }
  if (rad == 0.0) rad = 1.0;
  if (rad > bound) rad = bound;
  for (int k = 0; k < deg; ++k) {
    const double ang = 0.4 + 2.0 * M_PI * k / deg;
    re[k] = rad * cos(ang); im[k] = rad * sin(ang);
  }
  for (int it = 0; it < 500; ++it) {
    double moved = 0.0, size = 0.0;
    for (int k = 0; k < deg; ++k) {
      /* p(z_k) by Horner */
      double pr = 1.0, pi = 0.0;
      for (int i = deg - 1; i >= 0; --i) {
        const double tr = pr * re[k] - pi * im[k] + a[i], ti = pr * im[k] + pi * re[k];
        pr = tr; pi = ti;
      }
      double dr = 1.0, di = 0.0;
      for (int j = 0; j < deg; ++j) {
        if (j == k) continue;
        const double xr = re[k] - re[j], xi = im[k] - im[j];
        const double tr = dr * xr - di * xi, ti = dr * xi + di * xr;
        dr = tr; di = ti;
      }
      const double den = dr * dr + di * di;
      if (den == 0.0) continue;
      const double qr = (pr * dr + pi * di) / den, qi = (pi * dr - pr * di) / den;
      re[k] -= qr; im[k] -= qi;
      moved += fabs(qr) + fabs(qi);
      size += fabs(re[k]) + fabs(im[k]);
    }
    if (moved <= 1e-16 * (size + 1e-300)) break;
  }
}

/* ------------------------------------------------------------------ minimal solver */
/* Pose [R|t] (row-major 3x4) from three correspondences, selected among the (up to four)
 * solutions by the fourth; restates slam_oracle.p3p_solutions / p3p_solve.  obj4: 4x3,
 * img4: 4x2 pixels.  Returns 1 and fills Rt, or 0. */
int oracle_p3p_solve(const double *obj4, const double *img4, const double *K, double *Rt_out) {
  double Kinv[9];
  if (!mat3_inv(K, Kinv)) return 0;
  double f[3][3];
  for (int i = 0; i < 3; ++i) {
    const double u = img4[2 * i], v = img4[2 * i + 1];
    double b[3];
    for (int r = 0; r < 3; ++r) b[r] = Kinv[3 * r] * u + Kinv[3 * r + 1] * v + Kinv[3 * r + 2];
    const double s = norm3(b);
    f[i][0] = b[0] / s; f[i][1] = b[1] / s; f[i][2] = b[2] / s;
  }
  const double *X0 = obj4, *X1 = obj4 + 3, *X2 = obj4 + 6;
  double a12 = 0, a13 = 0, a23 = 0, c12 = 0, c13 = 0, c23 = 0;
  for (int k = 0; k < 3; ++k) {
    a12 += (X0[k] - X1[k]) * (X0[k] - X1[k]);
    a13 += (X0[k] - X2[k]) * (X0[k] - X2[k]);
    a23 += (X1[k] - X2[k]) * (X1[k] - X2[k]);
    c12 += f[0][k] * f[1][k]; c13 += f[0][k] * f[2][k]; c23 += f[1][k] * f[2][k];
  }
  /* polynomials in v, ascending coefficients */
  const double A1 = a13, B1 = -2 * a13 * c12, C1[3] = {a13 - a12, 2 * a12 * c13, -a12};
  const double A2 = a23 - a12, B2[2] = {-2 * a23 * c12, 2 * a12 * c23}, C2[3] = {a23, 0.0, -a12};
  double p[3], q[2], r[4], quart[5];
  for (int i = 0; i < 3; ++i) p[i] = A1 * C2[i] - A2 * C1[i];
  q[0] = A1 * B2[0] - A2 * B1; q[1] = A1 * B2[1];
  /* r = B1 * C2 - B2 * C1 */
  for (int i = 0; i < 4; ++i) r[i] = 0.0;
  for (int i = 0; i < 3; ++i) r[i] += B1 * C2[i];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j) r[i + j] -= B2[i] * C1[j];
  for (int i = 0; i < 5; ++i) quart[i] = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) quart[i + j] += p[i] * p[j];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j) quart[i + j] -= q[i] * r[j];
  int deg = 4;
  while (deg > 0 && quart[deg] == 0.0) --deg;
  if (deg == 0) return 0;
  double re[4], im[4];
  poly_roots(quart, deg, re, im);
  double Fw[9];
  triangle_frame(X0, X1, X2, Fw);
  /* the fourth correspondence in normalised image coordinates */
  double n4[3];
  for (int rr = 0; rr < 3; ++rr) n4[rr] = Kinv[3 * rr] * img4[6] + Kinv[3 * rr + 1] * img4[7] + Kinv[3 * rr + 2];
  const double n4x = n4[0] / n4[2], n4y = n4[1] / n4[2];
  int found = 0;
  double best_err = INFINITY;
  for (int k = 0; k < deg; ++k) {
    if (fabs(im[k]) > 1e-9 * fmax(1.0, fabs(re[k])) || re[k] <= 0.0) continue;
    double v = re[k];
    /* Newton polish on the real polynomial */
    for (int it = 0; it < 3; ++it) {
      double pv = 0.0, dv = 0.0;
      for (int i = deg; i >= 0; --i) { dv = dv * v + pv; pv = pv * v + quart[i]; }
      if (dv == 0.0) break;
      const double step = pv / dv;
      if (!(fabs(step) < 0.1 * fabs(v))) break;     /* near a double root: keep the iterate */
      v -= step;
    }
    const double qv = q[0] + q[1] * v;
    if (qv == 0.0) continue;
    const double u = -(p[0] + p[1] * v + p[2] * v * v) / qv;
    if (u <= 0.0) continue;
    const double d1 = sqrt(a12 / (1 + u * u - 2 * u * c12));
    double d[3] = {d1, u * d1, v * d1};
    int ok = 1;
    for (int it = 0; it < 3 && ok; ++it) {            /* Gauss-Newton polish of the depths */
      double res[3] = {d[0] * d[0] + d[1] * d[1] - 2 * d[0] * d[1] * c12 - a12,
                       d[0] * d[0] + d[2] * d[2] - 2 * d[0] * d[2] * c13 - a13,
                       d[1] * d[1] + d[2] * d[2] - 2 * d[1] * d[2] * c23 - a23};
      double J[9] = {2 * (d[0] - d[1] * c12), 2 * (d[1] - d[0] * c12), 0,
                     2 * (d[0] - d[2] * c13), 0, 2 * (d[2] - d[0] * c13),
                     0, 2 * (d[1] - d[2] * c23), 2 * (d[2] - d[1] * c23)};
      if (!solve_small(J, res, 3)) { ok = 0; break; }
      d[0] -= res[0]; d[1] -= res[1]; d[2] -= res[2];
    }
    if (!ok || !(d[0] > 0.0 && d[1] > 0.0 && d[2] > 0.0)) continue;
    double Y[3][3];
    for (int i = 0; i < 3; ++i)
      for (int c = 0; c < 3; ++c) Y[i][c] = d[i] * f[i][c];
    double Fc[9], FwT[9], R[9];
    triangle_frame(Y[0], Y[1], Y[2], Fc);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) FwT[3 * i + j] = Fw[3 * j + i];
    mat3_mul(Fc, FwT, R);
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = Y[0][i] - (R[3 * i] * X0[0] + R[3 * i + 1] * X0[1] + R[3 * i + 2] * X0[2]);
    const double *X3 = obj4 + 9;
    double c[3];
    for (int i = 0; i < 3; ++i) c[i] = R[3 * i] * X3[0] + R[3 * i + 1] * X3[1] + R[3 * i + 2] * X3[2] + t[i];
    const double ex = c[0] / c[2] - n4x, ey = c[1] / c[2] - n4y, err = ex * ex + ey * ey;
    if (err < best_err) {
      best_err = err;
      found = 1;
      for (int i = 0; i < 3; ++i) { Rt_out[4 * i] = R[3 * i]; Rt_out[4 * i + 1] = R[3 * i + 1]; Rt_out[4 * i + 2] = R[3 * i + 2]; Rt_out[4 * i + 3] = t[i]; }
    }
  }
  return found;
}

/* ------------------------------------------------------------------ consensus */
/* supporters of one hypothesis (pnp.py:133-146): both reprojection errors <= thresh.
 * X: n x 3, mL / mR: n x 2 (doubles).  Returns the count; mask may be NULL. */
static int hypothesis_mask(const double *K, const double *M_right, const double *Rt, const double *X,
                           const double *mL, const double *mR, int n, double thresh, uint8_t *mask) {
  double PL[12], RtR[12], PR[12];
  k_times_rt(K, Rt, PL);
  rt_compose(M_right, Rt, RtR);
  k_times_rt(K, RtR, PR);
  int count = 0;
  for (int i = 0; i < n; ++i) {
    const double x = X[3 * i], y = X[3 * i + 1], z = X[3 * i + 2];
    const double lw = PL[8] * x + PL[9] * y + PL[10] * z + PL[11];
    const double lu = (PL[0] * x + PL[1] * y + PL[2] * z + PL[3]) / lw - mL[2 * i];
    const double lv = (PL[4] * x + PL[5] * y + PL[6] * z + PL[7]) / lw - mL[2 * i + 1];
    const double rw = PR[8] * x + PR[9] * y + PR[10] * z + PR[11];
    const double ru = (PR[0] * x + PR[1] * y + PR[2] * z + PR[3]) / rw - mR[2 * i];
    const double rv = (PR[4] * x + PR[5] * y + PR[6] * z + PR[7]) / rw - mR[2 * i + 1];
    const int in = (sqrt(lu * lu + lv * lv) <= thresh) && (sqrt(ru * ru + rv * rv) <= thresh);
    if (mask) mask[i] = (uint8_t)in;
    count += in;
  }
  return count;
}

/* idx4 of iteration `it`: four distinct indices below n from a counter-based generator */
static uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static void draw4(uint64_t seed, int it, int n, int32_t *idx4) {
  uint64_t s = splitmix64(seed ^ ((uint64_t)it * 0xD1B54A32D192ED03ull));
  int got = 0;
  while (got < 4) {
    s = splitmix64(s);
    const int32_t c = (int32_t)(s % (uint64_t)n);
    int dup = 0;
    for (int k = 0; k < got; ++k) dup |= idx4[k] == c;
    if (!dup) idx4[got++] = c;
  }
}

/*
 * PnP.ransac_pnp_pair1 (pnp.py:102-160), one iteration at a time.
 *   samples  n_iter_max x 4 int32, or NULL: drawn from `seed`
 *   hyps     n_hyps x 12 doubles + hyp_valid n_hyps bytes, or NULL: the minimal solver runs
 *            (a prepared hypothesis list stands where solve_pnp stands, as in the fixture)
 *   early_inliers  < 0: off (pnp.py's early exit)
 * Outputs: mask (n bytes), Rt (12), info[4] = {best_index, best_count, iterations,
 * n_iter_required}.  Returns 0, or -2 where the reference raises (math.log(0): a
 * hypothesis that every point supports).
 */
int oracle_ransac_loop(const double *K, const double *M_right, const double *X, const double *mL,
                       const double *mR, int n, double thresh, double p_success, int n_iter_max,
                       int early_inliers, const int32_t *samples, uint64_t seed, const double *hyps,
                       const uint8_t *hyp_valid, int n_hyps, uint8_t *mask, double *Rt, int32_t *info) {
  int best_idx = -1, best_count = 0, iter_done = 0;
  long long n_req = n_iter_max;
  memset(mask, 0, (size_t)(n > 0 ? n : 0));
  info[0] = -1; info[1] = 0; info[2] = 0; info[3] = n_iter_max;
  if (n < 4) return 0;
  uint8_t *cur = (uint8_t *)malloc((size_t)n);
  while (iter_done < n_req && iter_done < n_iter_max && (hyps == NULL || iter_done < n_hyps)) {
    const int it = iter_done++;
    double H[12];
    int have;
    if (hyps != NULL) {
      have = hyp_valid == NULL || hyp_valid[it];
      memcpy(H, hyps + 12 * (size_t)it, sizeof H);
    } else {
      int32_t idx4[4];
      if (samples != NULL) memcpy(idx4, samples + 4 * (size_t)it, sizeof idx4);
      else draw4(seed, it, n, idx4);
      double obj4[12], img4[8];
      for (int k = 0; k < 4; ++k) {
        memcpy(obj4 + 3 * k, X + 3 * (size_t)idx4[k], 3 * sizeof(double));
        memcpy(img4 + 2 * k, mL + 2 * (size_t)idx4[k], 2 * sizeof(double));
      }
      have = oracle_p3p_solve(obj4, img4, K, H);
    }
    if (!have) continue;
    const int inl = hypothesis_mask(K, M_right, H, X, mL, mR, n, thresh, cur);
    if (inl > best_count) {
      best_count = inl; best_idx = it;
      memcpy(mask, cur, (size_t)n);
      memcpy(Rt, H, sizeof H);
      const double w = (double)best_count / (double)n;
      if (w > 0) {
        const double prob = w * w * w * w;
        if (1.0 - prob <= 0.0) { free(cur); return -2; }      /* math.log(0) */
        n_req = (long long)ceil(log(1.0 - p_success) / log(1.0 - prob));
      }
    }
    if (early_inliers >= 0 && best_count >= early_inliers) break;
  }
  free(cur);
  info[0] = best_idx; info[1] = best_count; info[2] = iter_done;
  info[3] = (int32_t)(n_req > INT_MAX ? INT_MAX : n_req);
  return 0;
}

/* ------------------------------------------------------------------ refinement */
static void rodrigues(const double *w, double *R) {
  const double th = norm3(w);
  double k[3] = {0, 0, 0}, s = 0.0, c = 1.0;
  if (th > 0.0) { k[0] = w[0] / th; k[1] = w[1] / th; k[2] = w[2] / th; s = sin(th); c = cos(th); }
  const double v = 1.0 - c;
  R[0] = c + k[0] * k[0] * v;        R[1] = k[0] * k[1] * v - k[2] * s; R[2] = k[0] * k[2] * v + k[1] * s;
  R[3] = k[1] * k[0] * v + k[2] * s; R[4] = c + k[1] * k[1] * v;        R[5] = k[1] * k[2] * v - k[0] * s;
  R[6] = k[2] * k[0] * v - k[1] * s; R[7] = k[2] * k[1] * v + k[0] * s; R[8] = c + k[2] * k[2] * v;
}
/* Gauss-Newton on the left reprojection error over the masked points, from Rt (in place):
 * the stationary point slam_oracle.refine_pose converges to.  Returns the iterations used. */
int oracle_refine_pose(const double *K, const double *X, const double *mL, const uint8_t *mask, int n,
                       double *Rt) {
  double R[9] = {Rt[0], Rt[1], Rt[2], Rt[4], Rt[5], Rt[6], Rt[8], Rt[9], Rt[10]}, t[3] = {Rt[3], Rt[7], Rt[11]};
  int it = 0;
  for (; it < 30; ++it) {
    double JtJ[36], Jtr[6];
    memset(JtJ, 0, sizeof JtJ);
    memset(Jtr, 0, sizeof Jtr);
    for (int i = 0; i < n; ++i) {
      if (mask && !mask[i]) continue;
      const double *x = X + 3 * (size_t)i;
      double Xc[3], p[3];
      for (int r = 0; r < 3; ++r) Xc[r] = R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2] + t[r];
      for (int r = 0; r < 3; ++r) p[r] = K[3 * r] * Xc[0] + K[3 * r + 1] * Xc[1] + K[3 * r + 2] * Xc[2];
      const double res[2] = {p[0] / p[2] - mL[2 * i], p[1] / p[2] - mL[2 * i + 1]};
      /* d(pixel)/dp (2x3), times K -> d(pixel)/dXc (2x3) */
      const double dp[2][3] = {{1 / p[2], 0, -p[0] / (p[2] * p[2])}, {0, 1 / p[2], -p[1] / (p[2] * p[2])}};
      double dX[2][3];
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 3; ++b) dX[a][b] = dp[a][0] * K[b] + dp[a][1] * K[3 + b] + dp[a][2] * K[6 + b];
      /* d(Xc)/d(w, v) = [-[Xc]x | I] */
      const double sk[3][3] = {{0, -Xc[2], Xc[1]}, {Xc[2], 0, -Xc[0]}, {-Xc[1], Xc[0], 0}};
      double J[2][6];
      for (int a = 0; a < 2; ++a) {
        for (int b = 0; b < 3; ++b) {
          J[a][b] = -(dX[a][0] * sk[0][b] + dX[a][1] * sk[1][b] + dX[a][2] * sk[2][b]);
          J[a][3 + b] = dX[a][b];
        }
      }
      for (int a = 0; a < 2; ++a)
        for (int r = 0; r < 6; ++r) {
          Jtr[r] += J[a][r] * res[a];
          for (int c = 0; c < 6; ++c) JtJ[6 * r + c] += J[a][r] * J[a][c];
        }
    }
    double delta[6];
    for (int r = 0; r < 6; ++r) delta[r] = -Jtr[r];
    if (!solve_small(JtJ, delta, 6)) break;
    double E[9], Rn[9], tn[3];
    rodrigues(delta, E);
    mat3_mul(E, R, Rn);
    for (int r = 0; r < 3; ++r) tn[r] = E[3 * r] * t[0] + E[3 * r + 1] * t[1] + E[3 * r + 2] * t[2] + delta[3 + r];
    memcpy(R, Rn, sizeof R);
    memcpy(t, tn, sizeof t);
    double m = 0.0;
    for (int r = 0; r < 6; ++r) if (fabs(delta[r]) > m) m = fabs(delta[r]);
    if (m < 1e-14) { ++it; break; }
  }
  for (int r = 0; r < 3; ++r) { Rt[4 * r] = R[3 * r]; Rt[4 * r + 1] = R[3 * r + 1]; Rt[4 * r + 2] = R[3 * r + 2]; Rt[4 * r + 3] = t[r]; }
  return it;
}

/* ------------------------------------------------------------------ matching, triangulation */
static inline int row_distance(const uint8_t *a, const uint8_t *b, int nbytes) {
  int d = 0, i = 0;
  for (; i + 8 <= nbytes; i += 8) {
    uint64_t x, y;
    memcpy(&x, a + i, 8);
    memcpy(&y, b + i, 8);
    d += __builtin_popcountll(x ^ y);
  }
  for (; i < nbytes; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
  return d;
}
/* BFMatcher.match (k = 1, lowest train index on ties) + sorted(key=distance) (stable):
 * fills q_sorted / t_sorted / d_sorted, returns the number of matches (nq, or 0 without train rows) */
static int match_sorted(const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes, int32_t *qs,
                        int32_t *ts, int32_t *ds, int32_t *tmp_t, int32_t *tmp_d) {
  if (nq == 0 || nt == 0) return 0;
  const int maxd = 8 * nbytes;
  int *hist = (int *)calloc((size_t)maxd + 2, sizeof(int));
  for (int i = 0; i < nq; ++i) {
    int best = INT_MAX, bi = -1;
    const uint8_t *qi = q + (size_t)i * nbytes;
    for (int j = 0; j < nt; ++j) {
      const int d = row_distance(qi, t + (size_t)j * nbytes, nbytes);
      if (d < best) { best = d; bi = j; }
    }
    tmp_t[i] = bi; tmp_d[i] = best;
    hist[best + 1]++;
  }
  for (int d = 0; d <= maxd; ++d) hist[d + 1] += hist[d];
  for (int i = 0; i < nq; ++i) {                       /* counting sort: stable */
    const int pos = hist[tmp_d[i]]++;
    qs[pos] = i; ts[pos] = tmp_t[i]; ds[pos] = tmp_d[i];
  }
  free(hist);
  return nq;
}
/* right singular vector of the smallest singular value of a 4x4 matrix (one-sided Jacobi) */
static void smallest_right_singular_vector(const double *A_in, double *x) {
  double A[16], V[16];
  memcpy(A, A_in, sizeof A);
  for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int r = 0; r < 4; ++r) { al += A[4 * r + p] * A[4 * r + p]; be += A[4 * r + q] * A[4 * r + q]; ga += A[4 * r + p] * A[4 * r + q]; }
        if (ga == 0.0) continue;
        const double lim = sqrt(al * be);
        if (fabs(ga) <= 1e-16 * lim) continue;
        off += fabs(ga) / (lim > 0 ? lim : 1.0);
        const double zeta = (be - al) / (2.0 * ga);
        const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
        for (int r = 0; r < 4; ++r) {
          const double ap = A[4 * r + p], aq = A[4 * r + q];
          A[4 * r + p] = c * ap - s * aq; A[4 * r + q] = s * ap + c * aq;
          const double vp = V[4 * r + p], vq = V[4 * r + q];
          V[4 * r + p] = c * vp - s * vq; V[4 * r + q] = s * vp + c * vq;
        }
      }
    if (off == 0.0) break;
  }
  int best = 0;
  double bn = INFINITY;
  for (int c = 0; c < 4; ++c) {
    double nn = 0;
    for (int r = 0; r < 4; ++r) nn += A[4 * r + c] * A[4 * r + c];
    if (nn < bn) { bn = nn; best = c; }
  }
  for (int r = 0; r < 4; ++r) x[r] = V[4 * r + best];
}
/* cv2.triangulatePoints row order + Xh[:3] / Xh[3] in float32 (triangulation.py:85-86) */
static void triangulate_f32(const double *PL, const double *PR, const double *pl, const double *pr, float *xyz) {
  double A[16], x[4];
  for (int c = 0; c < 4; ++c) {
    A[c] = pl[0] * PL[8 + c] - PL[c];
    A[4 + c] = pl[1] * PL[8 + c] - PL[4 + c];
    A[8 + c] = pr[0] * PR[8 + c] - PR[c];
    A[12 + c] = pr[1] * PR[8 + c] - PR[4 + c];
  }
  smallest_right_singular_vector(A, x);
  const float w = (float)x[3];
  xyz[0] = (float)x[0] / w; xyz[1] = (float)x[1] / w; xyz[2] = (float)x[2] / w;
}

/*
 * The whole front-end over F frames (fill_database.py:140-240 with the features given).
 * Frame f owns rows [offL[f], offL[f+1]) of kpL / desL and [offR[f], offR[f+1]) of kpR / desR.
 * Stereo stage: one frame per thread; temporal stage: one frame pair per thread.
 * Outputs, frame f at offset offL[f]: st_left / st_right / st_dist (st_count[f] sorted, filtered
 * matches), st_xyz (float32, 3 per match); pair p at offset offL[p]: tm_q / tm_t / tm_d
 * (tm_count[p] matches of pair (p, p+1), indices into the stereo inliers of each frame) and
 * mask; info[p * 4 ..] = {best_index, best_count, iterations, n_iter_required};
 * Rt_ransac / Rt: 12 doubles per pair, has_pose[p].
 * samples: (F-1) x n_iter_max x 4 int32 or NULL (drawn from seed + pair).
 */
int oracle_frontend_run(const double *K, const double *M_right, const double *P_left, const double *P_right,
                        const double *kpL, const double *kpR, const uint8_t *desL, const uint8_t *desR,
                        const int32_t *offL, const int32_t *offR, int F, int nbytes, double stereo_thresh,
                        double disparity_min, double pnp_thresh, double p_success, int n_iter_max,
                        const int32_t *samples, uint64_t seed, int refine, int threads,
                        int32_t *st_count, int32_t *st_left, int32_t *st_right, int32_t *st_dist, float *st_xyz,
                        int32_t *tm_count, int32_t *tm_q, int32_t *tm_t, int32_t *tm_d, uint8_t *mask,
                        int32_t *info, double *Rt_ransac, double *Rt, uint8_t *has_pose) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  const int nL_total = offL[F];
  uint8_t *st_des = (uint8_t *)malloc((size_t)(nL_total > 0 ? nL_total : 1) * nbytes);   /* inlier_des_left */
  double *st_pl = (double *)malloc((size_t)(nL_total > 0 ? nL_total : 1) * 2 * sizeof(double));
  double *st_pr = (double *)malloc((size_t)(nL_total > 0 ? nL_total : 1) * 2 * sizeof(double));
#pragma omp parallel for schedule(dynamic, 1)
  for (int f = 0; f < F; ++f) {
    const int l0 = offL[f], nl = offL[f + 1] - l0, r0 = offR[f], nr = offR[f + 1] - r0;
    int32_t *qs = (int32_t *)malloc((size_t)(nl + 1) * 5 * sizeof(int32_t));
    int32_t *ts = qs + nl + 1, *ds = ts + nl + 1, *tt = ds + nl + 1, *td = tt + nl + 1;
    const int m = match_sorted(desL + (size_t)l0 * nbytes, nl, desR + (size_t)r0 * nbytes, nr, nbytes, qs, ts, ds, tt, td);
    int cnt = 0;
    for (int k = 0; k < m; ++k) {
      const double *pl = kpL + 2 * (size_t)(l0 + qs[k]), *pr = kpR + 2 * (size_t)(r0 + ts[k]);
      if (!(fabs(pr[1] - pl[1]) <= stereo_thresh && (pl[0] - pr[0]) >= disparity_min)) continue;
      const size_t o = (size_t)l0 + cnt;
      st_left[o] = qs[k]; st_right[o] = ts[k]; st_dist[o] = ds[k];
      triangulate_f32(P_left, P_right, pl, pr, st_xyz + 3 * o);
      memcpy(st_des + o * nbytes, desL + (size_t)(l0 + qs[k]) * nbytes, (size_t)nbytes);
      st_pl[2 * o] = pl[0]; st_pl[2 * o + 1] = pl[1];
      st_pr[2 * o] = pr[0]; st_pr[2 * o + 1] = pr[1];
      ++cnt;
    }
    st_count[f] = cnt;
    free(qs);
  }
  int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int p = 0; p < F - 1; ++p) {
    const int a0 = offL[p], na = st_count[p], b0 = offL[p + 1], nb = st_count[p + 1];
    int32_t *tt = (int32_t *)malloc((size_t)(na + 1) * 2 * sizeof(int32_t)), *td = tt + na + 1;
    const int m = match_sorted(st_des + (size_t)a0 * nbytes, na, st_des + (size_t)b0 * nbytes, nb, nbytes,
                               tm_q + a0, tm_t + a0, tm_d + a0, tt, td);
    free(tt);
    tm_count[p] = m;
    double *X = (double *)malloc((size_t)(m + 1) * 7 * sizeof(double)), *mL = X + 3 * (size_t)(m + 1), *mR = mL + 2 * (size_t)(m + 1);
    for (int k = 0; k < m; ++k) {
      const size_t qa = (size_t)a0 + tm_q[a0 + k], tb = (size_t)b0 + tm_t[a0 + k];
      X[3 * k] = st_xyz[3 * qa]; X[3 * k + 1] = st_xyz[3 * qa + 1]; X[3 * k + 2] = st_xyz[3 * qa + 2];
      mL[2 * k] = st_pl[2 * tb]; mL[2 * k + 1] = st_pl[2 * tb + 1];
      mR[2 * k] = st_pr[2 * tb]; mR[2 * k + 1] = st_pr[2 * tb + 1];
    }
    double *H = Rt_ransac + 12 * (size_t)p;
    memset(H, 0, 12 * sizeof(double));
    const int r = oracle_ransac_loop(K, M_right, X, mL, mR, m, pnp_thresh, p_success, n_iter_max, -1,
                                     samples ? samples + (size_t)p * n_iter_max * 4 : NULL, seed + (uint64_t)p,
                                     NULL, NULL, 0, mask + a0, H, info + 4 * (size_t)p);
    if (r != 0) {
#pragma omp atomic write
      rc = r;
    }
    has_pose[p] = (uint8_t)(info[4 * (size_t)p] >= 0);
    memcpy(Rt + 12 * (size_t)p, H, 12 * sizeof(double));
    if (refine && has_pose[p] && info[4 * (size_t)p + 1] >= 3)
      oracle_refine_pose(K, X, mL, mask + a0, m, Rt + 12 * (size_t)p);
    free(X);
  }
  free(st_des); free(st_pl); free(st_pr);
  return rc;
}
