// Minimal pose solver (P3P + fourth-point disambiguation) for gfx950, one
// hypothesis per lane, fp64.
//
// Role in the reference: PnP.solve_pnp(obj4, img4, cv2.SOLVEPNP_AP3P) inside the
// RANSAC loop (frontend/geometry/pnp.py:36-43,126-130): the first three
// correspondences give up to four poses, the fourth picks the one that
// reprojects it best; failure is signalled to the caller (valid = 0), which then
// skips the iteration (pnp.py:129-130).
//
// Method (the same polynomial system every exact P3P solver solves):
//   unit bearings f_i = normalise(K^-1 [u_i, v_i, 1]); depths d_i along them;
//   d_i^2 + d_j^2 - 2 d_i d_j (f_i.f_j) = |X_i - X_j|^2 for the three pairs.
//   With d2 = u d1, d3 = v d1 two quadratics in (u, v) remain; their resultant
//   in u is a quartic in v whose coefficients are formed numerically by
//   polynomial products.  Its real roots come from the closed-form
//   factorisation through the resolvent cubic, polished by guarded Newton
//   steps on the quartic itself (quartic.h; straight-line code, no divergence);
//   u follows from the common-root relation; depths are polished by two
//   Gauss-Newton steps on the three distance equations; the pose is the rigid
//   alignment of the two triangles.
#include "common.h"

#include "p3p_device.h"

namespace {

using namespace slamhip_dev;

template <typename PT>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(SLAMHIP_P3P_WAVES, SLAMHIP_P3P_WAVES))) void p3p_kernel(const PT *__restrict__ pts,
                                                 const float *__restrict__ pix, int n_points,
                                                 const int32_t *__restrict__ idx4, int h, Kinv ki,
                                                 double *__restrict__ Rt_out,
                                                 uint8_t *__restrict__ valid_out) {
  const int hyp = blockIdx.x * blockDim.x + threadIdx.x;
  if (hyp >= h) return;
  double X[4][3], f[4][3], nrm4[2];
  long long rows[4];
  bool in_range = true;
  for (int k = 0; k < 4; ++k) {
    rows[k] = idx4 ? (long long)idx4[4 * (size_t)hyp + k] : 4ll * hyp + k;
    in_range = in_range && rows[k] >= 0 && rows[k] < (long long)n_points;
  }
  double *out = Rt_out + 12 * (size_t)hyp;
  if (!in_range) {            // a sample naming a row that does not exist: no pose, nothing read
    valid_out[hyp] = 0;
    for (int i = 0; i < 12; ++i) out[i] = 0.0;
    return;
  }
  for (int k = 0; k < 4; ++k) {
    const size_t r = (size_t)rows[k];
    for (int c = 0; c < 3; ++c) X[k][c] = (double)pts[3 * r + c];
    double nk[2];
    pixel_to_bearing(ki, (double)pix[2 * r + 0], (double)pix[2 * r + 1], f[k], nk);
    if (k == 3) { nrm4[0] = nk[0]; nrm4[1] = nk[1]; }
  }
  double Rt[12];
  const bool found = p3p_lane(X, f, nrm4, Rt);
  valid_out[hyp] = found ? 1 : 0;
  for (int i = 0; i < 12; ++i) out[i] = found ? Rt[i] : 0.0;
}

bool invert3(const double *K, double *o) {
  const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  if (det == 0.0) return false;
  const double s = 1.0 / det;
  o[0] = (e * i - f * h) * s; o[1] = (c * h - b * i) * s; o[2] = (b * f - c * e) * s;
  o[3] = (f * g - d * i) * s; o[4] = (a * i - c * g) * s; o[5] = (c * d - a * f) * s;
  o[6] = (d * h - e * g) * s; o[7] = (b * g - a * h) * s; o[8] = (a * e - b * d) * s;
  return true;
}

}  // namespace

extern "C" int slamhip_p3p_solve(const void *d_points, int points_fp64, const float *d_pixels,
                                 int n_points, const int32_t *d_idx4, int h, const double *K,
                                 double *d_Rt, uint8_t *d_valid, void *stream) {
  SLAMHIP_BEGIN();
  if (h < 0 || n_points < 0) return SLAMHIP_EINVAL;
  if (h == 0) return SLAMHIP_OK;
  if (!d_points || !d_pixels || !K || !d_Rt || !d_valid) return SLAMHIP_EINVAL;
  Kinv ki;
  if (!invert3(K, ki.m)) return SLAMHIP_EINVAL;
  const dim3 grid(slamhip_ceil_div(h, 64)), block(64);
  if (points_fp64)
    hipLaunchKernelGGL(p3p_kernel<double>, grid, block, 0, (hipStream_t)stream,
                       (const double *)d_points, d_pixels, n_points, d_idx4, h, ki, d_Rt, d_valid);
  else
    hipLaunchKernelGGL(p3p_kernel<float>, grid, block, 0, (hipStream_t)stream,
                       (const float *)d_points, d_pixels, n_points, d_idx4, h, ki, d_Rt, d_valid);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}
