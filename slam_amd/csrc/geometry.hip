// Rectified-stereo row/disparity filter fused with DLT triangulation for
// gfx950: one match per lane, everything in fp64 registers.
//
// Reference arithmetic being reproduced:
//   filter : frontend/geometry/triangulation.py:96-109
//            keep = abs(vR - vL) <= pixel_thresh and (uL - uR) >= disparity_min
//            (float32 pixel coordinates widened to fp64 before the arithmetic)
//   DLT    : frontend/geometry/triangulation.py:76-87 -> cv2.triangulatePoints:
//            rows  x*P[2] - P[0],  y*P[2] - P[1]  for the left then the right
//            camera, right singular vector of the smallest singular value,
//            then X = Xh[:3] / Xh[3].
// The 4x4 system is solved with one-sided (Hestenes) Jacobi rotations on the
// columns of A, accumulating V; it never forms A^T A, so the small singular
// value keeps full relative accuracy.
#include "common.h"

#include "dlt_device.h"

namespace {

using namespace slamhip_dev;

template <typename KP>
__global__ __launch_bounds__(256) void filter_triangulate_kernel(
    const KP *__restrict__ kp_left, int n_left, const KP *__restrict__ kp_right, int n_right,
    const int32_t *__restrict__ qidx, const int32_t *__restrict__ tidx, int m, CamPair cam,
    double pix_thresh, double disp_min, uint8_t *__restrict__ keep, void *__restrict__ xyz,
    int out_fp64) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const int li = qidx ? qidx[k] : k;
  const int ri = tidx ? tidx[k] : k;
  // a match naming a keypoint that does not exist is rejected and nothing is read for it
  const bool in_range = li >= 0 && li < n_left && ri >= 0 && ri < n_right;
  double uL = 0.0, vL = 0.0, uR = 0.0, vR = 0.0;
  if (in_range) {
    uL = (double)kp_left[2 * (size_t)li + 0];
    vL = (double)kp_left[2 * (size_t)li + 1];
    uR = (double)kp_right[2 * (size_t)ri + 0];
    vR = (double)kp_right[2 * (size_t)ri + 1];
  }

  bool ok = in_range;
  if (keep != nullptr) {
    ok = in_range && stereo_keep(uL, vL, uR, vR, pix_thresh, disp_min);
    keep[k] = ok ? 1 : 0;
  }

  double X = 0.0, Y = 0.0, Z = 0.0;
  float Xf = 0.f, Yf = 0.f, Zf = 0.f;
  if (ok) {
    double x[4];
    dlt_null_vector(cam, uL, vL, uR, vR, x);
    if (out_fp64) {  // 1: fp64 out; 2: fp64 division, result stored as fp32
      X = x[0] / x[3];
      Y = x[1] / x[3];
      Z = x[2] / x[3];
      Xf = (float)X; Yf = (float)Y; Zf = (float)Z;
    } else {
      const float x0 = (float)x[0], x1 = (float)x[1], x2 = (float)x[2], x3 = (float)x[3];
      Xf = x0 / x3;
      Yf = x1 / x3;
      Zf = x2 / x3;
    }
  }
  if (out_fp64 == 1) {
    double *o = reinterpret_cast<double *>(xyz) + 3 * (size_t)k;
    o[0] = X; o[1] = Y; o[2] = Z;
  } else {
    float *o = reinterpret_cast<float *>(xyz) + 3 * (size_t)k;
    o[0] = Xf; o[1] = Yf; o[2] = Zf;
  }
}

}  // namespace

extern "C" int slamhip_stereo_filter_triangulate(const void *d_kp_left, int n_left,
                                                 const void *d_kp_right, int n_right,
                                                 int kp_fp64, const int32_t *d_qidx,
                                                 const int32_t *d_tidx, int m,
                                                 const double *P_left, const double *P_right,
                                                 double pix_thresh, double disp_min,
                                                 uint8_t *d_keep, void *d_xyz, int out_fp64,
                                                 void *stream) {
  SLAMHIP_BEGIN();
  if (m < 0 || n_left < 0 || n_right < 0) return SLAMHIP_EINVAL;
  if (m == 0) return SLAMHIP_OK;
  if (!P_left || !P_right || !d_xyz) return SLAMHIP_EINVAL;
  if ((n_left > 0 && !d_kp_left) || (n_right > 0 && !d_kp_right)) return SLAMHIP_EINVAL;
  CamPair cam;
  for (int i = 0; i < 12; ++i) {
    cam.PL[i] = P_left[i];
    cam.PR[i] = P_right[i];
  }
  const dim3 grid(slamhip_ceil_div(m, 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (kp_fp64)
    hipLaunchKernelGGL(filter_triangulate_kernel<double>, grid, block, 0, st,
                       (const double *)d_kp_left, n_left, (const double *)d_kp_right, n_right, d_qidx, d_tidx, m,
                       cam, pix_thresh, disp_min, d_keep, d_xyz, out_fp64);
  else
    hipLaunchKernelGGL(filter_triangulate_kernel<float>, grid, block, 0, st,
                       (const float *)d_kp_left, n_left, (const float *)d_kp_right, n_right, d_qidx, d_tidx, m,
                       cam, pix_thresh, disp_min, d_keep, d_xyz, out_fp64);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}
