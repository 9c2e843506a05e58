// Pose refinement on the inlier set: Levenberg-Marquardt on the left-image
// reprojection error, one workgroup per problem, fp64.
//
// Role in the reference: PnP.solve_pnp(X[mask], pix_L[mask], cv2.SOLVEPNP_ITERATIVE)
// after RANSAC (backend/tracking/fill_database.py:204-207,
// backend/optimizers/loop_closure.py:222): the pose that minimises
//   sum_i | proj(K (R X_i + t)) - m_i |^2   over the selected correspondences.
// OpenCV reaches that minimum by LM from a DLT start; this kernel reaches the
// same minimum by LM from the RANSAC winner (or any given start).
//
// Parametrisation: left-multiplied twist, Xc' = exp(w) Xc + v.  Each lane
// accumulates the 21 + 6 normal-equation terms and the cost over its points;
// wave shuffles + LDS fold them; lane 0 solves the damped 6x6 system (Cholesky)
// and broadcasts the candidate pose through LDS.
#include "common.h"

#include "refine_device.h"

namespace {

using namespace slamhip_dev;

template <typename PT>
__global__ __launch_bounds__(kRefThreads) void refine_pose_kernel(
    const PT *__restrict__ pts_all, const float *__restrict__ pix_all,
    const uint8_t *__restrict__ mask_all, const int32_t *__restrict__ off,
    const int32_t *__restrict__ cnt, int n_single, const double *__restrict__ Rt_init, Intr in,
    int max_iters, double *__restrict__ Rt_out, double *__restrict__ info_out,
    const int32_t *__restrict__ ransac_info, int min_inliers) {
  __shared__ RefineLds L;
  const int prob = blockIdx.x;
  // A problem whose consensus set is smaller than the caller's acceptance threshold is
  // rejected by the caller whatever its refined pose (loop_closure.py:218-219 returns
  // before the refinement): hand the start back.
  if (ransac_info != nullptr && ransac_info[8 * (size_t)prob + 6] < min_inliers) {
    if (threadIdx.x < 12) Rt_out[12 * (size_t)prob + threadIdx.x] = Rt_init[12 * (size_t)prob + threadIdx.x];
    if (threadIdx.x == 0 && info_out != nullptr) {
      info_out[2 * (size_t)prob] = 0.0;
      info_out[2 * (size_t)prob + 1] = 0.0;
    }
    return;
  }
  int p0 = 0, n = n_single;
  if (off != nullptr) { p0 = off[prob]; n = cnt ? cnt[prob] : off[prob + 1] - p0; }
  const PT *pts = pts_all + 3 * (size_t)p0;
  const float *pix = pix_all + 2 * (size_t)p0;
  const uint8_t *mask = mask_all ? mask_all + p0 : nullptr;

  refine_problem<PT>(pts, pix, mask, n, Rt_init + 12 * (size_t)prob, in, max_iters, Rt_out + 12 * (size_t)prob,
                     info_out ? info_out + 2 * (size_t)prob : nullptr, L);
}

}  // namespace

extern "C" int slamhip_refine_pose(const void *d_points, int points_fp64, const float *d_pixels,
                                   const uint8_t *d_mask, int n, const int32_t *d_off,
                                   int n_problems, const double *K, const double *d_Rt_init,
                                   int max_iters, double *d_Rt_out, double *d_info, void *stream) {
  SLAMHIP_BEGIN();
  if (n < 0 || n_problems < 1 || max_iters < 0) return SLAMHIP_EINVAL;
  if (!d_points || !d_pixels || !K || !d_Rt_init || !d_Rt_out) return SLAMHIP_EINVAL;
  Intr in;
  for (int i = 0; i < 9; ++i) in.K[i] = K[i];
  const dim3 grid(n_problems), block(kRefThreads);
  if (points_fp64)
    hipLaunchKernelGGL(refine_pose_kernel<double>, grid, block, 0, (hipStream_t)stream,
                       (const double *)d_points, d_pixels, d_mask, d_off, (const int32_t *)nullptr,
                       d_off ? 0 : n, d_Rt_init, in, max_iters, d_Rt_out, d_info,
                       (const int32_t *)nullptr, 0);
  else
    hipLaunchKernelGGL(refine_pose_kernel<float>, grid, block, 0, (hipStream_t)stream,
                       (const float *)d_points, d_pixels, d_mask, d_off, (const int32_t *)nullptr,
                       d_off ? 0 : n, d_Rt_init, in, max_iters, d_Rt_out, d_info,
                       (const int32_t *)nullptr, 0);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

int slamhip_internal::refine_pose(const void *d_points, int points_fp64, const float *d_pixels,
                                  const uint8_t *d_mask, const int32_t *d_off, const int32_t *d_cnt,
                                  int n_problems, const double *K, const double *d_Rt_init,
                                  int max_iters, double *d_Rt_out, double *d_info, hipStream_t st,
                                  const int32_t *d_ransac_info, int min_inliers) {
  if (n_problems <= 0) return SLAMHIP_OK;
  Intr in;
  for (int i = 0; i < 9; ++i) in.K[i] = K[i];
  const dim3 grid(n_problems), block(kRefThreads);
  if (points_fp64)
    hipLaunchKernelGGL(refine_pose_kernel<double>, grid, block, 0, st, (const double *)d_points,
                       d_pixels, d_mask, d_off, d_cnt, 0, d_Rt_init, in, max_iters, d_Rt_out, d_info,
                       d_ransac_info, min_inliers);
  else
    hipLaunchKernelGGL(refine_pose_kernel<float>, grid, block, 0, st, (const float *)d_points,
                       d_pixels, d_mask, d_off, d_cnt, 0, d_Rt_init, in, max_iters, d_Rt_out, d_info,
                       d_ransac_info, min_inliers);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}
