// PnP-RANSAC consensus scoring for gfx950: hypotheses x points in fp64.
//
// Mapping: one lane per HYPOTHESIS.  The lane turns its [R|t] into the left and
// right projection matrices once (24 fp64 registers) and then walks the points,
// which are wave-uniform (staged in LDS as fp64, read with a broadcast
// address).  Each lane keeps its own inlier count, so scoring needs no
// cross-lane reduction at all; the grid splits hypotheses and point chunks over
// blocks and point chunks are combined with one integer atomic per lane.
//
// Reference arithmetic (frontend/geometry/pnp.py of the reference):
//   P_L  = K @ Rt                                   pnp.py:133
//   Rt_R = (homogenous(M_right) @ homogenous(Rt))[:3];  P_R = K @ Rt_R   :136-138
//   proj = (P @ [X,1])[:2] / (P @ [X,1])[2]          pnp.py:94-99 (no z>0 test)
//   err  = sqrt(dx*dx + dy*dy)                       utils/graph_utils.py:7-11
//   mask = (err_L <= thr) & (err_R <= thr)           pnp.py:86-91
// The common case is decided without a division:
//   (u*w - x)^2 + (v*w - y)^2 <= thr^2 * w^2
// and any point whose two sides agree to 1e-10 relative is re-evaluated with the
// reference's literal divide / sqrt / compare sequence, so the decision equals
// the literal formula's decision on the same projection values.
// For float32 points a packed-fp32 screen (ransac_device.h) decides most points two
// at a time and leaves the rest to the fp64 test; the counts do not change.
#include "common.h"

#include <type_traits>

#include "ransac_device.h"

namespace {

using namespace slamhip_dev;

template <typename PT>
__device__ __forceinline__ void stage_points(double *__restrict__ lds, const PT *__restrict__ pts,
                                             const float *__restrict__ mL,
                                             const float *__restrict__ mR, int p0, int count) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    const size_t p = (size_t)(p0 + i);
    double *o = lds + i * kPtStride;
    o[0] = (double)pts[3 * p + 0];
    o[1] = (double)pts[3 * p + 1];
    o[2] = (double)pts[3 * p + 2];
    o[3] = (double)mL[2 * p + 0];
    o[4] = (double)mL[2 * p + 1];
    o[5] = (double)mR[2 * p + 0];
    o[6] = (double)mR[2 * p + 1];
    o[7] = 0.0;
  }
}

// grid = (hypothesis blocks of 64, point chunks of kPtsPerChunk)
// MODE: 0 = fp64 only, 1 = packed-fp32 screen, 2 = the same for a rectified pair
// (1 and 2 need exact fp32 points).
template <typename PT, int MODE>
__global__ __launch_bounds__(64) void ransac_score_kernel(
    const PT *__restrict__ pts, const float *__restrict__ mL, const float *__restrict__ mR, int n,
    const double *__restrict__ Rt, const uint8_t *__restrict__ valid, int h, RansacCam cam,
    double thr, int32_t *__restrict__ counts) {
  static_assert(MODE == 0 || std::is_same<PT, float>::value, "screening needs fp32 points");
  __shared__ __align__(16) double lds[kPtsPerChunk * kPtStride];
  const int p0 = blockIdx.y * kPtsPerChunk;
  const int cnt = min(kPtsPerChunk, n - p0);
  unsigned umax = 0u;
  if constexpr (MODE != 0) {
    float *ldsf = reinterpret_cast<float *>(lds);
    for (int j = threadIdx.x; 2 * j < cnt; j += blockDim.x) {
      const size_t a = (size_t)p0 + 2 * j, b = (2 * j + 1 < cnt) ? a + 1 : a;
      unsigned s2max = 0u;   // not needed here
      stage_pair(ldsf + j * kPairStride, reinterpret_cast<const float *>(pts), mL, mR, a, b, umax, s2max);
    }
    umax = wave_max_u32(umax);
  } else {
    stage_points(lds, pts, mL, mR, p0, cnt);
  }
  __syncthreads();

  const int hyp = blockIdx.x * 64 + threadIdx.x;
  if (hyp >= h) return;
  if (valid != nullptr && valid[hyp] == 0) {
    if (blockIdx.y == 0) counts[hyp] = -1;
    return;
  }
  double PL[12], PR[12];
  make_projections(Rt + 12 * (size_t)hyp, cam, PL, PR);
  const double thr2 = threshold_squared(thr);
  int c;
  if constexpr (MODE != 0) {
    Hyp32 h32;
    make_hyp32(PL, PR, h32);
    const Screen32 sc = make_screen(h32, umax, (float)thr, (float)thr2);
    c = count_chunk_inliers_screened<MODE == 2>(reinterpret_cast<const float *>(lds), cnt, PL, PR,
                                                h32, sc, thr, thr2);
  } else {
    c = count_chunk_inliers(lds, cnt, PL, PR, thr, thr2);
  }
  if (gridDim.y == 1) counts[hyp] = c;
  else if (c != 0) atomicAdd(&counts[hyp], c);
}

// One hypothesis, one lane per point.
template <typename PT>
__global__ __launch_bounds__(256) void ransac_mask_kernel(
    const PT *__restrict__ pts, const float *__restrict__ mL, const float *__restrict__ mR, int n,
    const double *__restrict__ Rt_one, RansacCam cam, double thr, uint8_t *__restrict__ mask) {
  double PL[12], PR[12];
  make_projections(Rt_one, cam, PL, PR);
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const double X = (double)pts[3 * (size_t)p + 0];
  const double Y = (double)pts[3 * (size_t)p + 1];
  const double Z = (double)pts[3 * (size_t)p + 2];
  const double thr2 = threshold_squared(thr);
  const bool inL = inlier_one_camera(PL, X, Y, Z, (double)mL[2 * (size_t)p + 0],
                                     (double)mL[2 * (size_t)p + 1], thr, thr2);
  const bool inR = inlier_one_camera(PR, X, Y, Z, (double)mR[2 * (size_t)p + 0],
                                     (double)mR[2 * (size_t)p + 1], thr, thr2);
  mask[p] = (inL && inR) ? 1 : 0;
}

// Sequential replay of the adaptive stopping rule (pnp.py:123-158) by one lane.
__global__ void ransac_replay_kernel(const int32_t *__restrict__ counts, int h, int n_points,
                                     double p_success, int n_iter_max, int early_inliers,
                                     int32_t *__restrict__ result) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int best_count = 0, best_idx = -1, iter_done = 0, domain_error = 0;
  int n_req = n_iter_max;
  const double log1mp = log(1.0 - p_success);
  const int limit = min(h, n_iter_max);
  while (iter_done < n_req && iter_done < limit) {
    const int c = counts[iter_done];
    const int idx = iter_done;
    ++iter_done;
    if (c < 0) continue;  // failed minimal solve: iteration counted, nothing scored
    if (c > best_count) {
      best_count = c;
      best_idx = idx;
      const double w = (double)best_count / (double)n_points;
      if (w > 0.0) {
        const double w2 = w * w;
        const double den = log(1.0 - w2 * w2);  // w**4 ; -inf when every point is an inlier
        if (best_count >= n_points || !(p_success < 1.0)) domain_error = 1;  // math.log(0)
        const double r = ceil(log1mp / den);
        n_req = r > 2147483647.0 ? 2147483647 : (int)r;
      }
    }
    if (early_inliers >= 0 && best_count >= early_inliers) break;
  }
  result[0] = best_idx;
  result[1] = best_count;
  result[2] = iter_done;
  result[3] = n_req;
  result[4] = domain_error;
  result[5] = 0;
  result[6] = 0;
  result[7] = 0;
}

struct Proj { double P[12]; };

// PnP.project (pnp.py:94-99): (P @ [X,1])[:2] / (P @ [X,1])[2] in fp64.
template <typename PT>
__global__ __launch_bounds__(256) void project_kernel(const PT *__restrict__ pts, int n, Proj pr,
                                                      double *__restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const double X = (double)pts[3 * (size_t)p + 0];
  const double Y = (double)pts[3 * (size_t)p + 1];
  const double Z = (double)pts[3 * (size_t)p + 2];
  const double *P = pr.P;
  // The reference's P @ [X,1] goes through its BLAS, whose rounding (fused or not, and in
  // which order) is not specified; here: fused multiply-adds in column order.
  const double x = fma(P[2], Z, fma(P[1], Y, fma(P[0], X, P[3])));
  const double y = fma(P[6], Z, fma(P[5], Y, fma(P[4], X, P[7])));
  const double w = fma(P[10], Z, fma(P[9], Y, fma(P[8], X, P[11])));
  out[2 * (size_t)p + 0] = x / w;
  out[2 * (size_t)p + 1] = y / w;
}

// compute_err (utils/graph_utils.py:7-11) twice + PnP.supporters_pair (pnp.py:86-91).
__global__ __launch_bounds__(256) void supporters_kernel(const double *__restrict__ mL,
                                                         const double *__restrict__ pL,
                                                         const double *__restrict__ mR,
                                                         const double *__restrict__ pR, int n,
                                                         double thr, uint8_t *__restrict__ mask) {
  // no contraction of dx*dx + dy*dy into a fused multiply-add (NumPy rounds each step);
  // plain operators, because the pragma does not reach the __dmul_rn / __dadd_rn intrinsics
#pragma clang fp contract(off)
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const size_t i = 2 * (size_t)p;
  const double ax = mL[i] - pL[i], ay = mL[i + 1] - pL[i + 1];
  const double bx = mR[i] - pR[i], by = mR[i + 1] - pR[i + 1];
  const double eL = sqrt(ax * ax + ay * ay);
  const double eR = sqrt(bx * bx + by * by);
  mask[p] = (eL <= thr && eR <= thr) ? 1 : 0;
}

// compute_err (utils/graph_utils.py:7-11): np.linalg.norm(meas - proj, axis=1) for (n,2)
// fp64 arrays = sqrt(dx*dx + dy*dy), each step rounded on its own, the root correctly rounded.
__global__ __launch_bounds__(256) void compute_err_kernel(const double *__restrict__ meas,
                                                          const double *__restrict__ proj, int n,
                                                          double *__restrict__ err) {
#pragma clang fp contract(off)
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const size_t i = 2 * (size_t)p;
  const double dx = meas[i] - proj[i], dy = meas[i + 1] - proj[i + 1];
  const double sx = dx * dx, sy = dy * dy;
  err[p] = __dsqrt_rn(sx + sy);
}

void fill_cam(RansacCam &cam, const double *K, const double *M_right) {
  for (int i = 0; i < 9; ++i) cam.K[i] = K[i];
  for (int i = 0; i < 12; ++i) cam.MR[i] = M_right[i];
}

}  // namespace

int slamhip_internal::g_ransac_screen = 1;
int slamhip_internal::g_ransac_mode = 0;
int slamhip_internal::g_wide_rounds = 2;
int slamhip_internal::g_small_call_fusion = 1;
int slamhip_internal::g_small_call_span = 0;
int slamhip_internal::g_count_split = 1;
int slamhip_internal::g_xcd_remap = 1;
hipEvent_t slamhip_internal::g_dispatch_start = nullptr, slamhip_internal::g_dispatch_stop = nullptr;

extern "C" {

int slamhip_ransac_score(const void *d_points, int points_fp64, const float *d_meas_l,
                         const float *d_meas_r, int n, const double *d_Rt,
                         const uint8_t *d_valid, int h, const double *K, const double *M_right,
                         double thresh, int32_t *d_counts, void *stream) {
  SLAMHIP_BEGIN();
  if (n < 0 || h < 0) return SLAMHIP_EINVAL;
  if (h == 0) return SLAMHIP_OK;
  if (!d_Rt || !d_counts || !K || !M_right) return SLAMHIP_EINVAL;
  if (n > 0 && (!d_points || !d_meas_l || !d_meas_r)) return SLAMHIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  RansacCam cam;
  fill_cam(cam, K, M_right);
  const int chunks = n > 0 ? slamhip_ceil_div(n, kPtsPerChunk) : 1;
  if (chunks > 65535) return SLAMHIP_EUNSUPPORTED;
  if (chunks > 1) {
    if (hipMemsetAsync(d_counts, 0, sizeof(int32_t) * (size_t)h, st) != hipSuccess)
      return SLAMHIP_ELAUNCH;
  }
  const dim3 grid(slamhip_ceil_div(h, 64), chunks), block(64);
  if (points_fp64) {
    hipLaunchKernelGGL((ransac_score_kernel<double, 0>), grid, block, 0, st, (const double *)d_points,
                       d_meas_l, d_meas_r, n, d_Rt, d_valid, h, cam, thresh, d_counts);
  } else {
    const int mode = screening_mode(slamhip_internal::g_ransac_screen, K, M_right, thresh);
    auto kern = mode == 2 ? ransac_score_kernel<float, 2> : mode == 1 ? ransac_score_kernel<float, 1>
                                                                     : ransac_score_kernel<float, 0>;
    hipLaunchKernelGGL(kern, grid, block, 0, st, (const float *)d_points, d_meas_l, d_meas_r, n, d_Rt,
                       d_valid, h, cam, thresh, d_counts);
  }
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

int slamhip_ransac_mask(const void *d_points, int points_fp64, const float *d_meas_l,
                        const float *d_meas_r, int n, const double *d_Rt_one, const double *K,
                        const double *M_right, double thresh, uint8_t *d_mask, void *stream) {
  SLAMHIP_BEGIN();
  if (n < 0) return SLAMHIP_EINVAL;
  if (n == 0) return SLAMHIP_OK;
  if (!d_points || !d_meas_l || !d_meas_r || !d_Rt_one || !K || !M_right || !d_mask)
    return SLAMHIP_EINVAL;
  RansacCam cam;
  fill_cam(cam, K, M_right);
  const dim3 grid(slamhip_ceil_div(n, 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (points_fp64)
    hipLaunchKernelGGL(ransac_mask_kernel<double>, grid, block, 0, st, (const double *)d_points,
                       d_meas_l, d_meas_r, n, d_Rt_one, cam, thresh, d_mask);
  else
    hipLaunchKernelGGL(ransac_mask_kernel<float>, grid, block, 0, st, (const float *)d_points,
                       d_meas_l, d_meas_r, n, d_Rt_one, cam, thresh, d_mask);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

int slamhip_ransac_replay(const int32_t *d_counts, int h, int n_points, double p_success,
                          int n_iter_max, int early_inliers, int32_t *d_result, void *stream) {
  SLAMHIP_BEGIN();
  if (h < 0 || n_points < 1 || n_iter_max < 0) return SLAMHIP_EINVAL;
  if (!d_result || (h > 0 && !d_counts)) return SLAMHIP_EINVAL;
  hipLaunchKernelGGL(ransac_replay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_counts, h,
                     n_points, p_success, n_iter_max, early_inliers, d_result);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

int slamhip_project(const void *d_points, int points_fp64, int n, const double *P,
                    double *d_pixels, void *stream) {
  SLAMHIP_BEGIN();
  if (n < 0) return SLAMHIP_EINVAL;
  if (n == 0) return SLAMHIP_OK;
  if (!d_points || !P || !d_pixels) return SLAMHIP_EINVAL;
  Proj pr;
  for (int i = 0; i < 12; ++i) pr.P[i] = P[i];
  const dim3 grid(slamhip_ceil_div(n, 256)), block(256);
  if (points_fp64)
    hipLaunchKernelGGL(project_kernel<double>, grid, block, 0, (hipStream_t)stream,
                       (const double *)d_points, n, pr, d_pixels);
  else
    hipLaunchKernelGGL(project_kernel<float>, grid, block, 0, (hipStream_t)stream,
                       (const float *)d_points, n, pr, d_pixels);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

int slamhip_compute_err(const double *d_meas, const double *d_proj, int n, double *d_err,
                        void *stream) {
  SLAMHIP_BEGIN();
  if (n < 0) return SLAMHIP_EINVAL;
  if (n == 0) return SLAMHIP_OK;
  if (!d_meas || !d_proj || !d_err) return SLAMHIP_EINVAL;
  hipLaunchKernelGGL(compute_err_kernel, dim3(slamhip_ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, d_meas, d_proj, n, d_err);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

int slamhip_supporters_pair(const double *d_meas_l, const double *d_proj_l,
                            const double *d_meas_r, const double *d_proj_r, int n, double thresh,
                            uint8_t *d_mask, void *stream) {
  SLAMHIP_BEGIN();
  if (n < 0) return SLAMHIP_EINVAL;
  if (n == 0) return SLAMHIP_OK;
  if (!d_meas_l || !d_proj_l || !d_meas_r || !d_proj_r || !d_mask) return SLAMHIP_EINVAL;
  hipLaunchKernelGGL(supporters_kernel, dim3(slamhip_ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, d_meas_l, d_proj_l, d_meas_r, d_proj_r, n, thresh,
                     d_mask);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

int slamhip_version(void) { return SLAMHIP_VERSION; }

const char *slamhip_strerror(int code) {
  switch (code) {
    case SLAMHIP_OK: return "ok";
    case SLAMHIP_EINVAL: return "invalid argument";
    case SLAMHIP_EUNSUPPORTED: return "size outside the supported range";
    case SLAMHIP_EWORKSPACE: return "workspace missing or too small";
    case SLAMHIP_ELAUNCH: return "HIP launch failed";
    default: return "unknown error";
  }
}

}  // extern "C"
