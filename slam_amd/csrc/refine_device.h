// Device code of the pose refinement (Levenberg-Marquardt on the left-image reprojection
// error, one workgroup of kRefThreads threads per problem, fp64), shared by the stand-alone
// kernel (refine.hip) and by the kernel that runs it straight behind the RANSAC replay of
// the same pair (pipeline.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Floating-point contraction in this header is confined to what the source spells out (a product
// and the sum it feeds in one expression, or an explicit fma): the functions here are inlined into
// several kernels whose results are compared bit for bit, and the compiler's default for device
// code (fuse whatever it finds once everything is inlined) makes that depend on the surroundings.
#pragma clang fp contract(on)
namespace slamhip_dev {

constexpr int kRefThreads = 256;
constexpr int kRefTermStride = kRefThreads + 2;   // doubles between the rows of two terms in LDS

struct Intr { double K[9]; };

// Sum over the 64 lanes of a wavefront, the same value in every lane.  Within a row of 16
// lanes through DPP (two 32-bit moves per step: a few cycles each, where a shuffle through
// LDS hardware costs a round trip per step, 28 sums x 6 steps per pass); the four row sums
// through scalar registers.
__device__ __forceinline__ double dpp_add(double v, int ctrl) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  int olo, ohi;
  switch (ctrl) {   // the control word is an immediate
    case 0: olo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
            ohi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;   // lanes 1,0,3,2
    case 1: olo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true);
            ohi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;   // lanes 2,3,0,1
    case 2: olo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true);
            ohi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break;  // half-row mirror
    default: olo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true);
             ohi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break; // row mirror
  }
  return v + __hiloint2double(ohi, olo);
}
__device__ __forceinline__ double read_lane(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
  v = dpp_add(v, 0);
  v = dpp_add(v, 1);
  v = dpp_add(v, 2);
  v = dpp_add(v, 3);
  return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}

// exp of a rotation vector (Rodrigues), row-major 3x3
static __device__ void so3_exp(const double *w, double *R) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double th = sqrt(th2);   // used by the large-angle branch only
  double a, b;  // sin(th)/th, (1-cos(th))/th^2
  if (th2 < 0.0625) {
    // |w| < 0.25 rad (every step from a RANSAC-grade start): the two series to th^18 / th^20, whose first
    // dropped terms are below 1e-26 there -- a dozen multiply-adds in place of fp64 sin and cos with their
    // range reduction, in the one lane whose chain every pass of the kernel waits for
    const double x = th2;
    a = 1.0 + x * (-1.0 / 6 + x * (1.0 / 120 + x * (-1.0 / 5040 + x * (1.0 / 362880 + x * (-1.0 / 39916800 +
        x * (1.0 / 6227020800.0 + x * (-1.0 / 1307674368000.0 + x * (1.0 / 355687428096000.0))))))));
    b = 0.5 + x * (-1.0 / 24 + x * (1.0 / 720 + x * (-1.0 / 40320 + x * (1.0 / 3628800 + x * (-1.0 / 479001600 +
        x * (1.0 / 87178291200.0 + x * (-1.0 / 20922789888000.0 + x * (1.0 / 6402373705728000.0))))))));
  } else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
  const double wx = w[0], wy = w[1], wz = w[2];
  R[0] = 1.0 - b * (wy * wy + wz * wz); R[1] = -a * wz + b * wx * wy; R[2] = a * wy + b * wx * wz;
  R[3] = a * wz + b * wx * wy; R[4] = 1.0 - b * (wx * wx + wz * wz); R[5] = -a * wx + b * wy * wz;
  R[6] = -a * wy + b * wx * wz; R[7] = a * wx + b * wy * wz; R[8] = 1.0 - b * (wx * wx + wy * wy);
}

// Solve (H + lambda*diag(H)) x = -g for symmetric positive definite H (6x6,
// upper triangle packed row-wise in h[21]).  Returns false if not SPD.
static __device__ bool solve_damped(const double *h, const double *g, double lambda, double *x) {
  double A[6][6];
  int k = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) { A[i][j] = h[k]; A[j][i] = h[k]; ++k; }
  for (int i = 0; i < 6; ++i) A[i][i] += lambda * fmax(A[i][i], 1e-12);
  // Cholesky with the reciprocal of each pivot's root kept (one reciprocal square root per
  // row in place of a root and a division per entry below it: this runs in one lane, and
  // its 27 divisions were a third of a refinement pass)
  double L[6][6], inv[6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i][j];
      for (int m = 0; m < j; ++m) s -= L[i][m] * L[j][m];
      if (i == j) {
        if (!(s > 0.0)) return false;
        inv[i] = rsqrt(s);
        L[i][i] = s * inv[i];
      } else {
        L[i][j] = s * inv[j];
      }
    }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = -g[i];
    for (int m = 0; m < i; ++m) s -= L[i][m] * y[m];
    y[i] = s * inv[i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    for (int m = i + 1; m < 6; ++m) s -= L[m][i] * x[m];
    x[i] = s * inv[i];
  }
  return true;
}

// One pass over the points at pose `Rt`: normal equations + cost, folded over
// the workgroup into out[28] (valid in every lane of wave 0 after the call;
// lane 0 of the block reads it from LDS).
template <typename PT>
__device__ void accumulate(const PT *__restrict__ pts, const float *__restrict__ pix,
                           const uint8_t *__restrict__ mask, int n, const double *Rt,
                           const Intr &in, double *lds /* RefineLds::lds */, double *out /* [28] */) {
  double acc[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) acc[i] = 0.0;
  const double fx = in.K[0], sk = in.K[1], cx = in.K[2], fy = in.K[4], cy = in.K[5];
  for (int p = threadIdx.x; p < n; p += kRefThreads) {
    if (mask != nullptr && mask[p] == 0) continue;
    const double X = (double)pts[3 * (size_t)p], Y = (double)pts[3 * (size_t)p + 1],
                 Z = (double)pts[3 * (size_t)p + 2];
    const double xc = Rt[0] * X + Rt[1] * Y + Rt[2] * Z + Rt[3];
    const double yc = Rt[4] * X + Rt[5] * Y + Rt[6] * Z + Rt[7];
    const double zc = Rt[8] * X + Rt[9] * Y + Rt[10] * Z + Rt[11];
    const double iz = 1.0 / zc;
    const double xn = xc * iz, yn = yc * iz;
    const double ru = fx * xn + sk * yn + cx - (double)pix[2 * (size_t)p];
    const double rv = fy * yn + cy - (double)pix[2 * (size_t)p + 1];
    // d(u,v)/d(Xc): rows
    const double ux = fx * iz, uy = sk * iz, uz = -(fx * xn + sk * yn) * iz;
    const double vx = 0.0, vy = fy * iz, vz = -(fy * yn) * iz;
    // columns: w (rotation) then v (translation); dXc/dw = -[Xc]x
    double ju[6], jv[6];
    ju[0] = uy * (-zc) + uz * yc;  // d/dwx: Xc' = w x Xc -> (0,-zc,yc)
    ju[1] = ux * zc + uz * (-xc);  // d/dwy: (zc,0,-xc)
    ju[2] = ux * (-yc) + uy * xc;  // d/dwz: (-yc,xc,0)
    ju[3] = ux; ju[4] = uy; ju[5] = uz;
    jv[0] = vy * (-zc) + vz * yc;
    jv[1] = vx * zc + vz * (-xc);
    jv[2] = vx * (-yc) + vy * xc;
    jv[3] = vx; jv[4] = vy; jv[5] = vz;
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) { acc[k] += ju[i] * ju[j] + jv[i] * jv[j]; ++k; }
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] += ju[i] * ru + jv[i] * rv;
    acc[27] += ru * ru + rv * rv;
  }
  // The 28 sums over the workgroup's 256 lanes, through LDS: every lane lays its terms down
  // (term-major, a lane apart: no bank conflict on the way in), thread (term, part) adds the
  // 32 lanes of its part, and the eight parts of a term meet in a second, short step.  Some
  // hundred instructions a pass where 28 wavefront-wide sums through DPP took six hundred --
  // a quarter of a pass, in a kernel that is a chain of such passes.
  double *const terms = lds;                          // [28][kRefTermStride]
  double *const parts = lds + 28 * kRefTermStride;    // [28][8]
#pragma unroll
  for (int i = 0; i < 28; ++i) terms[i * kRefTermStride + threadIdx.x] = acc[i];
  __syncthreads();
  {
    const int term = threadIdx.x & 31, part = threadIdx.x >> 5;
    if (term < 28) {
      const double *src = terms + term * kRefTermStride + part * 32;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int r = 0; r < 32; r += 4) { s0 += src[r]; s1 += src[r + 1]; s2 += src[r + 2]; s3 += src[r + 3]; }
      parts[term * 8 + part] = (s0 + s1) + (s2 + s3);
    }
  }
  __syncthreads();
  if (threadIdx.x < 28) {
    const double *q = parts + threadIdx.x * 8;
    out[threadIdx.x] = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
  }
  __syncthreads();
}

// LDS of one refinement problem.
struct RefineLds {
  double lds[28 * kRefTermStride + 28 * 8];
  double sums[28];
  double pose[12];   // current accepted pose
  double cand[12];   // candidate pose
  int ctrl[2];       // [0] continue flag, [1] candidate exists
};

// One problem, worked by the whole workgroup (kRefThreads threads, every one of them calls):
// from the start pose Rt_init[12] to Rt_out[12]; info_out (may be null): final cost, passes.
template <typename PT>
__device__ __forceinline__ void refine_problem(const PT *__restrict__ pts, const float *__restrict__ pix,
                                               const uint8_t *__restrict__ mask, int n,
                                               const double *__restrict__ Rt_init, const Intr &in, int max_iters,
                                               double *__restrict__ Rt_out, double *__restrict__ info_out,
                                               RefineLds &L) {
  double *const lds = L.lds, *const sums = L.sums, *const pose = L.pose, *const cand = L.cand;
  int *const ctrl = L.ctrl;
  if (threadIdx.x < 12) pose[threadIdx.x] = Rt_init[threadIdx.x];
  __syncthreads();
  accumulate(pts, pix, mask, n, pose, in, lds, sums);

  // lane 0 state
  double H[21], g[6], cost = 0.0, lambda = 1e-6, last_step = 0.0;
  int iters = 0;
  if (threadIdx.x == 0) {
    for (int i = 0; i < 21; ++i) H[i] = sums[i];
    for (int i = 0; i < 6; ++i) g[i] = sums[21 + i];
    cost = sums[27];
  }
  for (int it = 0; it < max_iters; ++it) {
    if (threadIdx.x == 0) {
      double dx[6];
      bool ok = false;
      last_step = 0.0;
      for (int tries = 0; tries < 8 && !ok; ++tries) {
        ok = solve_damped(H, g, lambda, dx);
        if (!ok) lambda = fmax(lambda * 10.0, 1e-6);
      }
      double step = 0.0;
      if (ok) for (int i = 0; i < 6; ++i) step = fmax(step, fabs(dx[i]));
      last_step = step;
      // Convergence is quadratic near the minimum: a step below 1e-10 (rad, m) means
      // the pose already sits at it to that accuracy, so the pass that would only
      // confirm it is not spent.
      // A step below 1e-7 from an undamped system (no rejection behind it) is a Gauss-Newton step
      // next to the minimum: the pose it leads to is at the minimum to ~step^2, far inside what a
      // pass that only confirmed it could tell, so it is taken as it is and that pass is not spent.
      const bool last = ok && step >= 1e-10 && step < 1e-7 && lambda <= 1e-5;
      if (!ok || step < 1e-10) {
        ctrl[0] = 0;
      } else {
        double E[9];
        so3_exp(dx, E);
        for (int i = 0; i < 3; ++i) {
          for (int j = 0; j < 4; ++j)
            cand[4 * i + j] = E[3 * i] * pose[j] + E[3 * i + 1] * pose[4 + j] + E[3 * i + 2] * pose[8 + j];
          cand[4 * i + 3] += dx[3 + i];
        }
        ctrl[0] = 1;
        if (last) {
          for (int i = 0; i < 12; ++i) pose[i] = cand[i];
          ctrl[0] = 0;
          ++iters;
        }
      }
    }
    __syncthreads();
    if (ctrl[0] == 0) break;
    accumulate(pts, pix, mask, n, cand, in, lds, sums);
    if (threadIdx.x == 0) {
      ++iters;
      const double new_cost = sums[27];
      if (new_cost < cost) {
        const double rel = (cost - new_cost) / fmax(cost, 1e-300);
        for (int i = 0; i < 12; ++i) pose[i] = cand[i];
        for (int i = 0; i < 21; ++i) H[i] = sums[i];
        for (int i = 0; i < 6; ++i) g[i] = sums[21 + i];
        cost = new_cost;
        lambda = fmax(lambda * 0.1, 1e-12);
        ctrl[1] = rel < 1e-14 ? 0 : 1;
      } else {
        // a rejected step this small is rounding noise around the minimum, not a
        // sign that more damping is needed
        lambda *= 10.0;
        ctrl[1] = (lambda > 1e12 || last_step < 1e-8) ? 0 : 1;
      }
    }
    __syncthreads();
    if (ctrl[1] == 0) break;
  }
  __syncthreads();
  if (threadIdx.x < 12) Rt_out[threadIdx.x] = pose[threadIdx.x];
  if (threadIdx.x == 0 && info_out != nullptr) {
    info_out[0] = cost;
    info_out[1] = (double)iters;
  }
}

}  // namespace slamhip_dev
#pragma clang fp contract(fast)   // the compiler's default for device code, for whatever follows

