// Batched front-end: the data flow of TrackingPipeline.build()
// (backend/tracking/fill_database.py:140-240 of the reference) for a whole run
// of frames in a fixed number of launches, with no host round trip in between.
//
// Per frame f (stereo stage, fill_database.py:155-169):
//   match L->R (k=1) -> sort by distance (stable) -> row/disparity filter ->
//   triangulate the survivors -> gather their left descriptors / pixels
//   (create_links order: the order of the sorted, filtered matches).
// Per consecutive pair (f, f+1) (temporal stage, fill_database.py:185-207):
//   match inlier descriptors of f against those of f+1 -> sort ->
//   X_corr = X_f[queryIdx], pixels of f+1 by trainIdx -> adaptive RANSAC over
//   P3P hypotheses with stereo consensus -> LM refinement on the inliers.
//
// Every per-frame array lives at the frame's left-feature offset (a frame has at
// most as many stereo inliers as left features, and a pair has as many temporal
// matches as the earlier frame has inliers), so no global prefix sum is needed;
// counts are kept in device arrays and kernels take (offset, count) segments.
#include "common.h"

#include "dlt_device.h"
#include "p3p_device.h"
#include "ransac_device.h"

namespace {

using namespace slamhip_dev;

constexpr int kBlock = 256;

// --------------------------------------------------------------------------
// Stereo stage: filter + stable compaction + triangulation + gathers.
// One workgroup per frame walks the sorted matches in chunks of 256.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void stereo_compact_kernel(
    const float *__restrict__ kp_left, const float *__restrict__ kp_right,
    const int32_t *__restrict__ off_left, const int32_t *__restrict__ off_right,
    const int32_t *__restrict__ idx2, const int32_t *__restrict__ dist2,
    const int32_t *__restrict__ order, const uint32_t *__restrict__ packed_left, int W, CamPair cam,
    double pix_thresh, double disp_min, int32_t *__restrict__ st_count,
    int32_t *__restrict__ st_left, int32_t *__restrict__ st_right, int32_t *__restrict__ st_dist,
    float *__restrict__ st_xyz, float *__restrict__ st_pix_left, float *__restrict__ st_pix_right,
    uint32_t *__restrict__ st_desc) {
  __shared__ int wave_tot[kBlock / 64];
  __shared__ int running;
  const int f = blockIdx.x;
  const int base = off_left[f], nq = off_left[f + 1] - base, rbase = off_right[f];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  for (int chunk = 0; chunk < nq; chunk += kBlock) {
    const int r = chunk + (int)threadIdx.x;
    bool keep = false;
    int q = 0, t = -1, d = -1;
    double uL = 0, vL = 0, uR = 0, vR = 0;
    if (r < nq) {
      q = order[base + r];
      t = idx2[2 * (size_t)(base + q)];
      d = dist2[2 * (size_t)(base + q)];
      if (t >= 0) {
        uL = (double)kp_left[2 * (size_t)(base + q)];
        vL = (double)kp_left[2 * (size_t)(base + q) + 1];
        uR = (double)kp_right[2 * (size_t)(rbase + t)];
        vR = (double)kp_right[2 * (size_t)(rbase + t) + 1];
        keep = stereo_keep(uL, vL, uR, vR, pix_thresh, disp_min);
      }
    }
    const unsigned long long ballot = __ballot(keep);
    const int prefix = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(ballot);
    __syncthreads();
    int wave_base = 0, chunk_total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
      if (w < wave) wave_base += wave_tot[w];
      chunk_total += wave_tot[w];
    }
    const int pos = running + wave_base + prefix;
    if (keep) {
      double x[4];
      dlt_null_vector(cam, uL, vL, uR, vR, x);
      const float x0 = (float)x[0], x1 = (float)x[1], x2 = (float)x[2], x3 = (float)x[3];
      const size_t o = (size_t)(base + pos);
      st_left[o] = q;
      st_right[o] = t;
      st_dist[o] = d;
      st_xyz[3 * o + 0] = x0 / x3;
      st_xyz[3 * o + 1] = x1 / x3;
      st_xyz[3 * o + 2] = x2 / x3;
      st_pix_left[2 * o + 0] = (float)uL;
      st_pix_left[2 * o + 1] = (float)vL;
      st_pix_right[2 * o + 0] = (float)uR;
      st_pix_right[2 * o + 1] = (float)vR;
    }
    __syncthreads();
    if (threadIdx.x == 0) running += chunk_total;
    __syncthreads();
  }
  const int m = running;
  if (threadIdx.x == 0) st_count[f] = m;
  // descriptor rows of the inliers, in inlier order (global writes above are
  // visible to the whole workgroup after the barriers)
  for (int e = threadIdx.x; e < m * W; e += kBlock) {
    const int i = e / W, w = e - i * W;
    st_desc[(size_t)(base + i) * W + w] = packed_left[(size_t)(base + st_left[base + i]) * W + w];
  }
}

// --------------------------------------------------------------------------
// Temporal stage: sorted matches -> 3D-2D correspondences of each pair.
// --------------------------------------------------------------------------
// Pair p matches the rows [qbase[p], +qcnt[p]) (query side: 3-D points) against
// the rows [tbase[p], +tcnt[p]) (train side: pixels); its outputs live at
// obase[p] of the pair-output arrays.
template <typename PIX>
__global__ __launch_bounds__(kBlock) void temporal_gather_kernel(
    const int32_t *__restrict__ qbase, const int32_t *__restrict__ qcnt,
    const int32_t *__restrict__ tbase, const int32_t *__restrict__ tcnt,
    const int32_t *__restrict__ obase, const int32_t *__restrict__ tidx2,
    const int32_t *__restrict__ tdist2, const int32_t *__restrict__ order2,
    const float *__restrict__ xyz, const PIX *__restrict__ pix_left,
    const PIX *__restrict__ pix_right, int32_t *__restrict__ tm_query,
    int32_t *__restrict__ tm_train, int32_t *__restrict__ tm_dist, int32_t *__restrict__ tm_count,
    float *__restrict__ cx, float *__restrict__ cpl, float *__restrict__ cpr) {
  const int p = blockIdx.y;
  const int qb = qbase[p], tb = tbase[p], ob = obase[p];
  const int n = tcnt[p] > 0 ? qcnt[p] : 0;  // no train rows -> no matches
  if (blockIdx.x == 0 && threadIdx.x == 0) tm_count[p] = n;
  const int r = blockIdx.x * kBlock + (int)threadIdx.x;
  if (r >= n) return;
  const int q = order2[ob + r];
  const int t = tidx2[2 * (size_t)(ob + q)];
  const size_t o = (size_t)(ob + r);
  tm_query[o] = q;
  tm_train[o] = t;
  tm_dist[o] = tdist2[2 * (size_t)(ob + q)];
#pragma unroll
  for (int c = 0; c < 3; ++c) cx[3 * o + c] = xyz[3 * (size_t)(qb + q) + c];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    cpl[2 * o + c] = (float)pix_left[2 * (size_t)(tb + t) + c];
    cpr[2 * o + c] = (float)pix_right[2 * (size_t)(tb + t) + c];
  }
}

// --------------------------------------------------------------------------
// Adaptive RANSAC, one workgroup per pair.  256 hypotheses are sampled, solved
// (P3P) and scored against every point per round; lane 0 then replays the
// sequential stopping rule of pnp.py:123-158 over the round's ordered counts,
// so the outcome equals the one-hypothesis-at-a-time loop on the same samples.
// --------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

struct RansacParams {
  RansacCam cam;
  Kinv kinv;
  double thresh, p_success;
  int n_iter_max;
  int min_points;  // pairs with fewer correspondences are not attempted
  unsigned long long seed;
};

// Consensus count of one lane's pose against all n correspondences of the pair
// (both cameras).  Every lane of the block calls it: the points are staged
// through LDS chunk by chunk with block barriers.  Kept out of line so that its
// registers are allocated apart from the minimal solver's.
__device__ __noinline__ int score_round(bool valid, const double *rt_all, const RansacCam *cam,
                                        double *lds, const float *__restrict__ X,
                                        const float *__restrict__ mL,
                                        const float *__restrict__ mR, int n, double thresh) {
  double PL[12], PR[12];
  if (valid) {
    double Rt[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Rt[i] = rt_all[i * kBlock + threadIdx.x];
    make_projections(Rt, *cam, PL, PR);
  } else {
#pragma unroll
    for (int i = 0; i < 12; ++i) PL[i] = PR[i] = 0.0;
  }
  const double thr2 = thresh * thresh;
  int c = 0;
  for (int p0 = 0; p0 < n; p0 += kPtsPerChunk) {
    const int cnt = min(kPtsPerChunk, n - p0);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += kBlock) {
      const size_t g = (size_t)(p0 + i);
      double *o = lds + i * kPtStride;
      o[0] = (double)X[3 * g]; o[1] = (double)X[3 * g + 1]; o[2] = (double)X[3 * g + 2];
      o[3] = (double)mL[2 * g]; o[4] = (double)mL[2 * g + 1];
      o[5] = (double)mR[2 * g]; o[6] = (double)mR[2 * g + 1];
      o[7] = 0.0;
    }
    __syncthreads();
    if (valid) c += count_chunk_inliers(lds, cnt, PL, PR, thresh, thr2);
  }
  return c;
}

__global__ __launch_bounds__(kBlock) void ransac_pairs_kernel(
    const int32_t *__restrict__ obase, const int32_t *__restrict__ tm_count,
    const float *__restrict__ cx, const float *__restrict__ cpl, const float *__restrict__ cpr,
    const int32_t *__restrict__ samples, RansacParams prm, double *__restrict__ pose_ransac,
    uint8_t *__restrict__ tm_mask, int32_t *__restrict__ info) {
  __shared__ __align__(16) double lds[kPtsPerChunk * kPtStride];
  __shared__ int counts[kBlock];
  __shared__ double best_rt[12];
  __shared__ double rt_all[12 * kBlock];  // this round's poses, element-major: [i * kBlock + lane]
  __shared__ RansacCam cam_lds;
  __shared__ int st[8];  // best_count, best_idx, iter_done, n_req, stop, winner_lane, domain, mask_count
  const int p = blockIdx.x;
  const int base = obase[p];
  const int n = tm_count[p];
  const float *X = cx + 3 * (size_t)base;
  const float *mL = cpl + 2 * (size_t)base;
  const float *mR = cpr + 2 * (size_t)base;
  if (threadIdx.x == 0) {
    st[0] = 0; st[1] = -1; st[2] = 0; st[3] = prm.n_iter_max; st[4] = 0; st[5] = -1; st[6] = 0; st[7] = 0;
  }
  if (threadIdx.x < 12) best_rt[threadIdx.x] = (threadIdx.x % 5 == 0) ? 1.0 : 0.0;  // [I|0]
  if (threadIdx.x < 9) cam_lds.K[threadIdx.x] = prm.cam.K[threadIdx.x];
  if (threadIdx.x < 12) cam_lds.MR[threadIdx.x] = prm.cam.MR[threadIdx.x];
  __syncthreads();
  const double thr2 = prm.thresh * prm.thresh;
  const double log1mp = log(1.0 - prm.p_success);

  if (n >= 4 && n >= prm.min_points) {
    for (int round = 0;; ++round) {
      const int it = round * kBlock + (int)threadIdx.x;
      const bool active = it < prm.n_iter_max;
      bool valid = false;
      if (active) {
        double Rt[12];
        int s4[4];
        if (samples != nullptr) {
          const int32_t *sp = samples + ((size_t)p * prm.n_iter_max + it) * 4;
          for (int k = 0; k < 4; ++k) s4[k] = min(max(sp[k], 0), n - 1);
        } else {
          unsigned long long s = splitmix64(prm.seed ^ ((unsigned long long)p << 32) ^ (unsigned long long)it);
          for (int k = 0; k < 4; ++k) {
            int c = 0;
            for (int tries = 0; tries < 64; ++tries) {
              s = splitmix64(s);
              c = (int)(((s >> 32) * (unsigned long long)n) >> 32);
              bool dup = false;
              for (int j = 0; j < k; ++j) dup |= (s4[j] == c);
              if (!dup) break;
            }
            s4[k] = c;
          }
        }
        double Xs[4][3], f[4][3], nrm4[2];
        for (int k = 0; k < 4; ++k) {
          for (int c = 0; c < 3; ++c) Xs[k][c] = (double)X[3 * (size_t)s4[k] + c];
          double nk[2];
          pixel_to_bearing(prm.kinv, (double)mL[2 * (size_t)s4[k]], (double)mL[2 * (size_t)s4[k] + 1],
                           f[k], nk);
          if (k == 3) { nrm4[0] = nk[0]; nrm4[1] = nk[1]; }
        }
        valid = p3p_lane(Xs, f, nrm4, Rt);
        if (valid)
          for (int i = 0; i < 12; ++i) rt_all[i * kBlock + threadIdx.x] = Rt[i];
      }
      const int c = score_round(valid, rt_all, &cam_lds, lds, X, mL, mR, n, prm.thresh);
      counts[threadIdx.x] = active ? (valid ? c : -1) : -2;
      __syncthreads();
      if (threadIdx.x == 0) {
        int best_count = st[0], best_idx = st[1], iter_done = st[2], n_req = st[3];
        int winner = -1;
        while (iter_done < n_req && iter_done < prm.n_iter_max && iter_done < (round + 1) * kBlock) {
          const int cc = counts[iter_done - round * kBlock];
          const int idx = iter_done;
          ++iter_done;
          if (cc < 0) continue;
          if (cc > best_count) {
            best_count = cc;
            best_idx = idx;
            winner = idx - round * kBlock;
            const double w = (double)best_count / (double)n;
            const double w2 = w * w;
            const double den = log(1.0 - w2 * w2);
            if (best_count >= n) st[6] = 1;
            const double r = ceil(log1mp / den);
            n_req = r > 2147483647.0 ? 2147483647 : (int)r;
          }
        }
        st[0] = best_count; st[1] = best_idx; st[2] = iter_done; st[3] = n_req;
        st[5] = winner;
        st[4] = (iter_done < n_req && iter_done < prm.n_iter_max) ? 0 : 1;
      }
      __syncthreads();
      if (threadIdx.x < 12 && st[5] >= 0) best_rt[threadIdx.x] = rt_all[threadIdx.x * kBlock + st[5]];
      __syncthreads();
      if (st[4]) break;
    }
  }

  // winner's mask and its count
  {
    double Rt[12], PL[12], PR[12];
    for (int i = 0; i < 12; ++i) Rt[i] = best_rt[i];
    make_projections(Rt, prm.cam, PL, PR);
    const bool have = st[1] >= 0;
    for (int i = threadIdx.x; i < n; i += kBlock) {
      bool in = false;
      if (have) {
        const double Xd = (double)X[3 * (size_t)i], Yd = (double)X[3 * (size_t)i + 1],
                     Zd = (double)X[3 * (size_t)i + 2];
        in = inlier_one_camera(PL, Xd, Yd, Zd, (double)mL[2 * (size_t)i], (double)mL[2 * (size_t)i + 1],
                               prm.thresh, thr2) &&
             inlier_one_camera(PR, Xd, Yd, Zd, (double)mR[2 * (size_t)i], (double)mR[2 * (size_t)i + 1],
                               prm.thresh, thr2);
      }
      tm_mask[base + i] = in ? 1 : 0;
      if (in) atomicAdd(&st[7], 1);
    }
  }
  __syncthreads();
  if (threadIdx.x < 12) pose_ransac[12 * (size_t)p + threadIdx.x] = best_rt[threadIdx.x];
  if (threadIdx.x == 0) {
    int32_t *o = info + 8 * (size_t)p;
    o[0] = st[1]; o[1] = st[0]; o[2] = st[2]; o[3] = st[3]; o[4] = st[6]; o[5] = n; o[6] = st[7]; o[7] = 0;
  }
}

struct Layout {
  int W;
  size_t idx2, dist2, order, packL, packR, knn, tidx2, tdist2, order2, cx, cpl, cpr, tcount, total;
  size_t knn_bytes;
};

Layout make_layout(const slamhip_frontend_args *a) {
  Layout L;
  L.W = slamhip_packed_words(a->desc_bytes);
  const size_t NL = (size_t)a->n_left_total, NR = (size_t)a->n_right_total;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = slamhip_align_up(o + bytes, 256); return at; };
  L.idx2 = take(NL * 8); L.dist2 = take(NL * 8); L.order = take(NL * 4);
  L.packL = take(NL * L.W * 4); L.packR = take(NR * L.W * 4);
  const int F = a->n_frames > 0 ? a->n_frames : 1;
  const size_t k1 = slamhip_internal::knn2_workspace_bytes(a->n_left_total, a->n_right_total, F,
                                                           a->max_left, a->max_right, a->desc_bytes);
  const size_t k2 = slamhip_internal::knn2_workspace_bytes(a->n_left_total, a->n_left_total,
                                                           F > 1 ? F - 1 : 1, a->max_left, a->max_left,
                                                           a->desc_bytes);
  L.knn_bytes = k1 > k2 ? k1 : k2;
  L.knn = take(L.knn_bytes);
  L.tidx2 = take(NL * 8); L.tdist2 = take(NL * 8); L.order2 = take(NL * 4);
  L.cx = take(NL * 12); L.cpl = take(NL * 8); L.cpr = take(NL * 8);
  L.tcount = take((size_t)F * 4);
  L.total = o + 256;
  return L;
}

bool invert3x3(const double *K, double *o) {
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

extern "C" {

size_t slamhip_frontend_workspace_bytes(const slamhip_frontend_args *a) {
  if (!a || a->n_frames < 1 || a->n_left_total < 0 || a->n_right_total < 0 || a->desc_bytes < 1 ||
      a->desc_bytes > SLAMHIP_MAX_DESC_BYTES)
    return 0;
  return make_layout(a).total;
}

int slamhip_frontend_batch(const slamhip_frontend_args *a, void *stream) {
  SLAMHIP_BEGIN();
  if (!a || a->n_frames < 1 || a->n_left_total < 0 || a->n_right_total < 0) return SLAMHIP_EINVAL;
  if (a->desc_bytes < 1 || a->desc_stride < a->desc_bytes) return SLAMHIP_EINVAL;
  if (a->desc_bytes > SLAMHIP_MAX_DESC_BYTES || a->n_frames > 65535 ||
      a->max_left > SLAMHIP_MAX_TRAIN_ROWS || a->max_right > SLAMHIP_MAX_TRAIN_ROWS)
    return SLAMHIP_EUNSUPPORTED;
  if (a->max_left < 0 || a->max_right < 0 || a->n_iter_max < 1 || a->refine_iters < 0)
    return SLAMHIP_EINVAL;
  if (!a->d_off_left || !a->d_off_right || !a->d_st_count) return SLAMHIP_EINVAL;
  if (a->n_left_total > 0 &&
      (!a->d_kp_left || !a->d_des_left || !a->d_st_left || !a->d_st_right || !a->d_st_dist ||
       !a->d_st_xyz || !a->d_st_pix_left || !a->d_st_pix_right || !a->d_st_desc || !a->d_tm_query ||
       !a->d_tm_train || !a->d_tm_dist || !a->d_tm_mask))
    return SLAMHIP_EINVAL;
  if (a->n_right_total > 0 && (!a->d_kp_right || !a->d_des_right)) return SLAMHIP_EINVAL;
  if (a->n_frames > 1 && (!a->d_pose_ransac || !a->d_pose_refined || !a->d_ransac_info))
    return SLAMHIP_EINVAL;
  const Layout L = make_layout(a);
  if (!a->d_workspace || a->workspace_bytes < L.total) return SLAMHIP_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const uintptr_t base = ((uintptr_t)a->d_workspace + 255) & ~(uintptr_t)255;
  int32_t *idx2 = (int32_t *)(base + L.idx2), *dist2 = (int32_t *)(base + L.dist2);
  int32_t *order = (int32_t *)(base + L.order);
  uint32_t *packL = (uint32_t *)(base + L.packL), *packR = (uint32_t *)(base + L.packR);
  void *knn_ws = (void *)(base + L.knn);
  int32_t *tidx2 = (int32_t *)(base + L.tidx2), *tdist2 = (int32_t *)(base + L.tdist2);
  int32_t *order2 = (int32_t *)(base + L.order2);
  float *cx = (float *)(base + L.cx), *cpl = (float *)(base + L.cpl), *cpr = (float *)(base + L.cpr);
  int32_t *tcount = (int32_t *)(base + L.tcount);
  const int F = a->n_frames, W = L.W;
  int rc;

  if (a->n_left_total == 0 || a->max_left == 0) {
    if (hipMemsetAsync(a->d_st_count, 0, sizeof(int32_t) * F, st) != hipSuccess) return SLAMHIP_ELAUNCH;
    if (F > 1) {
      if (hipMemsetAsync(a->d_ransac_info, 0, sizeof(int32_t) * 8 * (F - 1), st) != hipSuccess)
        return SLAMHIP_ELAUNCH;
    }
    return SLAMHIP_OK;
  }

  // ---- stereo stage
  rc = slamhip_internal::pack_rows(a->d_des_left, a->n_left_total, packL, a->d_des_right,
                                   a->n_right_total, packR, a->desc_bytes, a->desc_stride, st);
  if (rc) return rc;
  rc = slamhip_internal::knn2((const uint8_t *)packL, a->n_left_total, (const uint8_t *)packR,
                              a->n_right_total, a->desc_bytes, W * 4, true, a->d_off_left, nullptr,
                              a->d_off_right, nullptr, F, a->max_left, a->max_right, idx2, dist2,
                              knn_ws, L.knn_bytes, st);
  if (rc) return rc;
  rc = slamhip_internal::sort_by_distance(dist2, 2, a->d_off_left, nullptr, F, a->max_left, order, st);
  if (rc) return rc;
  CamPair cam;
  for (int i = 0; i < 12; ++i) { cam.PL[i] = a->P_left[i]; cam.PR[i] = a->P_right[i]; }
  hipLaunchKernelGGL(stereo_compact_kernel, dim3(F), dim3(kBlock), 0, st, a->d_kp_left,
                     a->d_kp_right, a->d_off_left, a->d_off_right, idx2, dist2, order, packL, W, cam,
                     a->pixel_thresh, a->disparity_min, a->d_st_count, a->d_st_left, a->d_st_right,
                     a->d_st_dist, a->d_st_xyz, a->d_st_pix_left, a->d_st_pix_right, a->d_st_desc);
  SLAMHIP_CHECK_LAUNCH();
  if (F < 2) return SLAMHIP_OK;

  // ---- temporal stage: pair p = frames (p, p+1)
  rc = slamhip_internal::knn2((const uint8_t *)a->d_st_desc, a->n_left_total,
                              (const uint8_t *)a->d_st_desc, a->n_left_total, a->desc_bytes, W * 4,
                              true, a->d_off_left, a->d_st_count, a->d_off_left + 1,
                              a->d_st_count + 1, F - 1, a->max_left, a->max_left, tidx2, tdist2, knn_ws,
                              L.knn_bytes, st);
  if (rc) return rc;
  rc = slamhip_internal::sort_by_distance(tdist2, 2, a->d_off_left, a->d_st_count, F - 1, a->max_left,
                                          order2, st);
  if (rc) return rc;
  hipLaunchKernelGGL(temporal_gather_kernel<float>,
                     dim3(slamhip_ceil_div(a->max_left, kBlock), F - 1), dim3(kBlock), 0, st,
                     a->d_off_left, a->d_st_count, a->d_off_left + 1, a->d_st_count + 1, a->d_off_left,
                     tidx2, tdist2, order2, a->d_st_xyz, a->d_st_pix_left, a->d_st_pix_right,
                     a->d_tm_query, a->d_tm_train, a->d_tm_dist, tcount, cx, cpl, cpr);
  SLAMHIP_CHECK_LAUNCH();

  RansacParams prm;
  for (int i = 0; i < 9; ++i) prm.cam.K[i] = a->K[i];
  for (int i = 0; i < 12; ++i) prm.cam.MR[i] = a->M_right[i];
  if (!invert3x3(a->K, prm.kinv.m)) return SLAMHIP_EINVAL;
  prm.thresh = a->ransac_thresh;
  prm.p_success = a->p_success;
  prm.n_iter_max = a->n_iter_max;
  prm.min_points = 4;
  prm.seed = a->seed;
  hipLaunchKernelGGL(ransac_pairs_kernel, dim3(F - 1), dim3(kBlock), 0, st, a->d_off_left, tcount, cx,
                     cpl, cpr, a->d_samples, prm, a->d_pose_ransac, a->d_tm_mask, a->d_ransac_info);
  SLAMHIP_CHECK_LAUNCH();

  rc = slamhip_internal::refine_pose(cx, 0, cpl, a->d_tm_mask, a->d_off_left, tcount, F - 1, a->K,
                                     a->d_pose_ransac, a->refine_iters, a->d_pose_refined, nullptr, st);
  return rc;
}

// ---------------------------------------------------------------------------
// Pair verification (loop-closure candidates).
// ---------------------------------------------------------------------------
size_t slamhip_verify_workspace_bytes(const slamhip_verify_args *a) {
  if (!a || a->n_rows_total < 0 || a->n_pairs < 0 || a->n_out_total < 0 || a->desc_bytes < 1 ||
      a->desc_bytes > SLAMHIP_MAX_DESC_BYTES)
    return 0;
  const int W = slamhip_packed_words(a->desc_bytes);
  const size_t N = (size_t)a->n_rows_total, O = (size_t)a->n_out_total;
  const int chunk = a->n_pairs < 32768 ? (a->n_pairs > 0 ? a->n_pairs : 1) : 32768;
  size_t o = 0;
  auto take = [&](size_t bytes) { o = slamhip_align_up(o + bytes, 256); };
  take(N * W * 4);            // packed descriptors
  take(O * 8); take(O * 8);   // idx2, dist2
  take(O * 4);                // order
  take(O * 12); take(O * 8); take(O * 8);  // cx, cpl, cpr
  take((size_t)(a->n_pairs > 0 ? a->n_pairs : 1) * 4);  // counts
  take(slamhip_internal::knn2_workspace_bytes(a->n_rows_total, a->n_rows_total, chunk, a->max_q,
                                              a->max_t, a->desc_bytes, (long long)a->n_out_total,
                                              true));
  return o + 256;
}

int slamhip_verify_pairs(const slamhip_verify_args *a, void *stream) {
  SLAMHIP_BEGIN();
  if (!a || a->n_rows_total < 0 || a->n_pairs < 0 || a->n_out_total < 0) return SLAMHIP_EINVAL;
  if (a->desc_bytes < 1 || a->desc_stride < a->desc_bytes || a->n_iter_max < 1 ||
      a->refine_iters < 0 || a->max_q < 0 || a->max_t < 0)
    return SLAMHIP_EINVAL;
  if (a->desc_bytes > SLAMHIP_MAX_DESC_BYTES || a->max_q > SLAMHIP_MAX_TRAIN_ROWS ||
      a->max_t > SLAMHIP_MAX_TRAIN_ROWS)
    return SLAMHIP_EUNSUPPORTED;
  if (a->n_rows_total == 0) return SLAMHIP_OK;
  if (!a->d_kp_left || !a->d_kp_right || !a->d_des || !a->d_xyz) return SLAMHIP_EINVAL;
  const size_t need = slamhip_verify_workspace_bytes(a);
  if (!a->d_workspace || a->workspace_bytes < need) return SLAMHIP_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int W = slamhip_packed_words(a->desc_bytes);
  const size_t N = (size_t)a->n_rows_total, O = (size_t)a->n_out_total;
  uintptr_t cur = ((uintptr_t)a->d_workspace + 255) & ~(uintptr_t)255;
  auto take = [&](size_t bytes) { const uintptr_t at = cur; cur = slamhip_align_up(cur + bytes, 256); return at; };
  uint32_t *packed = (uint32_t *)take(N * W * 4);
  int32_t *idx2 = (int32_t *)take(O * 8), *dist2 = (int32_t *)take(O * 8);
  int32_t *order = (int32_t *)take(O * 4);
  float *cx = (float *)take(O * 12), *cpl = (float *)take(O * 8), *cpr = (float *)take(O * 8);
  int32_t *tcount = (int32_t *)take((size_t)(a->n_pairs > 0 ? a->n_pairs : 1) * 4);
  const int chunk_max = a->n_pairs < 32768 ? (a->n_pairs > 0 ? a->n_pairs : 1) : 32768;
  const size_t knn_bytes = slamhip_internal::knn2_workspace_bytes(
      a->n_rows_total, a->n_rows_total, chunk_max, a->max_q, a->max_t, a->desc_bytes,
      (long long)a->n_out_total, true);
  void *knn_ws = (void *)take(knn_bytes);

  // per keyframe: points of all stored features (loop_closure.py:179-180,237-249: fp64
  // triangulation, later cast to float32 at :209) and packed descriptors
  int rc = slamhip_stereo_filter_triangulate(a->d_kp_left, a->d_kp_right, 1, nullptr, nullptr,
                                             a->n_rows_total, a->P_left, a->P_right, 0.0, 0.0,
                                             nullptr, a->d_xyz, 2, stream);
  if (rc) return rc;
  rc = slamhip_internal::pack_rows(a->d_des, a->n_rows_total, packed, nullptr, 0, nullptr,
                                   a->desc_bytes, a->desc_stride, st);
  if (rc) return rc;
  if (a->n_pairs == 0 || a->n_out_total == 0) return SLAMHIP_OK;
  if (!a->d_qbase || !a->d_qcnt || !a->d_tbase || !a->d_tcnt || !a->d_obase || !a->d_tm_query ||
      !a->d_tm_train || !a->d_tm_dist || !a->d_tm_mask || !a->d_pose_ransac || !a->d_pose_refined ||
      !a->d_info)
    return SLAMHIP_EINVAL;

  RansacParams prm;
  for (int i = 0; i < 9; ++i) prm.cam.K[i] = a->K[i];
  for (int i = 0; i < 12; ++i) prm.cam.MR[i] = a->M_right[i];
  if (!invert3x3(a->K, prm.kinv.m)) return SLAMHIP_EINVAL;
  prm.thresh = a->ransac_thresh;
  prm.p_success = a->p_success;
  prm.n_iter_max = a->n_iter_max;
  prm.min_points = a->min_points;

  for (int c0 = 0; c0 < a->n_pairs; c0 += chunk_max) {
    const int np = a->n_pairs - c0 < chunk_max ? a->n_pairs - c0 : chunk_max;
    const int32_t *qb = a->d_qbase + c0, *qc = a->d_qcnt + c0, *tb = a->d_tbase + c0,
                  *tc = a->d_tcnt + c0, *ob = a->d_obase + c0;
    rc = slamhip_internal::knn2((const uint8_t *)packed, a->n_rows_total, (const uint8_t *)packed,
                                a->n_rows_total, a->desc_bytes, W * 4, true, qb, qc, tb, tc, np,
                                a->max_q, a->max_t, idx2, dist2, knn_ws, knn_bytes, st, ob,
                                a->n_out_total);
    if (rc) return rc;
    rc = slamhip_internal::sort_by_distance(dist2, 2, ob, qc, np, a->max_q, order, st);
    if (rc) return rc;
    hipLaunchKernelGGL(temporal_gather_kernel<double>, dim3(slamhip_ceil_div(a->max_q, kBlock), np),
                       dim3(kBlock), 0, st, qb, qc, tb, tc, ob, idx2, dist2, order,
                       (const float *)a->d_xyz, a->d_kp_left, a->d_kp_right, a->d_tm_query,
                       a->d_tm_train, a->d_tm_dist, tcount + c0, cx, cpl, cpr);
    SLAMHIP_CHECK_LAUNCH();
    prm.seed = a->seed + (unsigned long long)c0 * 0x9E3779B97F4A7C15ull;
    hipLaunchKernelGGL(ransac_pairs_kernel, dim3(np), dim3(kBlock), 0, st, ob, tcount + c0, cx, cpl,
                       cpr, (const int32_t *)nullptr, prm, a->d_pose_ransac + 12 * (size_t)c0,
                       a->d_tm_mask, a->d_info + 8 * (size_t)c0);
    SLAMHIP_CHECK_LAUNCH();
    rc = slamhip_internal::refine_pose(cx, 0, cpl, a->d_tm_mask, ob, tcount + c0, np, a->K,
                                       a->d_pose_ransac + 12 * (size_t)c0, a->refine_iters,
                                       a->d_pose_refined + 12 * (size_t)c0, nullptr, st);
    if (rc) return rc;
  }
  return SLAMHIP_OK;
}

}  // extern "C"
