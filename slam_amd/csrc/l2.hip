// Brute-force L2 top-2 matcher for float32 descriptors (gfx950).
//
// The reference's DescriptorMatcher picks cv2.NORM_L2 whenever the descriptors are
// not uint8 (frontend/vision/descriptor_matcher.py:23-28), which is what SuperPoint's
// 256-float descriptors get with matcher_type="bf".  This is that branch.
//
// Distance, as OpenCV's batchDistance defines it for NORM_L2 on float32: the square
// root of the sum of squared differences, a float32.  The sum is accumulated here in
// float32 in index order, every product and every addition rounded separately (no
// fused multiply-add), and the square root is correctly rounded, so the result is
// reproducible by a plain float32 loop on any machine.  (OpenCV's own summation order
// inside its SIMD kernel is not specified; distances can differ from it in the last
// bits.)  Ties go to the lowest train index, like the Hamming matcher.
//
// Block = 256 threads computing a 64 x 64 tile of (query, train) distances, 4 x 4 per
// thread, with 32 dimensions of both tiles staged in LDS per step (dimension-major, so
// that the four rows a thread needs are one 16-byte read: broadcast for the query
// side, consecutive for the train side).  Per dimension: 8 floats from LDS, 16 x
// (subtract, multiply, add).  The per-thread candidates of a query row meet in LDS;
// 64 threads keep the running (best, second) per query as 64-bit keys
// (distance bits << 32 | train index: non-negative floats order like their bits).
// grid = (query tiles, train splits); a second kernel folds the splits.
#include "common.h"

namespace {

constexpr int kTQ = 64, kTT = 64, kTK = 32;
constexpr unsigned long long kNone = ~0ull;
constexpr int kMaxL2Splits = 64;

__device__ __forceinline__ void push2(unsigned long long &b1, unsigned long long &b2,
                                      unsigned long long key) {
  if (key < b1) { b2 = b1; b1 = key; }
  else if (key < b2) { b2 = key; }
}

__global__ __launch_bounds__(256) void l2_top2_kernel(const float *__restrict__ q, int nq,
                                                      const float *__restrict__ t, int nt, int dim,
                                                      int stride, ulonglong2 *__restrict__ partial,
                                                      int32_t *__restrict__ idx2,
                                                      float *__restrict__ dist2) {
  __shared__ __align__(16) float qs[kTK][kTQ];
  __shared__ __align__(16) float ts[kTK][kTT];
  __shared__ ulonglong2 cand[kTQ][16];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int q0 = blockIdx.x * kTQ;
  const int n_splits = gridDim.y;
  const int tiles = (nt + kTT - 1) / kTT;
  const int tiles_per_split = (tiles + n_splits - 1) / n_splits;
  const int tile_begin = blockIdx.y * tiles_per_split;
  const int tile_end = min(tiles, tile_begin + tiles_per_split);
  unsigned long long best1 = kNone, best2 = kNone;     // threads 0..63: query row q0 + tid

  // staging role: row tid / 4 of the tile, dimensions (tid % 4) * 8 .. + 7 of the step
  const int lrow = tid >> 2, lk = (tid & 3) * 8;
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int t0 = tile * kTT;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
    for (int k0 = 0; k0 < dim; k0 += kTK) {
      __syncthreads();
      const int qr = q0 + lrow, tr = t0 + lrow;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int kk = k0 + lk + k;
        qs[lk + k][lrow] = (qr < nq && kk < dim) ? q[(size_t)qr * stride + kk] : 0.0f;
        ts[lk + k][lrow] = (tr < nt && kk < dim) ? t[(size_t)tr * stride + kk] : 0.0f;
      }
      __syncthreads();
      // dimensions past `dim` hold zeros on both sides: they add +0.0 and change nothing
#pragma unroll 8
      for (int k = 0; k < kTK; ++k) {
        const float4 a = *reinterpret_cast<const float4 *>(&qs[k][ty * 4]);
        const float4 b = *reinterpret_cast<const float4 *>(&ts[k][tx * 4]);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // no fused multiply-add: the product is rounded before it is added (hipcc
            // contracts a * b + c by default, also through __fmul_rn / __fadd_rn)
#pragma clang fp contract(off)
            const float d = av[i] - bv[j];
            const float sq = d * d;
            acc[i][j] = acc[i][j] + sq;
          }
      }
    }
    // this thread's four train columns for each of its four query rows, in index order
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned long long c1 = kNone, c2 = kNone;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = t0 + tx * 4 + j;
        if (col < nt) {
          // double square root rounded to float = the correctly rounded float square root
          const float dist = (float)__dsqrt_rn((double)acc[i][j]);
          push2(c1, c2, ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned)col);
        }
      }
      cand[ty * 4 + i][tx] = make_ulonglong2(c1, c2);
    }
    __syncthreads();
    if (tid < kTQ) {
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const ulonglong2 c = cand[tid][x];
        push2(best1, best2, c.x);
        push2(best1, best2, c.y);
      }
    }
  }
  if (tid < kTQ && q0 + tid < nq) {
    const int row = q0 + tid;
    if (n_splits > 1) {
      partial[(size_t)blockIdx.y * nq + row] = make_ulonglong2(best1, best2);
    } else {
      idx2[2 * row + 0] = best1 == kNone ? -1 : (int32_t)(best1 & 0xFFFFFFFFull);
      idx2[2 * row + 1] = best2 == kNone ? -1 : (int32_t)(best2 & 0xFFFFFFFFull);
      dist2[2 * row + 0] = best1 == kNone ? -1.0f : __uint_as_float((unsigned)(best1 >> 32));
      dist2[2 * row + 1] = best2 == kNone ? -1.0f : __uint_as_float((unsigned)(best2 >> 32));
    }
  }
}

__global__ __launch_bounds__(256) void l2_merge_kernel(const ulonglong2 *__restrict__ partial, int nq,
                                                       int n_splits, int32_t *__restrict__ idx2,
                                                       float *__restrict__ dist2) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nq) return;
  unsigned long long b1 = kNone, b2 = kNone;
  for (int s = 0; s < n_splits; ++s) {
    const ulonglong2 p = partial[(size_t)s * nq + row];
    push2(b1, b2, p.x);
    push2(b1, b2, p.y);
  }
  idx2[2 * row + 0] = b1 == kNone ? -1 : (int32_t)(b1 & 0xFFFFFFFFull);
  idx2[2 * row + 1] = b2 == kNone ? -1 : (int32_t)(b2 & 0xFFFFFFFFull);
  dist2[2 * row + 0] = b1 == kNone ? -1.0f : __uint_as_float((unsigned)(b1 >> 32));
  dist2[2 * row + 1] = b2 == kNone ? -1.0f : __uint_as_float((unsigned)(b2 >> 32));
}

// crossCheck=True fold for float distances: one thread per query row scans the
// train -> query result in ascending train order and keeps the closest train row that
// chose it, the first one on ties (the fold of cv2's batchDistance).
__global__ __launch_bounds__(256) void l2_cross_check_kernel(const int32_t *__restrict__ t2q_idx2,
                                                             const float *__restrict__ t2q_dist2,
                                                             int nt, int nq,
                                                             int32_t *__restrict__ match_idx,
                                                             float *__restrict__ match_dist) {
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= nq) return;
  int best_t = -1;
  float best_d = 0.0f;
  for (int t = 0; t < nt; ++t) {
    const int who = t2q_idx2[2 * t];  // wave-uniform address
    const float d = t2q_dist2[2 * t];
    if (who == qi && (best_t < 0 || d < best_d)) {
      best_d = d;
      best_t = t;
    }
  }
  match_idx[qi] = best_t;
  match_dist[qi] = best_t < 0 ? -1.0f : best_d;
}

__global__ __launch_bounds__(256) void l2_ratio_test_kernel(const float *__restrict__ dist2, int nq,
                                                            double ratio, uint8_t *__restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const float2 d = reinterpret_cast<const float2 *>(dist2)[i];
  keep[i] = (d.x >= 0.0f && d.y >= 0.0f && (double)d.x < ratio * (double)d.y) ? 1 : 0;
}

// Stable order by (distance, row) through rank counting on the keys (distance bits,
// row): non-negative floats order like their bits, and -1.0 (no match) has a larger bit
// pattern than any of them, so unmatched rows sort last.  One segment of n rows.
__global__ __launch_bounds__(256) void l2_rank_sort_kernel(const float *__restrict__ dist, int stride,
                                                           int n, int32_t *__restrict__ order) {
  __shared__ uint32_t keys[1024];
  const int i = blockIdx.x * 256 + (int)threadIdx.x;
  const uint32_t my = i < n ? __float_as_uint(dist[(size_t)i * stride]) : 0xFFFFFFFFu;
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 1024) {
    __syncthreads();
    for (int k = threadIdx.x; k < 1024; k += 256)
      keys[k] = j0 + k < n ? __float_as_uint(dist[(size_t)(j0 + k) * stride]) : 0xFFFFFFFFu;
    __syncthreads();
    const int lim = min(1024, n - j0);
    for (int k = 0; k < lim; ++k) {
      const uint32_t o = keys[k];
      rank += (o < my || (o == my && j0 + k < i)) ? 1 : 0;
    }
  }
  if (i < n) order[rank] = i;
}

int l2_splits(int nq, int nt) {
  const int qtiles = slamhip_ceil_div(nq > 0 ? nq : 1, kTQ);
  const int ttiles = slamhip_ceil_div(nt > 0 ? nt : 1, kTT);
  int s = slamhip_ceil_div(1024, qtiles);       // about four blocks per CU
  if (s > ttiles) s = ttiles;
  if (s > kMaxL2Splits) s = kMaxL2Splits;
  return s < 1 ? 1 : s;
}

}  // namespace

extern "C" {

size_t slamhip_l2_knn2_workspace_bytes(int nq, int nt, int dim) {
  if (nq < 0 || nt < 0 || dim < 1 || dim > SLAMHIP_MAX_L2_DIM) return 0;
  const int s = l2_splits(nq, nt);
  return (s > 1 ? (size_t)s * (size_t)nq * sizeof(ulonglong2) : 0) + 512;
}

int slamhip_l2_knn2(const float *d_query, int nq, const float *d_train, int nt, int dim,
                    int row_stride, int32_t *d_idx2, float *d_dist2, void *d_workspace,
                    size_t workspace_bytes, void *stream) {
  SLAMHIP_BEGIN();
  if (nq < 0 || nt < 0 || dim < 1 || row_stride < dim) return SLAMHIP_EINVAL;
  if (dim > SLAMHIP_MAX_L2_DIM || nt > SLAMHIP_MAX_TRAIN_ROWS || nq > SLAMHIP_MAX_TRAIN_ROWS)
    return SLAMHIP_EUNSUPPORTED;
  if (nq == 0) return SLAMHIP_OK;
  if (!d_query || !d_idx2 || !d_dist2 || (nt > 0 && !d_train)) return SLAMHIP_EINVAL;
  const int s = l2_splits(nq, nt);
  const size_t need = slamhip_l2_knn2_workspace_bytes(nq, nt, dim);
  if (s > 1 && (!d_workspace || workspace_bytes < need)) return SLAMHIP_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  ulonglong2 *partial = (ulonglong2 *)(((uintptr_t)d_workspace + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(l2_top2_kernel, dim3(slamhip_ceil_div(nq, kTQ), s), dim3(256), 0, st, d_query, nq,
                     d_train, nt, dim, row_stride, partial, d_idx2, d_dist2);
  SLAMHIP_CHECK_LAUNCH();
  if (s > 1) {
    hipLaunchKernelGGL(l2_merge_kernel, dim3(slamhip_ceil_div(nq, 256)), dim3(256), 0, st, partial, nq, s,
                       d_idx2, d_dist2);
    SLAMHIP_CHECK_LAUNCH();
  }
  return SLAMHIP_OK;
}

int slamhip_l2_cross_check(const int32_t *d_t2q_idx2, const float *d_t2q_dist2, int nt, int nq,
                           int32_t *d_match_idx, float *d_match_dist, void *stream) {
  SLAMHIP_BEGIN();
  if (nt < 0 || nq < 0) return SLAMHIP_EINVAL;
  if (nq == 0) return SLAMHIP_OK;
  if (!d_match_idx || !d_match_dist || (nt > 0 && (!d_t2q_idx2 || !d_t2q_dist2))) return SLAMHIP_EINVAL;
  hipLaunchKernelGGL(l2_cross_check_kernel, dim3(slamhip_ceil_div(nq, 256)), dim3(256), 0,
                     (hipStream_t)stream, d_t2q_idx2, d_t2q_dist2, nt, nq, d_match_idx, d_match_dist);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

int slamhip_ratio_test_f32(const float *d_dist2, int nq, double ratio, uint8_t *d_keep, void *stream) {
  SLAMHIP_BEGIN();
  if (nq < 0) return SLAMHIP_EINVAL;
  if (nq == 0) return SLAMHIP_OK;
  if (!d_dist2 || !d_keep) return SLAMHIP_EINVAL;
  hipLaunchKernelGGL(l2_ratio_test_kernel, dim3(slamhip_ceil_div(nq, 256)), dim3(256), 0,
                     (hipStream_t)stream, d_dist2, nq, ratio, d_keep);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

int slamhip_sort_by_distance_f32(const float *d_dist, int dist_stride, int n, int32_t *d_order,
                                 void *stream) {
  SLAMHIP_BEGIN();
  if (n < 0 || dist_stride < 1) return SLAMHIP_EINVAL;
  if (n > SLAMHIP_MAX_TRAIN_ROWS) return SLAMHIP_EUNSUPPORTED;
  if (n == 0) return SLAMHIP_OK;
  if (!d_dist || !d_order) return SLAMHIP_EINVAL;
  hipLaunchKernelGGL(l2_rank_sort_kernel, dim3(slamhip_ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, d_dist, dist_stride, n, d_order);
  SLAMHIP_CHECK_LAUNCH();
  return SLAMHIP_OK;
}

}  // extern "C"
