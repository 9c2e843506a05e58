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
// Block = 256 threads computing 64 x 64 tiles of (query, train) distances, 4 x 4 per
// thread, with 32 dimensions of both tiles staged in LDS per step (dimension-major, so
// that the four rows a thread needs are one 16-byte read: broadcast for the query
// side, consecutive for the train side; two buffers, one barrier per step, the next
// step's rows already on their way from memory while a step is computed).  Per
// dimension: 8 floats from LDS, 16 x (subtract, multiply, add) as 24 packed float32
// instructions.  Every thread keeps the running (best, second) of its four query rows
// over its own train columns as 64-bit keys (distance bits << 32 | train index:
// non-negative floats order like their bits) and drops a candidate before the square
// root when its sum of squares cannot beat the second best; the threads of a query row
// meet in LDS once, at the end of the block.  grid = (query tiles, train splits); a
// second kernel folds the splits.
#include "common.h"

namespace {

constexpr int kTQ = 64, kTT = 64, kTK = 32;
constexpr int kPitch = kTQ + 4;   // floats per staged dimension: keeps 16-byte alignment, spreads banks
constexpr unsigned long long kNone = ~0ull;
constexpr int kMaxL2Splits = 64;
constexpr int kL2TargetBlocks = 2048;   // 8192 x 8192 x 256: 1024 blocks 1180 us, 2048 blocks 1149 us
typedef float f2 __attribute__((ext_vector_type(2)));
// scheduling barrier mask: scalar, vector-memory and LDS instructions may cross, VALU may not
constexpr int kMayCross = 0x4 | 0x10 | 0x80;

// Branch-free on purpose: with branches the compiler keeps the pair in scratch memory and
// selects the slot to overwrite by address.
__device__ __forceinline__ void push2(unsigned long long &b1, unsigned long long &b2,
                                      unsigned long long key) {
  const unsigned long long hi = key > b1 ? key : b1;
  b2 = hi < b2 ? hi : b2;
  b1 = key < b1 ? key : b1;
}

// Eight floats of one descriptor row for the staging role of a thread: dimensions
// k0 + c*4 .. +3 and k0 + 16 + c*4 .. +3.  VEC: rows are 16-byte aligned and `dim` is
// a multiple of 4, so each half is one 16-byte load that is inside the row or not at
// all.  Rows and dimensions that do not exist read as zero.
template <bool VEC>
__device__ __forceinline__ void fetch8(const float *__restrict__ base, int row, int n_rows, int stride,
                                       int dim, int k, float4 &lo, float4 &hi) {
  lo = hi = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (row >= n_rows) return;
  const float *p = base + (size_t)row * stride;
  if (VEC) {
    if (k < dim) lo = *reinterpret_cast<const float4 *>(p + k);
    if (k + 16 < dim) hi = *reinterpret_cast<const float4 *>(p + k + 16);
  } else {
    if (k + 0 < dim) lo.x = p[k + 0];
    if (k + 1 < dim) lo.y = p[k + 1];
    if (k + 2 < dim) lo.z = p[k + 2];
    if (k + 3 < dim) lo.w = p[k + 3];
    if (k + 16 < dim) hi.x = p[k + 16];
    if (k + 17 < dim) hi.y = p[k + 17];
    if (k + 18 < dim) hi.z = p[k + 18];
    if (k + 19 < dim) hi.w = p[k + 19];
  }
}

__device__ __forceinline__ void stage8(float *s, int c, int lrow, const float4 &lo, const float4 &hi) {
  float *d = s + (c * 4) * kPitch + lrow;
  d[0 * kPitch] = lo.x;
  d[1 * kPitch] = lo.y;
  d[2 * kPitch] = lo.z;
  d[3 * kPitch] = lo.w;
  d += 16 * kPitch;
  d[0 * kPitch] = hi.x;
  d[1 * kPitch] = hi.y;
  d[2 * kPitch] = hi.z;
  d[3 * kPitch] = hi.w;
}

template <bool VEC>
__global__ __launch_bounds__(256) void l2_top2_kernel(const float *__restrict__ q, int nq,
                                                      const float *__restrict__ t, int nt, int dim,
                                                      int stride, ulonglong2 *__restrict__ partial,
                                                      int32_t *__restrict__ idx2,
                                                      float *__restrict__ dist2) {
  // two staging buffers per side (one barrier per step); the candidate exchange at the
  // end of the block reuses the space
  __shared__ __align__(16) float smem[2][2 * kTK * kPitch];
  static_assert(sizeof(smem) >= sizeof(ulonglong2) * kTQ * 16, "candidates fit");
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int q0 = blockIdx.x * kTQ;
  const int n_splits = gridDim.y;
  const int tiles = (nt + kTT - 1) / kTT;
  const int tiles_per_split = (tiles + n_splits - 1) / n_splits;
  const int tile_begin = blockIdx.y * tiles_per_split;
  const int tile_end = min(tiles, tile_begin + tiles_per_split);

  // Running (best, second) of this thread's four query rows over the train columns it
  // sees (tx*4 .. +3 of every tile, in ascending order), with the sums of squares they
  // came from.  The rounded square root is monotonic, so a candidate whose sum is not
  // below the second best's cannot come strictly before it, and on equal distance the
  // earlier column stays: such a candidate is dropped before its square root is taken.
  // NaN (no second best yet, or a NaN distance) lets everything through to the exact
  // comparison of keys.
  unsigned long long b1[4], b2[4];
  float sum1[4], sum2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b1[i] = b2[i] = kNone;
    sum1[i] = sum2[i] = __uint_as_float(0x7FC00000u);
  }

  // staging role: row tid / 4 of both tiles, dimensions (tid % 4) * 4 .. + 3 and 16 more
  const int lrow = tid >> 2, c = tid & 3;
  float4 gq0, gq1, gt0, gt1;                            // the step after the one being computed
  if (tile_begin < tile_end) {
    fetch8<VEC>(q, q0 + lrow, nq, stride, dim, c * 4, gq0, gq1);
    fetch8<VEC>(t, tile_begin * kTT + lrow, nt, stride, dim, c * 4, gt0, gt1);
  }
  int buf = 0;
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int t0 = tile * kTT;
    f2 acc[8];                                          // [query row i][column pair]: acc[2*i + j/2][j%2]
#pragma unroll
    for (int x = 0; x < 8; ++x) acc[x] = f2{0.0f, 0.0f};
    for (int k0 = 0; k0 < dim; k0 += kTK, buf ^= 1) {
      // the buffer written here was last read two steps ago, before the previous barrier
      float *qs = smem[buf], *ts = smem[buf] + kTK * kPitch;
      stage8(qs, c, lrow, gq0, gq1);
      stage8(ts, c, lrow, gt0, gt1);
      __syncthreads();
      // next step's rows travel while this one is computed
      if (k0 + kTK < dim) {
        fetch8<VEC>(q, q0 + lrow, nq, stride, dim, k0 + kTK + c * 4, gq0, gq1);
        fetch8<VEC>(t, t0 + lrow, nt, stride, dim, k0 + kTK + c * 4, gt0, gt1);
      } else if (tile + 1 < tile_end) {
        fetch8<VEC>(q, q0 + lrow, nq, stride, dim, c * 4, gq0, gq1);
        fetch8<VEC>(t, t0 + kTT + lrow, nt, stride, dim, c * 4, gt0, gt1);
      }
      // dimensions past `dim` hold zeros on both sides: they add +0.0 and change nothing
      float4 a = *reinterpret_cast<const float4 *>(&qs[ty * 4]);
      float4 b = *reinterpret_cast<const float4 *>(&ts[tx * 4]);
#pragma unroll 8
      for (int k = 0; k < kTK; ++k) {
        float4 an = a, bn = b;
        if (k + 1 < kTK) {
          an = *reinterpret_cast<const float4 *>(&qs[(k + 1) * kPitch + ty * 4]);
          bn = *reinterpret_cast<const float4 *>(&ts[(k + 1) * kPitch + tx * 4]);
        }
        {
          // No fused multiply-add: the product is rounded before it is added (hipcc
          // contracts a * b + c by default, also through __fmul_rn / __fadd_rn).  The
          // three phases are kept apart so that eight independent packed operations
          // follow each other: a dependent packed operation issued back to back waits
          // for its predecessor, and three waves per SIMD do not hide that.
#pragma clang fp contract(off)
          const float av[4] = {a.x, a.y, a.z, a.w};
          const f2 b01 = {b.x, b.y}, b23 = {b.z, b.w};
          f2 d[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f2 ai = {av[i], av[i]};
            d[2 * i + 0] = ai - b01;
            d[2 * i + 1] = ai - b23;
          }
          __builtin_amdgcn_sched_barrier(kMayCross);
#pragma unroll
          for (int x = 0; x < 8; ++x) d[x] = d[x] * d[x];
          __builtin_amdgcn_sched_barrier(kMayCross);
#pragma unroll
          for (int x = 0; x < 8; ++x) acc[x] = acc[x] + d[x];
          __builtin_amdgcn_sched_barrier(kMayCross);
        }
        a = an;
        b = bn;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = t0 + tx * 4 + j;
        if (col < nt && !(acc[2 * i + j / 2][j % 2] >= sum2[i])) {
          // correctly rounded float square root (hipcc's default for sqrtf)
          const float dist = sqrtf(acc[2 * i + j / 2][j % 2]);
          const unsigned long long key = ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned)col;
          if (key < b1[i]) {
            b2[i] = b1[i];
            sum2[i] = sum1[i];
            b1[i] = key;
            sum1[i] = acc[2 * i + j / 2][j % 2];
          } else if (key < b2[i]) {
            b2[i] = key;
            sum2[i] = acc[2 * i + j / 2][j % 2];
          }
        }
      }
  }
  __syncthreads();                                      // every thread is done with the staged rows
  ulonglong2 *cand = reinterpret_cast<ulonglong2 *>(&smem[0][0]);   // [kTQ][16]
#pragma unroll
  for (int i = 0; i < 4; ++i) cand[(ty * 4 + i) * 16 + tx] = make_ulonglong2(b1[i], b2[i]);
  __syncthreads();
  if (tid < kTQ && q0 + tid < nq) {
    unsigned long long best1 = kNone, best2 = kNone;
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const ulonglong2 cx = cand[tid * 16 + x];
      push2(best1, best2, cx.x);
      push2(best1, best2, cx.y);
    }
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
  int s = slamhip_ceil_div(kL2TargetBlocks, qtiles);
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
  // 16-byte loads when every row starts on a 16-byte boundary and holds whole groups of four
  const bool vec = dim % 4 == 0 && row_stride % 4 == 0 && ((uintptr_t)d_query & 15) == 0 &&
                   (nt == 0 || ((uintptr_t)d_train & 15) == 0);
  const dim3 grid(slamhip_ceil_div(nq, kTQ), s);
  if (vec)
    hipLaunchKernelGGL(l2_top2_kernel<true>, grid, dim3(256), 0, st, d_query, nq, d_train, nt, dim,
                       row_stride, partial, d_idx2, d_dist2);
  else
    hipLaunchKernelGGL(l2_top2_kernel<false>, grid, dim3(256), 0, st, d_query, nq, d_train, nt, dim,
                       row_stride, partial, d_idx2, d_dist2);
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
