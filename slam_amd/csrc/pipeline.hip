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
#include "refine_device.h"
#include "sort_device.h"

namespace {

using namespace slamhip_dev;

constexpr int kBlock = 256;

// Timing experiment (-DSLAMHIP_STAMPS): the first workgroup of the fused launches of a small call
// notes the 100 MHz clock at its phase boundaries; tools/stamps_small_call.py prints the phases.
#ifdef SLAMHIP_STAMPS
static __device__ unsigned long long g_stamps[64];
#define SLAMHIP_STAMP(i)                                                                      \
  do {                                                                                        \
    __syncthreads();                                                                          \
    if (threadIdx.x == 0 && blockIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define SLAMHIP_STAMP(i)
#endif

// --------------------------------------------------------------------------
// Stereo stage, first half: row/disparity filter + stable compaction (ballot
// prefix) of the sorted matches.  One workgroup per frame walks them in chunks
// of 1024 so that the survivors keep their order (a chunk is a chain of three
// dependent loads: sixteen wavefronts have four times as many of them in flight
// as four, and a frame of 1500 features is two chunks instead of six).
// --------------------------------------------------------------------------
constexpr int kCompactBlock = 1024;
constexpr int kFusedSortMax = 8192;   // matches of a frame / pair that one block's LDS sorts
__global__ __launch_bounds__(kCompactBlock) void stereo_compact_kernel(
    const float *__restrict__ kp_left, const float *__restrict__ kp_right,
    const int32_t *__restrict__ off_left, const int32_t *__restrict__ off_right,
    const int32_t *__restrict__ idx2, const int32_t *__restrict__ dist2,
    const int32_t *__restrict__ order, double pix_thresh, double disp_min,
    int32_t *__restrict__ st_count, int32_t *__restrict__ st_left, int32_t *__restrict__ st_right,
    int32_t *__restrict__ st_dist, float *__restrict__ st_pix_left,
    float *__restrict__ st_pix_right) {
  __shared__ int wave_tot[kCompactBlock / 64];
  __shared__ int running;
  const int f = blockIdx.x;
  const int base = off_left[f], nq = off_left[f + 1] - base, rbase = off_right[f];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  for (int chunk = 0; chunk < nq; chunk += kCompactBlock) {
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
    for (int w = 0; w < kCompactBlock / 64; ++w) {
      if (w < wave) wave_base += wave_tot[w];
      chunk_total += wave_tot[w];
    }
    const int pos = running + wave_base + prefix;
    if (keep) {
      const size_t o = (size_t)(base + pos);
      st_left[o] = q;
      st_right[o] = t;
      st_dist[o] = d;
      st_pix_left[2 * o + 0] = (float)uL;
      st_pix_left[2 * o + 1] = (float)vL;
      st_pix_right[2 * o + 0] = (float)uR;
      st_pix_right[2 * o + 1] = (float)vR;
    }
    __syncthreads();
    if (threadIdx.x == 0) running += chunk_total;
    __syncthreads();
  }
  if (threadIdx.x == 0) st_count[f] = running;
}

// The sort by distance and the compaction above in one launch, for frames whose matches fit one
// block's LDS: the block sorts the frame's keys (sort_device.h) and walks them where they lie,
// so the order never goes through memory and a launch is saved.  Same results as
// sort_by_distance + stereo_compact_kernel.
template <int CAP>
__global__ __launch_bounds__(kCompactBlock) void stereo_sort_compact_kernel(
    const float *__restrict__ kp_left, const float *__restrict__ kp_right,
    const int32_t *__restrict__ off_left, const int32_t *__restrict__ off_right,
    const int32_t *__restrict__ idx2, const int32_t *__restrict__ dist2, double pix_thresh,
    double disp_min, int32_t *__restrict__ st_count, int32_t *__restrict__ st_left,
    int32_t *__restrict__ st_right, int32_t *__restrict__ st_dist, float *__restrict__ st_pix_left,
    float *__restrict__ st_pix_right) {
  __shared__ RadixSortLds<kCompactBlock, CAP> lds;
  __shared__ int wave_tot[kCompactBlock / 64];
  __shared__ int running;
  const int f = blockIdx.x;
  const int base = off_left[f], nq = off_left[f + 1] - base, rbase = off_right[f];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) running = 0;
  if (nq <= 0 || nq > CAP) {       // block-uniform (the host launches this kernel only when every frame fits)
    if (threadIdx.x == 0) st_count[f] = 0;
    return;
  }
  for (int i = threadIdx.x; i < nq; i += kCompactBlock)
    lds.keys[0][i] = sort_key(dist2[2 * (size_t)(base + i)], i);
  radix_sort_lds<kCompactBlock, CAP>(lds, nq);
  for (int chunk = 0; chunk < nq; chunk += kCompactBlock) {
    const int r = chunk + (int)threadIdx.x;
    bool keep = false;
    int q = 0, t = -1, d = -1;
    double uL = 0, vL = 0, uR = 0, vR = 0;
    if (r < nq) {
      q = (int)(lds.keys[0][r] & kSortIdxMask);
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
    for (int w = 0; w < kCompactBlock / 64; ++w) {
      if (w < wave) wave_base += wave_tot[w];
      chunk_total += wave_tot[w];
    }
    const int pos = running + wave_base + prefix;
    if (keep) {
      const size_t o = (size_t)(base + pos);
      st_left[o] = q;
      st_right[o] = t;
      st_dist[o] = d;
      st_pix_left[2 * o + 0] = (float)uL;
      st_pix_left[2 * o + 1] = (float)vL;
      st_pix_right[2 * o + 0] = (float)uR;
      st_pix_right[2 * o + 1] = (float)vR;
    }
    __syncthreads();
    if (threadIdx.x == 0) running += chunk_total;
    __syncthreads();
  }
  if (threadIdx.x == 0) st_count[f] = running;
}

// A call of a few frames (the frame-at-a-time use of the reference's own loop,
// fill_database.py:140) is a chain of dependent launches of one or a few workgroups each, and
// every launch in the chain costs some 3.5 us before it does anything.  For such calls the
// whole stereo stage behind the matcher is ONE launch, one workgroup per frame: fold of the
// matcher's train splits (its merge launch is left out), the row / disparity filter, the sort
// by distance of the SURVIVORS only (compacted in left-row order first: sorting after the filter
// then gives the order that filtering the sorted matches gives, for two fifths of the keys), then what
// stereo_points_kernel does for them.  The survivors' right rows, pixels and sorted order stay
// in LDS instead of going through memory between launches.  Same results as the separate launches.
//   part != nullptr: part[split * part_rows + off_left[f] + i] holds the (best, second best) key
//   of left row i within train split `split` (key = distance << 21 | right row, 0xFFFFFFFF =
//   none); else the folded result is in idx2 / dist2.
template <int CAP>
struct StereoTailLds {
  RadixSortLds<kCompactBlock, CAP> sort;
  int32_t train[CAP];                    // survivors' right rows, by left row
  float4 pix[CAP];                       // survivors' (uL, vL, uR, vR), by left row
  int wave_tot[2][kCompactBlock / 64];
};
template <int CAP>
__global__ __launch_bounds__(kCompactBlock) void stereo_tail_kernel(
    const float *__restrict__ kp_left, const float *__restrict__ kp_right,
    const int32_t *__restrict__ off_left, const int32_t *__restrict__ off_right,
    const int32_t *__restrict__ idx2, const int32_t *__restrict__ dist2,
    const uint2 *__restrict__ part, int splits, int part_rows, double pix_thresh, double disp_min,
    int32_t *__restrict__ st_count, int32_t *__restrict__ st_left, int32_t *__restrict__ st_right,
    int32_t *__restrict__ st_dist, float *__restrict__ st_pix_left, float *__restrict__ st_pix_right,
    const uint32_t *__restrict__ packed_left, int W, const uint8_t *__restrict__ raw_left, int raw_stride,
    int desc_bytes, size_t raw_bytes, CamPair cam, float *__restrict__ st_xyz, uint32_t *__restrict__ st_desc) {
  __shared__ StereoTailLds<CAP> lds;
  SLAMHIP_STAMP(0);
  const int f = blockIdx.x;
  const int base = off_left[f], nq = off_left[f + 1] - base, rbase = off_right[f];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (nq <= 0 || nq > CAP) {       // block-uniform (the host launches this kernel only when every frame fits)
    if (threadIdx.x == 0) st_count[f] = 0;
    return;
  }
  // Every left row's match first -- fold of the train splits -- and its verdict; the survivors' keys
  // go into LDS in left-row order (the radix passes below sort by the distance field alone and
  // leave equal distances in the order they start in), their right rows and pixels beside them,
  // indexed by the left row.  Two chunks of kCompactBlock rows are judged before the one barrier
  // that makes the wavefronts' survivor counts known.
  constexpr int kGroup = 2;
  int running = 0;
  for (int i0 = 0; i0 < nq; i0 += kGroup * kCompactBlock) {
    bool keep[kGroup];
    int prefix[kGroup], dist[kGroup];
#pragma unroll
    for (int c = 0; c < kGroup; ++c) {
      const int i = i0 + c * kCompactBlock + (int)threadIdx.x;
      int t = -1, d = -1;
      if (i < nq) {
        if (part != nullptr) {
          uint32_t best = 0xFFFFFFFFu;           // keys are unique: the smallest is (closest, lowest right row)
          for (int s0 = 0; s0 < splits; s0 += 16) {   // sixteen loads in flight (the matcher's usual split count)
            uint32_t k[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) k[j] = part[(size_t)min(s0 + j, splits - 1) * part_rows + base + i].x;
#pragma unroll
            for (int j = 0; j < 16; ++j) best = min(best, k[j]);
          }
          t = best == 0xFFFFFFFFu ? -1 : (int)(best & kSortIdxMask);
          d = best == 0xFFFFFFFFu ? -1 : (int)(best >> kSortKeyShift);
        } else {
          t = idx2[2 * (size_t)(base + i)];
          d = dist2[2 * (size_t)(base + i)];
        }
      }
      keep[c] = false;
      dist[c] = d;
      if (t >= 0) {
        const float4 px = make_float4(kp_left[2 * (size_t)(base + i)], kp_left[2 * (size_t)(base + i) + 1],
                                      kp_right[2 * (size_t)(rbase + t)], kp_right[2 * (size_t)(rbase + t) + 1]);
        keep[c] = stereo_keep((double)px.x, (double)px.y, (double)px.z, (double)px.w, pix_thresh, disp_min);
        if (keep[c]) {
          lds.train[i] = t;
          lds.pix[i] = px;
        }
      }
      const unsigned long long ballot = __ballot(keep[c]);
      prefix[c] = __popcll(ballot & ((1ull << lane) - 1ull));
      if (lane == 0) lds.wave_tot[c][wave] = __popcll(ballot);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kGroup; ++c) {
      int wave_base = 0, chunk_total = 0;
#pragma unroll
      for (int w = 0; w < kCompactBlock / 64; ++w) {
        const int v = lds.wave_tot[c][w];
        if (w < wave) wave_base += v;
        chunk_total += v;
      }
      if (keep[c])
        lds.sort.keys[0][running + wave_base + prefix[c]] = sort_key(dist[c], i0 + c * kCompactBlock + (int)threadIdx.x);
      running += chunk_total;
    }
    __syncthreads();     // the counts have been read before the next group overwrites them
  }
  const int m = running;
  SLAMHIP_STAMP(1);
  radix_sort_lds<kCompactBlock, CAP>(lds.sort, m);
  SLAMHIP_STAMP(2);
  uint32_t *const left_of = lds.sort.keys[1];    // survivor -> left row, once the sort is done with the array
  if (threadIdx.x == 0) st_count[f] = m;
  // survivor r in sorted order: its record, and its point from the pixels kept in LDS
  for (int r = threadIdx.x; r < m; r += kCompactBlock) {
    const uint32_t key = lds.sort.keys[0][r];
    const int q = (int)(key & kSortIdxMask);
    const float4 px = lds.pix[q];
    const size_t o = (size_t)(base + r);
    left_of[r] = (uint32_t)q;
    st_left[o] = q;
    st_right[o] = lds.train[q];
    st_dist[o] = (int)(key >> kSortKeyShift);
    st_pix_left[2 * o + 0] = px.x;
    st_pix_left[2 * o + 1] = px.y;
    st_pix_right[2 * o + 0] = px.z;
    st_pix_right[2 * o + 1] = px.w;
    double x[4];
    dlt_null_vector(cam, (double)px.x, (double)px.y, (double)px.z, (double)px.w, x);
    const float x0 = (float)x[0], x1 = (float)x[1], x2 = (float)x[2], x3 = (float)x[3];
    st_xyz[3 * o + 0] = x0 / x3;
    st_xyz[3 * o + 1] = x1 / x3;
    st_xyz[3 * o + 2] = x2 / x3;
  }
  __syncthreads();      // left_of is complete
  SLAMHIP_STAMP(3);
  SLAMHIP_STAMP(4);
  // and their descriptor rows, in survivor order (as stereo_points_kernel)
  const int lw = W == 8 ? 3 : (W == 16 ? 4 : 5);
  const int cpr = W >> 2, lc = lw - 2;
  uint4 *dst = reinterpret_cast<uint4 *>(st_desc);
  if (raw_left == nullptr) {
    const uint4 *src = reinterpret_cast<const uint4 *>(packed_left);
#pragma unroll 4
    for (int e = threadIdx.x; e < m * cpr; e += kCompactBlock) {
      const int r = e >> lc, c = e & (cpr - 1);
      dst[(size_t)(base + r) * cpr + c] = src[(size_t)(base + (int)left_of[r]) * cpr + c];
    }
  } else {
#pragma unroll 4
    for (int e = threadIdx.x; e < m * cpr; e += kCompactBlock) {
      const int r = e >> lc, c = e & (cpr - 1);
      const int valid = desc_bytes - 16 * c;          // descriptor bytes from this chunk on
      const size_t at = (size_t)(base + (int)left_of[r]) * raw_stride + 16 * c;
      const size_t from = at + 16 <= raw_bytes ? at : raw_bytes - 16;
      unsigned __int128 x;
      __builtin_memcpy(&x, raw_left + from, 16);
      const int back = (int)(at - from);
      x >>= 8 * (back < 15 ? back : 15);
      auto mk = [&](int j) {
        const int nb = valid - 4 * j;
        return nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : (1u << (8 * nb)) - 1u);
      };
      dst[(size_t)(base + r) * cpr + c] = make_uint4((uint32_t)x & mk(0), (uint32_t)(x >> 32) & mk(1),
                                                     (uint32_t)(x >> 64) & mk(2), (uint32_t)(x >> 96) & mk(3));
    }
  }
  SLAMHIP_STAMP(5);
}

// Second half of the stereo stage, over the compacted inliers only (every lane
// has work): DLT triangulation of inlier i of frame f from its two pixels, and
// the gather of its packed left descriptor row.  grid = (chunks of 256, frames).
__global__ __launch_bounds__(kBlock) void stereo_points_kernel(
    const int32_t *__restrict__ off_left, const int32_t *__restrict__ st_count,
    const int32_t *__restrict__ st_left, const float *__restrict__ st_pix_left,
    const float *__restrict__ st_pix_right, const uint32_t *__restrict__ packed_left, int W,
    const uint8_t *__restrict__ raw_left, int raw_stride, int desc_bytes, size_t raw_bytes,
    CamPair cam, float *__restrict__ st_xyz, uint32_t *__restrict__ st_desc) {
  const int f = blockIdx.y;
  const int base = off_left[f], m = st_count[f];
  const int i0 = blockIdx.x * kBlock;
  if (i0 >= m) return;
  const int i = i0 + (int)threadIdx.x;
  if (i < m) {
    const size_t o = (size_t)(base + i);
    double x[4];
    dlt_null_vector(cam, (double)st_pix_left[2 * o], (double)st_pix_left[2 * o + 1],
                    (double)st_pix_right[2 * o], (double)st_pix_right[2 * o + 1], x);
    const float x0 = (float)x[0], x1 = (float)x[1], x2 = (float)x[2], x3 = (float)x[3];
    st_xyz[3 * o + 0] = x0 / x3;
    st_xyz[3 * o + 1] = x1 / x3;
    st_xyz[3 * o + 2] = x2 / x3;
  }
  // descriptor rows of this block's inliers, in inlier order (W is a power of two)
  const int rows = min(kBlock, m - i0);
  const int lw = W == 8 ? 3 : (W == 16 ? 4 : 5);
  // one 16-byte chunk of a row per step (W / 4 chunks a row): few, wide, independent loads
  const int cpr = W >> 2, lc = lw - 2;
  if (raw_left == nullptr) {
    const uint4 *src = reinterpret_cast<const uint4 *>(packed_left);
    uint4 *dst = reinterpret_cast<uint4 *>(st_desc);
#pragma unroll 4
    for (int e = threadIdx.x; e < rows * cpr; e += kBlock) {
      const int r = i0 + (e >> lc), c = e & (cpr - 1);
      dst[(size_t)(base + r) * cpr + c] = src[(size_t)(base + st_left[base + r]) * cpr + c];
    }
  } else {
    // the caller's rows (any stride and alignment; raw_bytes >= 16): 16-byte loads that never
    // leave the array -- one that would is moved back to end at its end and shifted down -- and
    // zeros behind the descriptor's last byte, which is what a packed row holds
    uint4 *dst = reinterpret_cast<uint4 *>(st_desc);
#pragma unroll 4
    for (int e = threadIdx.x; e < rows * cpr; e += kBlock) {
      const int r = i0 + (e >> lc), c = e & (cpr - 1);
      const int valid = desc_bytes - 16 * c;          // descriptor bytes from this chunk on
      const size_t at = (size_t)(base + st_left[base + r]) * raw_stride + 16 * c;
      const size_t from = at + 16 <= raw_bytes ? at : raw_bytes - 16;
      unsigned __int128 x;
      __builtin_memcpy(&x, raw_left + from, 16);
      const int back = (int)(at - from);
      x >>= 8 * (back < 15 ? back : 15);
      auto m = [&](int j) {
        const int nb = valid - 4 * j;
        return nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : (1u << (8 * nb)) - 1u);
      };
      dst[(size_t)(base + r) * cpr + c] = make_uint4((uint32_t)x & m(0), (uint32_t)(x >> 32) & m(1),
                                                     (uint32_t)(x >> 64) & m(2), (uint32_t)(x >> 96) & m(3));
    }
  }
}

// --------------------------------------------------------------------------
// crossCheck=True for many pairs at once.  The matcher runs with the roles
// swapped (every train row finds its best query row, lowest query index on
// ties); each train row then bids (distance, its own index) for that query row
// with one atomic minimum on a key per query row, so a query keeps the closest
// of the train rows that chose it, lowest train index on ties -- the fold of
// cv2's batchDistance -- in O(rows) instead of a scan per query.
// Pair p: its tcnt[p] train rows hold their choice in r_idx2 / r_dist2 at rows
// rbase[p] + t (the rows themselves in the front-end; a scratch row per (pair, row)
// where pairs share rows); the bids of its query rows are at qbase[p] + query.
// --------------------------------------------------------------------------
constexpr uint32_t kBidShift = 21, kBidIdxMask = (1u << kBidShift) - 1u, kNoBid = 0xFFFFFFFFu;

__global__ __launch_bounds__(kBlock) void cross_bid_kernel(
    const int32_t *__restrict__ tbase /* = rbase */, const int32_t *__restrict__ tcnt_or_null,
    const int32_t *__restrict__ qbase, const int32_t *__restrict__ r_idx2,
    const int32_t *__restrict__ r_dist2, uint32_t *__restrict__ bid) {
  const int p = blockIdx.y;
  const int tb = tbase[p];
  const int nt = tcnt_or_null ? tcnt_or_null[p] : tbase[p + 1] - tb;
  const int t = blockIdx.x * kBlock + (int)threadIdx.x;
  if (t >= nt) return;
  const int who = r_idx2[2 * (size_t)(tb + t)];
  if (who < 0) return;
  const uint32_t d = (uint32_t)r_dist2[2 * (size_t)(tb + t)];
  atomicMin(&bid[(size_t)qbase[p] + who], (d << kBidShift) | (uint32_t)t);
}

// bids -> the (index, distance) layout the forward matcher writes (second neighbour absent)
__global__ __launch_bounds__(kBlock) void cross_decode_kernel(const uint32_t *__restrict__ bid, int n,
                                                              int32_t *__restrict__ idx2,
                                                              int32_t *__restrict__ dist2) {
  const int i = blockIdx.x * kBlock + (int)threadIdx.x;
  if (i >= n) return;
  const uint32_t b = bid[i];
  int2 oi, od;
  oi.x = b == kNoBid ? -1 : (int32_t)(b & kBidIdxMask);
  od.x = b == kNoBid ? -1 : (int32_t)(b >> kBidShift);
  oi.y = od.y = -1;
  reinterpret_cast<int2 *>(idx2)[i] = oi;
  reinterpret_cast<int2 *>(dist2)[i] = od;
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
  if (blockIdx.x == 0 && threadIdx.x == 0 && n == 0) tm_count[p] = 0;
  const int r = blockIdx.x * kBlock + (int)threadIdx.x;
  if (r >= n) return;
  const int q = order2[ob + r];
  const int t = tidx2[2 * (size_t)(ob + q)];
  // query rows without a match (cross-check) sort last: the matches are a prefix of
  // the sorted rows, and the row that ends it writes the count
  const bool next_matched = r + 1 < n && tidx2[2 * (size_t)(ob + order2[ob + r + 1])] >= 0;
  if (t < 0) {
    if (r == 0) tm_count[p] = 0;
    return;
  }
  if (!next_matched) tm_count[p] = r + 1;
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

// The sort by distance and the gather above in one launch, one block per pair, for pairs whose
// matches fit one block's LDS (see stereo_sort_compact_kernel).  Same results as
// sort_by_distance + temporal_gather_kernel.
template <typename PIX, int T, int CAP>
__global__ __launch_bounds__(T) void temporal_sort_gather_kernel(
    const int32_t *__restrict__ qbase, const int32_t *__restrict__ qcnt,
    const int32_t *__restrict__ tbase, const int32_t *__restrict__ tcnt,
    const int32_t *__restrict__ obase, const int32_t *__restrict__ tidx2,
    const int32_t *__restrict__ tdist2, const float *__restrict__ xyz,
    const PIX *__restrict__ pix_left, const PIX *__restrict__ pix_right,
    int32_t *__restrict__ tm_query, int32_t *__restrict__ tm_train, int32_t *__restrict__ tm_dist,
    int32_t *__restrict__ tm_count, float *__restrict__ cx, float *__restrict__ cpl,
    float *__restrict__ cpr) {
  __shared__ RadixSortLds<T, CAP> lds;
  const int p = blockIdx.x;
  const int qb = qbase[p], tb = tbase[p], ob = obase[p];
  const int nsort = qcnt[p];
  const int n = tcnt[p] > 0 ? nsort : 0;  // no train rows -> no matches
  if (n <= 0 || n > CAP) {                // block-uniform
    if (threadIdx.x == 0) tm_count[p] = 0;
    return;
  }
  for (int i = threadIdx.x; i < n; i += T) lds.keys[0][i] = sort_key(tdist2[2 * (size_t)(ob + i)], i);
  radix_sort_lds<T, CAP>(lds, n);
  for (int r = threadIdx.x; r < n; r += T) {
    const int q = (int)(lds.keys[0][r] & kSortIdxMask);
    const int t = tidx2[2 * (size_t)(ob + q)];
    // query rows without a match (cross-check) sort last: the matches are a prefix of
    // the sorted rows, and the row that ends it writes the count
    const bool next_matched =
        r + 1 < n && tidx2[2 * (size_t)(ob + (int)(lds.keys[0][r + 1] & kSortIdxMask))] >= 0;
    if (t < 0) {
      if (r == 0) tm_count[p] = 0;
      continue;
    }
    if (!next_matched) tm_count[p] = r + 1;
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
}

// The temporal stage behind the matcher as one launch for calls of a few pairs (see
// stereo_tail_kernel): fold of the matcher's train splits, sort, gather.  Same results as the
// merge launch + temporal_sort_gather_kernel.
template <typename PIX, int T, int CAP>
struct TemporalTailLds {
  RadixSortLds<T, CAP> sort;
  int32_t train[CAP];
};
template <typename PIX, int T, int CAP>
__global__ __launch_bounds__(T) void temporal_tail_kernel(
    const int32_t *__restrict__ qbase, const int32_t *__restrict__ qcnt,
    const int32_t *__restrict__ tbase, const int32_t *__restrict__ tcnt,
    const int32_t *__restrict__ obase, const int32_t *__restrict__ tidx2,
    const int32_t *__restrict__ tdist2, const uint2 *__restrict__ part, int splits, int part_rows,
    const float *__restrict__ xyz, const PIX *__restrict__ pix_left, const PIX *__restrict__ pix_right,
    int32_t *__restrict__ tm_query, int32_t *__restrict__ tm_train, int32_t *__restrict__ tm_dist,
    int32_t *__restrict__ tm_count, float *__restrict__ cx, float *__restrict__ cpl,
    float *__restrict__ cpr, int32_t *__restrict__ clear, int clear_per_pair) {
  __shared__ TemporalTailLds<PIX, T, CAP> lds;
  SLAMHIP_STAMP(8);
  const int p = blockIdx.x;
  // the pair's slots of the RANSAC count table, for a counting launch that only adds (may be null)
  if (clear != nullptr)
    for (int i = threadIdx.x; i < clear_per_pair; i += T) clear[(size_t)p * clear_per_pair + i] = 0;
  const int qb = qbase[p], tb = tbase[p], ob = obase[p];
  const int nsort = qcnt[p];
  const int n = tcnt[p] > 0 ? nsort : 0;  // no train rows -> no matches
  if (n <= 0 || n > CAP) {                // block-uniform
    if (threadIdx.x == 0) tm_count[p] = 0;
    return;
  }
  for (int i = threadIdx.x; i < n; i += T) {
    int t, d;
    if (part != nullptr) {
      uint32_t best = 0xFFFFFFFFu;
      for (int s0 = 0; s0 < splits; s0 += 8) {
        uint32_t k[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) k[j] = part[(size_t)min(s0 + j, splits - 1) * part_rows + ob + i].x;
#pragma unroll
        for (int j = 0; j < 8; ++j) best = min(best, k[j]);
      }
      t = best == 0xFFFFFFFFu ? -1 : (int)(best & kSortIdxMask);
      d = best == 0xFFFFFFFFu ? -1 : (int)(best >> kSortKeyShift);
    } else {
      t = tidx2[2 * (size_t)(ob + i)];
      d = tdist2[2 * (size_t)(ob + i)];
    }
    lds.train[i] = t;
    lds.sort.keys[0][i] = sort_key(d, i);
  }
  SLAMHIP_STAMP(9);
  radix_sort_lds<T, CAP>(lds.sort, n);
  SLAMHIP_STAMP(10);
  for (int r = threadIdx.x; r < n; r += T) {
    const uint32_t key = lds.sort.keys[0][r];
    const int q = (int)(key & kSortIdxMask);
    const int t = lds.train[q];
    // query rows without a match sort last: the matches are a prefix of the sorted rows, and
    // the row that ends it writes the count
    const bool next_matched = r + 1 < n && lds.train[(int)(lds.sort.keys[0][r + 1] & kSortIdxMask)] >= 0;
    if (t < 0) {
      if (r == 0) tm_count[p] = 0;
      continue;
    }
    if (!next_matched) tm_count[p] = r + 1;
    const size_t o = (size_t)(ob + r);
    tm_query[o] = q;
    tm_train[o] = t;
    tm_dist[o] = (int)(key >> kSortKeyShift);
#pragma unroll
    for (int c = 0; c < 3; ++c) cx[3 * o + c] = xyz[3 * (size_t)(qb + q) + c];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      cpl[2 * o + c] = (float)pix_left[2 * (size_t)(tb + t) + c];
      cpr[2 * o + c] = (float)pix_right[2 * (size_t)(tb + t) + c];
    }
  }
  SLAMHIP_STAMP(11);
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// Adaptive RANSAC over many pairs, run as rounds of kHypPerRound hypotheses
// per pair.  A round is two launches over all pairs that are still running:
// "solve" (sample + P3P, one lane per hypothesis) and "count" (every
// hypothesis against every correspondence, both cameras, partial counts added
// atomically).  The next solve launch (or the finishing launch) first replays
// the sequential stopping rule of pnp.py:123-158 over the round's counts in
// hypothesis order, so the outcome equals the one-hypothesis-at-a-time loop on
// the same samples; a pair that has stopped costs later launches one load.
// Splitting the rounds into launches keeps every CU busy however unevenly the
// pairs' iteration counts fall: a pair that needs ten rounds does not hold a
// CU while the rest of the chip has nothing left to do.
//
// Per-pair state is the caller's 8-int info record: best_idx, best_count,
// iterations done, iterations required, domain-error flag, n, mask count,
// stopped flag (cleared again by the finishing launch).
constexpr int kHypPerRound = kBlock;
constexpr int kPairGroup = 4096;   // pairs per group of launches (bounds the pose table)

// Iterations required once the best hypothesis has `count` of n inliers
// (pnp.py:150-156): ceil(log(1-p) / log(1 - w^4)), clamped to int.
__device__ __forceinline__ int iterations_required(int count, int n, double log1mp) {
  const double w = (double)count / (double)n;
  const double w2 = w * w;
  const double den = log(1.0 - w2 * w2);   // -inf when every point is an inlier
  const double r = ceil(log1mp / den);
  return r > 2147483647.0 ? 2147483647 : (int)r;
}

// Inclusive scan over the kBlock threads of the workgroup (max if kMax else
// min); wave shuffles, then the four wave totals through LDS.  `wt` holds
// kBlock / 64 ints.  Also returns the exclusive value (identity for thread 0).
template <bool kMax>
__device__ __forceinline__ int block_scan(int v, int identity, int *wt, int &exclusive) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // within the wavefront through DPP, as sort_device.h's wave_inclusive_sum: lanes without a source take
  // the identity
  auto fold = [&](int t) { v = kMax ? max(v, t) : min(v, t); };
  fold(__builtin_amdgcn_update_dpp(identity, v, 0x111, 0xF, 0xF, false));   // row_shr:1
  fold(__builtin_amdgcn_update_dpp(identity, v, 0x112, 0xF, 0xF, false));   // row_shr:2
  fold(__builtin_amdgcn_update_dpp(identity, v, 0x114, 0xF, 0xF, false));   // row_shr:4
  fold(__builtin_amdgcn_update_dpp(identity, v, 0x118, 0xF, 0xF, false));   // row_shr:8
  fold(__builtin_amdgcn_update_dpp(identity, v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
  fold(__builtin_amdgcn_update_dpp(identity, v, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
  const int ex = __builtin_amdgcn_update_dpp(identity, v, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0: identity)
  if (lane == 63) wt[wave] = v;
  __syncthreads();
  int before = identity;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w)
    if (w < wave) before = kMax ? max(before, wt[w]) : min(before, wt[w]);
  __syncthreads();
  exclusive = kMax ? max(ex, before) : min(ex, before);
  return kMax ? max(v, before) : min(v, before);
}

// The sequential stopping rule of pnp.py:123-158 over hypotheses
// [round*kHypPerRound, +kHypPerRound) of one pair, evaluated by the whole
// workgroup (thread i looks at hypothesis i) with the same outcome as the
// one-at-a-time loop:
//   a hypothesis is a record if its count exceeds every earlier count and the
//   best so far; the requirement in force when hypothesis i is reached is the
//   minimum of the carried one and those of the earlier records; the loop
//   reaches i iff i lies below that requirement and below the cap.  The
//   requirement never grows, so the hypotheses reached form a prefix.
// Updates the pair's state record and returns the lane of a new best
// hypothesis of this round or -1 (same value in every thread).
// `red` holds 4 + kBlock / 64 ints of LDS.
// `s` may be the pair's record in memory or a copy of it in LDS; `c` = counts[threadIdx.x], loaded
// by the caller (which may have asked for it well before).
__device__ int replay_round(int32_t *s, const int32_t *__restrict__ counts, const int c, int round,
                            int n, int n_iter_max, double log1mp, int *red) {
  const int i = (int)threadIdx.x;
  const int best0 = s[1], done0 = s[2], req0 = s[3];
  const int g = round * kHypPerRound + i;   // c: -1 = no pose from this sample, -2 = not generated
  if (i < 4) red[i] = (i == 1) ? -1 : (i == 2 ? 2147483647 : 0);  // reached, winner, requirement, domain
  int before;
  block_scan<true>(max(c, best0), best0, red + 4, before);
  const bool record = c > before;
  const int req_i = record ? iterations_required(c, n, log1mp) : 2147483647;
  int in_force;
  block_scan<false>(min(req_i, req0), req0, red + 4, in_force);
  const bool reached = g < in_force && g < n_iter_max;
  if (reached) {
    atomicAdd(&red[0], 1);
    if (record) {
      atomicMax(&red[1], i);
      atomicMin(&red[2], req_i);
      if (c >= n) red[3] = 1;              // log(0): the reference raises here
    }
  }
  __syncthreads();
  // the requirement reported is that of the latest record (the smallest of this round's,
  // as it falls with the count); it may exceed the cap, which bounds the loop separately
  const int n_reached = red[0], winner = red[1], req = red[1] >= 0 ? red[2] : req0, domain = red[3];
  const int best_count = winner >= 0 ? counts[winner] : best0;
  __syncthreads();                         // red[] is reused by the caller
  if (i == 0) {
    const int done = done0 + n_reached;
    if (winner >= 0) { s[0] = round * kHypPerRound + winner; s[1] = best_count; }
    s[2] = done;
    s[3] = req;
    if (domain) s[4] = 1;
    s[7] = (done < req && done < n_iter_max) ? 0 : 1;
  }
  return winner;
}

struct RansacParams {
  RansacCam cam;
  Kinv kinv;
  double thresh, p_success;
  int n_iter_max;
  int min_points;  // pairs with fewer correspondences are not attempted
  unsigned long long seed;
  long long pair0;  // index of the launch's first pair among the pairs of this call (sample table row)
  long long seq0;   // index of this call's first pair in the caller's numbering of the whole job (sampler)
  const long long *pair_ids;   // or, when not null: pair_ids[pair0 + p] is pair p's number in the whole job
};

// Hypothesis `it` of pair p of the launch: its four sample points (from the caller's table, or
// drawn from the sampler keyed by the pair's number in the whole job, so that batching, chunking
// and sharding over ranks do not change a pair's samples) and the pose the minimal solver gives
// them.  X / mL: the pair's n correspondences.  One function for every launch that generates
// hypotheses, so that they all generate the same ones.
__device__ __forceinline__ bool sample_and_solve(const float *__restrict__ X, const float *__restrict__ mL, int n,
                                                 const int32_t *__restrict__ samples, const RansacParams &prm,
                                                 int p, int it, int (&s4)[4], double (&Rt)[12]) {
  const unsigned long long gp = (unsigned long long)(prm.pair0 + p);
  if (samples != nullptr) {
    const int32_t *sp = samples + ((size_t)gp * prm.n_iter_max + it) * 4;
    for (int k = 0; k < 4; ++k) s4[k] = min(max(sp[k], 0), n - 1);
  } else {
    const unsigned long long sp_ = (unsigned long long)(prm.pair_ids != nullptr ? prm.pair_ids[prm.pair0 + p]
                                                                                 : prm.seq0 + prm.pair0 + p);
    unsigned long long r = splitmix64(prm.seed ^ (sp_ << 32) ^ (unsigned long long)it);
    for (int k = 0; k < 4; ++k) {
      int c = 0;
      for (int tries = 0; tries < 64; ++tries) {
        r = splitmix64(r);
        c = (int)(((r >> 32) * (unsigned long long)n) >> 32);
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
    pixel_to_bearing(prm.kinv, (double)mL[2 * (size_t)s4[k]], (double)mL[2 * (size_t)s4[k] + 1], f[k], nk);
    if (k == 3) { nrm4[0] = nk[0]; nrm4[1] = nk[1]; }
  }
  return p3p_lane(Xs, f, nrm4, Rt);
}

// grid (pairs, slots).  A launch covers `slots` consecutive blocks of kHypPerRound
// hypotheses per pair, starting at block `first_block`: 1 in the round scheme; all of them
// (first_block 0, no replay in between) when a call has so few pairs that launches, not
// arithmetic, bound it; the rounds left after the first few of a long run in one go (the
// merged tail).  `replay`: first replay block first_block - 1 (one slot per pair only: the
// replay updates the pair's state); `generate`: sample and solve (a launch with replay
// alone brings the state up to date for a multi-slot launch that follows).
// Pose table layout: rt[((pair * slots + slot) * 12 + element) * kHypPerRound + lane].
// `list` (may be null): list[0] counts, list[1] holds the largest correspondence count among, and
// list[2..] names, the table rows this launch
// generated hypotheses for; a launch with `generate` off clears the count instead.
// `hyp32` (may be null): per valid hypothesis the fp32 image of its projection matrices and
// its error-bound constants (Hyp32), hyp32[(row * kHyp32Floats + field) * kHypPerRound + lane].
// `cpr_own` (may be null; needs hyp32): the right-image pixels; the launch then counts, per
// hypothesis, the inliers among the staged pairs (points 2r, 2r + 1) that hold its own sample
// points and starts the hypothesis' count there instead of at zero: the counting launch for
// mostly unrelated pairs leaves exactly those pairs out (count_chunk_inliers_sparse).
constexpr int kHyp32Floats = 32;   // 31 used: PL[12], PR[12], A3xy, Atxy, A3w, Atw, cR, and the four sample
                                   // indices, 16 bits each, in two more (kSampleIndexMax)
constexpr int kSampleIndexMax = 65535;
__global__ __launch_bounds__(kBlock)
__attribute__((amdgpu_waves_per_eu(SLAMHIP_P3P_WAVES, SLAMHIP_P3P_WAVES))) void ransac_solve_kernel(
    int first_block, int slots, int replay, int generate, const int32_t *__restrict__ obase,
    const int32_t *__restrict__ tm_count, const float *__restrict__ cx, const float *__restrict__ cpl,
    const int32_t *__restrict__ samples, RansacParams prm, double *__restrict__ rt,
    int32_t *__restrict__ counts, double *__restrict__ pose_ransac, int32_t *__restrict__ info,
    int32_t *__restrict__ list, float *__restrict__ hyp32, const float *__restrict__ cpr_own) {
  __shared__ int sh[3];  // stopped, winner lane of the replayed round, iterations required
  __shared__ int red[4 + kBlock / 64];
  const int p = blockIdx.x, slot = blockIdx.y;
  const size_t row = (size_t)p * slots + slot;      // this block's row of the pose / count tables
  int32_t *s = info + 8 * (size_t)p;
  const int n = tm_count[p];
  int winner = -1;
  if (first_block == 0) {
    // every slot's block works out the same start state; slot 0 records it
    const int skip = (n >= 4 && n >= prm.min_points) ? 0 : 1;
    if (threadIdx.x == 0) {
      if (slot == 0) {
        s[0] = -1; s[1] = 0; s[2] = 0; s[3] = prm.n_iter_max; s[4] = 0; s[5] = n; s[6] = 0;
        s[7] = skip;
      }
      sh[0] = skip;
      sh[1] = -1;
      sh[2] = prm.n_iter_max;
    }
  } else {
    if (replay && !s[7])   // slots == 1; s[7]: same value in every thread, written by an earlier launch
      winner = replay_round(s, counts + row * kHypPerRound, counts[row * kHypPerRound + threadIdx.x],
                            first_block - 1, n, prm.n_iter_max, log(1.0 - prm.p_success), red);
    if (threadIdx.x == 0) {
      sh[0] = s[7];
      sh[1] = winner;
      sh[2] = s[3];
    }
  }
  __syncthreads();
  if (threadIdx.x < 12 && slot == 0) {
    if (first_block == 0)
      pose_ransac[12 * (size_t)p + threadIdx.x] = (threadIdx.x % 5 == 0) ? 1.0 : 0.0;  // [I|0]
    else if (sh[1] >= 0)
      pose_ransac[12 * (size_t)p + threadIdx.x] = rt[(row * 12 + threadIdx.x) * kHypPerRound + sh[1]];
  }
  if (list != nullptr && !generate && p == 0 && threadIdx.x == 0) { list[0] = 0; list[1] = 0; }   // the next launch appends
  if (sh[0] || !generate) return;
  __syncthreads();  // the winner has been read before its slot is overwritten

  // The required iteration count only ever shrinks, so a hypothesis at or beyond the
  // current requirement can never be consumed by the replay: it is not generated.
  const int it = (first_block + slot) * kHypPerRound + (int)threadIdx.x;
  const bool active = it < prm.n_iter_max && it < sh[2];
  if (list != nullptr && threadIdx.x == 0 && active) {    // lane 0 holds the block's first hypothesis
    list[2 + atomicAdd(&list[0], 1)] = (int32_t)row;
    atomicMax(&list[1], n);
  }
  bool valid = false;
  int own = 0;   // inliers among the staged pairs of the hypothesis' own sample points (cpr_own)
  if (active) {
    const float *X = cx + 3 * (size_t)obase[p];
    const float *mL = cpl + 2 * (size_t)obase[p];
    int s4[4];
    double Rt[12];
    valid = sample_and_solve(X, mL, n, samples, prm, p, it, s4, Rt);
    if (valid) {
      for (int i = 0; i < 12; ++i) rt[(row * 12 + i) * kHypPerRound + threadIdx.x] = Rt[i];
      if (hyp32 != nullptr) {
        // what the counting launch's fp32 screen needs of this hypothesis, worked out once here
        // instead of once per (64 hypotheses, 128 points) block there
        double PL[12], PR[12];
        make_projections(Rt, prm.cam, PL, PR);
        Hyp32 h;
        make_hyp32(PL, PR, h);
        float *o = hyp32 + row * kHyp32Floats * kHypPerRound + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 12; ++i) { o[i * kHypPerRound] = h.PL[i]; o[(12 + i) * kHypPerRound] = h.PR[i]; }
        o[24 * kHypPerRound] = h.A3xy; o[25 * kHypPerRound] = h.Atxy; o[26 * kHypPerRound] = h.A3w;
        o[27 * kHypPerRound] = h.Atw;  o[28 * kHypPerRound] = h.cR;
        o[29 * kHypPerRound] = __int_as_float((s4[0] & 0xFFFF) | (s4[1] << 16));
        o[30 * kHypPerRound] = __int_as_float((s4[2] & 0xFFFF) | (s4[3] << 16));
        if (cpr_own != nullptr) {
          const float *mR = cpr_own + 2 * (size_t)obase[p];
          const double thr2 = threshold_squared(prm.thresh);
          // straight-line: the eight points' loads go out together (indices clamped, results
          // of pairs seen before or past the end dropped)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r = (s4[k] & 0xFFFF) >> 1;   // as the counting launch reads it back
            bool fresh = true;
#pragma unroll
            for (int j = 0; j < k; ++j) fresh = fresh && ((s4[j] & 0xFFFF) >> 1) != r;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
              const size_t g = (size_t)min(2 * r + pt, n - 1);
              // the staged form of the point (stage_pair), so that the decision is
              // inlier_exact_staged's, as for any point the screen leaves undecided
              float f[13];
              f[0] = X[3 * g]; f[2] = X[3 * g + 1]; f[4] = X[3 * g + 2];
              f[6] = mL[2 * g]; f[8] = mL[2 * g + 1];
              f[10] = mR[2 * g]; f[12] = mR[2 * g + 1];
              const bool in = inlier_exact_staged<true>(f, PL, PR, prm.thresh, thr2);
              own += (fresh && 2 * r + pt < n && in) ? 1 : 0;
            }
          }
        }
      }
    }
  }
  counts[row * kHypPerRound + threadIdx.x] = active ? (valid ? own : -1) : -2;
}

// grid (point spans, slots * kHypPerRound / 64, pairs), one wavefront per block; a span is
// `span_points` correspondences (even), staged in LDS kPtsPerChunk at a time.  The first
// rounds, which most pairs need, use one chunk per block for parallelism; later
// rounds, which few pairs reach, use one block per 64 hypotheses so that a launch
// with nothing left to do costs a few thousand empty blocks instead of tens of
// thousands.
// MODE: 0 = fp64 only, 1 = packed-fp32 screen, 2 = the same for a rectified pair, 3 = as 2 for pairs
// that are mostly unrelated (count_chunk_inliers_sparse; needs the solve launch's hyp32 table).
constexpr int kWideRounds = 3;   // rounds whose counting launch has one block per chunk (calls without a merged tail)
// Rounds of a front-end call before its merged tail (slamhip_internal::g_wide_rounds, 1..3).
static int wide_rounds() {
  const int w = slamhip_internal::g_wide_rounds;
  return w < 1 ? 1 : (w > kWideRounds ? kWideRounds : w);
}
// A counting launch over hypothesis groups 1 .. 3 of a block whose group 0 (hypotheses 0 .. 63 of
// the block) has been counted by the launch in front: the replay of the stopping rule never
// reaches hypothesis 64 of block b when the requirement in force there is at most 256 b + 64 --
// the carried requirement or that of the largest count among the first 64 (the requirement falls
// as the count grows, and a record is a count above all earlier ones) -- or the cap is.  Those
// pairs' other hypotheses need no count: whatever their table entries hold is never looked at
// (replay_round: `reached` is false from there on, and only reached hypotheses set the state).
struct CountAhead {
  int block = -1;          // the block's number among the pair's blocks of 256 hypotheses; < 0: no such check
  int n_iter_max = 0;
  double p_success = 0.0;
};
__device__ __forceinline__ bool stops_within_first_group(const int32_t *__restrict__ row_counts,
                                                         const int32_t *__restrict__ s, int n,
                                                         const CountAhead &ahead) {
  const int at = ahead.block * kHypPerRound + 64;          // the first hypothesis behind the group
  if (ahead.n_iter_max <= at) return true;
  const int best0 = s[1], req0 = s[3];
  int c = __hip_atomic_load(row_counts + (threadIdx.x & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c = max(c, __shfl_xor(c, d, 64));
  if (req0 <= at) return true;
  // (the logarithm as the replaying launches take it)
  return c > best0 && iterations_required(c, n, log(1.0 - ahead.p_success)) <= at;
}

// One wavefront's share: hypotheses [64 * hyp_group, +64) of table row `row` (a block of
// pair p's hypotheses) against the pair's correspondences [span0, span0 + span_len).
template <int MODE>
__device__ __forceinline__ void ransac_count_item(
    double *lds, int p, size_t row, int hyp_group, int span0, int span_len,
    const int32_t *__restrict__ obase, const int32_t *__restrict__ tm_count,
    const float *__restrict__ cx, const float *__restrict__ cpl, const float *__restrict__ cpr,
    const RansacCam &cam, double thresh, const double *__restrict__ rt, int32_t *__restrict__ counts,
    const int32_t *__restrict__ info, const float *__restrict__ hyp32) {
  constexpr bool screened = MODE != 0;
  if (info[8 * (size_t)p + 7]) return;
  const int n = tm_count[p];
  if (span0 >= n) return;
  const int span1 = min(n, span0 + span_len);
  const int hyp = hyp_group * 64 + (int)threadIdx.x;
  int32_t *slot = counts + row * kHypPerRound + hyp;
  // a valid hypothesis' slot starts at 0 and only grows; invalid ones hold -1 / -2
  const bool valid = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0;
  if (!__any(valid)) return;
  // the hypothesis' fp64 projections: the fp64 path's; the screened loop keeps their fp32 image
  // only (as the solve launch left it, or worked out here) and comes back for these for the
  // one point in thousands it cannot decide -- 48 registers less through the loop
  auto projections = [&](double (&PL_)[12], double (&PR_)[12]) {
    double Rt[12];
#pragma unroll
    for (int i = 0; i < 12; ++i)
      Rt[i] = valid ? rt[(row * 12 + i) * kHypPerRound + hyp] : 0.0;
    make_projections(Rt, cam, PL_, PR_);
  };
  double PL[12], PR[12];
  Hyp32 h32;
  int s4[4] = {0, 0, 0, 0};   // MODE 3: the hypothesis' sample points (indices within the pair)
  if (!screened) {
    projections(PL, PR);
  } else if (hyp32 != nullptr) {
    const float *o = hyp32 + row * kHyp32Floats * kHypPerRound + hyp;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      h32.PL[i] = valid ? o[i * kHypPerRound] : 0.0f;
      h32.PR[i] = (MODE == 1 && valid) ? o[(12 + i) * kHypPerRound] : 0.0f;
    }
    if (MODE == 3) {
      const unsigned s01 = valid ? __float_as_uint(o[29 * kHypPerRound]) : 0u;
      const unsigned s23 = valid ? __float_as_uint(o[30 * kHypPerRound]) : 0u;
      s4[0] = (int)(s01 & 0xFFFFu); s4[1] = (int)(s01 >> 16); s4[2] = (int)(s23 & 0xFFFFu); s4[3] = (int)(s23 >> 16);
    }
    h32.A3xy = valid ? o[24 * kHypPerRound] : 0.0f; h32.Atxy = valid ? o[25 * kHypPerRound] : 0.0f;
    h32.A3w = valid ? o[26 * kHypPerRound] : 0.0f;  h32.Atw = valid ? o[27 * kHypPerRound] : 0.0f;
    h32.cR = valid ? o[28 * kHypPerRound] : 0.0f;
  } else {
    double PL_[12], PR_[12];
    projections(PL_, PR_);
    make_hyp32(PL_, PR_, h32);
  }
  const double thr2 = threshold_squared(thresh);
  float *ldsf = reinterpret_cast<float *>(lds);
  int c = 0;
  for (int p0 = span0; p0 < span1; p0 += kPtsPerChunk) {
    const size_t base = (size_t)obase[p] + p0;
    const int cnt = min(kPtsPerChunk, span1 - p0);
    if (p0 != span0) __syncthreads();
    Screen32 sc = {0.0f, 0.0f, 0.0f};
    unsigned s2max_chunk = 0u;   // MODE 3: bits of the largest staged s^2 of the chunk
    if (screened) {
      unsigned umax = 0u, s2max = 0u;
      for (int j = threadIdx.x; 2 * j < cnt; j += 64) {
        const size_t a = base + 2 * j, b = (2 * j + 1 < cnt) ? a + 1 : a;
        stage_pair(ldsf + j * kPairStride, cx, cpl, cpr, a, b, umax, s2max);
      }
      sc = make_screen(h32, wave_max_u32(umax), (float)thresh, (float)thr2);
      if (MODE == 3) s2max_chunk = wave_max_u32(s2max);
    } else {
      for (int i = threadIdx.x; i < cnt; i += 64) {
        const size_t g = base + i;
        double *o = lds + i * kPtStride;
        o[0] = (double)cx[3 * g]; o[1] = (double)cx[3 * g + 1]; o[2] = (double)cx[3 * g + 2];
        o[3] = (double)cpl[2 * g]; o[4] = (double)cpl[2 * g + 1];
        o[5] = (double)cpr[2 * g]; o[6] = (double)cpr[2 * g + 1];
        o[7] = 0.0;
      }
    }
    __syncthreads();
    if (valid) {
      auto exact = [&](const float *f) {
        double PL_[12], PR_[12];
        projections(PL_, PR_);
        return inlier_exact_staged<MODE >= 2>(f, PL_, PR_, thresh, thr2);
      };
      if (MODE == 3) {
        unsigned long long mine = 0ull;   // staged pairs of this chunk that hold one of the lane's sample points
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int d = s4[k] - p0;
          if (d >= 0 && d < cnt) mine |= 1ull << (d >> 1);
        }
        c += count_chunk_inliers_sparse<true>(ldsf, cnt, h32, sc, thr2, s2max_chunk, mine, exact);
      } else if (screened)
        c += count_chunk_inliers_screened<MODE >= 2>(ldsf, cnt, h32, sc, thr2, exact);
      else
        c += count_chunk_inliers(lds, cnt, PL, PR, thresh, thr2);
    }
  }
  if (valid && c) atomicAdd(slot, c);
}

// The same for hypothesis groups group0 .. 3 of the row over ONE chunk of correspondences (span0,
// at most kPtsPerChunk of them), one wavefront of the block per group (a third of the blocks to
// hand out, which is what a launch costs whose blocks mostly leave at once): the launch behind the
// one that counted group 0 (CountAhead).  Every wavefront stages the chunk (the same values to
// the same places).  Rectified pairs with the solve launch's fp32 table only.
__device__ __forceinline__ void ransac_count_rest_item(
    double *lds, int p, size_t row, int group0, int span0, const int32_t *__restrict__ obase,
    const int32_t *__restrict__ tm_count, const float *__restrict__ cx, const float *__restrict__ cpl,
    const float *__restrict__ cpr, const RansacCam &cam, double thresh, const double *__restrict__ rt,
    int32_t *__restrict__ counts, const int32_t *__restrict__ info, const float *__restrict__ hyp32,
    const CountAhead &ahead) {
  if (info[8 * (size_t)p + 7]) return;
  const int n = tm_count[p];
  if (span0 >= n) return;
  if (ahead.block >= 0 && stops_within_first_group(counts + row * kHypPerRound, info + 8 * (size_t)p, n, ahead))
    return;
  const int cnt = min(kPtsPerChunk, n - span0);
  const size_t base = (size_t)obase[p] + span0;
  const double thr2 = threshold_squared(thresh);
  float *ldsf = reinterpret_cast<float *>(lds);
  unsigned umax = 0u, s2max = 0u;
  const int lane = (int)threadIdx.x & 63;
  for (int j = lane; 2 * j < cnt; j += 64) {
    const size_t a = base + 2 * j, b = (2 * j + 1 < cnt) ? a + 1 : a;
    stage_pair(ldsf + j * kPairStride, cx, cpl, cpr, a, b, umax, s2max);
  }
  umax = wave_max_u32(umax);
  __syncthreads();
  {
    const int g = group0 + ((int)threadIdx.x >> 6);
    const int hyp = g * 64 + lane;
    int32_t *slot = counts + row * kHypPerRound + hyp;
    const bool valid = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0;
    if (!__any(valid)) return;
    const float *o = hyp32 + row * kHyp32Floats * kHypPerRound + hyp;
    Hyp32 h32;
#pragma unroll
    for (int i = 0; i < 12; ++i) { h32.PL[i] = valid ? o[i * kHypPerRound] : 0.0f; h32.PR[i] = 0.0f; }
    h32.A3xy = valid ? o[24 * kHypPerRound] : 0.0f; h32.Atxy = valid ? o[25 * kHypPerRound] : 0.0f;
    h32.A3w = valid ? o[26 * kHypPerRound] : 0.0f;  h32.Atw = valid ? o[27 * kHypPerRound] : 0.0f;
    h32.cR = valid ? o[28 * kHypPerRound] : 0.0f;
    const Screen32 sc = make_screen(h32, umax, (float)thresh, (float)thr2);
    if (valid) {
      auto exact = [&](const float *f) {
        double Rt[12], PL_[12], PR_[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) Rt[i] = rt[(row * 12 + i) * kHypPerRound + hyp];
        make_projections(Rt, cam, PL_, PR_);
        return inlier_exact_staged<true>(f, PL_, PR_, thresh, thr2);
      };
      const int c = count_chunk_inliers_screened<true>(ldsf, cnt, h32, sc, thr2, exact);
      if (c) atomicAdd(slot, c);
    }
  }
}

template <int MODE>
__device__ __forceinline__ void ransac_count_body(
    const int32_t *__restrict__ obase, const int32_t *__restrict__ tm_count,
    const float *__restrict__ cx, const float *__restrict__ cpl, const float *__restrict__ cpr,
    const RansacCam &cam, double thresh, const double *__restrict__ rt, int32_t *__restrict__ counts,
    const int32_t *__restrict__ info, int span_points, int slots, const float *__restrict__ hyp32,
    int n_pairs) {
  __shared__ __align__(16) double lds[kPtsPerChunk * kPtStride];
  // n_pairs > 0: grid (pairs rounded up to a multiple of 8, point spans, slots * kHypPerRound / 64):
  // the pair is the fastest index and the hardware hands consecutive workgroups to consecutive XCDs,
  // so all blocks of a pair run on XCD pair % 8 and its points and hypothesis tables come into one
  // L2.  n_pairs == 0 (calls with few pairs, which dealt by pair would load the XCDs unevenly):
  // grid (point spans, slots * kHypPerRound / 64, pairs), a pair's blocks spread over the chip.
  const bool by_pair = n_pairs > 0;
  const int p = by_pair ? blockIdx.x : blockIdx.z;
  if (by_pair && p >= n_pairs) return;
  const int span = by_pair ? blockIdx.y : blockIdx.x, gy = by_pair ? blockIdx.z : blockIdx.y;
  ransac_count_item<MODE>(lds, p, (size_t)p * slots + gy / (kHypPerRound / 64), gy % (kHypPerRound / 64),
                          span * span_points, span_points, obase,
                          tm_count, cx, cpl, cpr, cam, thresh, rt, counts, info, hyp32);
}

// The merged tail's counting launch: a fixed grid of wavefronts strides over the work the
// solve launch listed -- `list[0]` rows (pair * slots + slot) from list[1] on, each
// kHypPerRound / 64 hypothesis groups x `chunks` chunks of correspondences -- so that pairs
// which stopped long ago cost neither blocks nor time, and the few that go on get the whole
// chip.  The counts are integer sums: the order of the list does not show in them.
template <int MODE>
__device__ __forceinline__ void ransac_count_list_body(
    const int32_t *__restrict__ obase, const int32_t *__restrict__ tm_count,
    const float *__restrict__ cx, const float *__restrict__ cpl, const float *__restrict__ cpr,
    const RansacCam &cam, double thresh, const double *__restrict__ rt, int32_t *__restrict__ counts,
    const int32_t *__restrict__ info, int chunks, int slots, const float *__restrict__ hyp32, int,
    const int32_t *__restrict__ list) {
  __shared__ __align__(16) double lds[kPtsPerChunk * kPtStride];
  constexpr int kGroups = kHypPerRound / 64;
  // chunks any listed pair has correspondences for (the host's `chunks` comes from its row bound)
  chunks = min(chunks, (list[1] + kPtsPerChunk - 1) / kPtsPerChunk);
  const long long n_items = (long long)list[0] * kGroups * chunks;
  for (long long w = blockIdx.x; w < n_items; w += gridDim.x) {
    const int chunk = (int)(w % chunks);
    const int hyp_group = (int)((w / chunks) % kGroups);
    const int row = list[2 + w / ((long long)chunks * kGroups)];
    ransac_count_item<MODE>(lds, row / slots, (size_t)row, hyp_group, chunk * kPtsPerChunk, kPtsPerChunk,
                            obase, tm_count, cx, cpl, cpr, cam, thresh, rt, counts, info, hyp32);
    __syncthreads();   // the staged chunk has been read by every lane before the next item's overwrites it
  }
}

#define SLAMHIP_COUNT_ARGS                                                                         \
  const int32_t *__restrict__ obase, const int32_t *__restrict__ tm_count,                         \
      const float *__restrict__ cx, const float *__restrict__ cpl, const float *__restrict__ cpr,  \
      RansacCam cam, double thresh, const double *__restrict__ rt, int32_t *__restrict__ counts,   \
      const int32_t *__restrict__ info, int span_points, int slots, const float *__restrict__ hyp32, \
      int n_pairs
#define SLAMHIP_COUNT_PASS \
  obase, tm_count, cx, cpl, cpr, cam, thresh, rt, counts, info, span_points, slots, hyp32, n_pairs

__global__ __launch_bounds__(64) void ransac_count_fp64_kernel(SLAMHIP_COUNT_ARGS) {
  ransac_count_body<0>(SLAMHIP_COUNT_PASS);
}
__global__ __launch_bounds__(64) void ransac_count_screened_kernel(SLAMHIP_COUNT_ARGS) {
  ransac_count_body<1>(SLAMHIP_COUNT_PASS);
}
// The rectified-pair loop needs a few registers less than the rare fp64 branch inside
// it; capping the kernel at a quarter of the register file (four wavefronts per SIMD)
// moves those few out of the loop's way.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
void ransac_count_rectified_kernel(SLAMHIP_COUNT_ARGS) {
  ransac_count_body<2>(SLAMHIP_COUNT_PASS);
}
// ... the launch over hypothesis groups group0 .. of a block whose group 0 has been counted
// (NG wavefronts per block: groups group0 .. group0 + NG - 1; NG = 4 with group0 = 0 and no check
// ahead is a whole block of hypotheses, one workgroup per pair and chunk)
template <int NG>
__global__ __launch_bounds__(64 * NG) __attribute__((amdgpu_waves_per_eu(4, 4)))
void ransac_count_rectified_rest_kernel(SLAMHIP_COUNT_ARGS, int group0, CountAhead ahead) {
  // grid (pairs rounded up to a multiple of 8, chunks) with n_pairs > 0, else (chunks, 1, pairs);
  // kHypPerRound / 64 - group0 wavefronts per block; slots == 1, span_points == kPtsPerChunk
  __shared__ __align__(16) double lds[kPtsPerChunk * kPtStride];
  const bool by_pair = n_pairs > 0;
  const int p = by_pair ? blockIdx.x : blockIdx.z;
  if (by_pair && p >= n_pairs) return;
  const int span = by_pair ? blockIdx.y : blockIdx.x;
  ransac_count_rest_item(lds, p, (size_t)p, group0, span * kPtsPerChunk, obase, tm_count, cx, cpl, cpr, cam, thresh,
                         rt, counts, info, hyp32, ahead);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
void ransac_count_rectified_sparse_kernel(SLAMHIP_COUNT_ARGS) {
  ransac_count_body<3>(SLAMHIP_COUNT_PASS);
}
__global__ __launch_bounds__(64) void ransac_count_list_fp64_kernel(SLAMHIP_COUNT_ARGS, const int32_t *__restrict__ list) {
  ransac_count_list_body<0>(SLAMHIP_COUNT_PASS, list);
}
__global__ __launch_bounds__(64) void ransac_count_list_screened_kernel(SLAMHIP_COUNT_ARGS, const int32_t *__restrict__ list) {
  ransac_count_list_body<1>(SLAMHIP_COUNT_PASS, list);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
void ransac_count_list_rectified_kernel(SLAMHIP_COUNT_ARGS, const int32_t *__restrict__ list) {
  ransac_count_list_body<2>(SLAMHIP_COUNT_PASS, list);
}

// A call of a pair or two (the single pass over every hypothesis of the cap, see below) waits for
// two chains in a row: the solve launch's (sample, minimal solver: the longest chain of arithmetic
// in the call) and the counting launch's.  Here every counting wavefront -- 64 hypotheses x one
// span of correspondences -- solves its own 64 hypotheses first, with the function the solve
// launch uses, and counts them straight from registers: one launch and no pose / fp32-image tables
// in between, at the price of every span's wavefront solving the same hypotheses again, which an
// otherwise idle chip does not notice.  The wavefront of span 0 also writes the poses the finishing
// launch picks the winner from; the block (span 0, group 0) writes the pair's start state.
// The counts are added atomically into slots that are zero at launch (cleared by the launch in
// front, temporal_tail_kernel): a hypothesis without a pose, or beyond the cap, keeps 0 where the
// solve launch notes -1 / -2 -- the replay treats all three alike (none can exceed the best so
// far, and the cap bounds the loop separately).  Rectified pairs with the fp32 screen only.
// grid (spans, slots * kHypPerRound / 64, pairs), one wavefront per block.
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(SLAMHIP_P3P_WAVES, SLAMHIP_P3P_WAVES))) void ransac_solve_count_kernel(
    int slots, const int32_t *__restrict__ obase, const int32_t *__restrict__ tm_count,
    const float *__restrict__ cx, const float *__restrict__ cpl, const float *__restrict__ cpr,
    const int32_t *__restrict__ samples, RansacParams prm, double *__restrict__ rt,
    int32_t *__restrict__ counts, double *__restrict__ pose_ransac, int32_t *__restrict__ info,
    int span_points) {
  __shared__ __align__(16) double lds[kPtsPerChunk * kPtStride];
  const int span = blockIdx.x, gy = blockIdx.y, p = blockIdx.z;
  const int slot = gy / (kHypPerRound / 64), hyp_group = gy % (kHypPerRound / 64);
  const size_t row = (size_t)p * slots + slot;
  const int n = tm_count[p];
  const int skip = (n >= 4 && n >= prm.min_points) ? 0 : 1;
  if (span == 0 && gy == 0) {
    int32_t *s = info + 8 * (size_t)p;
    if (threadIdx.x == 0) {
      s[0] = -1; s[1] = 0; s[2] = 0; s[3] = prm.n_iter_max; s[4] = 0; s[5] = n; s[6] = 0;
      s[7] = skip;
    }
    if (threadIdx.x < 12) pose_ransac[12 * (size_t)p + threadIdx.x] = (threadIdx.x % 5 == 0) ? 1.0 : 0.0;  // [I|0]
  }
  const int span0 = span * span_points;
  if (skip || span0 >= n) return;
  const int span1 = min(n, span0 + span_points);
  const int hyp = hyp_group * 64 + (int)threadIdx.x;
  const int it = slot * kHypPerRound + hyp;
  const size_t base0 = (size_t)obase[p];
  double Rt[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) Rt[i] = 0.0;
  bool valid = false;
  if (it < prm.n_iter_max) {
    int s4[4];
    valid = sample_and_solve(cx + 3 * base0, cpl + 2 * base0, n, samples, prm, p, it, s4, Rt);
    if (!valid) {
#pragma unroll
      for (int i = 0; i < 12; ++i) Rt[i] = 0.0;
    }
  }
  if (valid && span == 0)
    for (int i = 0; i < 12; ++i) rt[(row * 12 + i) * kHypPerRound + hyp] = Rt[i];
  if (!__any(valid)) return;
  Hyp32 h32;
  {
    double PL[12], PR[12];
    make_projections(Rt, prm.cam, PL, PR);
    make_hyp32(PL, PR, h32);
  }
  const double thr2 = threshold_squared(prm.thresh);
  float *ldsf = reinterpret_cast<float *>(lds);
  int c = 0;
  for (int p0 = span0; p0 < span1; p0 += kPtsPerChunk) {
    const size_t base = base0 + p0;
    const int cnt = min(kPtsPerChunk, span1 - p0);
    if (p0 != span0) __syncthreads();
    unsigned umax = 0u, s2max = 0u;
    for (int j = threadIdx.x; 2 * j < cnt; j += 64) {
      const size_t a = base + 2 * j, b = (2 * j + 1 < cnt) ? a + 1 : a;
      stage_pair(ldsf + j * kPairStride, cx, cpl, cpr, a, b, umax, s2max);
    }
    const Screen32 sc = make_screen(h32, wave_max_u32(umax), (float)prm.thresh, (float)thr2);
    __syncthreads();
    if (valid) {
      auto exact = [&](const float *f) {
        double PL_[12], PR_[12];
        make_projections(Rt, prm.cam, PL_, PR_);
        return inlier_exact_staged<true>(f, PL_, PR_, prm.thresh, thr2);
      };
      c += count_chunk_inliers_screened<true>(ldsf, cnt, h32, sc, thr2, exact);
    }
  }
  if (valid && c) atomicAdd(counts + row * kHypPerRound + hyp, c);
}

// Replays the hypothesis blocks not replayed yet (the last round's one, or all
// `slots` of a single-pass call, in order), then writes the winner's mask, its
// count and the final info record.  `first_block` = index of the first of them.
// `seg_end` (may be null): seg_end[p] = one past the last mask row that belongs to pair p
// (pairs' segments follow each other); the rows behind the pair's matches are cleared, and the
// launch's last pair also clears the segment after its own when `clear_after_last` is set (the
// last frame of a front-end call, which is the earlier frame of no pair) -- the caller's mask
// array then needs no fill before the call.
// REFINE: the pose refinement on the winner's inliers (refine.hip) follows in the same workgroup,
// for calls whose launches, not arithmetic, bound them; same results as the separate launch.
template <bool REFINE>
__global__ __launch_bounds__(kBlock) void ransac_finish_kernel(
    int first_block, int slots, const int32_t *__restrict__ obase, const int32_t *__restrict__ tm_count,
    const float *__restrict__ cx, const float *__restrict__ cpl, const float *__restrict__ cpr,
    RansacParams prm, const double *__restrict__ rt, const int32_t *__restrict__ counts,
    double *__restrict__ pose_ransac, uint8_t *__restrict__ tm_mask, int32_t *__restrict__ info,
    const int32_t *__restrict__ seg_end, int clear_after_last, Intr intr, int refine_iters,
    double *__restrict__ pose_refined) {
  __shared__ int sh[3];  // winner lane, have a best hypothesis, mask count
  __shared__ double best_rt[12];
  __shared__ RefineLds refine_lds;
  SLAMHIP_STAMP(16);
  const int p = blockIdx.x;
  int32_t *s = info + 8 * (size_t)p;
  const int n = tm_count[p];
  __shared__ int red[4 + kBlock / 64];
  // The pair's record stays in LDS while the blocks are replayed (a round trip through memory per
  // block otherwise), and the loads that do not depend on the outcome are asked for up front: the next
  // block's counts before the current one is replayed, this thread's first correspondences before any.
  __shared__ int32_t state[8];
  if (threadIdx.x < 8) state[threadIdx.x] = s[threadIdx.x];
  const size_t base = (size_t)obase[p];
  const float *X = cx + 3 * base, *mL = cpl + 2 * base, *mR = cpr + 2 * base;
  constexpr int kAhead = 3;               // correspondences per thread fetched ahead (768 a pair)
  float pre[kAhead][7];
#pragma unroll
  for (int j = 0; j < kAhead; ++j) {
    const int i = (int)threadIdx.x + j * kBlock;
    const size_t g = (size_t)(i < n ? i : 0);
#pragma unroll
    for (int c = 0; c < 3; ++c) pre[j][c] = n > 0 ? X[3 * g + c] : 0.0f;
#pragma unroll
    for (int c = 0; c < 2; ++c) { pre[j][3 + c] = n > 0 ? mL[2 * g + c] : 0.0f; pre[j][5 + c] = n > 0 ? mR[2 * g + c] : 0.0f; }
  }
  int winner = -1, winner_slot = 0;
  const double log1mp = log(1.0 - prm.p_success);
  const int32_t *const my_counts = counts + (size_t)p * slots * kHypPerRound + threadIdx.x;
  int c_next = my_counts[0];
  __syncthreads();
  for (int k = 0; k < slots; ++k) {
    if (state[7]) break;   // same value in every thread: written before the barrier below
    const int c = c_next;
    if (k + 1 < slots) c_next = my_counts[(size_t)(k + 1) * kHypPerRound];
    const int w = replay_round(state, counts + ((size_t)p * slots + k) * kHypPerRound, c, first_block + k, n,
                               prm.n_iter_max, log1mp, red);
    if (w >= 0) { winner = w; winner_slot = k; }
    __syncthreads();       // thread 0's state update is visible to the next pass
  }
  SLAMHIP_STAMP(17);
  if (threadIdx.x == 0) {
    sh[0] = winner;
    sh[1] = state[0] >= 0;
    sh[2] = 0;
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    const double v = sh[0] >= 0 ? rt[(((size_t)p * slots + winner_slot) * 12 + threadIdx.x) * kHypPerRound + sh[0]]
                                : pose_ransac[12 * (size_t)p + threadIdx.x];
    best_rt[threadIdx.x] = v;
    pose_ransac[12 * (size_t)p + threadIdx.x] = v;
  }
  __syncthreads();
  double Rt[12], PL[12], PR[12];
  for (int i = 0; i < 12; ++i) Rt[i] = best_rt[i];
  make_projections(Rt, prm.cam, PL, PR);
  const double thr2 = threshold_squared(prm.thresh);
  const bool have = sh[1] != 0;
  auto judge = [&](int i, float x, float y, float z, float ul, float vl, float ur, float vr) {
    bool in = false;
    if (have) {
      const double Xd = (double)x, Yd = (double)y, Zd = (double)z;
      in = inlier_one_camera(PL, Xd, Yd, Zd, (double)ul, (double)vl, prm.thresh, thr2) &&
           inlier_one_camera(PR, Xd, Yd, Zd, (double)ur, (double)vr, prm.thresh, thr2);
    }
    tm_mask[base + i] = in ? 1 : 0;
    if (in) atomicAdd(&sh[2], 1);
  };
#pragma unroll
  for (int j = 0; j < kAhead; ++j) {
    const int i = (int)threadIdx.x + j * kBlock;
    if (i < n) judge(i, pre[j][0], pre[j][1], pre[j][2], pre[j][3], pre[j][4], pre[j][5], pre[j][6]);
  }
  for (int i = (int)threadIdx.x + kAhead * kBlock; i < n; i += kBlock)
    judge(i, X[3 * (size_t)i], X[3 * (size_t)i + 1], X[3 * (size_t)i + 2], mL[2 * (size_t)i], mL[2 * (size_t)i + 1],
          mR[2 * (size_t)i], mR[2 * (size_t)i + 1]);
  if (seg_end != nullptr) {
    const int last = (clear_after_last && p == (int)gridDim.x - 1) ? seg_end[p + 1] : seg_end[p];
    for (size_t i = base + n + threadIdx.x; i < (size_t)last; i += kBlock) tm_mask[i] = 0;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int i = threadIdx.x;
    s[i] = i == 5 ? n : (i == 6 ? sh[2] : (i == 7 ? 0 : state[i]));
  }
  SLAMHIP_STAMP(18);
  if (REFINE)   // the mask rows this workgroup has written come back from memory behind the barrier above
    refine_problem<float>(X, mL, tm_mask + base, n, best_rt, intr, refine_iters, pose_refined + 12 * (size_t)p,
                          (double *)nullptr, refine_lds);
  SLAMHIP_STAMP(19);
}

// A call with few pairs is bound by its dependent launches, not by arithmetic: it
// solves and counts every hypothesis of the cap in one pass (kMaxSlots blocks of
// kHypPerRound at most) and replays them all in the finishing launch -- 3 launches
// instead of 2 per round.  Same samples, same replay order, same outcome.
constexpr int kMaxSlots = 16;
constexpr int kPairsPerXcdDeal = 64;     // pairs from which the counting launch deals its blocks to the XCDs by pair
constexpr int kTailGrid = 4096;         // wavefronts of the merged tail's counting launch: 4 per SIMD
constexpr int kSinglePassPairs = 64;    // measured break-even against the adaptive rounds at ~40 % inliers
constexpr int kSmallCallBlocks = 2048;  // one-wavefront blocks the consensus count of a call of a few pairs aims at
constexpr int kFusedSolveCountPairs = 2;   // pairs up to which the counting launch solves its own hypotheses

int single_pass_slots(int n_pairs, int n_iter_max) {
  const int slots = slamhip_ceil_div(n_iter_max > 0 ? n_iter_max : 1, kHypPerRound);
  const int mode = slamhip_internal::g_ransac_mode;      // 0 auto, 1 rounds always, 2 rounds with the merged tail
  if (slots < 2 || slots > kMaxSlots || mode == 1 || mode == 2) return 1;
  if (n_pairs > kSinglePassPairs) return 1;
  return slots;
}

// The rounds after the first wide_rounds() reach few pairs (those with a small consensus set):
// each would be two launches that find nearly nothing to do.  With `merge_tail` they run as ONE
// multi-slot launch pair without replays in between (as the single pass does), behind a launch
// that only replays the last wide round; the finishing launch replays the tail's blocks in
// order, so the outcome is that of the rounds one by one.  Its pose / count tables have a row
// per (pair, tail block).
int tail_slots(int n_iter_max, bool merge_tail) {
  const int rounds = slamhip_ceil_div(n_iter_max > 0 ? n_iter_max : 1, kHypPerRound);
  return (merge_tail && rounds >= wide_rounds() + 2) ? rounds - wide_rounds() : 0;
}

// Where the tables of a call lie in its workspace: pose / count / fp32-image tables of
// `rows` rows (a row = one block of kHypPerRound hypotheses of one pair), then the merged
// tail's pose / count tables of `tail_rows` rows and its work list.  Every offset grows with
// the row counts, so a call lays its own, smaller tables out inside a workspace sized for the
// caller's largest call.
struct RansacTables {
  size_t rt, counts, hyp32, rt_tail, counts_tail, list, total;
};
RansacTables ransac_tables(size_t rows, size_t tail_rows) {
  RansacTables t;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = slamhip_align_up(o + bytes, 256); return at; };
  t.rt = take(rows * 12 * kHypPerRound * sizeof(double));
  t.counts = take(rows * kHypPerRound * sizeof(int32_t));
  t.hyp32 = take(rows * kHyp32Floats * kHypPerRound * sizeof(float));
  t.rt_tail = take(tail_rows * 12 * kHypPerRound * sizeof(double));
  t.counts_tail = take(tail_rows * kHypPerRound * sizeof(int32_t));
  t.list = take((tail_rows + 2) * sizeof(int32_t));
  t.total = o;
  return t;
}

// Sized for any call with at most n_pairs pairs (a caller may run its last, shorter
// batch through the same workspace, and a short batch may take the single pass).
size_t ransac_workspace_bytes(int n_pairs, int n_iter_max, bool merge_tail = false) {
  const size_t g = (size_t)(n_pairs < kPairGroup ? (n_pairs > 0 ? n_pairs : 1) : kPairGroup);
  size_t rows = g;
  const int most = n_pairs < kSinglePassPairs ? n_pairs : kSinglePassPairs;   // largest single-pass call
  const size_t single = (size_t)most * single_pass_slots(most, n_iter_max);
  if (single > rows) rows = single;
  return ransac_tables(rows, g * (size_t)tail_slots(n_iter_max, merge_tail)).total;
}

// RANSAC for pairs [0, n_pairs) described by obase / tm_count; `prm.pair0` is
// the first pair's index for sampling.  max_points bounds tm_count.
// Correspondences per one-wavefront counting block of a single-pass call (see run_pair_ransac).
int small_call_span(int np, int groups, int max_points) {
  const long long work = (long long)np * groups * (max_points > 0 ? max_points : 1);
  int sp = (int)((work / kSmallCallBlocks + 31) / 32 * 32);
  if (slamhip_internal::g_small_call_span > 0) sp = slamhip_internal::g_small_call_span;
  if (sp < 32) sp = 32;
  return sp < kPtsPerChunk ? sp : kPtsPerChunk;
}

// Whether a call of n_pairs pairs may solve its hypotheses inside the counting launch
// (ransac_solve_count_kernel): the single pass, the fp32 screen of a rectified pair, and the knob.
bool fused_solve_count_applies(int n_pairs, const RansacParams &prm) {
  return slamhip_internal::g_small_call_fusion != 0 && n_pairs <= kFusedSolveCountPairs &&
         single_pass_slots(n_pairs, prm.n_iter_max) > 1 &&
         screening_mode(slamhip_internal::g_ransac_screen, prm.cam.K, prm.cam.MR, prm.thresh) == 2 &&
         slamhip_internal::g_ransac_screen != 3;
}

// The count table of a single-pass call of n_pairs pairs in its workspace, and its length in
// entries (for the launch that clears it ahead of ransac_solve_count_kernel).
int32_t *single_pass_counts(void *ws, int n_pairs, int n_iter_max, size_t *entries) {
  const int slots = single_pass_slots(n_pairs, n_iter_max);
  const RansacTables T = ransac_tables((size_t)n_pairs * slots, 0);
  *entries = (size_t)n_pairs * slots * kHypPerRound;
  return (int32_t *)((uintptr_t)ws + T.counts);
}

// What the finishing launch does beyond the replay and the mask (ransac_finish_kernel).
struct FinishExtras {
  const int32_t *seg_end = nullptr;   // per pair: one past its last mask row; rows behind the matches are cleared
  bool clear_after_last = false;      // the call's last pair also clears the segment behind its own
  bool refine = false;                // the refinement follows in the same launch ...
  Intr intr;                          // ... with these intrinsics,
  int refine_iters = 0;               // this pass limit,
  double *pose_refined = nullptr;     // into this array (12 doubles a pair)
  bool counts_cleared = false;        // the single pass' count table (single_pass_counts) is zero: the launch
                                      // in front has cleared it, and hypotheses may be solved where they are counted
};

int run_pair_ransac(const int32_t *obase, const int32_t *tm_count, const float *cx, const float *cpl,
                    const float *cpr, const int32_t *samples, RansacParams prm, int n_pairs,
                    int max_points, double *pose_ransac, uint8_t *tm_mask, int32_t *info, void *ws,
                    hipStream_t st, bool merge_tail = false, bool mostly_unrelated = false,
                    const FinishExtras &fx = FinishExtras()) {
  if (n_pairs <= 0) return SLAMHIP_OK;
  const int group = n_pairs < kPairGroup ? n_pairs : kPairGroup;
  const int slots = single_pass_slots(n_pairs, prm.n_iter_max);
  const int tail = (slots > 1 || slamhip_internal::g_ransac_mode == 1) ? 0 : tail_slots(prm.n_iter_max, merge_tail);
  const RansacTables T = ransac_tables((size_t)group * slots, (size_t)group * tail);
  double *rt = (double *)((uintptr_t)ws + T.rt);
  int32_t *counts = (int32_t *)((uintptr_t)ws + T.counts);
  double *rt_tail = (double *)((uintptr_t)ws + T.rt_tail);
  int32_t *counts_tail = (int32_t *)((uintptr_t)ws + T.counts_tail);
  int32_t *list = (int32_t *)((uintptr_t)ws + T.list);
  const int wide = wide_rounds();
  const int rounds = slots > 1 ? 1 : (tail ? wide : slamhip_ceil_div(prm.n_iter_max, kHypPerRound));
  const int chunks = slamhip_ceil_div(max_points > 0 ? max_points : 1, kPtsPerChunk);
  const long long pair0 = prm.pair0;
  const int mode = screening_mode(slamhip_internal::g_ransac_screen, prm.cam.K, prm.cam.MR, prm.thresh);
  // the screened counting launches take each hypothesis' fp32 image from the solve launch (the
  // merged tail, which few pairs reach, works it out itself and needs no table)
  float *hyp32 = mode != 0 ? (float *)((uintptr_t)ws + T.hyp32) : nullptr;
  for (int g0 = 0; g0 < n_pairs; g0 += group) {
    const int np = n_pairs - g0 < group ? n_pairs - g0 : group;
    prm.pair0 = pair0 + g0;
    // tune knob 2: 1 = the sparse count where the caller expects mostly unrelated pairs,
    // 2 = never, 3 = always
    const int scr = slamhip_internal::g_ransac_screen;
    const bool sparse = mode == 2 && max_points <= kSampleIndexMax && (scr == 3 || (scr == 1 && mostly_unrelated));
    auto count = sparse ? ransac_count_rectified_sparse_kernel
                 : mode == 2 ? ransac_count_rectified_kernel
                 : mode == 1 ? ransac_count_screened_kernel : ransac_count_fp64_kernel;
    if (fx.counts_cleared && fused_solve_count_applies(n_pairs, prm) && !sparse) {
      // (np == n_pairs: such a call is one group)
      const int groups = slots * (kHypPerRound / 64);
      const int span_points = small_call_span(np, groups, max_points);
      const int spans = slamhip_ceil_div(max_points > 0 ? max_points : 1, span_points);
      hipLaunchKernelGGL(ransac_solve_count_kernel, dim3(spans, groups, np), dim3(64), 0, st, slots, obase + g0,
                         tm_count + g0, cx, cpl, cpr, samples, prm, rt, counts, pose_ransac + 12 * (size_t)g0,
                         info + 8 * (size_t)g0, span_points);
    } else
    for (int r = 0; r < rounds; ++r) {
      hipLaunchKernelGGL(ransac_solve_kernel, dim3(np, slots), dim3(kBlock), 0, st, r, slots, r > 0 ? 1 : 0, 1,
                         obase + g0, tm_count + g0, cx, cpl, samples, prm, rt, counts,
                         pose_ransac + 12 * (size_t)g0, info + 8 * (size_t)g0, (int32_t *)nullptr, hyp32,
                         sparse ? cpr : (const float *)nullptr);
      // narrow only where that still leaves at least a thousand wavefronts of work
      const int per_span = (r < kWideRounds || np * (kHypPerRound / 64) < 1024) ? 1 : chunks;
      // a pair's blocks on one XCD where there are pairs enough to load the eight evenly (measured:
      // 499 pairs of ~600 points 59 -> 41 us per launch; 59 ragged pairs 201 -> 256 us)
      const bool by_pair = np >= kPairsPerXcdDeal;
      const int groups = slots * (kHypPerRound / 64);
      // Correspondences per one-wavefront block: a chunk (or all of them, above).  A call of a pair
      // or two leaves most of the chip idle with that and waits for each wavefront's walk over its
      // 128 correspondences: there a block takes fewer, down to 32, while the launch stays within
      // some two thousand blocks.
      int span_points = per_span * kPtsPerChunk;
      if (slots > 1 && slamhip_internal::g_small_call_fusion) span_points = min(span_points, small_call_span(np, groups, max_points));
      const int spans = slamhip_ceil_div(max_points > 0 ? max_points : 1, span_points);
      // The first round of a long call as two launches: the first 64 hypotheses of every pair, then
      // the other 192 where the stopping rule does not end within those 64 (CountAhead).
      const bool split = r == 0 && slots == 1 && mode == 2 && !sparse && slamhip_internal::g_count_split != 0 &&
                         prm.n_iter_max > 64 && span_points == kPtsPerChunk;
      auto cgrid = [&](int gr) { return by_pair ? dim3((np + 7) / 8 * 8, spans, gr) : dim3(spans, gr, np); };
      // ... and a later round of such a call with one workgroup of four wavefronts per pair and chunk
      const bool whole = r > 0 && slots == 1 && mode == 2 && !sparse && slamhip_internal::g_count_split != 0 &&
                         span_points == kPtsPerChunk;
      if (whole)
        hipLaunchKernelGGL(ransac_count_rectified_rest_kernel<kHypPerRound / 64>, cgrid(1), dim3(kHypPerRound), 0, st,
                           obase + g0, tm_count + g0, cx, cpl, cpr, prm.cam, prm.thresh, rt, counts,
                           info + 8 * (size_t)g0, span_points, slots, (const float *)hyp32, by_pair ? np : 0, 0,
                           CountAhead());
      else if (split)
        hipLaunchKernelGGL(ransac_count_rectified_rest_kernel<1>, cgrid(1), dim3(64), 0, st,
                           obase + g0, tm_count + g0, cx, cpl, cpr, prm.cam, prm.thresh, rt, counts,
                           info + 8 * (size_t)g0, span_points, slots, (const float *)hyp32, by_pair ? np : 0, 0,
                           CountAhead());
      else
      hipLaunchKernelGGL(count, cgrid(split ? 1 : groups), dim3(64), 0, st, obase + g0, tm_count + g0, cx, cpl, cpr,
                         prm.cam, prm.thresh, rt, counts, info + 8 * (size_t)g0, span_points, slots,
                         (const float *)hyp32, by_pair ? np : 0);
      if (split) {
        CountAhead ahead;
        ahead.block = r;
        ahead.n_iter_max = prm.n_iter_max;
        ahead.p_success = prm.p_success;
        hipLaunchKernelGGL(ransac_count_rectified_rest_kernel<kHypPerRound / 64 - 1>, cgrid(1),
                           dim3(kHypPerRound - 64), 0, st, obase + g0,
                           tm_count + g0, cx, cpl, cpr, prm.cam, prm.thresh, rt, counts, info + 8 * (size_t)g0,
                           span_points, slots, (const float *)hyp32, by_pair ? np : 0, 1, ahead);
      }
    }
    if (tail) {
      // replay of the last wide round (state and winner pose up to date), then every block left
      hipLaunchKernelGGL(ransac_solve_kernel, dim3(np, 1), dim3(kBlock), 0, st, wide, 1, 1, 0,
                         obase + g0, tm_count + g0, cx, cpl, samples, prm, rt, counts,
                         pose_ransac + 12 * (size_t)g0, info + 8 * (size_t)g0, list, (float *)nullptr,
                         (const float *)nullptr);
      hipLaunchKernelGGL(ransac_solve_kernel, dim3(np, tail), dim3(kBlock), 0, st, wide, tail, 0, 1,
                         obase + g0, tm_count + g0, cx, cpl, samples, prm, rt_tail, counts_tail,
                         pose_ransac + 12 * (size_t)g0, info + 8 * (size_t)g0, list, (float *)nullptr,
                         (const float *)nullptr);
      auto count_list = mode == 2 ? ransac_count_list_rectified_kernel
                        : mode == 1 ? ransac_count_list_screened_kernel : ransac_count_list_fp64_kernel;
      const long long most = (long long)np * tail * (kHypPerRound / 64) * chunks;
      hipLaunchKernelGGL(count_list, dim3((unsigned)(most < kTailGrid ? most : kTailGrid)), dim3(64), 0, st,
                         obase + g0, tm_count + g0, cx, cpl, cpr, prm.cam, prm.thresh, rt_tail, counts_tail,
                         info + 8 * (size_t)g0, chunks, tail, (const float *)nullptr, np, (const int32_t *)list);
    }
    {
      const int32_t *seg_end = fx.seg_end != nullptr ? fx.seg_end + g0 : nullptr;
      const int clear_after = (fx.clear_after_last && g0 + np == n_pairs) ? 1 : 0;
      double *refined = fx.pose_refined != nullptr ? fx.pose_refined + 12 * (size_t)g0 : nullptr;
      auto finish = fx.refine ? ransac_finish_kernel<true> : ransac_finish_kernel<false>;
      hipLaunchKernelGGL(finish, dim3(np), dim3(kBlock), 0, st, tail ? wide : rounds - 1, tail ? tail : slots,
                         obase + g0, tm_count + g0, cx, cpl, cpr, prm, tail ? rt_tail : rt,
                         tail ? counts_tail : counts, pose_ransac + 12 * (size_t)g0, tm_mask,
                         info + 8 * (size_t)g0, seg_end, clear_after, fx.intr, fx.refine_iters, refined);
    }
    SLAMHIP_CHECK_LAUNCH();
  }
  return SLAMHIP_OK;
}

struct Layout {
  int W;
  size_t idx2, dist2, order, packL, packR, knn, tidx2, tdist2, order2, cx, cpl, cpr, tcount, ransac, total;
  size_t ridx2, rdist2, bid;   // cross-check only: the swapped matcher's result, the bids
  size_t knn_bytes;
  bool raw;                    // the stereo matcher reads the caller's descriptor rows: no packed copy
};

constexpr int kInPlaceRows = 1 << 20;  // features per image up to which the stereo matcher reads unpacked rows (no limit in practice)
constexpr int kSmallCallFrames = 3;  // frames up to which a front-end call runs its stages as fused launches
constexpr int kSmallCallRows = 4096; // ... and features per image

Layout make_layout(const slamhip_frontend_args *a) {
  Layout L;
  L.W = slamhip_packed_words(a->desc_bytes);
  const size_t NL = (size_t)a->n_left_total, NR = (size_t)a->n_right_total;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = slamhip_align_up(o + bytes, 256); return at; };
  L.idx2 = take(NL * 8); L.dist2 = take(NL * 8); L.order = take(NL * 4);
  const int F = a->n_frames > 0 ? a->n_frames : 1;
  // frames whose stereo stage this call runs (the others' outputs are in place: first_stereo_frame)
  const int Fs = (a->first_stereo_frame > 0 && a->first_stereo_frame < F) ? F - a->first_stereo_frame : F;
  // In place wherever the matcher's kernel can read the rows as they are (no packed copy of the
  // descriptors, no packing launch).  Measured per pass, in place against packed first, with the
  // loader that needs no shift: 60 frames of 2000-8000 features 0.763 / 0.773 ms, 100 of 5000
  // 0.804 / 0.801, 30 of 8000 0.789 / 0.797.
  L.raw = a->max_left <= kInPlaceRows && a->max_right <= kInPlaceRows &&
          slamhip_internal::knn2_reads_raw(a->n_left_total, a->n_right_total, Fs, a->max_left,
                                           a->max_right, a->desc_bytes) &&
          (!a->stereo_cross_check ||
           slamhip_internal::knn2_reads_raw(a->n_right_total, a->n_left_total, Fs, a->max_right,
                                            a->max_left, a->desc_bytes));
  L.packL = take(L.raw ? 0 : NL * L.W * 4); L.packR = take(L.raw ? 0 : NR * L.W * 4);
  const size_t k1 = slamhip_internal::knn2_workspace_bytes(a->n_left_total, a->n_right_total, Fs,
                                                           a->max_left, a->max_right, a->desc_bytes, -1, !L.raw);
  const size_t k2 = slamhip_internal::knn2_workspace_bytes(a->n_left_total, a->n_left_total,
                                                           F > 1 ? F - 1 : 1, a->max_left, a->max_left,
                                                           a->desc_bytes, -1, true);
  L.knn_bytes = k1 > k2 ? k1 : k2;
  L.ridx2 = L.rdist2 = L.bid = 0;
  if (a->stereo_cross_check || a->temporal_cross_check) {
    const size_t k3 = slamhip_internal::knn2_workspace_bytes(a->n_right_total, a->n_left_total, Fs,
                                                             a->max_right, a->max_left, a->desc_bytes, -1, !L.raw);
    if (k3 > L.knn_bytes) L.knn_bytes = k3;
    const size_t rows = NL > NR ? NL : NR;
    L.ridx2 = take(rows * 8); L.rdist2 = take(rows * 8); L.bid = take(NL * 4);
  }
  L.knn = take(L.knn_bytes);
  L.tidx2 = take(NL * 8); L.tdist2 = take(NL * 8); L.order2 = take(NL * 4);
  L.cx = take(NL * 12); L.cpl = take(NL * 8); L.cpr = take(NL * 8);
  L.tcount = take((size_t)F * 4);
  L.ransac = take(ransac_workspace_bytes(F - 1, a->n_iter_max, true));
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
  if (!a || a->struct_size != sizeof(slamhip_frontend_args) || a->n_frames < 1 || a->n_left_total < 0 || a->n_right_total < 0 || a->desc_bytes < 1 ||
      a->desc_bytes > SLAMHIP_MAX_DESC_BYTES)
    return 0;
  return make_layout(a).total;
}

int slamhip_frontend_batch(const slamhip_frontend_args *a, void *stream) {
  SLAMHIP_BEGIN();
  if (!a || a->struct_size != sizeof(slamhip_frontend_args)) return SLAMHIP_EINVAL;   // built against another header
  if (a->n_frames < 1 || a->n_left_total < 0 || a->n_right_total < 0) return SLAMHIP_EINVAL;
  if (a->desc_bytes < 1 || a->desc_stride < a->desc_bytes) return SLAMHIP_EINVAL;
  if (a->desc_bytes > SLAMHIP_MAX_DESC_BYTES || a->n_frames > 65535 ||
      a->max_left > SLAMHIP_MAX_TRAIN_ROWS || a->max_right > SLAMHIP_MAX_TRAIN_ROWS)
    return SLAMHIP_EUNSUPPORTED;
  if (a->max_left < 0 || a->max_right < 0 || a->n_iter_max < 1 || a->refine_iters < 0)
    return SLAMHIP_EINVAL;
  if (a->first_stereo_frame < 0 || a->first_stereo_frame >= a->n_frames) return SLAMHIP_EINVAL;
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
  // the stereo stage runs for frames [s0, F); the earlier frames' stereo outputs are in place
  const int s0 = a->first_stereo_frame, Fs = F - s0;
  const int32_t *const s_off_left = a->d_off_left + s0, *const s_off_right = a->d_off_right + s0;
  int32_t *const s_st_count = a->d_st_count + s0;
  int rc;

  if (a->n_left_total == 0 || a->max_left == 0) {
    if (hipMemsetAsync(s_st_count, 0, sizeof(int32_t) * Fs, st) != hipSuccess) return SLAMHIP_ELAUNCH;
    if (F > 1) {
      if (hipMemsetAsync(a->d_ransac_info, 0, sizeof(int32_t) * 8 * (F - 1), st) != hipSuccess)
        return SLAMHIP_ELAUNCH;
    }
    return SLAMHIP_OK;
  }

  // ---- stereo stage
  // The stereo matcher reads the caller's rows as they are where its kernel can; otherwise
  // from a packed copy (W dwords a row, zero padded) made here.
  const uint8_t *rowsL = a->d_des_left, *rowsR = a->d_des_right;
  int rows_stride = a->desc_stride;
  if (!L.raw) {
    rc = slamhip_internal::pack_rows(a->d_des_left, a->n_left_total, packL, a->d_des_right,
                                     a->n_right_total, packR, a->desc_bytes, a->desc_stride, st);
    if (rc) return rc;
    rowsL = (const uint8_t *)packL; rowsR = (const uint8_t *)packR;
    rows_stride = W * 4;
  }
  int32_t *ridx2 = (int32_t *)(base + L.ridx2), *rdist2 = (int32_t *)(base + L.rdist2);
  uint32_t *bid = (uint32_t *)(base + L.bid);
  // A call of a few frames runs its stages as fused launches (stereo_tail_kernel,
  // temporal_tail_kernel, refinement behind the RANSAC replay): same results, a shorter chain.
  const bool fused_sort = slamhip_internal::sort_choice() == 0 && a->max_left <= kFusedSortMax;
  // (up to 4096 features an image: measured per call of one pair, fused against separate launches,
  // 300 features 73 / 86 us, 1500 77 / 85, 3000 103 / 108, 6000 157 / 148, 8000 185 / 174 -- with
  // thousands of survivors the one workgroup a frame has in the fused stage is the longer chain)
  const bool small_call = slamhip_internal::g_small_call_fusion != 0 && F <= kSmallCallFrames && fused_sort &&
                          a->max_left <= kSmallCallRows;
  slamhip_internal::Knn2Splits unfolded = {nullptr, 1, 0};
  if (!a->stereo_cross_check) {
    rc = slamhip_internal::knn2(rowsL, a->n_left_total, rowsR,
                                a->n_right_total, a->desc_bytes, rows_stride, !L.raw, s_off_left, nullptr,
                                s_off_right, nullptr, Fs, a->max_left, a->max_right, idx2, dist2,
                                knn_ws, L.knn_bytes, st, nullptr, -1, true, 0, small_call ? &unfolded : nullptr);
    if (rc) return rc;
  } else {
    if (hipMemsetAsync(bid, 0xFF, sizeof(uint32_t) * (size_t)a->n_left_total, st) != hipSuccess)
      return SLAMHIP_ELAUNCH;
    if (a->n_right_total > 0 && a->max_right > 0) {
      rc = slamhip_internal::knn2(rowsR, a->n_right_total, rowsL,
                                  a->n_left_total, a->desc_bytes, rows_stride, !L.raw, s_off_right, nullptr,
                                  s_off_left, nullptr, Fs, a->max_right, a->max_left, ridx2, rdist2,
                                  knn_ws, L.knn_bytes, st, nullptr, -1, true);
      if (rc) return rc;
      hipLaunchKernelGGL(cross_bid_kernel, dim3(slamhip_ceil_div(a->max_right, kBlock), Fs), dim3(kBlock),
                         0, st, s_off_right, (const int32_t *)nullptr, s_off_left, ridx2, rdist2,
                         bid);
    }
    hipLaunchKernelGGL(cross_decode_kernel, dim3(slamhip_ceil_div(a->n_left_total, kBlock)),
                       dim3(kBlock), 0, st, bid, a->n_left_total, idx2, dist2);
    SLAMHIP_CHECK_LAUNCH();
  }
  // sort by distance + filter + compaction: one launch where a frame's matches fit a block's LDS
  // (and no sort algorithm is forced), else the sort's launch and the compaction's
  CamPair cam;
  for (int i = 0; i < 12; ++i) { cam.PL[i] = a->P_left[i]; cam.PR[i] = a->P_right[i]; }
  const size_t raw_bytes = (size_t)(a->n_left_total - 1) * a->desc_stride + a->desc_bytes;
  if (small_call) {
    auto k = a->max_left <= 2048 ? stereo_tail_kernel<2048> : stereo_tail_kernel<kSmallCallRows>;
    hipLaunchKernelGGL(k, dim3(Fs), dim3(kCompactBlock), 0, st, a->d_kp_left, a->d_kp_right, s_off_left,
                       s_off_right, idx2, dist2, unfolded.part, unfolded.splits, unfolded.rows,
                       a->pixel_thresh, a->disparity_min, s_st_count, a->d_st_left, a->d_st_right,
                       a->d_st_dist, a->d_st_pix_left, a->d_st_pix_right, packL, W,
                       L.raw ? a->d_des_left : nullptr, a->desc_stride, a->desc_bytes, raw_bytes, cam,
                       a->d_st_xyz, a->d_st_desc);
  } else if (fused_sort) {
    auto k = a->max_left <= 2048 ? stereo_sort_compact_kernel<2048> : stereo_sort_compact_kernel<kFusedSortMax>;
    hipLaunchKernelGGL(k, dim3(Fs), dim3(kCompactBlock), 0, st, a->d_kp_left, a->d_kp_right, s_off_left,
                       s_off_right, idx2, dist2, a->pixel_thresh, a->disparity_min, s_st_count,
                       a->d_st_left, a->d_st_right, a->d_st_dist, a->d_st_pix_left, a->d_st_pix_right);
  } else {
    rc = slamhip_internal::sort_by_distance(dist2, 2, s_off_left, nullptr, Fs, a->max_left, order, st);
    if (rc) return rc;
    hipLaunchKernelGGL(stereo_compact_kernel, dim3(Fs), dim3(kCompactBlock), 0, st, a->d_kp_left,
                       a->d_kp_right, s_off_left, s_off_right, idx2, dist2, order,
                       a->pixel_thresh, a->disparity_min, s_st_count, a->d_st_left, a->d_st_right,
                       a->d_st_dist, a->d_st_pix_left, a->d_st_pix_right);
  }
  if (!small_call)
    hipLaunchKernelGGL(stereo_points_kernel, dim3(slamhip_ceil_div(a->max_left, kBlock), Fs), dim3(kBlock),
                       0, st, s_off_left, s_st_count, a->d_st_left, a->d_st_pix_left,
                       a->d_st_pix_right, packL, W, L.raw ? a->d_des_left : nullptr, a->desc_stride,
                       a->desc_bytes, raw_bytes, cam, a->d_st_xyz, a->d_st_desc);
  SLAMHIP_CHECK_LAUNCH();
  if (F < 2) return SLAMHIP_OK;

  // ---- temporal stage: pair p = frames (p, p+1)
  RansacParams prm;
  for (int i = 0; i < 9; ++i) prm.cam.K[i] = a->K[i];
  for (int i = 0; i < 12; ++i) prm.cam.MR[i] = a->M_right[i];
  if (!invert3x3(a->K, prm.kinv.m)) return SLAMHIP_EINVAL;
  prm.thresh = a->ransac_thresh;
  prm.p_success = a->p_success;
  prm.n_iter_max = a->n_iter_max;
  prm.min_points = 4;
  prm.seed = a->seed;
  prm.pair0 = 0;
  prm.seq0 = a->first_pair;
  prm.pair_ids = nullptr;
  // a call of a pair or two solves its hypotheses inside the counting launch, which only adds to its
  // count table: the temporal tail clears the table on its way
  size_t clear_entries = 0;
  int32_t *clear = nullptr;
  if (small_call && fused_solve_count_applies(F - 1, prm))
    clear = single_pass_counts((void *)(base + L.ransac), F - 1, prm.n_iter_max, &clear_entries);
  if (!a->temporal_cross_check) {
    rc = slamhip_internal::knn2((const uint8_t *)a->d_st_desc, a->n_left_total,
                                (const uint8_t *)a->d_st_desc, a->n_left_total, a->desc_bytes, W * 4,
                                true, a->d_off_left, a->d_st_count, a->d_off_left + 1,
                                a->d_st_count + 1, F - 1, a->max_left, a->max_left, tidx2, tdist2, knn_ws,
                                L.knn_bytes, st, nullptr, -1, true, 0, small_call ? &unfolded : nullptr);
    if (rc) return rc;
  } else {
    // the inliers of frame p+1 choose among those of frame p, then bid for them
    if (hipMemsetAsync(bid, 0xFF, sizeof(uint32_t) * (size_t)a->n_left_total, st) != hipSuccess)
      return SLAMHIP_ELAUNCH;
    rc = slamhip_internal::knn2((const uint8_t *)a->d_st_desc, a->n_left_total,
                                (const uint8_t *)a->d_st_desc, a->n_left_total, a->desc_bytes, W * 4,
                                true, a->d_off_left + 1, a->d_st_count + 1, a->d_off_left,
                                a->d_st_count, F - 1, a->max_left, a->max_left, ridx2, rdist2, knn_ws,
                                L.knn_bytes, st, nullptr, -1, true);
    if (rc) return rc;
    hipLaunchKernelGGL(cross_bid_kernel, dim3(slamhip_ceil_div(a->max_left, kBlock), F - 1), dim3(kBlock),
                       0, st, a->d_off_left + 1, a->d_st_count + 1, a->d_off_left, ridx2, rdist2, bid);
    hipLaunchKernelGGL(cross_decode_kernel, dim3(slamhip_ceil_div(a->n_left_total, kBlock)),
                       dim3(kBlock), 0, st, bid, a->n_left_total, tidx2, tdist2);
    SLAMHIP_CHECK_LAUNCH();
  }
  if (small_call) {
    if (a->temporal_cross_check) unfolded = {nullptr, 1, 0};
    auto k = a->max_left <= 2048 ? temporal_tail_kernel<float, 1024, 2048>
                                 : temporal_tail_kernel<float, 1024, kSmallCallRows>;
    hipLaunchKernelGGL(k, dim3(F - 1), dim3(1024), 0, st, a->d_off_left, a->d_st_count, a->d_off_left + 1,
                       a->d_st_count + 1, a->d_off_left, tidx2, tdist2, unfolded.part, unfolded.splits,
                       unfolded.rows, a->d_st_xyz, a->d_st_pix_left, a->d_st_pix_right, a->d_tm_query,
                       a->d_tm_train, a->d_tm_dist, tcount, cx, cpl, cpr, clear, (int)(clear_entries / (size_t)(F - 1)));
  } else if (fused_sort) {
    // many short pairs: 256 threads a block; few long ones: 1024
    auto k = a->max_left <= 2048 ? temporal_sort_gather_kernel<float, 256, 2048>
                                 : temporal_sort_gather_kernel<float, 1024, kFusedSortMax>;
    hipLaunchKernelGGL(k, dim3(F - 1), dim3(a->max_left <= 2048 ? 256 : 1024), 0, st, a->d_off_left,
                       a->d_st_count, a->d_off_left + 1, a->d_st_count + 1, a->d_off_left, tidx2, tdist2,
                       a->d_st_xyz, a->d_st_pix_left, a->d_st_pix_right, a->d_tm_query, a->d_tm_train,
                       a->d_tm_dist, tcount, cx, cpl, cpr);
  } else {
    rc = slamhip_internal::sort_by_distance(tdist2, 2, a->d_off_left, a->d_st_count, F - 1, a->max_left,
                                            order2, st);
    if (rc) return rc;
    hipLaunchKernelGGL(temporal_gather_kernel<float>,
                       dim3(slamhip_ceil_div(a->max_left, kBlock), F - 1), dim3(kBlock), 0, st,
                       a->d_off_left, a->d_st_count, a->d_off_left + 1, a->d_st_count + 1, a->d_off_left,
                       tidx2, tdist2, order2, a->d_st_xyz, a->d_st_pix_left, a->d_st_pix_right,
                       a->d_tm_query, a->d_tm_train, a->d_tm_dist, tcount, cx, cpl, cpr);
  }
  SLAMHIP_CHECK_LAUNCH();

  // the finishing launch leaves no mask row of the call unwritten (the caller's array needs no fill)
  // and, in a call of a few frames, goes on to refine the winner
  FinishExtras fx;
  fx.seg_end = a->d_off_left + 1;
  fx.clear_after_last = true;
  fx.refine = small_call;
  for (int i = 0; i < 9; ++i) fx.intr.K[i] = a->K[i];
  fx.refine_iters = a->refine_iters;
  fx.pose_refined = a->d_pose_refined;
  fx.counts_cleared = clear != nullptr;
  rc = run_pair_ransac(a->d_off_left, tcount, cx, cpl, cpr, a->d_samples, prm, F - 1, a->max_left,
                       a->d_pose_ransac, a->d_tm_mask, a->d_ransac_info, (void *)(base + L.ransac), st,
                       /*merge_tail=*/true, /*mostly_unrelated=*/false, fx);
  if (rc || small_call) return rc;

  rc = slamhip_internal::refine_pose(cx, 0, cpl, a->d_tm_mask, a->d_off_left, tcount, F - 1, a->K,
                                     a->d_pose_ransac, a->refine_iters, a->d_pose_refined, nullptr, st);
  return rc;
}

// ---------------------------------------------------------------------------
// Pair verification (loop-closure candidates).
// ---------------------------------------------------------------------------
constexpr int kVerifyChunk = 32768;   // pairs per group of launches (grid z limit is 65535)

// The pair list runs in chunks of kVerifyChunk pairs plus a shorter last chunk.  Left to
// itself the matcher would pick more train splits for a launch with fewer pairs, and
// its partial buffer (splits x output rows) would outgrow a workspace sized for the
// full chunk; every chunk of a call therefore runs with the split count of the
// largest one.
static int verify_chunk(const slamhip_verify_args *a) {
  return a->n_pairs < kVerifyChunk ? (a->n_pairs > 0 ? a->n_pairs : 1) : kVerifyChunk;
}
static int verify_splits(const slamhip_verify_args *a, bool swapped) {
  return slamhip_internal::knn2_plan_splits(verify_chunk(a), swapped ? a->max_t : a->max_q,
                                            swapped ? a->max_q : a->max_t, a->desc_bytes);
}
// workspace of the matcher calls of a chunk of pairs (forward, and swapped when cross-checking)
static size_t verify_knn_bytes(const slamhip_verify_args *a) {
  const int chunk = verify_chunk(a);
  size_t k = slamhip_internal::knn2_workspace_bytes(a->n_rows_total, a->n_rows_total, chunk, a->max_q,
                                                    a->max_t, a->desc_bytes, (long long)a->n_out_total, true,
                                                    verify_splits(a, false));
  if (a->cross_check) {
    const size_t k2 = slamhip_internal::knn2_workspace_bytes(
        a->n_rows_total, a->n_rows_total, chunk, a->max_t, a->max_q, a->desc_bytes,
        (long long)(a->n_tout_total > 0 ? a->n_tout_total : 0), true, verify_splits(a, true));
    if (k2 > k) k = k2;
  }
  return k;
}

size_t slamhip_verify_workspace_bytes(const slamhip_verify_args *a) {
  if (!a || a->struct_size != sizeof(slamhip_verify_args) || a->n_rows_total < 0 || a->n_pairs < 0 || a->n_out_total < 0 || a->desc_bytes < 1 ||
      a->desc_bytes > SLAMHIP_MAX_DESC_BYTES)
    return 0;
  const int W = slamhip_packed_words(a->desc_bytes);
  const size_t N = (size_t)a->n_rows_total, O = (size_t)a->n_out_total;
  const int chunk = verify_chunk(a);
  size_t o = 0;
  auto take = [&](size_t bytes) { o = slamhip_align_up(o + bytes, 256); };
  take(N * W * 4);            // packed descriptors
  take(O * 8); take(O * 8);   // idx2, dist2
  take(O * 4);                // order
  take(O * 12); take(O * 8); take(O * 8);  // cx, cpl, cpr
  take((size_t)(a->n_pairs > 0 ? a->n_pairs : 1) * 4);  // counts
  take(verify_knn_bytes(a));
  take(ransac_workspace_bytes(chunk, a->n_iter_max));
  if (a->cross_check) {
    const size_t TO = (size_t)(a->n_tout_total > 0 ? a->n_tout_total : 0);
    take(TO * 8); take(TO * 8); take(O * 4);   // swapped matcher's result, bids
  }
  return o + 256;
}

int slamhip_verify_pairs(const slamhip_verify_args *a, void *stream) {
  SLAMHIP_BEGIN();
  if (!a || a->struct_size != sizeof(slamhip_verify_args)) return SLAMHIP_EINVAL;   // built against another header
  if (a->n_rows_total < 0 || a->n_pairs < 0 || a->n_out_total < 0) return SLAMHIP_EINVAL;
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
  const int chunk_max = verify_chunk(a);
  const size_t knn_bytes = verify_knn_bytes(a);
  const int splits_fwd = verify_splits(a, false), splits_swp = verify_splits(a, true);
  void *knn_ws = (void *)take(knn_bytes);
  void *ransac_ws = (void *)take(ransac_workspace_bytes(chunk_max, a->n_iter_max));
  int32_t *ridx2 = nullptr, *rdist2 = nullptr;
  uint32_t *bid = nullptr;
  if (a->cross_check) {
    if (a->n_tout_total < 0 || (a->n_pairs > 0 && !a->d_tobase)) return SLAMHIP_EINVAL;
    const size_t TO = (size_t)a->n_tout_total;
    ridx2 = (int32_t *)take(TO * 8);
    rdist2 = (int32_t *)take(TO * 8);
    bid = (uint32_t *)take(O * 4);
  }

  // per keyframe: points of all stored features (loop_closure.py:179-180,237-249: fp64
  // triangulation, later cast to float32 at :209) and packed descriptors
  int rc = slamhip_stereo_filter_triangulate(a->d_kp_left, a->n_rows_total, a->d_kp_right,
                                             a->n_rows_total, 1, nullptr, nullptr, a->n_rows_total,
                                             a->P_left, a->P_right, 0.0, 0.0, nullptr, a->d_xyz, 2,
                                             stream);
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
    if (!a->cross_check) {
      rc = slamhip_internal::knn2((const uint8_t *)packed, a->n_rows_total, (const uint8_t *)packed,
                                  a->n_rows_total, a->desc_bytes, W * 4, true, qb, qc, tb, tc, np,
                                  a->max_q, a->max_t, idx2, dist2, knn_ws, knn_bytes, st, ob,
                                  a->n_out_total, true, splits_fwd);
      if (rc) return rc;
    } else {
      // the rows of the current keyframe choose among the candidate's, then bid for them
      const int32_t *tob = a->d_tobase + c0;
      if (c0 == 0 && hipMemsetAsync(bid, 0xFF, sizeof(uint32_t) * O, st) != hipSuccess)
        return SLAMHIP_ELAUNCH;
      if (a->max_t > 0 && a->n_tout_total > 0) {
        rc = slamhip_internal::knn2((const uint8_t *)packed, a->n_rows_total, (const uint8_t *)packed,
                                    a->n_rows_total, a->desc_bytes, W * 4, true, tb, tc, qb, qc, np,
                                    a->max_t, a->max_q, ridx2, rdist2, knn_ws, knn_bytes, st, tob,
                                    a->n_tout_total, true, splits_swp);
        if (rc) return rc;
        hipLaunchKernelGGL(cross_bid_kernel, dim3(slamhip_ceil_div(a->max_t, kBlock), np), dim3(kBlock), 0,
                           st, tob, tc, ob, ridx2, rdist2, bid);
      }
      hipLaunchKernelGGL(cross_decode_kernel, dim3(slamhip_ceil_div(a->n_out_total, kBlock)), dim3(kBlock),
                         0, st, bid, a->n_out_total, idx2, dist2);
      SLAMHIP_CHECK_LAUNCH();
    }
    rc = slamhip_internal::sort_by_distance(dist2, 2, ob, qc, np, a->max_q, order, st);
    if (rc) return rc;
    hipLaunchKernelGGL(temporal_gather_kernel<double>, dim3(slamhip_ceil_div(a->max_q, kBlock), np),
                       dim3(kBlock), 0, st, qb, qc, tb, tc, ob, idx2, dist2, order,
                       (const float *)a->d_xyz, a->d_kp_left, a->d_kp_right, a->d_tm_query,
                       a->d_tm_train, a->d_tm_dist, tcount + c0, cx, cpl, cpr);
    SLAMHIP_CHECK_LAUNCH();
    prm.seed = a->seed;
    prm.pair0 = 0;
    prm.seq0 = c0;
    prm.pair_ids = a->d_pair_ids != nullptr ? reinterpret_cast<const long long *>(a->d_pair_ids) + c0 : nullptr;
    // a given sample table is indexed by the pair's place in this call's list
    const int32_t *samples = a->d_samples != nullptr ? a->d_samples + (size_t)c0 * a->n_iter_max * 4 : nullptr;
    rc = run_pair_ransac(ob, tcount + c0, cx, cpl, cpr, samples, prm, np, a->max_q,
                         a->d_pose_ransac + 12 * (size_t)c0, a->d_tm_mask,
                         a->d_info + 8 * (size_t)c0, ransac_ws, st, /*merge_tail=*/false,
                         /*mostly_unrelated=*/true);
    if (rc) return rc;
    rc = slamhip_internal::refine_pose(cx, 0, cpl, a->d_tm_mask, ob, tcount + c0, np, a->K,
                                       a->d_pose_ransac + 12 * (size_t)c0, a->refine_iters,
                                       a->d_pose_refined + 12 * (size_t)c0, nullptr, st,
                                       a->d_info + 8 * (size_t)c0, a->min_points);
    if (rc) return rc;
  }
  return SLAMHIP_OK;
}

}  // extern "C"

#ifdef SLAMHIP_STAMPS
extern "C" int slamhip_debug_stamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)); }
#endif
#ifdef SLAMHIP_DEBUG_SPARSE
extern "C" int slamhip_debug_counters(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(slamhip_dev::g_dbg), 16); }
#endif
