// Shared helpers for the libslamhip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/slamhip.h"

// Kernel family the matcher uses when nothing is forced: 1 = matrix cores, 0 = popcount.
// (8192 x 8192 x 61 B: 17 us against 61 us per top-2 kernel; same results bit for bit.)
#ifndef SLAMHIP_DEFAULT_MFMA
#define SLAMHIP_DEFAULT_MFMA 1
#endif

// hipGetLastError() is sticky per thread: an error left behind by unrelated HIP
// calls of the host application would otherwise be reported as ours.
#define SLAMHIP_BEGIN() ((void)hipGetLastError())

#define SLAMHIP_CHECK_LAUNCH()                         \
  do {                                                 \
    hipError_t e__ = hipGetLastError();                \
    if (e__ != hipSuccess) return SLAMHIP_ELAUNCH;     \
  } while (0)

static inline int slamhip_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t slamhip_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Number of 32-bit words a packed descriptor row occupies: 8, 16 or 32
// (32-, 64- or 128-byte rows; shorter descriptors are zero padded, which is
// neutral under XOR + popcount).
static inline int slamhip_packed_words(int desc_bytes) {
  if (desc_bytes <= 32) return 8;
  if (desc_bytes <= 64) return 16;
  return 32;
}

// Cross-translation-unit entry points (not part of the C ABI).
namespace slamhip_internal {
// 1: consensus counting screens in packed fp32 and re-evaluates undecided points in
// fp64 (same counts), with the sparse scheme (count_chunk_inliers_sparse) where the caller
// says its pairs are mostly unrelated; 2 / 3: the same with that scheme never / always;
// 0: fp64 only.  Set through slamhip_tune(2, value).
extern int g_ransac_screen;
// Pair RANSAC launch scheme: 0 = chosen from the number of pairs (single pass for few
// pairs; three rounds and a merged tail for the front-end's long calls), 1 = always rounds
// of 256 hypotheses, 2 = rounds with the merged tail.  Set through slamhip_tune(4, value).
extern int g_ransac_mode;
// 1 (default): batched matrix-core matcher launches deal their blocks to the XCDs pair by pair;
// 0: plain three-dimensional grid.  Set through slamhip_tune(11, value).
extern int g_xcd_remap;
// Rounds of 256 hypotheses a long front-end call runs as launch pairs over all pairs before
// the rest of the iteration cap goes through the merged tail (1..3, default 2: a third round finds
// next to nothing to do on typical data and still dispatches its whole grid; measured 1-3 % of a
// pass on five workloads).  Same results.  Set through slamhip_tune(13, value), before a workspace
// is sized.
extern int g_wide_rounds;
// Front-end calls of a few frames (frame-at-a-time use): 1 (default) = their stages run as fused
// launches (matcher splits folded by the stage behind the matcher, sort + filter + triangulation in
// one launch per stage, refinement behind the RANSAC replay); 0 = the launches of a long call.
// Same results.  Set through slamhip_tune(14, value).
extern int g_small_call_fusion;
// Correspondences per one-wavefront block of the consensus count of such calls (0 = chosen from
// the sizes).  Set through slamhip_tune(15, value).
extern int g_small_call_span;
// The first round's consensus count of a long call as two launches (1, the default): the first 64
// hypotheses of every pair, then the other 192 for the pairs that the stopping rule does not end
// within those 64 (at the reference's published inlier ratio most pairs need fewer than 64).
// 0 = one launch over all 256.  Same results.  Set through slamhip_tune(16, value).
extern int g_count_split;

// Duration probe of the matrix-core top-2 kernel by the dispatch's own timestamps: while both are
// set, the next launch of that kernel goes out through hipExtLaunchKernelGGL with these two events,
// which then hold the start and the end of the kernel itself (what a kernel trace reports), with
// no event record of its own on the stream.  Set and cleared by the matcher's host code around
// the launch (slamhip_tune(10, 0) selects this probe).
extern hipEvent_t g_dispatch_start, g_dispatch_stop;

// splits > 0 pins the number of train splits (see knn2_plan_splits); 0 = chosen from the sizes.
size_t knn2_workspace_bytes(int nq_total, int nt_total, int n_pairs, int max_nq, int max_nt,
                            int desc_bytes, long long n_out = -1, bool prepacked = false,
                            int splits = 0);
// Whether the matcher reads such a call's rows as the caller stores them (any stride, no packing
// launch, no packed copy in its workspace) when they are not handed over prepacked.
bool knn2_reads_raw(int nq_total, int nt_total, int n_pairs, int max_nq, int max_nt, int desc_bytes);
// Number of train splits the matcher chooses for a launch of n_pairs pairs of at most
// max_nq x max_nt rows.  A caller that runs several launches through one workspace
// passes the value of its largest launch to every one of them.
int knn2_plan_splits(int n_pairs, int max_nq, int max_nt, int desc_bytes);
// Batched top-2 matcher.  prepacked: rows are already W-dword packed (row_stride
// = 4*W, 16-byte aligned).  Pair p owns query rows [q_off[p], q_off[p] + count)
// with count = q_cnt[p] (or q_off[p+1]-q_off[p] when q_cnt is NULL); same for t.
// d_out_off != NULL: the result of query row r of pair p is written at output row
// d_out_off[p] + r of idx2/dist2, which then hold n_out_total rows (the same
// query set may take part in many pairs).
// `unfolded` (may be null): the caller's next kernel folds the train splits itself.  A call that
// ran with more than one split then leaves out its merge launch and reports where the per-split
// results lie: part[split * rows + output row] = (smallest, second smallest) key of the split,
// key = distance << 21 | train row within the pair, 0xFFFFFFFF = none; idx2 / dist2 are not
// written.  With splits == 1 the results are in idx2 / dist2 as usual.
struct Knn2Splits {
  const uint2 *part;
  int splits, rows;
};
int knn2(const uint8_t *d_query, int nq_total, const uint8_t *d_train, int nt_total, int desc_bytes,
         int row_stride, bool prepacked, const int32_t *d_q_off, const int32_t *d_q_cnt,
         const int32_t *d_t_off, const int32_t *d_t_cnt, int n_pairs, int max_nq, int max_nt,
         int32_t *d_idx2, int32_t *d_dist2, void *d_workspace, size_t workspace_bytes,
         hipStream_t st, const int32_t *d_out_off = nullptr, int n_out_total = -1,
         bool best_only = false,    // best_only: second-best entries come back as -1
         int splits = 0, Knn2Splits *unfolded = nullptr);
// Matrix-core matcher (hamming_mfma.hip): same arguments as the popcount kernel's launch.
// W = packed dwords per row, G = groups of 32 queries per wave (1 or 2; 32-word rows: 1),
// NW = waves per block (4, or 8 with G = 2 and rows of at most 16 words).
// arrivals != nullptr (grid.x * grid.z counters, zero at launch) with more than one split: the
// kernel folds the splits itself (the last block to arrive of each query block does) and
// applies the ratio test when keep != nullptr; otherwise the splits go to `part` for the merge kernel.
// Descriptor rows as the caller stores them, for the matrix-core kernel's direct read (no packing
// launch): qp / tp then point at the first row of each array.  Rows are at least 16 bytes long
// (every 16-byte load of the kernel lies inside a row).
struct RawRows {
  int row_stride, desc_bytes;
};
int mfma_max_split_rows(int W);    // train rows one block may scan (width of its row index)
void launch_hamming_mfma(int W, int G, int NW, bool best, dim3 grid, hipStream_t st, const uint32_t *qp,
                         const uint32_t *tp, const int32_t *q_off, const int32_t *q_cnt,
                         const int32_t *t_off, const int32_t *t_cnt, const int32_t *out_off, int nq,
                         int nt, int nq_total, void *part, int32_t *idx2, int32_t *dist2,
                         unsigned *arrivals = nullptr, double ratio = 0.0, uint8_t *keep = nullptr,
                         const RawRows *raw = nullptr);
int pack_rows(const uint8_t *d_a, int n_a, uint32_t *d_dst_a, const uint8_t *d_b, int n_b,
              uint32_t *d_dst_b, int desc_bytes, int row_stride, hipStream_t st);
// The sort algorithm forced through slamhip_tune(5, value): 0 = the library's choice.
int sort_choice();
int sort_by_distance(const int32_t *d_dist, int dist_stride, const int32_t *d_off,
                     const int32_t *d_cnt, int n_segments, int max_n, int32_t *d_order,
                     hipStream_t st);
int refine_pose(const void *d_points, int points_fp64, const float *d_pixels, const uint8_t *d_mask,
                const int32_t *d_off, const int32_t *d_cnt, int n_problems, const double *K,
                const double *d_Rt_init, int max_iters, double *d_Rt_out, double *d_info,
                hipStream_t st, const int32_t *d_ransac_info = nullptr, int min_inliers = 0);
}  // namespace slamhip_internal
