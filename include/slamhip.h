/*
 * slamhip.h - C ABI of libslamhip.so, the MI355X (gfx950) implementation of the
 * stereo-SLAM front-end hot path: Hamming brute-force kNN descriptor matching,
 * rectified-stereo row/disparity filter, 4x4 DLT triangulation and PnP-RANSAC
 * hypothesis consensus scoring.
 *
 * Conventions
 *  - Every pointer named d_* is a DEVICE pointer (hipMalloc / torch ROCm tensor
 *    data_ptr()).  Small parameter blocks (camera matrices) are HOST pointers and
 *    are copied by value into the kernel arguments.
 *  - The caller owns every buffer.  The library allocates no device memory;
 *    kernels that need scratch take a caller-provided workspace whose size is
 *    returned by the matching *_workspace_bytes() function.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *    calls are asynchronous on that stream and return after enqueueing.
 *  - Return value: SLAMHIP_OK (0) or a negative SLAMHIP_E* code.  Nothing throws
 *    across this boundary.  slamhip_strerror() maps a code to text.
 *  - Thread-safety: the compute entry points keep no global mutable state;
 *    calls on different streams/devices may run concurrently.  Exceptions: the
 *    diagnostics entry points slamhip_tune() and slamhip_profile_begin()/_end()
 *    set process-global, unsynchronised state and must not be called while
 *    another thread is inside the library.
 *
 * Each entry point cites the reference interface (GuyShmueli/SLAM, paths
 * relative to the reference root) whose arithmetic it replaces.
 */
#ifndef SLAMHIP_H
#define SLAMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAMHIP_VERSION 103 /* 0.1.3: slamhip_frontend_args.first_stereo_frame (frame-at-a-time calls) */

enum {
  SLAMHIP_OK = 0,
  SLAMHIP_EINVAL = -1,      /* null pointer / negative size / bad flag          */
  SLAMHIP_EUNSUPPORTED = -2,/* size outside what the kernels are built for      */
  SLAMHIP_EWORKSPACE = -3,  /* workspace missing or too small                   */
  SLAMHIP_ELAUNCH = -4      /* HIP runtime reported an error at launch          */
};

/* Limits the kernels are built for. */
#define SLAMHIP_MAX_DESC_BYTES 128      /* descriptor row length in bytes        */
#define SLAMHIP_MAX_TRAIN_ROWS (1 << 21)/* train rows per pair (packed-key width)*/
#define SLAMHIP_MAX_L2_DIM 4096         /* floats per row of the L2 matcher      */

int slamhip_version(void);
const char *slamhip_strerror(int code);

/* Diagnostics only (kernel-variant benchmarks): knob 0 = matcher kernel variant
 * (-1 auto; 0..5 popcount kernels; 8 matrix-core kernel, 9 / 10 the same with
 * 1 / 2 groups of 32 queries per wave), knob 1 = forced number of train splits (0 auto), knob 2 = packed-fp32
 * screening in the consensus counting kernels (1 = on, the default: pair verification, whose
 * pairs are mostly unrelated, dismisses most hypothesis-point combinations by the left image
 * alone and counts each hypothesis' own sample points where it is solved; 2 = on, every
 * combination screened everywhere; 3 = the pair-verification scheme everywhere; 0 = fp64 only;
 * the counts are identical in all four), knob 3 = matcher block size (0 auto, 256 or
 * 128 queries per block), knob 4 = pair RANSAC launch scheme (0 auto: calls with at
 * most 64 pairs solve and count every hypothesis in one pass, longer front-end calls run
 * two rounds of 256 hypotheses (knob 13) and the rest of the cap as one merged launch; 1 always
 * rounds of 256; 2 rounds with the merged tail whatever the pair count; identical results;
 * set it before sizing a workspace), knob 5 = sort
 * algorithm (0 auto: radix passes in LDS where the segments fit, 1 rank counting, 2 bitonic
 * network where the segments fit it; same order), knob 6 / 7 = matrix-core matcher: blocks a launch aims at / train rows
 * per block it does not exceed when blocks abound (0 = default), knob 8 = where the
 * matrix-core matcher folds its train splits (2, the default: inside the launch for small
 * calls, a merge launch for large ones; 1: always inside; 0: always a merge launch; set it
 * before sizing a workspace; same results), knob 9 = the matrix-core matcher reads the
 * caller's rows as they are (1, the default) or rows packed by a launch in front (0), knob 10 =
 * launches of the top-2 kernel per call while the duration probe below is on (1..64, default 1:
 * identical launches back to back between the probe's two events, which spreads the events' own
 * cost; slamhip_profile_end then reports the time of all of them; 0: one launch, and the probe's
 * two events are handed to that launch itself (hipExtLaunchKernelGGL), so that they hold the start and
 * the end of the dispatch - the duration a kernel trace reports - with no event record on the stream
 * (the merge launch behind it is then given events of its own as well, never read: a dispatch with a
 * profiling signal among dispatches without one reads 1-2 us longer); matrix-core kernel only; -2 .. -8: that many launches of the kernel in every call, probe or not, so that a
 * caller can time its step with one more launch in it), knob 11 = batched launches
 * of the matrix-core matcher (64 pairs or more) deal their blocks to the chip's eight XCDs pair
 * by pair, so that a pair's rows come into one L2 (1, the default), or use the plain grid (0);
 * same results; knob 13 = rounds of 256 hypotheses a long front-end call runs over all its pairs
 * before the rest of the iteration cap goes through the merged launch (1..3, default 2; set it before
 * sizing a workspace; identical results); knob 14 = front-end calls of up to three frames of up to 4096
 * features run their stages as fused launches (1, the default: the matcher's train splits folded by
 * the stage behind it, sort + filter + triangulation + gather in one launch per stage, the refinement
 * behind the RANSAC replay) or as the launches of a long call (0); identical results; knob 15 =
 * correspondences per one-wavefront block of the consensus count of such calls (0, the default: from
 * the sizes, 32 for one pair; else an even number); identical results; knob 16 = the first round's
 * consensus count of a long front-end call as two launches (1, the default: the first 64 hypotheses of
 * every pair, then the other 192 for the pairs the stopping rule does not end within those 64; later
 * rounds one workgroup per pair and chunk) or as one launch of one-wavefront workgroups (0); identical
 * results; knob 0 = 12: the matrix-core matcher with four query groups per wave and one wave
 * per SIMD (64-byte rows only; measured slower).  Process-global; not for use while other threads
 * are inside the library. */
int slamhip_tune(int knob, int value);

/* Which kernel family the Hamming matcher uses when knob 0 forces nothing:
 * 1 = matrix cores (FP4 block-scaled MFMA emits the sort key), 0 = xor + popcount on
 * the vector ALU.  Both give the same results bit for bit.  The environment variable
 * SLAMHIP_MATCHER ("mfma" / "popcount"), read once, overrides the built-in default. */
int slamhip_matcher_kind(void);

/* Diagnostics only: kernel-duration probe.  After begin(max_launches) the
 * matcher brackets its top-2 kernel with HIP events recorded on the launch
 * stream.  end() waits for them, stores their summed duration (milliseconds)
 * and returns how many launches were bracketed (negative = error). */
int slamhip_profile_begin(int max_launches);
int slamhip_profile_end(double *total_ms);

/* ------------------------------------------------------------------------- */
/* Hamming brute-force k=2 nearest neighbours.                               */
/* Replaces cv2.BFMatcher(NORM_HAMMING).match / knnMatch(k=2) as called from */
/* frontend/vision/descriptor_matcher.py:23-34 (match, k=1 = column 0 of the */
/* result) and the commented knn_match/ratio_test call site at               */
/* backend/optimizers/loop_closure.py:193-194.                               */
/*                                                                           */
/* For every query row i: the two train rows with the smallest Hamming       */
/* distance, ties resolved towards the LOWEST train index (the order a       */
/* sequential scan with a strict `<` insertion produces).                    */
/*   d_idx2 [nq*2] int32 : train index of best, second best (-1 if absent)   */
/*   d_dist2[nq*2] int32 : their distances in bits          (-1 if absent)   */
/* Rows are `desc_bytes` long (1..128) and `row_stride` bytes apart.         */
/* ------------------------------------------------------------------------- */
/* Workspace size for a call with these totals; max_nq / max_nt are the largest
 * single-pair row counts (= nq / nt for the unbatched call).  0 = bad input. */
size_t slamhip_hamming_knn2_workspace_bytes(int nq_total, int nt_total, int n_pairs,
                                            int max_nq, int max_nt, int desc_bytes);

int slamhip_hamming_knn2(const uint8_t *d_query, int nq, const uint8_t *d_train, int nt,
                         int desc_bytes, int row_stride, int32_t *d_idx2, int32_t *d_dist2,
                         void *d_workspace, size_t workspace_bytes, void *stream);

/* k=2 neighbours and the Lowe ratio test in one call (the ratio decision is
 * taken where the split results are folded; same outputs as
 * slamhip_hamming_knn2 followed by slamhip_ratio_test). */
int slamhip_hamming_knn2_ratio(const uint8_t *d_query, int nq, const uint8_t *d_train, int nt,
                               int desc_bytes, int row_stride, double ratio, int32_t *d_idx2,
                               int32_t *d_dist2, uint8_t *d_keep, void *d_workspace,
                               size_t workspace_bytes, void *stream);

/* The same call for callers that cannot allocate device memory or carry a
 * workspace: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  It has no counterpart in the reference.  Same arguments
 * and results as slamhip_hamming_knn2_ratio minus the workspace: the matcher's
 * scratch is a file-scope device array of SLAMHIP_STATIC_SCRATCH_BYTES inside
 * the library (one copy per GPU, held until the process ends), so the call
 * allocates and frees nothing.  nq and nt may be at most
 * SLAMHIP_STATIC_MAX_ROWS; larger calls return SLAMHIP_EUNSUPPORTED.  All
 * calls on one GPU share that GPU's array: calls that overlap in time on one
 * GPU (different streams or threads) race and must be serialised by the
 * caller.  The first call on a device looks up the array's address
 * (hipGetSymbolAddress); make that call outside stream capture. */
#define SLAMHIP_STATIC_MAX_ROWS 8192
#define SLAMHIP_STATIC_SCRATCH_BYTES (8u << 20)
int slamhip_hamming_knn2_ratio_static(const uint8_t *d_query, int nq, const uint8_t *d_train,
                                      int nt, int desc_bytes, int row_stride, double ratio,
                                      int32_t *d_idx2, int32_t *d_dist2, uint8_t *d_keep,
                                      void *stream);

/* Batched form: n_pairs independent (query set, train set) problems in one
 * launch.  Pair p owns query rows [q_off[p], q_off[p+1]) and train rows
 * [t_off[p], t_off[p+1]) of the concatenated descriptor arrays; train indices
 * in d_idx2 are LOCAL to the pair.  d_q_off / d_t_off are device arrays of
 * n_pairs+1 int32; max_nq / max_nt are host-side upper bounds on any single
 * pair's row counts (they size the grid; a bound that is too small is an
 * error the caller must avoid, it is not detectable on the host). */
int slamhip_hamming_knn2_batched(const uint8_t *d_query, int nq_total, const uint8_t *d_train,
                                 int nt_total, int desc_bytes, int row_stride,
                                 const int32_t *d_q_off, const int32_t *d_t_off, int n_pairs,
                                 int max_nq, int max_nt, int32_t *d_idx2, int32_t *d_dist2,
                                 void *d_workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- */
/* L2 brute-force k=2 nearest neighbours for float32 descriptors.            */
/* Replaces cv2.BFMatcher(NORM_L2).match / knnMatch(k=2): the norm the       */
/* reference selects for descriptors that are not uint8                      */
/* (frontend/vision/descriptor_matcher.py:23-28; SuperPoint's 256 floats     */
/* with matcher_type="bf").                                                  */
/*   distance = sqrt(sum_k (q_k - t_k)^2) as float32: the sum accumulated in */
/*   float32 in index order, products and additions rounded separately, the  */
/*   square root correctly rounded; ties -> lowest train index.              */
/*   d_idx2 [nq*2] int32   : train index of best, second best (-1 if absent) */
/*   d_dist2[nq*2] float32 : their distances                (-1.0 if absent) */
/* Rows are `dim` floats long (1..SLAMHIP_MAX_L2_DIM), `row_stride` floats   */
/* apart.                                                                    */
/* ------------------------------------------------------------------------- */
size_t slamhip_l2_knn2_workspace_bytes(int nq, int nt, int dim);
int slamhip_l2_knn2(const float *d_query, int nq, const float *d_train, int nt, int dim,
                    int row_stride, int32_t *d_idx2, float *d_dist2, void *d_workspace,
                    size_t workspace_bytes, void *stream);
/* The float-distance counterparts of slamhip_cross_check, slamhip_ratio_test and
 * slamhip_sort_by_distance (one segment): crossCheck fold of the train->query result
 * (d_match_idx[nq] = train row or -1, d_match_dist[nq] = distance or -1.0), Lowe ratio
 * test on a k=2 result, and the stable order by (distance, row) with unmatched rows
 * (distance -1.0) last - the `sorted(matches, key=distance)` of descriptor_matcher.py:34. */
int slamhip_l2_cross_check(const int32_t *d_t2q_idx2, const float *d_t2q_dist2, int nt, int nq,
                           int32_t *d_match_idx, float *d_match_dist, void *stream);
int slamhip_ratio_test_f32(const float *d_dist2, int nq, double ratio, uint8_t *d_keep, void *stream);
int slamhip_sort_by_distance_f32(const float *d_dist, int dist_stride, int n, int32_t *d_order,
                                 void *stream);

/* Lowe ratio test on the k=2 result (loop_closure.py:194, `ratio_test(raw, ratio)`):
 * keep[i] = 1 iff a second neighbour exists and
 *           (double)dist2[2i] < ratio * (double)dist2[2i+1]. */
int slamhip_ratio_test(const int32_t *d_dist2, int nq, double ratio, uint8_t *d_keep,
                       void *stream);

/* cv2.BFMatcher(crossCheck=True) semantics (descriptor_matcher.py:26 with
 * cross_check=True).  Input: for every TRAIN row its best QUERY row and distance
 * (the k=1 column of a train->query knn2 call, element stride 2).  Output per
 * query row: among the train rows that chose it, the one with the smallest
 * distance (lowest train index on ties), or -1.  */
int slamhip_cross_check(const int32_t *d_t2q_idx2, const int32_t *d_t2q_dist2, int nt, int nq,
                        int32_t *d_match_idx, int32_t *d_match_dist, void *stream);

/* Stable ordering of matches by (distance, query index): the order of
 * `sorted(matches, key=lambda x: x.distance)` at descriptor_matcher.py:34.
 * d_dist has element stride `dist_stride` (2 to read column 0 of d_dist2).
 * Entries with dist < 0 (no match) sort last.  d_order[r] = query index of the
 * r-th match.  Segmented like the batched matcher when d_off != NULL
 * (d_order values are local to the segment, written at segment offset). */
int slamhip_sort_by_distance(const int32_t *d_dist, int dist_stride, int n_total,
                             const int32_t *d_off, int n_segments, int max_n,
                             int32_t *d_order, void *stream);

/* ------------------------------------------------------------------------- */
/* Rectified-stereo filter + DLT triangulation, one match per lane.          */
/* Replaces Triangulation.create_inlier_outlier_matches                      */
/* (frontend/geometry/triangulation.py:96-109) and                           */
/* Triangulation.triangulate_opencv -> cv2.triangulatePoints                 */
/* (triangulation.py:76-87; loop_closure.py:237-249 for the fp64 form).      */
/*                                                                           */
/* Match k pairs left keypoint d_qidx[k] with right keypoint d_tidx[k]       */
/* (NULL index arrays mean the identity: k-th left with k-th right).         */
/*   keep[k] = |vR - vL| <= pix_thresh  &&  (uL - uR) >= disp_min  (fp64)    */
/*   xyz[k]  = dehomogenised null vector of the 4x4 DLT system built from    */
/*             P_left/P_right (host, 12 doubles each, row major), solved by  */
/*             one-sided Jacobi in fp64.  out_fp64=0: the unit 4-vector is   */
/*             rounded to fp32 and divided in fp32 (what float32 input gives */
/*             in OpenCV); out_fp64=1: divided in fp64.                      */
/* keypoints are (n,2) float32 (fp64 if kp_fp64=1).  d_keep may be NULL      */
/* (no filtering: every match is triangulated).  If d_keep is given, xyz of  */
/* rejected matches is written as 0.  n_left / n_right = rows of the two     */
/* keypoint arrays: a match naming a row outside them (e.g. the matcher's -1 */
/* for "no neighbour") is rejected (keep 0, xyz 0) and nothing is read.      */
/* ------------------------------------------------------------------------- */
int slamhip_stereo_filter_triangulate(const void *d_kp_left, int n_left, const void *d_kp_right,
                                      int n_right, int kp_fp64, const int32_t *d_qidx,
                                      const int32_t *d_tidx, int m, const double *P_left,
                                      const double *P_right, double pix_thresh,
                                      double disp_min, uint8_t *d_keep, void *d_xyz,
                                      int out_fp64, void *stream);

/* ------------------------------------------------------------------------- */
/* PnP-RANSAC consensus scoring.                                             */
/* Replaces the body of the loop in PnP.ransac_pnp_pair1                     */
/* (frontend/geometry/pnp.py:132-143): PnP.project (pnp.py:94-99) for the    */
/* left and right camera, compute_err (utils/graph_utils.py:7-11) and        */
/* PnP.supporters_pair (pnp.py:86-91), for H hypotheses at once.             */
/*                                                                           */
/*   d_points  (n,3) float32 (fp64 if points_fp64=1) landmark coordinates    */
/*   d_meas_l / d_meas_r (n,2) float32 measured pixels                       */
/*   d_Rt      (h,3,4) fp64 hypotheses [R|t] (left camera extrinsics)        */
/*   d_valid   (h) uint8 or NULL; 0 marks a failed minimal solve (count=-1)  */
/*   K (9 doubles) and M_right (12 doubles) are HOST pointers.               */
/*   d_counts  (h) int32 : number of points with                             */
/*             err_L <= thresh && err_R <= thresh (no cheirality test).      */
/* ------------------------------------------------------------------------- */
int slamhip_ransac_score(const void *d_points, int points_fp64, const float *d_meas_l,
                         const float *d_meas_r, int n, const double *d_Rt,
                         const uint8_t *d_valid, int h, const double *K,
                         const double *M_right, double thresh, int32_t *d_counts,
                         void *stream);

/* Inlier mask of ONE hypothesis (the `mask` of pnp.py:141-142). */
int slamhip_ransac_mask(const void *d_points, int points_fp64, const float *d_meas_l,
                        const float *d_meas_r, int n, const double *d_Rt_one,
                        const double *K, const double *M_right, double thresh,
                        uint8_t *d_mask, void *stream);

/* Minimal pose hypotheses: replaces PnP.solve_pnp(obj4, img4, cv2.SOLVEPNP_AP3P)
 * (pnp.py:36-43) as called once per RANSAC iteration at pnp.py:126-128.
 * Hypothesis i uses the four correspondences d_idx4[4i..4i+3] (rows of d_points
 * (n,3) and d_pixels (n,2) float32); d_idx4 == NULL means rows 4i..4i+3.  The
 * first three give up to four poses (P3P); the fourth selects the one with the
 * smallest reprojection error in normalised image coordinates.  d_Rt (h,3,4)
 * fp64, d_valid (h) uint8 (0 = no real solution with positive depths: the
 * reference's `None`).  n_points = number of rows of d_points / d_pixels: a
 * hypothesis naming a row outside [0, n_points) is reported invalid and nothing
 * is read for it.  K is a HOST pointer (9 doubles). */
int slamhip_p3p_solve(const void *d_points, int points_fp64, const float *d_pixels, int n_points,
                      const int32_t *d_idx4, int h, const double *K, double *d_Rt,
                      uint8_t *d_valid, void *stream);

/* Pose refinement: replaces PnP.solve_pnp(X[mask], pix_L[mask],
 * cv2.SOLVEPNP_ITERATIVE) (fill_database.py:204-207, loop_closure.py:222):
 * Levenberg-Marquardt on the left-image reprojection error from a given start.
 * One workgroup per problem.  d_off != NULL: problem p owns rows
 * [d_off[p], d_off[p+1]) of d_points (.,3) / d_pixels (.,2) float32 / d_mask
 * (uint8, NULL = all rows); d_off == NULL: every problem uses the same n rows
 * (several starts on one data set).  d_Rt_init / d_Rt_out: (n_problems,3,4)
 * fp64; d_info (n_problems,2) fp64 = final sum of squared pixel errors and
 * iterations used, may be NULL.  K: HOST pointer, 9 doubles, upper triangular
 * with K[8] = 1. */
int slamhip_refine_pose(const void *d_points, int points_fp64, const float *d_pixels,
                        const uint8_t *d_mask, int n, const int32_t *d_off, int n_problems,
                        const double *K, const double *d_Rt_init, int max_iters,
                        double *d_Rt_out, double *d_info, void *stream);

/* The two small helpers the reference exposes on its own:
 * PnP.project (pnp.py:94-99): pixels (n,2) fp64 = dehomogenised P @ [X,1];
 * P is a HOST pointer to 12 doubles.  PnP.supporters_pair (pnp.py:86-91) with
 * compute_err (utils/graph_utils.py:7-11): all inputs (n,2) fp64 on the device;
 * errors are computed like slamhip_compute_err does. */
int slamhip_project(const void *d_points, int points_fp64, int n, const double *P,
                    double *d_pixels, void *stream);
int slamhip_supporters_pair(const double *d_meas_l, const double *d_proj_l,
                            const double *d_meas_r, const double *d_proj_r, int n,
                            double thresh, uint8_t *d_mask, void *stream);
/* compute_err on its own (utils/graph_utils.py:7-11): d_err[i] = the Euclidean distance
 * between pixel rows i of two (n,2) fp64 device arrays, sqrt(dx*dx + dy*dy) with every
 * step rounded separately (what np.linalg.norm(meas - proj, axis=1) computes). */
int slamhip_compute_err(const double *d_meas, const double *d_proj, int n, double *d_err,
                        void *stream);

/* Sequential replay of the adaptive stopping rule of pnp.py:123-158 over the
 * ordered hypothesis counts (count < 0 = failed solve, iteration still
 * counted).  Writes 8 int32: d_result[0] = index of the winning hypothesis (-1
 * if none had count > 0), [1] = its count, [2] = iterations executed, [3] =
 * final n_iter_required, [4] = 1 if the reference would have raised
 * `ValueError: math domain error` (every point an inlier -> log(0)); the
 * replay then stops as if zero further iterations were required.  early_inliers < 0 disables the early exit.  Runs on the
 * device so that no host round trip sits between scoring and the mask.  */
int slamhip_ransac_replay(const int32_t *d_counts, int h, int n_points, double p_success,
                          int n_iter_max, int early_inliers, int32_t *d_result,
                          void *stream);

/* ------------------------------------------------------------------------- */
/* Batched front-end: the per-frame data flow of TrackingPipeline.build()    */
/* (backend/tracking/fill_database.py:140-240) for n_frames consecutive      */
/* frames in a fixed number of launches, features given.                     */
/*                                                                           */
/* Stereo stage per frame f (fill_database.py:155-169): match L->R (k=1),    */
/* sort by distance (stable), row/disparity filter, triangulate survivors,   */
/* gather their left descriptors (TrackingDB.create_links order,             */
/* backend/tracking/database.py:279-283).                                    */
/* Temporal stage per pair (f, f+1) (fill_database.py:185-207): match inlier */
/* descriptors of f (query) against those of f+1 (train), sort, build        */
/* X_corr / pix_L / pix_R, adaptive RANSAC over P3P hypotheses with stereo   */
/* consensus (pnp.py:102-160), LM refinement on the inliers.                 */
/*                                                                           */
/* All per-frame output arrays are indexed from the frame's LEFT offset      */
/* d_off_left[f]; counts are in d_st_count / d_ransac_info.                  */
/* ------------------------------------------------------------------------- */
typedef struct slamhip_frontend_args {
  uint32_t struct_size;                  /* = sizeof(slamhip_frontend_args) of the header the caller was
                                            built against.  Every entry point that takes this struct
                                            returns SLAMHIP_EINVAL (0 for the *_bytes queries) when it
                                            differs from the library's own: fields have been added over
                                            the versions, and a shorter struct read as the longer one
                                            would hand the kernels stray pointers.  Zero-initialise the
                                            struct, set this, then the fields in use. */
  /* inputs (device unless noted) */
  int32_t n_frames;                      /* F >= 1 */
  int32_t n_left_total, n_right_total;   /* rows of the concatenated arrays */
  int32_t max_left, max_right;           /* largest per-frame feature counts (host bound) */
  int32_t desc_bytes, desc_stride;
  const float *d_kp_left, *d_kp_right;       /* (.,2) float32 pixels */
  const uint8_t *d_des_left, *d_des_right;   /* (.,desc_stride) uint8 */
  const int32_t *d_off_left, *d_off_right;   /* F+1 row offsets */
  /* camera and parameters (host values) */
  double K[9], M_right[12], P_left[12], P_right[12];
  double pixel_thresh, disparity_min;    /* stereo filter (triangulation.py:104) */
  double ransac_thresh, p_success;       /* pnp.py:91, :104 */
  int32_t n_iter_max, refine_iters;
  int32_t stereo_cross_check, temporal_cross_check; /* cv2.BFMatcher(crossCheck=True) for the
                                            stereo / consecutive-frame matcher
                                            (fill_database.py:69-70; both default to 0 as
                                            run_slam.py:34-35): a query keeps, among the train
                                            rows whose own best match it is, the closest one;
                                            queries nobody chose have no match */
  uint64_t seed;                         /* device sampler seed */
  const int32_t *d_samples;  /* optional (F-1, n_iter_max, 4) sample indices into each pair's
                                sorted temporal matches; NULL = device RNG */
  /* outputs */
  int32_t *d_st_count;                   /* F: stereo inliers per frame */
  int32_t *d_st_left, *d_st_right, *d_st_dist; /* n_left_total: inlier (left kp, right kp, dist) */
  float *d_st_xyz;                       /* n_left_total*3: triangulated points */
  float *d_st_pix_left, *d_st_pix_right; /* n_left_total*2: inlier pixels */
  uint32_t *d_st_desc;                   /* n_left_total*W dwords: inlier left descriptors, zero
                                            padded to W = 8/16/32 dwords */
  int32_t *d_tm_query, *d_tm_train, *d_tm_dist; /* n_left_total: sorted temporal matches of pair f
                                            at d_off_left[f] (indices into the inlier sets) */
  uint8_t *d_tm_mask;                    /* n_left_total: RANSAC inlier mask, same order; a call with
                                            F >= 2 writes every row (zero behind a pair's matches and
                                            for the last frame): no fill needed beforehand */
  double *d_pose_ransac, *d_pose_refined;/* (F-1)*12: winning hypothesis, refined [R|t] */
  int32_t *d_ransac_info;                /* (F-1)*8: best index, best count, iterations, final
                                            n_iter_required, math-domain flag, N, mask count, 0 */
  void *d_workspace;
  size_t workspace_bytes;
  int64_t first_pair;        /* index of this batch's first pair in the caller's numbering of the
                                whole sequence: the device sampler draws pair p's samples from
                                (seed, first_pair + p), so a sequence cut into batches or shards
                                gets the samples of the uncut run.  d_samples rows stay local. */
  int32_t first_stereo_frame;/* 0 .. F-1: the stereo stage runs for frames [first_stereo_frame, F)
                                only; the d_st_* rows and d_st_count entries of the earlier frames
                                are taken as they stand (written by an earlier call on the same
                                arrays).  A caller that gets one frame at a time (the reference's
                                own loop, fill_database.py:140-240, which carries the previous
                                frame's stereo stage over, :234-240) keeps the last frame's rows
                                in place, appends the new frame behind them and calls with F = 2,
                                first_stereo_frame = 1, first_pair = the pair's number: the
                                results are those of one call over the whole sequence. */
  int32_t reserved0;
} slamhip_frontend_args;

size_t slamhip_frontend_workspace_bytes(const slamhip_frontend_args *args);
int slamhip_frontend_batch(const slamhip_frontend_args *args, void *stream);

/* Hand-over record of a finished batch: the rows in use of frames [frame_lo, frame_hi) and
 * pairs [pair_lo, pair_hi) of the output arrays in `args`, compacted into one byte string on the
 * device, so that a rank's share of a sequence reaches the rank that runs the back-end in one
 * gather.  It holds what the reference passes to TrackingDB.add_frame per frame
 * (backend/tracking/fill_database.py:231: links' pixels, inlier descriptors, temporal matches,
 * inlier mask) and the poses the back-end reads (run_slam.py:88-89).
 * Layout (little endian): int64 {frames, pairs, first_frame, first_pair, desc_bytes}, int64
 * stereo counts[frames], int64 temporal counts[pairs]; flat over the frames: left index, right
 * index, distance (int32), xyz (3 float32), left pixel, right pixel (2 float32 each), inlier
 * descriptor (desc_bytes uint8); flat over the pairs: query index, train index, distance
 * (int32), mask (uint8); per pair: refined pose, RANSAC pose (12 float64 each), the 8 int32 of
 * d_ransac_info.  d_st_out_off / d_tm_out_off: device arrays of frames+1 / pairs+1 int64, the
 * exclusive prefix sums of the owned frames' d_st_count and of the owned pairs' match counts
 * (d_ransac_info[8 p + 5]), which the caller has read back to size the record: their last
 * entries are stereo_rows / temporal_rows.  first_frame / first_pair: the caller's numbering of
 * the first owned frame / pair in the whole sequence. */
size_t slamhip_frontend_record_bytes(int n_frames, int n_pairs, long long stereo_rows, long long temporal_rows,
                                     int desc_bytes);
int slamhip_frontend_pack_record(const slamhip_frontend_args *args, int frame_lo, int frame_hi, int pair_lo,
                                 int pair_hi, int64_t first_frame, int64_t first_pair,
                                 const int64_t *d_st_out_off, const int64_t *d_tm_out_off, int64_t stereo_rows,
                                 int64_t temporal_rows, uint8_t *d_record, size_t record_bytes, void *stream);

/* One new frame from the host in ONE copy, for a caller that pushes frames one at a time (the
 * shape of the reference's loop, backend/tracking/fill_database.py:140-240).  No counterpart in
 * the reference, where the frame's arrays are host arrays already.  d_packet (4-byte aligned,
 * device memory the caller copied the packet into): six int32 -- the window's d_off_left[0..3)
 * and d_off_right[0..3) for a call with F <= 2 -- then the frame's left pixels (n_left x 2
 * float32), right pixels (n_right x 2), left descriptors (n_left x desc_bytes uint8, rows
 * packed) and right descriptors, back to back.  The call moves each piece to the device address
 * given (the frame's first row in the caller's arrays; descriptor rows desc_stride bytes apart)
 * and the six offsets to d_off (skipped when NULL), on `stream`, in one launch. */
int slamhip_frontend_unpack_frame(const uint8_t *d_packet, int n_left, int n_right, int desc_bytes,
                                  int desc_stride, float *d_kp_left, float *d_kp_right, uint8_t *d_des_left,
                                  uint8_t *d_des_right, int32_t *d_off, void *stream);

/* ------------------------------------------------------------------------- */
/* Pair verification: LoopClosure.consensus_pnp_match                        */
/* (backend/optimizers/loop_closure.py:167-235) for a LIST of                */
/* (candidate, current) keyframe pairs at once.                              */
/*                                                                           */
/* Per keyframe (loop_closure.py:152-164,179-180,237-249): left/right        */
/* keypoints rebuilt from the stored links, (x_left,y) / (x_right,y), fp64;  */
/* every stored feature is triangulated in fp64 and kept as float32 (:209).  */
/* Per pair (:192-222): match candidate descriptors (query) against the      */
/* current keyframe's (train), k=1, sorted by distance; P3D = candidate      */
/* points by queryIdx, pixels of the current keyframe by trainIdx (float32); */
/* pairs with fewer than min_points matches are not attempted (:206-207);    */
/* adaptive P3P RANSAC with stereo consensus; LM refinement on the inliers   */
/* of the pairs whose consensus set has at least min_points members (the     */
/* others get their RANSAC pose back: the reference returns before refining, */
/* :218-219).  The caller applies the same `mask.sum() < min_inliers` test   */
/* to d_info[8p+6] and inverts the pose (:226-228).                          */
/*                                                                           */
/* Pair p reads query rows [d_qbase[p], +d_qcnt[p]) and train rows           */
/* [d_tbase[p], +d_tcnt[p]) of the per-keyframe arrays; its per-match        */
/* outputs are at d_obase[p] (the caller lays pairs out back to back:        */
/* n_out_total = sum of d_qcnt).                                             */
/*                                                                           */
/* cross_check != 0: the matcher of :192 was built with crossCheck           */
/* (LoopClosure's frame_cross_check): a candidate row keeps, among the rows  */
/* of the current keyframe that chose it, the closest (lowest row on ties);  */
/* rows nobody chose have no match, so a pair yields d_info[8p+5] <= d_qcnt  */
/* matches.  The swapped matcher needs one scratch row per (pair, current-   */
/* keyframe row): d_tobase[p] = sum of d_tcnt over the pairs before p,       */
/* n_tout_total = sum of d_tcnt.                                             */
/* ------------------------------------------------------------------------- */
typedef struct slamhip_verify_args {
  uint32_t struct_size;                  /* = sizeof(slamhip_verify_args); checked like
                                            slamhip_frontend_args.struct_size */
  int32_t n_rows_total;                  /* rows of the per-keyframe arrays */
  int32_t desc_bytes, desc_stride;
  int32_t n_pairs, n_out_total, max_q, max_t;
  const double *d_kp_left, *d_kp_right;  /* (n_rows_total,2) fp64 */
  const uint8_t *d_des;                  /* (n_rows_total,desc_stride) uint8 */
  const int32_t *d_qbase, *d_qcnt, *d_tbase, *d_tcnt, *d_obase; /* n_pairs each (device) */
  double K[9], M_right[12], P_left[12], P_right[12];   /* host values */
  double ransac_thresh, p_success;
  int32_t n_iter_max, min_points, refine_iters, cross_check;
  uint64_t seed;
  /* outputs */
  float *d_xyz;                          /* n_rows_total*3: keyframe points */
  int32_t *d_tm_query, *d_tm_train, *d_tm_dist;   /* n_out_total: sorted matches */
  uint8_t *d_tm_mask;                    /* n_out_total */
  double *d_pose_ransac, *d_pose_refined;/* n_pairs*12 */
  int32_t *d_info;                       /* n_pairs*8, as d_ransac_info above */
  void *d_workspace;
  size_t workspace_bytes;
  /* cross_check only */
  const int32_t *d_tobase;               /* n_pairs (device) */
  int32_t n_tout_total, reserved;
  /* The device sampler keys pair p's samples by (seed, d_pair_ids[p]), or by (seed, p) when this
   * is NULL.  A pair list cut into shards (one per GPU) passes each pair's index in the whole
   * list here: every pair then draws the samples, and gets the result, of the uncut run. */
  const int64_t *d_pair_ids;             /* n_pairs (device) or NULL */
  /* Optional (n_pairs, n_iter_max, 4) int32 on the device: the idx4 sequence of
   * ransac_pnp_pair1's rng.choice(N, 4) (frontend/geometry/pnp.py:126, called from
   * loop_closure.py:215-216) for every pair, as indices into the pair's sorted matches; values
   * are clamped to [0, N).  Iteration i of pair p uses row (p, i).  NULL = device sampler.  With
   * it a CPU run of the reference's loop on the same sequence can be compared count for count. */
  const int32_t *d_samples;
} slamhip_verify_args;

size_t slamhip_verify_workspace_bytes(const slamhip_verify_args *args);
int slamhip_verify_pairs(const slamhip_verify_args *args, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SLAMHIP_H */
