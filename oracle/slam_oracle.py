"""
CPU oracle for the stereo-SLAM front-end hot path.  TEST INFRASTRUCTURE ONLY.

This module restates, in plain NumPy, the arithmetic of the reference
(GuyShmueli/SLAM; paths below are relative to the reference root).  It is the
checker for the HIP kernels: only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import it.  Nothing under slam_amd/ does.

Pinning status
  * project / compute_err / supporters_pair / homogenous / non_homogenous, the
    stereo filter decision, the own-SVD DLT builder and CameraModel are pinned
    against the reference's own functions (oracle/make_golden.py imports them
    from /root/reference with cv2/gtsam absent and commits the vectors under
    tests/golden/).
  * Everything whose arithmetic lives in OpenCV (cv2.BFMatcher.match/knnMatch,
    cv2.triangulatePoints, cv2.solvePnP) is PARITY UNPINNED: opencv-python is a
    dependency the reference does not pin (no requirements file) and it is not
    installed here, and the reference holds no tests or golden vectors for it.
    Those functions restate OpenCV's published algorithm:
      - BFMatcher: modules/core/src/batch_distance.cpp (batchDistHamming +
        the K-nearest insertion loop with strict `<`, train index ascending) and
        modules/features2d/src/matchers.cpp (BFMatcher::knnMatchImpl);
      - triangulatePoints: modules/calib3d/src/triangulate.cpp (rows
        x*P[2]-P[0], y*P[2]-P[1] per camera, SVD, last right singular vector).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------
# Hamming brute-force matching (cv2.BFMatcher(NORM_HAMMING) semantics)
# reference call site: frontend/vision/descriptor_matcher.py:23-34
# --------------------------------------------------------------------------


def hamming_distance_matrix(q: np.ndarray, t: np.ndarray) -> np.ndarray:
    """(nq, nt) int32 matrix of bit differences between uint8 rows."""
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    nq, nt = q.shape[0], t.shape[0]
    out = np.empty((nq, nt), dtype=np.int32)
    step = max(1, (1 << 24) // max(1, nt * max(1, q.shape[1])))
    for s in range(0, nq, step):
        x = np.bitwise_xor(q[s:s + step, None, :], t[None, :, :])
        out[s:s + step] = np.bitwise_count(x).sum(axis=2, dtype=np.int32)
    return out


def hamming_knn2(q: np.ndarray, t: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Top-2 neighbours per query row, lowest train index on ties.

    Equivalent to the sequential insertion of batch_distance.cpp: scanning train
    rows in ascending order, a candidate displaces an entry only when strictly
    smaller, so among equal distances the earliest index stays first.
    Returns (idx2, dist2), both (nq, 2) int32, -1 where there is no neighbour.
    """
    nq, nt = q.shape[0], t.shape[0]
    idx2 = np.full((nq, 2), -1, dtype=np.int32)
    dist2 = np.full((nq, 2), -1, dtype=np.int32)
    if nq == 0 or nt == 0:
        return idx2, dist2
    step = max(1, (1 << 22) // max(1, nt))
    rows = np.arange(nq)
    for s in range(0, nq, step):
        d = hamming_distance_matrix(q[s:s + step], t)
        r = np.arange(d.shape[0])
        i1 = d.argmin(axis=1)  # first occurrence = lowest index
        idx2[s:s + step, 0] = i1
        dist2[s:s + step, 0] = d[r, i1]
        if nt > 1:
            d[r, i1] = np.iinfo(np.int32).max
            i2 = d.argmin(axis=1)
            idx2[s:s + step, 1] = i2
            dist2[s:s + step, 1] = d[r, i2]
    del rows
    return idx2, dist2


def hamming_knn2_sequential(q: np.ndarray, t: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Literal restatement of the K=2 insertion loop (small inputs only)."""
    nq, nt = q.shape[0], t.shape[0]
    idx2 = np.full((nq, 2), -1, dtype=np.int32)
    dist2 = np.full((nq, 2), -1, dtype=np.int32)
    big = np.iinfo(np.int32).max
    for i in range(nq):
        bd = [big, big]
        bi = [-1, -1]
        for j in range(nt):
            d = int(np.bitwise_count(np.bitwise_xor(q[i], t[j])).sum())
            if d < bd[1]:
                k = 1
                while k > 0 and bd[k - 1] > d:
                    bd[k] = bd[k - 1]
                    bi[k] = bi[k - 1]
                    k -= 1
                bd[k] = d
                bi[k] = j
        for k in range(2):
            if bi[k] >= 0:
                idx2[i, k] = bi[k]
                dist2[i, k] = bd[k]
    return idx2, dist2


def l2_distance_matrix(q: np.ndarray, t: np.ndarray) -> np.ndarray:
    """(nq, nt) float32 L2 distances the way cv2.BFMatcher(NORM_L2) defines them for
    float32 descriptors (the norm descriptor_matcher.py:23-28 selects for non-uint8
    descriptors): sqrt of the sum of squared differences.  The sum runs in float32 in
    index order with separately rounded products and additions; OpenCV's own order
    inside its SIMD kernel is unspecified (PARITY UNPINNED in the last bits)."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    t = np.ascontiguousarray(t, dtype=np.float32)
    acc = np.zeros((q.shape[0], t.shape[0]), dtype=np.float32)
    for k in range(q.shape[1]):
        d = q[:, k, None] - t[None, :, k]
        acc = acc + d * d
    return np.sqrt(acc)


def l2_knn2(q: np.ndarray, t: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Top-2 L2 neighbours per query row, lowest train index on ties.  Returns
    (idx2 (nq,2) int32, dist2 (nq,2) float32), -1 where there is no neighbour."""
    nq, nt = q.shape[0], t.shape[0]
    idx2 = np.full((nq, 2), -1, dtype=np.int32)
    dist2 = np.full((nq, 2), -1, dtype=np.float32)
    if nq == 0 or nt == 0:
        return idx2, dist2
    d = l2_distance_matrix(q, t)
    r = np.arange(nq)
    i1 = d.argmin(axis=1)                      # first occurrence = lowest index
    idx2[:, 0], dist2[:, 0] = i1, d[r, i1]
    if nt > 1:
        d[r, i1] = np.inf
        i2 = d.argmin(axis=1)
        idx2[:, 1], dist2[:, 1] = i2, d[r, i2]
    return idx2, dist2


def bf_match_l2(q: np.ndarray, t: np.ndarray, cross_check: bool = False):
    """cv2.BFMatcher(NORM_L2, crossCheck).match + the reference's stable sort by
    distance (descriptor_matcher.py:33-34): (query_idx, train_idx int32, dist float32)."""
    nq, nt = q.shape[0], t.shape[0]
    if nq == 0 or nt == 0:
        z = np.zeros(0, np.int32)
        return z, z, np.zeros(0, np.float32)
    if not cross_check:
        idx2, dist2 = l2_knn2(q, t)
        qi, ti, di = np.arange(nq, dtype=np.int32), idx2[:, 0], dist2[:, 0]
    else:
        ridx, rdist = l2_knn2(t, q)            # every train row votes for its nearest query
        ti = np.full(nq, -1, np.int32)
        di = np.full(nq, np.inf, np.float32)
        for j in range(nt):                    # first voter wins ties (strict <, ascending train index)
            who, dj = ridx[j, 0], rdist[j, 0]
            if dj < di[who]:
                di[who], ti[who] = dj, j
        keep = ti >= 0
        qi, ti, di = np.flatnonzero(keep).astype(np.int32), ti[keep], di[keep]
    order = np.argsort(di, kind="stable")
    return qi[order], ti[order], di[order]


def ratio_test(dist2: np.ndarray, ratio: float) -> np.ndarray:
    """keep[i] = second neighbour exists and d1 < ratio * d2 (fp64)."""
    d1 = dist2[:, 0].astype(np.float64)
    d2 = dist2[:, 1].astype(np.float64)
    return (dist2[:, 0] >= 0) & (dist2[:, 1] >= 0) & (d1 < ratio * d2)


def cross_check_fold(t2q_idx: np.ndarray, t2q_dist: np.ndarray, nq: int):
    """crossCheck=True fold of batch_distance.cpp: every train row votes for its
    nearest query; a query keeps the voter with the smallest distance, first
    voter on ties (strict `<` while scanning train index ascending)."""
    match_idx = np.full(nq, -1, dtype=np.int32)
    match_dist = np.full(nq, -1, dtype=np.int32)
    best = np.full(nq, np.iinfo(np.int32).max, dtype=np.int64)
    for t in range(len(t2q_idx)):
        qi = int(t2q_idx[t])
        if qi < 0:
            continue
        d = int(t2q_dist[t])
        if d < best[qi]:
            best[qi] = d
            match_idx[qi] = t
            match_dist[qi] = d
    return match_idx, match_dist


def bf_match(q: np.ndarray, t: np.ndarray, cross_check: bool = False):
    """cv2.BFMatcher(NORM_HAMMING, crossCheck).match followed by the reference's
    `sorted(matches, key=lambda x: x.distance)` (descriptor_matcher.py:33-34).
    Returns (query_idx, train_idx, dist) int32 arrays in the sorted order."""
    if not cross_check:
        idx2, dist2 = hamming_knn2(q, t)
        qi = np.arange(q.shape[0], dtype=np.int32)
        ti, di = idx2[:, 0], dist2[:, 0]
    else:
        t2q_idx, t2q_dist = hamming_knn2(t, q)
        ti, di = cross_check_fold(t2q_idx[:, 0], t2q_dist[:, 0], q.shape[0])
        qi = np.arange(q.shape[0], dtype=np.int32)
    ok = ti >= 0
    qi, ti, di = qi[ok], ti[ok], di[ok]
    order = np.argsort(di, kind="stable")  # Python's sorted() is stable
    return qi[order], ti[order], di[order]


def sort_by_distance(dist: np.ndarray) -> np.ndarray:
    """Stable order by distance; entries < 0 (no match) last."""
    d = np.where(dist < 0, np.iinfo(np.int32).max, dist)
    return np.argsort(d, kind="stable").astype(np.int32)


# --------------------------------------------------------------------------
# Camera model  (frontend/io/camera_model.py:31-36, 59-73)
# --------------------------------------------------------------------------


class CameraModel:
    def __init__(self, K, M_left, M_right):
        self.K = np.asarray(K, dtype=np.float64)
        self.M_left = np.asarray(M_left, dtype=np.float64)
        self.M_right = np.asarray(M_right, dtype=np.float64)
        self.P_left = self.K @ self.M_left
        self.P_right = self.K @ self.M_right

    @classmethod
    def from_projection_rows(cls, p0: Sequence[float], p1: Sequence[float]):
        m1 = np.array(p0, dtype=np.float64).reshape(3, 4)
        m2 = np.array(p1, dtype=np.float64).reshape(3, 4)
        k = m1[:, :3]
        return cls(k, np.linalg.inv(k) @ m1, np.linalg.inv(k) @ m2)


# --------------------------------------------------------------------------
# Stereo filter  (frontend/geometry/triangulation.py:96-109)
# --------------------------------------------------------------------------


def stereo_filter(kp_left, kp_right, qidx, tidx, pixel_thresh, disparity_min) -> np.ndarray:
    """keep = abs(vR - vL) <= pixel_thresh and (uL - uR) >= disparity_min, on
    float32 pixels widened to fp64 (cv2.KeyPoint.pt yields Python floats)."""
    kl = np.asarray(kp_left)[np.asarray(qidx)].astype(np.float64)
    kr = np.asarray(kp_right)[np.asarray(tidx)].astype(np.float64)
    return (np.abs(kr[:, 1] - kl[:, 1]) <= pixel_thresh) & ((kl[:, 0] - kr[:, 0]) >= disparity_min)


# --------------------------------------------------------------------------
# DLT triangulation
# --------------------------------------------------------------------------


def dlt_matrices_cv(P_left, P_right, pts_left, pts_right) -> np.ndarray:
    """(M,4,4) fp64 systems in cv2.triangulatePoints row order."""
    pl = np.asarray(pts_left, dtype=np.float64)
    pr = np.asarray(pts_right, dtype=np.float64)
    PL = np.asarray(P_left, dtype=np.float64)
    PR = np.asarray(P_right, dtype=np.float64)
    A = np.empty((pl.shape[0], 4, 4), dtype=np.float64)
    A[:, 0] = pl[:, 0:1] * PL[2] - PL[0]
    A[:, 1] = pl[:, 1:2] * PL[2] - PL[1]
    A[:, 2] = pr[:, 0:1] * PR[2] - PR[0]
    A[:, 3] = pr[:, 1:2] * PR[2] - PR[1]
    return A


def triangulate_cv(P_left, P_right, pts_left, pts_right, out_fp64: bool = False) -> np.ndarray:
    """cv2.triangulatePoints + `Xh[:3] / Xh[3]` (triangulation.py:85-86;
    loop_closure.py:245-248).  float32 points give a float32 homogeneous result
    that is then divided in float32; float64 points stay in float64."""
    A = dlt_matrices_cv(P_left, P_right, pts_left, pts_right)
    if A.shape[0] == 0:
        return np.zeros((0, 3), dtype=np.float64 if out_fp64 else np.float32)
    _, _, Vt = np.linalg.svd(A)
    Xh = Vt[:, -1, :]
    if out_fp64:
        return Xh[:, :3] / Xh[:, 3:4]
    Xh32 = Xh.astype(np.float32)
    return Xh32[:, :3] / Xh32[:, 3:4]


def dlt_matrices_own(P_left, P_right, pts_left, pts_right) -> np.ndarray:
    """Row order of the reference's own builder (triangulation.py:32-46):
    v*P[2]-P[1], P[0]-u*P[2] per camera."""
    pl = np.asarray(pts_left, dtype=np.float64)
    pr = np.asarray(pts_right, dtype=np.float64)
    PL = np.asarray(P_left, dtype=np.float64)
    PR = np.asarray(P_right, dtype=np.float64)
    A = np.empty((pl.shape[0], 4, 4), dtype=np.float64)
    A[:, 0] = pl[:, 1:2] * PL[2] - PL[1]
    A[:, 1] = PL[0] - pl[:, 0:1] * PL[2]
    A[:, 2] = pr[:, 1:2] * PR[2] - PR[1]
    A[:, 3] = PR[0] - pr[:, 0:1] * PR[2]
    return A


def triangulate_own(P_left, P_right, pts_left, pts_right) -> np.ndarray:
    """triangulate_matches_batched (triangulation.py:49-73) with the final
    cv2.convertPointsFromHomogeneous written as the division its comment gives."""
    A = dlt_matrices_own(P_left, P_right, pts_left, pts_right)
    if A.shape[0] == 0:
        return np.zeros((0, 3))
    _, _, Vt = np.linalg.svd(A)
    Xh = Vt[:, -1, :]
    return Xh[:, :3] / Xh[:, 3:4]


# --------------------------------------------------------------------------
# PnP-RANSAC consensus  (frontend/geometry/pnp.py:86-160, utils/graph_utils.py:7-35)
# --------------------------------------------------------------------------


def compute_err(meas_pts, proj_pts):
    return np.linalg.norm(meas_pts - proj_pts, axis=1)


def homogenous(Rt):
    return np.vstack([Rt, [0, 0, 0, 1]])


def non_homogenous(T):
    return T[:3, :]


def project(Xw, P):
    Xw_h = np.column_stack([Xw, np.ones(len(Xw))])
    pix = (P @ Xw_h.T).T
    return pix[:, :2] / pix[:, 2:3]


def supporters_pair(meas_L, proj_L, meas_R, proj_R, pixel_thresh):
    err_L = compute_err(meas_L, proj_L)
    err_R = compute_err(meas_R, proj_R)
    return (err_L <= pixel_thresh) & (err_R <= pixel_thresh)


def hypothesis_errors(K, M_right, Rt, obj_pts, meas_L, meas_R):
    """(err_L, err_R) of one hypothesis, pnp.py:133-142 step by step."""
    M_right_h = homogenous(M_right)
    P_L = K @ Rt
    proj_L = project(obj_pts, P_L)
    Rt_R = non_homogenous(M_right_h @ homogenous(Rt))
    P_R = K @ Rt_R
    proj_R = project(obj_pts, P_R)
    with np.errstate(all="ignore"):
        return compute_err(meas_L, proj_L), compute_err(meas_R, proj_R)


def hypothesis_mask(K, M_right, Rt, obj_pts, meas_L, meas_R, pixel_thresh):
    with np.errstate(all="ignore"):
        eL, eR = hypothesis_errors(K, M_right, Rt, obj_pts, meas_L, meas_R)
        return (eL <= pixel_thresh) & (eR <= pixel_thresh)


def score_hypotheses(K, M_right, Rts, valid, obj_pts, meas_L, meas_R, pixel_thresh):
    """Inlier count per hypothesis (-1 where valid == 0)."""
    counts = np.empty(len(Rts), dtype=np.int32)
    for h, Rt in enumerate(Rts):
        if valid is not None and not valid[h]:
            counts[h] = -1
            continue
        counts[h] = int(hypothesis_mask(K, M_right, Rt, obj_pts, meas_L, meas_R, pixel_thresh).sum())
    return counts


def ransac_replay(counts: Sequence[int], n_points: int, p_success: float = 0.99999,
                  n_iter_max: int = 2500, early_inliers: Optional[int] = None):
    """The control flow of ransac_pnp_pair1 (pnp.py:110-160) driven by a given
    sequence of per-iteration inlier counts (count < 0 = solvePnP failed).
    Returns (best_index, best_count, iter_done, n_iter_required).  Raises
    ValueError exactly where math.log would (every point an inlier)."""
    m = 4
    best_idx, best_count, iter_done = -1, 0, 0
    n_iter_required = n_iter_max
    while iter_done < n_iter_required and iter_done < n_iter_max and iter_done < len(counts):
        c = int(counts[iter_done])
        idx = iter_done
        iter_done += 1
        if c < 0:
            continue
        if c > best_count:
            best_count, best_idx = c, idx
            inlier_ratio = best_count / n_points
            if inlier_ratio > 0:
                prob = inlier_ratio ** m
                n_iter_required = math.ceil(math.log(1 - p_success) / math.log(1 - prob))
        if early_inliers is not None and best_count >= early_inliers:
            break
    return best_idx, best_count, iter_done, n_iter_required


def sample_indices(n_points: int, n_iter: int, seed: int = 12345) -> np.ndarray:
    """idx4 sequence with the call pattern of pnp.py:126 and the seed of the
    commented-out line pnp.py:120."""
    rng = np.random.default_rng(seed)
    return np.stack([rng.choice(n_points, 4, replace=False) for _ in range(n_iter)])


# --------------------------------------------------------------------------
# Minimal solver (stands where cv2.solvePnP(..., SOLVEPNP_AP3P) stands in
# pnp.py:36-43).  PARITY UNPINNED: OpenCV is absent and the reference holds no
# vectors for it.  Any exact P3P solver returns the same (up to four) poses for
# three correspondences; this one forms the classical two-quadric system and
# takes the quartic's roots from numpy.roots, independently of the bracketing
# root finder of the HIP kernel.  The fourth correspondence selects the pose.
# --------------------------------------------------------------------------


def _triangle_frame(p1, p2, p3):
    a = p2 - p1
    a = a / np.linalg.norm(a)
    n = np.cross(a, p3 - p1)
    n = n / np.linalg.norm(n)
    return np.column_stack([a, np.cross(n, a), n])


def p3p_solutions(obj3, img3, K):
    """All poses [R|t] (3x4) with positive depths for three correspondences."""
    Kinv = np.linalg.inv(K)
    b = (Kinv @ np.column_stack([img3, np.ones(3)]).T).T
    f = b / np.linalg.norm(b, axis=1, keepdims=True)
    X = np.asarray(obj3, dtype=np.float64)
    a12 = np.sum((X[0] - X[1]) ** 2)
    a13 = np.sum((X[0] - X[2]) ** 2)
    a23 = np.sum((X[1] - X[2]) ** 2)
    c12, c13, c23 = f[0] @ f[1], f[0] @ f[2], f[1] @ f[2]
    P = np.polynomial.polynomial
    A1, B1, C1 = a13, -2 * a13 * c12, np.array([a13 - a12, 2 * a12 * c13, -a12])
    A2, B2, C2 = a23 - a12, np.array([-2 * a23 * c12, 2 * a12 * c23]), np.array([a23, 0.0, -a12])
    p = A1 * C2 - A2 * C1
    q = P.polysub(A1 * B2, [A2 * B1])
    r = P.polysub(B1 * C2, P.polymul(B2, C1))
    quart = P.polysub(P.polymul(p, p), P.polymul(q, r))
    roots = np.roots(quart[::-1])
    sols = []
    Fw = _triangle_frame(X[0], X[1], X[2])
    for v in roots:
        if abs(v.imag) > 1e-9 * max(1.0, abs(v.real)) or v.real <= 0:
            continue
        v = v.real
        qv = P.polyval(v, q)
        if qv == 0:
            continue
        u = -P.polyval(v, p) / qv
        if u <= 0:
            continue
        d1 = np.sqrt(a12 / (1 + u * u - 2 * u * c12))
        d = np.array([d1, u * d1, v * d1])
        for _ in range(3):  # Gauss-Newton polish of the depths
            res = np.array([d[0] ** 2 + d[1] ** 2 - 2 * d[0] * d[1] * c12 - a12,
                            d[0] ** 2 + d[2] ** 2 - 2 * d[0] * d[2] * c13 - a13,
                            d[1] ** 2 + d[2] ** 2 - 2 * d[1] * d[2] * c23 - a23])
            J = np.array([[2 * (d[0] - d[1] * c12), 2 * (d[1] - d[0] * c12), 0],
                          [2 * (d[0] - d[2] * c13), 0, 2 * (d[2] - d[0] * c13)],
                          [0, 2 * (d[1] - d[2] * c23), 2 * (d[2] - d[1] * c23)]])
            d = d - np.linalg.solve(J, res)
        if (d <= 0).any():
            continue
        Y = d[:, None] * f
        R = _triangle_frame(Y[0], Y[1], Y[2]) @ Fw.T
        t = Y[0] - R @ X[0]
        sols.append(np.column_stack([R, t]))
    return sols


def p3p_solve(obj4, img4, K):
    """Pose selected by the fourth correspondence, or None."""
    obj4 = np.asarray(obj4, dtype=np.float64)
    img4 = np.asarray(img4, dtype=np.float64)
    sols = p3p_solutions(obj4[:3], img4[:3], K)
    if not sols:
        return None
    n4 = np.linalg.inv(K) @ np.array([img4[3, 0], img4[3, 1], 1.0])
    n4 = n4[:2] / n4[2]
    best, best_err = None, np.inf
    for Rt in sols:
        c = Rt[:, :3] @ obj4[3] + Rt[:, 3]
        err = np.sum((c[:2] / c[2] - n4) ** 2)
        if err < best_err:
            best, best_err = Rt, err
    return best


def refine_pose(obj, img, K, Rt0, mask=None):
    """Least-squares pose (minimum of the left reprojection error) from a start,
    standing where cv2.solvePnP(..., SOLVEPNP_ITERATIVE) stands
    (fill_database.py:204-207).  PARITY UNPINNED (OpenCV absent).

    SciPy's MINPACK LM brings the start into the basin (its finite-difference
    Jacobian limits it to ~1e-7 in weak directions); Gauss-Newton steps with the
    analytic Jacobian (numpy.linalg.lstsq on the stacked 2n x 6 system) then
    converge to the stationary point.  Returns ([R|t], sum of squared errors)."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation

    obj = np.asarray(obj, dtype=np.float64)
    img = np.asarray(img, dtype=np.float64)
    if mask is not None:
        obj, img = obj[mask], img[mask]
    x0 = np.concatenate([Rotation.from_matrix(Rt0[:, :3]).as_rotvec(), Rt0[:, 3]])

    def fun(x):
        R = Rotation.from_rotvec(x[:3]).as_matrix()
        return (project(obj, K @ np.column_stack([R, x[3:]])) - img).ravel()

    sol = least_squares(fun, x0, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15)
    R = Rotation.from_rotvec(sol.x[:3]).as_matrix()
    t = sol.x[3:].copy()
    for _ in range(20):
        Xc = obj @ R.T + t
        p = Xc @ K.T
        res = (p[:, :2] / p[:, 2:3] - img).ravel()
        # d(pixel)/d(Xc) = (d(p/pz)/dp) K ;  d(Xc)/d(w, v) = [-[Xc]x | I]
        n = len(obj)
        dpix_dp = np.zeros((n, 2, 3))
        dpix_dp[:, 0, 0] = 1 / p[:, 2]
        dpix_dp[:, 1, 1] = 1 / p[:, 2]
        dpix_dp[:, 0, 2] = -p[:, 0] / p[:, 2] ** 2
        dpix_dp[:, 1, 2] = -p[:, 1] / p[:, 2] ** 2
        dpix_dX = dpix_dp @ K
        skew = np.zeros((n, 3, 3))
        skew[:, 0, 1], skew[:, 0, 2] = -Xc[:, 2], Xc[:, 1]
        skew[:, 1, 0], skew[:, 1, 2] = Xc[:, 2], -Xc[:, 0]
        skew[:, 2, 0], skew[:, 2, 1] = -Xc[:, 1], Xc[:, 0]
        J = np.concatenate([dpix_dX @ (-skew), dpix_dX], axis=2).reshape(2 * n, 6)
        delta = np.linalg.lstsq(J, -res, rcond=None)[0]
        E = Rotation.from_rotvec(delta[:3]).as_matrix()
        R, t = E @ R, E @ t + delta[3:]
        if np.abs(delta).max() < 1e-14:
            break
    Rt = np.column_stack([R, t])
    final = (project(obj, K @ Rt) - img).ravel()
    return Rt, float(np.sum(final ** 2))
