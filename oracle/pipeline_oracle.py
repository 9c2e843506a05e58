"""
CPU restatement of the per-frame data flow of TrackingPipeline.build()
(backend/tracking/fill_database.py:102-240 of the reference), features given.
TEST INFRASTRUCTURE ONLY.  The data flow restated here is pinned against the
reference's own TrackingPipeline.build() and LoopClosure.consensus_pnp_match(), run as
they are with their OpenCV leaves standing on oracle/slam_oracle.py
(oracle/make_golden_flow.py -> tests/golden/ref_build_flow.npz, ref_consensus_flow.npz,
tests/test_reference_flow.py); the steps that live in OpenCV themselves are parity
unpinned (see oracle/slam_oracle.py for the status of each).

It is deliberately written the way the reference is: one frame at a time, one
RANSAC iteration at a time, so that it also serves as the shape of the CPU
baseline ("a Python loop over frames calling array kernels").
"""
from __future__ import annotations

import numpy as np

from . import slam_oracle as so


def _bf_match(q, t, knn2=None, cross_check=False):
    """bf_match with an optional faster top-2 routine (the C oracle)."""
    if knn2 is None or cross_check:
        return so.bf_match(q, t, cross_check=cross_check)
    if len(q) == 0 or len(t) == 0:
        z = np.zeros(0, np.int32)
        return z, z, z
    idx2, dist2 = knn2(q, t)
    order = np.argsort(dist2[:, 0], kind="stable")
    return order.astype(np.int32), idx2[order, 0], dist2[order, 0]


def stereo_stage(cam, frame, pixel_thresh=4.0, disparity_min=1.2, knn2=None, cross_check=False):
    """fill_database.py:155-169 for one frame (cross_check: the stereo matcher's
    crossCheck, fill_database.py:69)."""
    kp_l, kp_r, des_l, des_r = frame
    qi, ti, di = _bf_match(des_l, des_r, knn2, cross_check)                 # :155
    keep = so.stereo_filter(kp_l, kp_r, qi, ti, pixel_thresh, disparity_min)  # :158-160
    qi, ti, di = qi[keep], ti[keep], di[keep]
    X = so.triangulate_cv(cam.P_left, cam.P_right, kp_l[qi], kp_r[ti])      # :163
    return {"left_idx": qi, "right_idx": ti, "dist": di, "xyz": X,
            "inlier_des_left": des_l[qi],                                   # create_links, :166-169
            "pix_left": kp_l[qi], "pix_right": kp_r[ti]}


def ransac_loop(cam, X_corr, pix_L, pix_R, pixel_thresh, samples=None, solver=None,
                p_success=0.99999, n_iter_max=2500, rng=None):
    """PnP.ransac_pnp_pair1 (frontend/geometry/pnp.py:102-160), one iteration at
    a time.  `samples` (n_iter,4) fixes the idx4 sequence (else rng.choice as in
    pnp.py:126); `solver(obj4,img4)` is the minimal solver (default: oracle P3P)."""
    import math
    N = len(X_corr)
    out = {"mask": np.zeros(N, bool), "best_index": -1, "best_count": 0, "iterations": 0,
           "Rt_ransac": None, "hyps": [], "n_iter_required": n_iter_max}
    if N < 4:
        return out
    if solver is None:
        solver = lambda o, i: so.p3p_solve(o, i, cam.K)
    if rng is None:
        rng = np.random.default_rng()
    best_mask, best_count, best_idx, best_Rt = None, 0, -1, None
    iter_done, n_req = 0, n_iter_max
    while iter_done < n_req and iter_done < n_iter_max:
        idx4 = samples[iter_done] if samples is not None else rng.choice(N, 4, replace=False)
        it = iter_done
        iter_done += 1
        Rt = solver(X_corr[idx4], pix_L[idx4])
        out["hyps"].append(Rt)
        if Rt is None:
            continue
        mask = so.hypothesis_mask(cam.K, cam.M_right, Rt, X_corr, pix_L, pix_R, pixel_thresh)
        inl = int(mask.sum())
        if inl > best_count:
            best_count, best_mask, best_idx, best_Rt = inl, mask, it, Rt
            w = best_count / N
            if w > 0:
                n_req = math.ceil(math.log(1 - p_success) / math.log(1 - w ** 4))
    out.update(mask=best_mask if best_mask is not None else out["mask"], best_index=best_idx,
               best_count=best_count, iterations=iter_done, Rt_ransac=best_Rt, n_iter_required=n_req)
    return out


def temporal_stage(cam, prev, curr, pixel_thresh=4.0, samples=None, solver=None,
                   p_success=0.99999, n_iter_max=2500, rng=None, knn2=None, cross_check=False):
    """fill_database.py:185-207 for one pair (cross_check: the frame matcher's
    crossCheck, fill_database.py:70)."""
    qi, ti, di = _bf_match(prev["inlier_des_left"], curr["inlier_des_left"], knn2, cross_check)  # :185
    if len(curr["inlier_des_left"]) == 0:
        qi, ti, di = qi[:0], ti[:0], di[:0]
    X_corr = prev["xyz"][qi]                                                      # :189-190
    pix_L, pix_R = curr["pix_left"][ti], curr["pix_right"][ti]                    # :195-197
    out = {"query_idx": qi, "train_idx": ti, "dist": di, "n_matches": len(X_corr), "X_corr": X_corr,
           "pix_L": pix_L, "pix_R": pix_R}
    out.update(ransac_loop(cam, X_corr, pix_L, pix_R, pixel_thresh, samples, solver, p_success,
                           n_iter_max, rng))
    return out


def run(cam, frames, pixel_thresh=4.0, disparity_min=1.2, samples=None, refine=True,
        stereo_cross_check=False, frame_cross_check=False, **kw):
    stereo = [stereo_stage(cam, f, pixel_thresh, disparity_min, knn2=kw.get("knn2"),
                           cross_check=stereo_cross_check) for f in frames]
    kw = dict(kw, cross_check=frame_cross_check)
    pairs = []
    for p in range(len(frames) - 1):
        s = None if samples is None else samples[p]
        t = temporal_stage(cam, stereo[p], stereo[p + 1], pixel_thresh, samples=s, **kw)
        if refine and t["Rt_ransac"] is not None and t["mask"].sum() >= 3:
            t["Rt"], _ = so.refine_pose(t["X_corr"], t["pix_L"], cam.K, t["Rt_ransac"], t["mask"])
        pairs.append(t)
    return stereo, pairs


def consensus_pnp_match(cam, kf_cand, kf_curr, pixel_thresh=1.5, min_inliers_loop=60,
                        p_success=0.99999, n_iter_max=1500, rng=None, knn2=None, refine=True,
                        cross_check=False, samples=None, solver=None):
    """LoopClosure.consensus_pnp_match (backend/optimizers/loop_closure.py:167-235)
    up to the refined pose.  kf = (links (n,3) [x_left, x_right, y], des_left);
    cross_check = LoopClosure's frame_cross_check (the matcher of :192).
    samples (n_iter_max, 4): the idx4 sequence of the RANSAC loop instead of rng.choice
    (pnp.py:126); solver(obj4, img4): the minimal solver instead of the oracle's P3P."""
    links_i, des_i = kf_cand
    links_n, des_n = kf_curr
    # frame_data, :152-164: cv2.KeyPoint(x, y, 1) keeps float32 coordinates
    f32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
    kpL_i, kpR_i = f32(links_i[:, [0, 2]]), f32(links_i[:, [1, 2]])
    kpL_n, kpR_n = f32(links_n[:, [0, 2]]), f32(links_n[:, [1, 2]])
    obj = so.triangulate_cv(cam.P_left, cam.P_right, kpL_i, kpR_i, out_fp64=True)   # :179-180
    qi, ti, di = _bf_match(des_i, des_n, knn2, cross_check)             # :192-195
    out = {"query_idx": qi, "train_idx": ti, "dist": di, "obj": obj, "ok": False, "Rt": None,
           "mask": np.zeros(len(qi), bool)}
    if len(qi) < min_inliers_loop:                                      # :206-207
        return out
    P3D = obj[qi].astype(np.float32)                                    # :209
    ptsL = kpL_n[ti].astype(np.float32)
    ptsR = kpR_n[ti].astype(np.float32)
    out.update(P3D=P3D, ptsL=ptsL, ptsR=ptsR)
    r = ransac_loop(cam, P3D, ptsL, ptsR, pixel_thresh, samples, solver, p_success, n_iter_max, rng)  # :215-216
    out.update(r)
    if r["mask"].sum() < min_inliers_loop:                              # :218-219
        return out
    if refine:
        out["Rt"], _ = so.refine_pose(P3D, ptsL, cam.K, r["Rt_ransac"], r["mask"])   # :222
    out["ok"] = True
    return out
