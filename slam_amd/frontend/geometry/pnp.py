"""PnP-RANSAC behind the reference's `PnP` interface (frontend/geometry/pnp.py
of the reference).

The reference evaluates one hypothesis per Python iteration.  Here the sample
indices of a whole block of iterations are drawn first (same generator call per
iteration, `rng.choice(N, 4, replace=False)`), all their minimal solutions are
scored against every point in one kernel launch, and the reference's adaptive
stopping rule is then replayed over the ordered counts.  For a given sequence
of samples the returned mask is the one the sequential loop would return.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from ... import kernels
from ...types import keypoints_to_array

# numeric values of the cv2 flags the reference passes to solve_pnp
SOLVEPNP_ITERATIVE = 0
SOLVEPNP_AP3P = 5


class PnP:
    def __init__(self, cam, pixel_thresh):
        self.K = cam.K
        self.M_left = cam.M_left
        self.M_right = cam.M_right
        self.P_left = cam.P_left
        self.P_right = cam.P_right
        self.pixel_thresh = pixel_thresh

    # --------------------------------------------------------------- solve_pnp
    def _rotation_vec_to_mat(self, R_vec, t_vec):
        """[R|t] (3,4) from a rotation vector and a translation (pnp.py:29-33; the
        reference calls cv2.Rodrigues).  Host arithmetic on three numbers: the device
        solvers hand out [R|t] directly and never go through a rotation vector; this is
        here for callers that hold one."""
        r = np.asarray(R_vec, dtype=np.float64).reshape(3)
        t = np.asarray(t_vec, dtype=np.float64).reshape(3, 1)
        theta = float(np.linalg.norm(r))
        if theta < np.finfo(np.float64).eps:
            R = np.eye(3)
        else:
            k = r / theta
            Kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
            c, s_ = np.cos(theta), np.sin(theta)
            R = c * np.eye(3) + (1.0 - c) * np.outer(k, k) + s_ * Kx
        return np.hstack((R, t))

    def solve_pnp(self, obj_pts, img_pts, flag):
        """3x4 [R|t] or None.  flag = SOLVEPNP_AP3P: minimal solve from the first
        three correspondences, the fourth picks the root.  flag =
        SOLVEPNP_ITERATIVE: least-squares refinement over all correspondences."""
        obj = np.asarray(obj_pts, dtype=np.float64).reshape(-1, 3)
        img = np.asarray(img_pts, dtype=np.float64).reshape(-1, 2)
        if flag == SOLVEPNP_AP3P:
            if obj.shape[0] != 4:
                return None
            Rts, valid = kernels.p3p_solve(obj[None], img[None], self.K)
            return Rts[0].cpu().numpy() if bool(valid[0].item()) else None
        if flag == SOLVEPNP_ITERATIVE:
            return kernels.refine_pose(obj, img, self.K)
        raise ValueError(f"unsupported solvePnP flag: {flag}")

    # ---------------------------------------------------------- index plumbing
    def _convert_left_to_right(self, intra_matches):
        return {m.queryIdx: m.trainIdx for m in intra_matches}

    def create_L_R_pixels(self, matches01, intra_matches, intra_inlier_matches_indices,
                          kps_L, kps_R, pair_0_or_1):
        """Left and right pixels (float32) of the stereo matches that the
        frame-to-frame matches refer to."""
        if pair_0_or_1 not in (0, 1):
            raise ValueError("pair_0_or_1 should be either 0 or 1")
        from ...types import matches_to_arrays
        q01, t01 = matches_to_arrays(matches01)
        sel = q01 if pair_0_or_1 == 0 else t01
        inl = np.asarray(intra_inlier_matches_indices, dtype=np.int64)
        L_index = inl[sel]
        qi, ti = matches_to_arrays(intra_matches)
        kl, kr = keypoints_to_array(kps_L), keypoints_to_array(kps_R)
        l_to_r = np.full(max(kl.shape[0], int(qi.max()) + 1 if len(qi) else 0), -1, dtype=np.int64)
        l_to_r[qi] = ti  # later entries win, as in a dict comprehension
        R_index = l_to_r[L_index]
        if (R_index < 0).any():
            raise KeyError(int(L_index[np.flatnonzero(R_index < 0)[0]]))
        return (kl[L_index].astype(np.float32).reshape(-1, 2),
                kr[R_index].astype(np.float32).reshape(-1, 2))

    # ---------------------------------------------------------------- geometry
    def supporters_pair(self, meas_L, proj_L, meas_R, proj_R):
        return kernels.supporters_pair(meas_L, proj_L, meas_R, proj_R,
                                       self.pixel_thresh).cpu().numpy().astype(bool)

    @staticmethod
    def project(Xw, P):
        return kernels.project(np.asarray(Xw), P).cpu().numpy()

    # ------------------------------------------------------------------ RANSAC
    @staticmethod
    def _replay(counts, N, p_success, n_iter_max, early_inliers, state):
        """Sequential adaptive-stop logic over `counts`, continuing `state` =
        [best_idx, best_count, iter_done, n_iter_required]."""
        m = 4
        best_idx, best_count, iter_done, n_req = state
        base = iter_done
        stop = False
        while iter_done < n_req and iter_done < n_iter_max and iter_done - base < len(counts):
            c = int(counts[iter_done - base])
            idx = iter_done
            iter_done += 1
            if c >= 0 and c > best_count:
                best_count, best_idx = c, idx
                inlier_ratio = best_count / N
                if inlier_ratio > 0:
                    prob_hit_all_inliers = inlier_ratio ** m
                    n_req = math.ceil(math.log(1 - p_success) / math.log(1 - prob_hit_all_inliers))
            if early_inliers is not None and best_count >= early_inliers:
                stop = True
                break
        if not (iter_done < n_req and iter_done < n_iter_max):
            stop = True
        return [best_idx, best_count, iter_done, n_req], stop

    def ransac_pnp_pair1(self, obj_pts, meas_L1, meas_R1, p_success=0.99999, n_iter_max=2500,
                         early_inliers=None, *, rng=None, hypotheses=None, block=256):
        """Boolean inlier mask of the best hypothesis, or None if no hypothesis
        had any inlier.  `rng` defaults to an unseeded NumPy generator like the
        reference; `hypotheses` = (Rts (H,3,4), valid (H,)) bypasses sampling and
        the minimal solver (iteration i evaluates hypothesis i)."""
        obj = np.ascontiguousarray(obj_pts)
        N = len(obj)
        pts_dev = kernels.to_device(obj, torch.float64 if obj.dtype == np.float64 else torch.float32)
        mL = kernels.to_device(np.asarray(meas_L1), torch.float32)
        mR = kernels.to_device(np.asarray(meas_R1), torch.float32)
        state = [-1, 0, 0, n_iter_max]
        best_Rt = None
        if hypotheses is not None:
            Rts_all = kernels.to_device(hypotheses[0], torch.float64)
            valid_all = None if hypotheses[1] is None else kernels.to_device(hypotheses[1], torch.uint8)
            n_avail = Rts_all.shape[0]
        else:
            if rng is None:
                rng = np.random.default_rng()
            n_avail = n_iter_max
        while True:
            start = state[2]
            if start >= n_avail:
                break
            nblk = min(block, n_avail - start, max(1, n_iter_max - start))
            if hypotheses is not None:
                Rts = Rts_all[start:start + nblk]
                valid = None if valid_all is None else valid_all[start:start + nblk]
            else:
                idx4 = np.stack([rng.choice(N, 4, replace=False) for _ in range(nblk)])
                Rts, valid = kernels.p3p_solve(obj[idx4], np.asarray(meas_L1)[idx4], self.K)
            counts = kernels.ransac_score(pts_dev, mL, mR, Rts, valid, self.K, self.M_right,
                                          self.pixel_thresh).cpu().numpy()
            prev_best = state[0]
            state, stop = self._replay(counts, N, p_success, n_iter_max, early_inliers, state)
            if state[0] != prev_best:
                best_Rt = Rts[state[0] - start].clone()
            if stop:
                break
        if best_Rt is None:
            return None
        mask = kernels.ransac_mask(pts_dev, mL, mR, best_Rt, self.K, self.M_right, self.pixel_thresh)
        return mask.cpu().numpy().astype(bool)
