"""Stereo row/disparity filter and DLT triangulation behind the reference's
`Triangulation` interface (frontend/geometry/triangulation.py of the
reference), executed by the fused filter + Jacobi-DLT HIP kernel."""
from __future__ import annotations

import numpy as np

from ... import kernels
from ...types import MatchList, keypoints_to_array, matches_to_arrays


class Triangulation:
    def __init__(self, cam, disparity_min, pixel_thresh):
        self.cam = cam
        self.P_left = cam.P_left
        self.P_right = cam.P_right
        self.disparity_min = disparity_min
        self.pixel_thresh = pixel_thresh

    # ------------------------------------------------------------------ arrays
    def filter_triangulate_arrays(self, kp_left, kp_right, qidx, tidx):
        """keep (m,) uint8 and xyz (m,3) float32 device tensors for matches
        (qidx[k], tidx[k]); xyz rows of rejected matches are zero."""
        return kernels.stereo_filter_triangulate(
            kp_left, kp_right, qidx, tidx, self.P_left, self.P_right,
            self.pixel_thresh, self.disparity_min, do_filter=True, fp64=False)

    # ------------------------------------------------- reference-shaped methods
    def create_inlier_outlier_matches(self, matches, kpR, kpL):
        """(inlier_matches, outlier_matches).  As in the reference, the first
        keypoint argument is indexed by queryIdx and the second by trainIdx (the
        caller passes left then right), and match order is preserved."""
        q, t = matches_to_arrays(matches)
        if len(q) == 0:
            return [], []
        keep, _ = kernels.stereo_filter_triangulate(
            keypoints_to_array(kpR), keypoints_to_array(kpL), q, t, self.P_left, self.P_right,
            self.pixel_thresh, self.disparity_min, do_filter=True, fp64=False)
        keep = keep.cpu().numpy().astype(bool)
        if isinstance(matches, MatchList):
            return matches[np.flatnonzero(keep)], matches[np.flatnonzero(~keep)]
        inl = [m for m, k in zip(matches, keep) if k]
        out = [m for m, k in zip(matches, keep) if not k]
        return inl, out

    def triangulate_opencv(self, kps_left, kps_right, matches):
        """(N,3) float32 points in the left camera frame (what
        cv2.triangulatePoints + dehomogenisation gives for float32 pixels)."""
        q, t = matches_to_arrays(matches)
        _, xyz = kernels.stereo_filter_triangulate(
            keypoints_to_array(kps_left), keypoints_to_array(kps_right), q, t,
            self.P_left, self.P_right, do_filter=False, fp64=False)
        return xyz.cpu().numpy()

    # The two builders of the reference's own DLT (triangulation.py:32-46), for callers
    # that assemble a system themselves.  triangulate_matches_batched below does not go
    # through them: the kernel builds every point's 4 x 4 system in registers.
    def _build_single_camera_A_matrix(self, x_pix, y_pix, camera_matrix):
        """(2,4) block of one camera for one point: rows y*P[2] - P[1] and P[0] - x*P[2]."""
        P = np.asarray(camera_matrix, dtype=np.float64)
        return np.vstack([(y_pix * P[2]) - P[1], P[0] - (x_pix * P[2])])

    def _build_A_matrix(self, A1, A2):
        """(4,4) system of one point from the two cameras' blocks."""
        return np.vstack([A1, A2])

    def triangulate_matches_batched(self, kps_left, kps_right, matches):
        """(N,3) float64 points: the reference's own least-squares DLT.  Same
        null space as triangulate_opencv's system, kept in fp64 throughout."""
        q, t = matches_to_arrays(matches)
        kl = keypoints_to_array(kps_left).astype(np.float64)
        kr = keypoints_to_array(kps_right).astype(np.float64)
        _, xyz = kernels.stereo_filter_triangulate(kl, kr, q, t, self.P_left, self.P_right,
                                                   do_filter=False, fp64=True)
        return xyz.cpu().numpy()

    def compare_triangulations(self, X_custom, X_cv):
        dists = np.linalg.norm(np.asarray(X_custom) - np.asarray(X_cv), axis=1)
        print(f"Median L2 distance (custom vs OpenCV): {np.median(dists):.8f}")
