"""Seeded synthetic inputs of KITTI shape for tests, smoke and benchmarks:
AKAZE-like binary descriptors and a stereo sequence (1226x370 images, KITTI
04-12 calibration, 0.8 m/frame forward motion) observed with pixel noise,
descriptor bit noise and clutter.  Pure NumPy; produces inputs only.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import numpy as np

from ..frontend.io.camera_model import CameraModel

IMG_W, IMG_H = 1226, 370


def random_descriptors(rng: np.random.Generator, n: int, nbytes: int = 61) -> np.ndarray:
    d = rng.integers(0, 256, size=(n, nbytes), dtype=np.uint8)
    if nbytes == 61:
        d[:, 60] &= 0x3F  # AKAZE MLDB has 486 valid bits
    return d


def flip_bits(rng: np.random.Generator, desc: np.ndarray, p: float) -> np.ndarray:
    """Flip each bit independently with probability ~p (positions drawn with
    replacement: a position drawn twice flips back, which is immaterial here)."""
    out = np.ascontiguousarray(desc).copy()
    total_bits = out.size * 8
    if total_bits == 0 or p <= 0:
        return out
    k = int(rng.binomial(total_bits, p))
    pos = rng.integers(0, total_bits, k)
    flat = out.reshape(-1)
    np.bitwise_xor.at(flat, pos >> 3, (1 << (pos & 7)).astype(np.uint8))
    if desc.shape[1] == 61:
        out[:, 60] &= 0x3F
    return out


def matcher_workload(nq: int = 8192, nt: int = 8192, nbytes: int = 61, seed: int = 0
                     ) -> Tuple[np.ndarray, np.ndarray]:
    """Query/train descriptor sets with structure: 70 % of the queries are noisy
    copies (8 % bit flips) of distinct train rows, 30 % are unrelated, and 1 % of
    the train rows are exact duplicates of other train rows so that equal
    distances (ties) occur for best and second best."""
    rng = np.random.default_rng(seed)
    train = random_descriptors(rng, nt, nbytes)
    n_dup = max(1, nt // 100) if nt > 1 else 0
    if n_dup:
        src = rng.integers(0, nt, n_dup)
        dst = rng.integers(0, nt, n_dup)
        train[dst] = train[src]
    query = random_descriptors(rng, nq, nbytes)
    if nt > 0:
        n_copy = int(0.7 * nq)
        which = rng.permutation(nq)[:n_copy]
        from_rows = rng.integers(0, nt, n_copy)
        query[which] = flip_bits(rng, train[from_rows], 0.08)
    return query, train


@dataclass
class StereoFrame:
    kp_left: np.ndarray    # (nL,2) float32
    kp_right: np.ndarray   # (nR,2) float32
    des_left: np.ndarray   # (nL,D) uint8
    des_right: np.ndarray  # (nR,D) uint8
    pose_w2c: np.ndarray   # (3,4) fp64 world -> left camera


def _yaw(a: float) -> np.ndarray:
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


class StereoSequence:
    """A corridor of static landmarks seen by a forward-moving stereo rig."""

    def __init__(self, cam: CameraModel, n_frames: int, n_features=1500, seed: int = 0,
                 nbytes: int = 61, clutter: float = 0.35, pixel_sigma: float = 0.5,
                 bit_noise: float = 0.03, landmark_density: float = 1.0):
        """`n_features` is the number of keypoints per image: an int, or a (lo, hi)
        pair for a per-frame count drawn uniformly from [lo, hi].
        `landmark_density` scales the number of landmarks in the corridor.  Every frame
        shows a random choice of the landmarks in view, so with many more in view than
        shown (1.0: about 2.5 times as many) consecutive frames share about 40 % of their
        true features; fewer landmarks make the frames share more of them."""
        if isinstance(n_features, (tuple, list)):
            self.n_features_range = (int(n_features[0]), int(n_features[1]))
            n_features = self.n_features_range[1]
        else:
            self.n_features_range = None
        self.cam, self.n_frames, self.n_features = cam, n_frames, n_features
        self.nbytes, self.clutter = nbytes, clutter
        self.pixel_sigma, self.bit_noise, self.seed = pixel_sigma, bit_noise, seed
        rng = np.random.default_rng(seed)
        length = 120.0 + 0.8 * n_frames
        # landmark density chosen so that ~n_features*(1-clutter) are in view
        n_land = int(n_features * (1 - clutter) * length / 45.0 * landmark_density) + 64
        self.land_xyz = np.column_stack([rng.uniform(-25, 25, n_land), rng.uniform(-3, 2, n_land),
                                         rng.uniform(2, length, n_land)])
        self.land_des = random_descriptors(rng, n_land, nbytes)
        # heading = bounded weave (+-3 deg, 200-frame period) plus jitter, so the rig
        # stays inside the landmark corridor however long the sequence is
        self.poses = []
        pos = np.zeros(3)
        jitter = np.deg2rad(rng.uniform(-0.3, 0.3, n_frames))
        for k in range(n_frames):
            R = _yaw(np.deg2rad(3.0) * np.sin(2 * np.pi * k / 200.0) + jitter[k])
            self.poses.append(np.hstack([R.T, (-R.T @ pos)[:, None]]))  # world -> camera
            pos = pos + R @ np.array([0.0, 0.0, 0.8])

    def frame(self, k: int) -> StereoFrame:
        rng = np.random.default_rng((self.seed + 1) * 1_000_003 + k)
        cam, Rt = self.cam, self.poses[k]
        n_feat = self.n_features
        if self.n_features_range is not None:
            n_feat = int(rng.integers(self.n_features_range[0], self.n_features_range[1] + 1))
        Xc = self.land_xyz @ Rt[:, :3].T + Rt[:, 3]
        vis = (Xc[:, 2] > 4.0) & (Xc[:, 2] < 120.0)
        Xc, des = Xc[vis], self.land_des[vis]
        Xh = np.column_stack([Xc, np.ones(len(Xc))])
        pl = Xh @ cam.P_left.T
        pr = Xh @ cam.P_right.T
        uvl = pl[:, :2] / pl[:, 2:3]
        uvr = pr[:, :2] / pr[:, 2:3]
        inside = ((uvl[:, 0] > 0) & (uvl[:, 0] < IMG_W) & (uvl[:, 1] > 0) & (uvl[:, 1] < IMG_H) &
                  (uvr[:, 0] > 0) & (uvr[:, 0] < IMG_W))
        uvl, uvr, des = uvl[inside], uvr[inside], des[inside]
        n_true = min(len(uvl), int(n_feat * (1 - self.clutter)))
        sel = rng.permutation(len(uvl))[:n_true]
        uvl, uvr, des = uvl[sel], uvr[sel], des[sel]
        uvl = uvl + rng.normal(0, self.pixel_sigma, uvl.shape)
        uvr = uvr + rng.normal(0, self.pixel_sigma, uvr.shape)
        des_l = flip_bits(rng, des, self.bit_noise)
        des_r = flip_bits(rng, des, self.bit_noise)
        n_cl = n_feat - n_true

        def clutter_pts(n):
            return np.column_stack([rng.uniform(0, IMG_W, n), rng.uniform(0, IMG_H, n)])

        kp_l = np.vstack([uvl, clutter_pts(n_cl)])
        kp_r = np.vstack([uvr, clutter_pts(n_cl)])
        d_l = np.vstack([des_l, random_descriptors(rng, n_cl, self.nbytes)])
        d_r = np.vstack([des_r, random_descriptors(rng, n_cl, self.nbytes)])
        pl_, pr_ = rng.permutation(len(kp_l)), rng.permutation(len(kp_r))
        return StereoFrame(kp_l[pl_].astype(np.float32), kp_r[pr_].astype(np.float32),
                           np.ascontiguousarray(d_l[pl_]), np.ascontiguousarray(d_r[pr_]), Rt)

    def relative_pose(self, k: int) -> np.ndarray:
        """[R|t] taking frame k-1 left-camera coordinates to frame k's."""
        A = np.vstack([self.poses[k - 1], [0, 0, 0, 1]])
        B = np.vstack([self.poses[k], [0, 0, 0, 1]])
        return (B @ np.linalg.inv(A))[:3]


def perturbed_hypotheses(rng: np.random.Generator, Rt_true: np.ndarray, h: int,
                         good_fraction: float = 0.3) -> Tuple[np.ndarray, np.ndarray]:
    """(h,3,4) fp64 pose hypotheses around a true pose: a fraction within a few
    millimetres / millidegrees, the rest grossly off; ~2 % flagged invalid."""
    out = np.empty((h, 3, 4), dtype=np.float64)
    for i in range(h):
        good = rng.random() < good_fraction
        ang = rng.normal(0, 2e-4 if good else 3e-2, 3)
        dt = rng.normal(0, 2e-3 if good else 0.5, 3)
        ax, ay, az = ang
        Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
        Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
        R = Rx @ _yaw(ay) @ Rz @ Rt_true[:, :3]
        out[i] = np.hstack([R, (Rt_true[:, 3] + dt)[:, None]])
    valid = (rng.random(h) > 0.02).astype(np.uint8)
    return out, valid
