"""Stereo rig description: K, [R|t] of both cameras, and P = K [R|t].

Mirrors the attribute names of the reference's CameraModel
(frontend/io/camera_model.py:18-36) because the geometry classes read
`cam.K`, `cam.M_left`, `cam.M_right`, `cam.P_left`, `cam.P_right`.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np


class CameraModel:
    def __init__(self, K: np.ndarray, M_left: np.ndarray, M_right: np.ndarray):
        self.K = np.asarray(K, dtype=np.float64)
        self.M_left = np.asarray(M_left, dtype=np.float64)
        self.M_right = np.asarray(M_right, dtype=np.float64)
        if self.K.shape != (3, 3) or self.M_left.shape != (3, 4) or self.M_right.shape != (3, 4):
            raise ValueError("K must be 3x3 and M_left/M_right 3x4")
        self.P_left = self.K @ self.M_left
        self.P_right = self.K @ self.M_right

    @classmethod
    def from_projections(cls, P0, P1) -> "CameraModel":
        """From the two 3x4 KITTI projection matrices (calib.txt rows P0, P1)."""
        P0 = np.asarray(P0, dtype=np.float64).reshape(3, 4)
        P1 = np.asarray(P1, dtype=np.float64).reshape(3, 4)
        K = P0[:, :3]
        K_inv = np.linalg.inv(K)
        return cls(K, K_inv @ P0, K_inv @ P1)

    @classmethod
    def from_kitti(cls, calib_txt) -> "CameraModel":
        """Parse the first two rows of a KITTI calib.txt (same contract as
        frontend/io/camera_model.py:39-73: FileNotFoundError if absent)."""
        path = Path(calib_txt)
        if not path.exists():
            raise FileNotFoundError(f"Calibration file not found: {path}")
        with open(path, "r") as fh:
            rows = [fh.readline().split()[1:] for _ in range(2)]
        return cls.from_projections([float(v) for v in rows[0]], [float(v) for v in rows[1]])

    @classmethod
    def kitti_like(cls, fx=707.0912, fy=707.0912, cx=601.8873, cy=183.1104,
                   baseline=0.5372) -> "CameraModel":
        """Synthetic rig with KITTI 04-12 shaped intrinsics (for benchmarks)."""
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)
        M_l = np.hstack([np.eye(3), np.zeros((3, 1))])
        M_r = np.hstack([np.eye(3), np.array([[-baseline], [0.0], [0.0]])])
        return cls(K, M_l, M_r)
