"""Pre-extracted stereo features on disk, one `.npz` per frame.

The reference extracts AKAZE features from the KITTI images inside its frame
loop (backend/tracking/fill_database.py:140-152, via
frontend/vision/feature_extractor.py).  Detection is outside this package, so
a sequence is handed over as a directory of per-frame files holding exactly
what `detect_and_compute` returns for the two images:

    kp_left  (nL, 2) float32   keypoint.pt of the left image
    kp_right (nR, 2) float32
    des_left (nL, D) uint8     D = 61 for AKAZE, 32 for ORB
    des_right(nR, D) uint8

`FeatureCache` is a lazy sequence: indexing or slicing reads only the frames
asked for, so every rank of a sharded run loads just its own frame range.
Files are read with `numpy.load(allow_pickle=False)`.
"""
from __future__ import annotations

import os
import re
from typing import List, Optional, Sequence, Tuple

import numpy as np

Frame = Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]
_NAME = re.compile(r"^(\d+)\.npz$")


def frame_path(directory: str, k: int) -> str:
    return os.path.join(directory, f"{k:06d}.npz")


def save_frame(directory: str, k: int, kp_left, kp_right, des_left, des_right) -> str:
    """Write frame k.  Accepts arrays or lists of objects with a `.pt` attribute."""
    def pts(kp):
        if isinstance(kp, np.ndarray):
            return np.ascontiguousarray(kp, dtype=np.float32).reshape(-1, 2)
        return np.array([k_.pt for k_ in kp], dtype=np.float32).reshape(-1, 2)

    kl, kr = pts(kp_left), pts(kp_right)
    dl = np.ascontiguousarray(des_left, dtype=np.uint8)
    dr = np.ascontiguousarray(des_right, dtype=np.uint8)
    if dl.ndim != 2 or dr.ndim != 2 or dl.shape[1] != dr.shape[1]:
        raise ValueError("descriptors must be (n, D) uint8 with the same D for both images")
    if len(kl) != len(dl) or len(kr) != len(dr):
        raise ValueError("one descriptor row per keypoint is required")
    os.makedirs(directory, exist_ok=True)
    path = frame_path(directory, k)
    np.savez(path, kp_left=kl, kp_right=kr, des_left=dl, des_right=dr)
    return path


def load_frame(path: str) -> Frame:
    with np.load(path, allow_pickle=False) as z:
        return (np.ascontiguousarray(z["kp_left"], dtype=np.float32),
                np.ascontiguousarray(z["kp_right"], dtype=np.float32),
                np.ascontiguousarray(z["des_left"], dtype=np.uint8),
                np.ascontiguousarray(z["des_right"], dtype=np.uint8))


class FeatureCache(Sequence):
    """Frames 0..n-1 of a directory written by `save_frame` (gaps are an error)."""

    def __init__(self, directory: str, max_frames: Optional[int] = None):
        if not os.path.isdir(directory):
            raise FileNotFoundError(f"feature cache directory not found: {directory}")
        ids = sorted(int(m.group(1)) for m in (_NAME.match(n) for n in os.listdir(directory)) if m)
        if not ids:
            raise FileNotFoundError(f"no NNNNNN.npz frame files in {directory}")
        if ids != list(range(len(ids))):
            raise ValueError(f"frame files in {directory} must be numbered 0..n-1 without gaps")
        self.directory = directory
        self.n = len(ids) if max_frames is None else min(len(ids), int(max_frames))

    def __len__(self) -> int:
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [load_frame(frame_path(self.directory, k)) for k in range(*i.indices(self.n))]
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        return load_frame(frame_path(self.directory, i))


def write_sequence(directory: str, frames: Sequence[Frame]) -> List[str]:
    return [save_frame(directory, k, *f) for k, f in enumerate(frames)]
