"""Plain-Python stand-ins for the two OpenCV value types that cross the
reference's call sites, so nothing here needs cv2: `DMatch` (queryIdx, trainIdx,
imgIdx, distance) and `KeyPoint` (.pt).  Real cv2 objects work wherever these
do, because only those attributes are read."""
from __future__ import annotations

from typing import Iterable, Sequence, Tuple

import numpy as np


class DMatch:
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx: int = -1, trainIdx: int = -1, imgIdx: int = 0,
                 distance: float = float("inf")):
        self.queryIdx = int(queryIdx)
        self.trainIdx = int(trainIdx)
        self.imgIdx = int(imgIdx)
        self.distance = float(distance)

    def __repr__(self):
        return f"DMatch(q={self.queryIdx}, t={self.trainIdx}, d={self.distance:g})"


class KeyPoint:
    __slots__ = ("pt", "size")

    def __init__(self, x: float, y: float, size: float = 1.0):
        # cv2.KeyPoint stores float32 coordinates and returns them as Python floats.
        self.pt = (float(np.float32(x)), float(np.float32(y)))
        self.size = float(size)


def keypoints_to_array(kps) -> np.ndarray:
    """(n,2) float32 pixel array from a keypoint sequence or an array."""
    if isinstance(kps, np.ndarray):
        return np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 2)
    if len(kps) == 0:
        return np.zeros((0, 2), dtype=np.float32)
    return np.array([k.pt for k in kps], dtype=np.float32).reshape(-1, 2)


def matches_to_arrays(matches: Iterable) -> Tuple[np.ndarray, np.ndarray]:
    """(queryIdx, trainIdx) int32 arrays of a match sequence."""
    if isinstance(matches, MatchList):
        return matches.query_idx, matches.train_idx
    q = np.fromiter((m.queryIdx for m in matches), dtype=np.int32)
    t = np.fromiter((m.trainIdx for m in matches), dtype=np.int32)
    return q, t


class MatchList(Sequence):
    """A list-of-DMatch view over three arrays.  Behaves like the Python list the
    reference returns (len, index, slice, iterate) but builds DMatch objects only
    when an element is touched; array consumers read the arrays directly."""

    def __init__(self, query_idx, train_idx, distance):
        self.query_idx = np.ascontiguousarray(query_idx, dtype=np.int32)
        self.train_idx = np.ascontiguousarray(train_idx, dtype=np.int32)
        self.distance = np.ascontiguousarray(distance, dtype=np.float32)

    def __len__(self):
        return int(self.query_idx.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return MatchList(self.query_idx[i], self.train_idx[i], self.distance[i])
        if isinstance(i, (np.ndarray, list)):
            i = np.asarray(i)
            return MatchList(self.query_idx[i], self.train_idx[i], self.distance[i])
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return DMatch(self.query_idx[i], self.train_idx[i], 0, self.distance[i])

    def __repr__(self):
        return f"MatchList(n={len(self)})"
