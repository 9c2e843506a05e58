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

`PackedFrames` is the same content as ONE array set: the four arrays of all frames
concatenated, plus row offsets per frame.  It is the shape the device wants (the
batched front-end takes concatenated arrays and offsets), so handing it over costs
four copies instead of a concatenation of four arrays per frame; `pin()` moves it
into page-locked memory, from which those copies run asynchronously at the full
host-to-device rate.  `save` / `load` keep a whole sequence in one `.npz`.
"""
from __future__ import annotations

import os
import re
from typing import List, Optional, Sequence, Tuple

import numpy as np

Frame = Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]
_NAME = re.compile(r"^(\d+)\.npz$")


def binary_descriptors(des, what: str = "descriptors") -> np.ndarray:
    """The descriptor rows as a contiguous uint8 array.  The batched front-end and the pair
    verification match under the Hamming norm only: the reference's matcher takes that norm
    exactly when the descriptors are uint8 (descriptor_matcher.py:23-28) and the L2 norm
    otherwise, so anything else (SuperPoint's float rows) is refused here instead of being cast
    -- `DescriptorMatcher.match` matches such rows one pair at a time."""
    a = np.asarray(des)
    if a.dtype != np.uint8 and a.size:
        raise TypeError(f"{what} must be uint8 (binary descriptors, Hamming norm), got {a.dtype}: float descriptors "
                        f"are matched under the L2 norm by DescriptorMatcher.match, one pair per call")
    return np.ascontiguousarray(a, dtype=np.uint8)


def frame_path(directory: str, k: int) -> str:
    return os.path.join(directory, f"{k:06d}.npz")


def save_frame(directory: str, k: int, kp_left, kp_right, des_left, des_right) -> str:
    """Write frame k.  Accepts arrays or lists of objects with a `.pt` attribute."""
    def pts(kp):
        if isinstance(kp, np.ndarray):
            return np.ascontiguousarray(kp, dtype=np.float32).reshape(-1, 2)
        return np.array([k_.pt for k_ in kp], dtype=np.float32).reshape(-1, 2)

    kl, kr = pts(kp_left), pts(kp_right)
    dl, dr = binary_descriptors(des_left, "des_left"), binary_descriptors(des_right, "des_right")
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


class PackedFrames(Sequence):
    """A run of frames as concatenated arrays + offsets (frame f owns rows
    off_left[f]:off_left[f+1] of kp_left / des_left, likewise on the right)."""

    def __init__(self, kp_left, kp_right, des_left, des_right, off_left, off_right, _pinned=None):
        self.kp_left, self.kp_right = kp_left, kp_right
        self.des_left, self.des_right = des_left, des_right
        self.off_left = np.ascontiguousarray(off_left, dtype=np.int32)
        self.off_right = np.ascontiguousarray(off_right, dtype=np.int32)
        self._pinned = _pinned        # the page-locked torch tensors the arrays are views of, if any
        n = len(self.off_left) - 1
        if n < 0 or len(self.off_right) != n + 1:
            raise ValueError("offsets must have n_frames + 1 entries")
        if self.off_left[0] != 0 or self.off_right[0] != 0:
            raise ValueError("offsets start at 0")
        if len(kp_left) != self.off_left[-1] or len(des_left) != self.off_left[-1] or \
                len(kp_right) != self.off_right[-1] or len(des_right) != self.off_right[-1]:
            raise ValueError("array lengths do not match the offsets")

    @classmethod
    def from_frames(cls, frames: Sequence[Frame]) -> "PackedFrames":
        if len(frames) < 1:
            raise ValueError("need at least one frame")
        for f in frames:
            if len(f[2]) != len(f[0]) or len(f[3]) != len(f[1]):
                raise ValueError("keypoints and descriptors differ in length")
        d = np.asarray(frames[0][2]).shape[1]
        off_l = np.concatenate([[0], np.cumsum([len(f[0]) for f in frames])]).astype(np.int32)
        off_r = np.concatenate([[0], np.cumsum([len(f[1]) for f in frames])]).astype(np.int32)
        def cat(i, dt, w):
            rows = [binary_descriptors(f[i]) if dt == np.uint8 else np.asarray(f[i], dtype=dt) for f in frames]
            return np.concatenate([r.reshape(-1, w) for r in rows])
        return cls(cat(0, np.float32, 2), cat(1, np.float32, 2), cat(2, np.uint8, d), cat(3, np.uint8, d),
                   off_l, off_r)

    @property
    def is_pinned(self) -> bool:
        return self._pinned is not None

    def pin(self) -> "PackedFrames":
        """A copy in page-locked memory (needs a GPU runtime; returns self where there is none)."""
        import torch
        if self._pinned is not None or not torch.cuda.is_available():
            return self
        t = {}
        for name in ("kp_left", "kp_right", "des_left", "des_right"):
            a = np.ascontiguousarray(getattr(self, name))
            t[name] = torch.empty(a.shape, dtype=torch.from_numpy(a[:0]).dtype, pin_memory=True)
            t[name].numpy()[...] = a
        return PackedFrames(t["kp_left"].numpy(), t["kp_right"].numpy(), t["des_left"].numpy(),
                            t["des_right"].numpy(), self.off_left, self.off_right, _pinned=t)

    def tensors(self):
        """The four arrays as torch CPU tensors (page-locked ones after pin())."""
        import torch
        if self._pinned is not None:
            return tuple(self._pinned[k] for k in ("kp_left", "kp_right", "des_left", "des_right"))
        return tuple(torch.from_numpy(np.ascontiguousarray(a))
                     for a in (self.kp_left, self.kp_right, self.des_left, self.des_right))

    def __len__(self) -> int:
        return len(self.off_left) - 1

    def __getitem__(self, i):
        n = len(self)
        if isinstance(i, slice):
            lo, hi, step = i.indices(n)
            if step != 1:
                raise ValueError("only contiguous frame ranges")
            hi = max(hi, lo)
            l0, l1, r0, r1 = self.off_left[lo], self.off_left[hi], self.off_right[lo], self.off_right[hi]
            pinned = None
            if self._pinned is not None:      # row ranges of page-locked tensors stay page-locked
                pinned = {"kp_left": self._pinned["kp_left"][l0:l1], "kp_right": self._pinned["kp_right"][r0:r1],
                          "des_left": self._pinned["des_left"][l0:l1], "des_right": self._pinned["des_right"][r0:r1]}
            return PackedFrames(self.kp_left[l0:l1], self.kp_right[r0:r1], self.des_left[l0:l1],
                                self.des_right[r0:r1], self.off_left[lo:hi + 1] - l0,
                                self.off_right[lo:hi + 1] - r0, _pinned=pinned)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        ls, rs = slice(self.off_left[i], self.off_left[i + 1]), slice(self.off_right[i], self.off_right[i + 1])
        return (self.kp_left[ls], self.kp_right[rs], self.des_left[ls], self.des_right[rs])

    def save(self, path: str) -> str:
        np.savez(path, kp_left=self.kp_left, kp_right=self.kp_right, des_left=self.des_left,
                 des_right=self.des_right, off_left=self.off_left, off_right=self.off_right)
        return path if path.endswith(".npz") else path + ".npz"

    @classmethod
    def load(cls, path: str) -> "PackedFrames":
        with np.load(path, allow_pickle=False) as z:
            return cls(np.ascontiguousarray(z["kp_left"], dtype=np.float32).reshape(-1, 2),
                       np.ascontiguousarray(z["kp_right"], dtype=np.float32).reshape(-1, 2),
                       np.ascontiguousarray(z["des_left"], dtype=np.uint8),
                       np.ascontiguousarray(z["des_right"], dtype=np.uint8), z["off_left"], z["off_right"])


def write_sequence(directory: str, frames: Sequence[Frame]) -> List[str]:
    return [save_frame(directory, k, *f) for k, f in enumerate(frames)]
