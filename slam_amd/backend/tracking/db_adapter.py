"""Hand-over from the batched front-end to the reference's unchanged back-end.

The reference's TrackingPipeline.build() ends each iteration with
`db.add_frame(links, inlier_des_L, matches01, inlier_mask)` and finally pickles
`np.stack(relative_Rts)` to a file named `relative_Rts`
(backend/tracking/fill_database.py:133,231,249-255).  Track ids are assigned
sequentially inside TrackingDB.add_frame (backend/tracking/database.py:313-377),
so that bookkeeping is replayed in frame order on the host from the gathered
records; everything that feeds it was computed on the GPUs.
"""
from __future__ import annotations

import pickle
from typing import Callable, Optional

import numpy as np

from ...types import MatchList


class Link:
    """Same three fields as the reference's Link (database.py:20-32)."""
    __slots__ = ("x_left", "x_right", "y")

    def __init__(self, x_left, x_right, y):
        self.x_left, self.x_right, self.y = x_left, x_right, y

    def left_keypoint(self):
        return np.array([self.x_left, self.y])

    def right_keypoint(self):
        return np.array([self.x_right, self.y])


def frame_links(frame: dict, link_factory: Callable = Link) -> list:
    """Links of one frame in inlier order: Link(x_l, x_r, (y_l + y_r) / 2), the
    expression of TrackingDB.create_links (database.py:276) on float pixels."""
    pl, pr = frame["pix_left"], frame["pix_right"]
    return [link_factory(float(a[0]), float(b[0]), (float(a[1]) + float(b[1])) / 2)
            for a, b in zip(pl, pr)]


def dense_matches(pair: dict, n_prev: int):
    """The reference's densification (fill_database.py:209-223), used when the frame
    matcher cross-checks: one entry per previous inlier row, in row order; rows without
    a match hold DMatch(i, -1, 0, inf) and are no inliers.  Returns (MatchList, mask)."""
    q = np.arange(n_prev, dtype=np.int32)
    t = np.full(n_prev, -1, dtype=np.int32)
    d = np.full(n_prev, np.inf, dtype=np.float32)
    ok = np.zeros(n_prev, dtype=bool)
    rows = np.asarray(pair["query_idx"], dtype=np.int64)
    t[rows] = pair["train_idx"]
    d[rows] = np.asarray(pair["dist"], dtype=np.float32)
    ok[rows] = np.asarray(pair["mask"], dtype=bool)
    return MatchList(q, t, d), ok


def replay_into_db(db, result, link_factory: Callable = Link, n_frames: Optional[int] = None,
                   frame_cross_check: Optional[bool] = None):
    """Feed a TrackingDB-like object (anything with the reference's
    `add_frame(links, left_features, matches_to_previous_left=None, inliers=None)`)
    with the frames of `result` (FrontendResult or SequenceResult), in order.
    Pass the reference's own Link class as `link_factory` when the DB will be
    pickled for the reference's loader.  `frame_cross_check` (default: what `result`
    records, else False) selects the reference's two hand-over shapes: without it the
    compact matches in sorted order, one per previous row; with it the densified list
    (`dense_matches`), as add_frame requires one entry per previous row (also used when a frame
    without stereo inliers leaves the previous rows without matches)."""
    if n_frames is None:
        n_frames = result.n_frames if hasattr(result, "n_frames") else result.inputs.n_frames
    if frame_cross_check is None:
        frame_cross_check = bool(getattr(result, "frame_cross_check", False))
    n_prev = 0
    for f in range(n_frames):
        fr = result.frame(f)
        links = frame_links(fr, link_factory)
        des = np.ascontiguousarray(fr["inlier_des_left"])
        if f == 0:
            db.add_frame(links, des, None)
        else:
            pr = result.pair(f - 1)
            if frame_cross_check or len(pr["query_idx"]) != n_prev:
                # (without cross-check every previous row has a match unless this frame kept no
                # stereo inlier at all; add_frame wants one entry per previous row either way, :332)
                matches, mask = dense_matches(pr, n_prev)
            else:
                matches = MatchList(pr["query_idx"], pr["train_idx"], pr["dist"].astype(np.float32))
                mask = pr["mask"]
            db.add_frame(links, des, matches, mask)
        n_prev = len(des)
    return db


def save_relative_Rts(result, path: str = "relative_Rts") -> np.ndarray:
    """Pickle the (F-1,3,4) relative poses the way fill_database.py:252-255 does."""
    rel = result.relative_Rts() if callable(getattr(result, "relative_Rts")) else result.relative_Rts
    rel = np.asarray(rel, dtype=np.float64)
    with open(path, "wb") as fh:
        pickle.dump(rel, fh)
    return rel


_FRAME_KEYS = ("pix_left", "pix_right", "xyz", "inlier_des_left")
_PAIR_KEYS = ("query_idx", "train_idx", "dist", "mask")


def save_records(result, path: str, n_frames: Optional[int] = None) -> str:
    """Everything `replay_into_db` reads, as one `.npz` of concatenated arrays
    plus per-frame / per-pair offsets, so that the replay into the reference's
    TrackingDB can run later, in another process, without a GPU."""
    if n_frames is None:
        n_frames = result.n_frames if hasattr(result, "n_frames") else result.inputs.n_frames
    frames = [result.frame(f) for f in range(n_frames)]
    pairs = [result.pair(p) for p in range(max(n_frames - 1, 0))]
    out = {"n_frames": np.int64(n_frames),
           "frame_off": np.concatenate([[0], np.cumsum([len(f["pix_left"]) for f in frames])]).astype(np.int64),
           "pair_off": np.concatenate([[0], np.cumsum([len(p["mask"]) for p in pairs])]).astype(np.int64)}
    for k in _FRAME_KEYS:
        out[k] = (np.concatenate([np.asarray(f[k]) for f in frames]) if frames
                  else np.zeros((0,), np.float32))
    for k in _PAIR_KEYS:
        out[k] = (np.concatenate([np.asarray(p[k]) for p in pairs]) if pairs
                  else np.zeros((0,), np.int32))
    out["Rt"] = (np.stack([np.asarray(p["Rt"], dtype=np.float64) for p in pairs]) if pairs
                 else np.zeros((0, 3, 4)))
    out["frame_cross_check"] = np.bool_(getattr(result, "frame_cross_check", False))
    np.savez(path, **out)
    return path


class SavedRecords:
    """Read-back of `save_records` with the accessors `replay_into_db` uses."""

    def __init__(self, path: str):
        with np.load(path, allow_pickle=False) as z:
            self._a = {k: z[k] for k in z.files}
        self.n_frames = int(self._a["n_frames"])
        self.frame_cross_check = bool(self._a.get("frame_cross_check", False))

    def frame(self, f: int) -> dict:
        lo, hi = self._a["frame_off"][f], self._a["frame_off"][f + 1]
        return {k: self._a[k][lo:hi] for k in _FRAME_KEYS}

    def pair(self, p: int) -> dict:
        lo, hi = self._a["pair_off"][p], self._a["pair_off"][p + 1]
        d = {k: self._a[k][lo:hi] for k in _PAIR_KEYS}
        d["mask"] = d["mask"].astype(bool)
        d["Rt"] = self._a["Rt"][p]
        return d

    def relative_Rts(self) -> np.ndarray:
        return self._a["Rt"]
