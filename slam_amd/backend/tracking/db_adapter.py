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


def replay_into_db(db, result, link_factory: Callable = Link, n_frames: Optional[int] = None):
    """Feed a TrackingDB-like object (anything with the reference's
    `add_frame(links, left_features, matches_to_previous_left=None, inliers=None)`)
    with the frames of `result` (FrontendResult or SequenceResult), in order.
    Pass the reference's own Link class as `link_factory` when the DB will be
    pickled for the reference's loader."""
    if n_frames is None:
        n_frames = result.n_frames if hasattr(result, "n_frames") else result.inputs.n_frames
    for f in range(n_frames):
        fr = result.frame(f)
        links = frame_links(fr, link_factory)
        des = np.ascontiguousarray(fr["inlier_des_left"])
        if f == 0:
            db.add_frame(links, des, None)
            continue
        pr = result.pair(f - 1)
        matches = MatchList(pr["query_idx"], pr["train_idx"], pr["dist"].astype(np.float32))
        db.add_frame(links, des, matches, pr["mask"])
    return db


def save_relative_Rts(result, path: str = "relative_Rts") -> np.ndarray:
    """Pickle the (F-1,3,4) relative poses the way fill_database.py:252-255 does."""
    rel = result.relative_Rts() if callable(getattr(result, "relative_Rts")) else result.relative_Rts
    rel = np.asarray(rel, dtype=np.float64)
    with open(path, "wb") as fh:
        pickle.dump(rel, fh)
    return rel
