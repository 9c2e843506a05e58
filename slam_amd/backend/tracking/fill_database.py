"""`TrackingPipeline` with the reference's constructor and `build()`
(backend/tracking/fill_database.py:19-259), for a caller that switches the import of the
pipeline itself rather than of the three classes it is made of:

    from slam_amd.backend.tracking.fill_database import TrackingPipeline

`build()` walks the image sequence as the reference does -- read a stereo pair, extract
features in both images (fill_database.py:141-145) -- and hands every frame's features to
the GPU front-end: one frame at a time (`OnlineFrontend.push`, the reference's own loop
shape with the previous frame carried over, :234-240) or, with `batched=True`, all frames in
one call once they are extracted.  Both give the same results.  Those two paths are the
Hamming ones; float descriptors (the reference's `--feature-type superpoint` with the "bf"
matcher, which BFMatcher pairs under the L2 norm, descriptor_matcher.py:23-28) take the
reference's own loop instead, one call per stage and frame through the three classes
(`_build_stagewise`): every stage still runs on the device, without the fused launches.
The per-frame records then
go through the reference's unchanged `TrackingDB.add_frame` in frame order (:133, :231), the
DB is serialised and the relative poses are pickled to the file `relative_Rts` (:249-255).

What is NOT in this module: feature extraction and image reading.  They run on the CPU
through OpenCV in the reference (frontend/vision/feature_extractor.py,
frontend/io/image_sequence.py) and stay there; the constructor takes the reference's own
`ImageSequence` and builds the reference's own `FeatureExtractor` when the `slam` package
is importable, or takes any objects with the same two methods (`img_seq[i] -> (left, right)`,
`detect_and_compute(img) -> (keypoints, descriptors)`).
"""
from __future__ import annotations

import itertools
from typing import Callable, Optional

import numpy as np

from . import db_adapter
from .frontend import TrackingFrontend
from .online import OnlineFrontend, OnlineRecords
from ...frontend.geometry.pnp import PnP, SOLVEPNP_ITERATIVE
from ...frontend.geometry.triangulation import Triangulation
from ...frontend.vision.descriptor_matcher import DescriptorMatcher
from ...types import MatchList


def _pixels(kps) -> np.ndarray:
    """(n, 2) float32 from a keypoint list (objects with `.pt`, as cv2.KeyPoint) or an array."""
    if isinstance(kps, np.ndarray):
        return np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 2)
    return np.array([k.pt for k in kps], dtype=np.float32).reshape(-1, 2)


class TrackingPipeline:
    """Arguments as the reference's (fill_database.py:27-38).  `feature_type` / `akaze_thresh`
    / `use_gpu` configure the reference's FeatureExtractor, which is only built when no
    `feature_extractor` is given; `matcher_type` must be "bf" (SuperGlue is an external neural
    package, as for DescriptorMatcher); `plot` keeps the per-frame match / inlier counts
    (`matches_per_frame`, `inliers_per_frame`) that the reference plots, without plotting.
    Extra keyword arguments: `feature_extractor`, `db_factory` / `link_factory` (default: the
    reference's TrackingDB and Link), `batched`, `seed` (of the device sampler; the
    reference's own sampler is unseeded, pnp.py:120-122)."""

    def __init__(self, cam, img_seq, akaze_thresh, disparity_min, pixel_thresh, stereo_cross_check,
                 frame_cross_check, feature_type: str, matcher_type: str, plot: bool, use_gpu=True, *,
                 feature_extractor=None, db_factory: Optional[Callable] = None,
                 link_factory: Optional[Callable] = None, batched: bool = False, seed: int = 0):
        if matcher_type not in ("bf", "superglue"):
            raise ValueError("matcher_type must be 'bf' or 'superglue'")       # descriptor_matcher.py:21
        if matcher_type == "superglue":
            raise ImportError("the SuperGlue matcher needs the external superpoint_superglue_deployment "
                              "package; this pipeline runs the brute-force matcher")
        self.cam, self.img_seq, self.plot = cam, img_seq, plot
        self.matcher_type = matcher_type
        self.relative_Rts = None            # cached in build(), fill_database.py:42
        self.batched = bool(batched)
        self._params = dict(disparity_min=disparity_min, pixel_thresh=pixel_thresh, seed=seed,
                            stereo_cross_check=bool(stereo_cross_check), frame_cross_check=bool(frame_cross_check))
        if feature_extractor is None:
            try:
                from slam.frontend.vision.feature_extractor import FeatureExtractor   # needs OpenCV
            except ImportError as e:
                raise ImportError("no feature_extractor given and the reference's FeatureExtractor "
                                  "(slam.frontend.vision.feature_extractor, OpenCV) cannot be imported") from e
            feature_extractor = FeatureExtractor(
                threshold=akaze_thresh, detector_type=feature_type,
                superpoint_cfg={"keypoint_threshold": 0.005, "max_keypoints": 2048, "use_gpu": use_gpu})
        self.feature_extractor = feature_extractor
        self._db_factory, self._link_factory = db_factory, link_factory
        self.matches_per_frame, self.inliers_per_frame = [], []

    def _frame(self, idx):
        img_l, img_r = self.img_seq[idx]                                        # :102, :141
        kp_l, des_l = self.feature_extractor.detect_and_compute(img_l)          # :104-105, :144-145
        kp_r, des_r = self.feature_extractor.detect_and_compute(img_r)
        # (an image without features: OpenCV hands back None for its descriptors)
        width = next((d.shape[1] for d in (des_l, des_r) if d is not None and np.ndim(d) == 2), 61)
        des_l, des_r = (np.zeros((0, width), np.uint8) if d is None else np.ascontiguousarray(d) for d in (des_l, des_r))
        return _pixels(kp_l), _pixels(kp_r), des_l, des_r

    def _build_stagewise(self, frames) -> OnlineRecords:
        """The reference's loop (:101-240) over the three classes, for descriptors the fused
        front-end does not take (anything but uint8): stereo match -> row / disparity filter
        -> triangulate per frame, then match the previous frame's stereo inliers to this
        frame's, RANSAC over the correspondences and refine on the inliers.  `frames`
        yields (kp_left, kp_right, des_left, des_right); the records have the layout of the
        fused paths'.  The sampler is a NumPy generator seeded with `seed`."""
        P = self._params
        stereo = DescriptorMatcher(cross_check=P["stereo_cross_check"])           # :55-57
        temporal = DescriptorMatcher(cross_check=P["frame_cross_check"])          # :58-60
        triang = Triangulation(self.cam, P["disparity_min"], P["pixel_thresh"])   # :62-63
        pnp = PnP(self.cam, P["pixel_thresh"])                                    # :64
        rng = np.random.default_rng(P["seed"])
        out = OnlineRecords(P["frame_cross_check"])
        prev = None
        for f, (kp_l, kp_r, des_l, des_r) in enumerate(frames):
            lr = stereo.match(des_l, des_r)                                       # :114, :155
            lr = triang.create_inlier_outlier_matches(lr, kp_l, kp_r)[0]          # :116-118, :158-160
            if len(lr) == 0:
                lr = MatchList([], [], [])
            q, t = lr.query_idx, lr.train_idx
            cur = {"left_idx": q, "right_idx": t, "dist": lr.distance,
                   "xyz": triang.triangulate_opencv(kp_l, kp_r, lr).reshape(-1, 3),   # :121, :163
                   "pix_left": kp_l[q], "pix_right": kp_r[t],
                   "inlier_des_left": np.ascontiguousarray(des_l[q])}             # create_links, :124-131
            out._frames.append(cur)
            if prev is not None:
                m01 = temporal.match(prev["inlier_des_left"], cur["inlier_des_left"])   # :185
                X = prev["xyz"][m01.query_idx]                                    # :189-190
                pix_l, pix_r = cur["pix_left"][m01.train_idx], cur["pix_right"][m01.train_idx]   # :195-197
                mask = pnp.ransac_pnp_pair1(X, pix_l, pix_r, rng=rng) if len(X) >= 4 else None   # :200-202
                if mask is None:
                    # fewer correspondences than a sample, or no hypothesis with an inlier: the reference
                    # fails here (it indexes with the missing mask); as in the fused paths the pair is
                    # recorded without inliers and with the identity as its pose
                    mask, Rt = np.zeros(len(X), dtype=bool), np.hstack([np.eye(3), np.zeros((3, 1))])
                else:
                    Rt = pnp.solve_pnp(X[mask], pix_l[mask], SOLVEPNP_ITERATIVE)  # :204-207
                out._pairs.append({"query_idx": m01.query_idx, "train_idx": m01.train_idx,
                                   "dist": m01.distance, "mask": mask, "Rt": np.asarray(Rt, dtype=np.float64)})
            prev = cur
        return out

    def build(self, max_frame=2_600, out_path="kitti05_tracks", rel_path="relative_Rts"):
        """As the reference's build() (:85-255): a TrackingDB over frames [0, max_frame), serialised
        to `out_path`, the relative poses cached and pickled to `rel_path`.  Returns the DB."""
        total = min(max_frame, len(self.img_seq))
        if total < 2:
            raise ValueError("Need at least 2 frames to build tracking database.")   # :98-99
        db_factory, link_factory = self._db_factory, self._link_factory
        if db_factory is None or link_factory is None:
            try:
                from slam.backend.tracking.database import TrackingDB, Link       # the reference's own DB
            except ImportError as e:
                raise ImportError("no db_factory / link_factory given and the reference's TrackingDB "
                                  "(slam.backend.tracking.database) cannot be imported") from e
            db_factory, link_factory = db_factory or TrackingDB, link_factory or Link
        first = self._frame(0)
        rest = (self._frame(idx) for idx in range(1, total))                    # :140
        if first[2].dtype != np.uint8:
            result = self._build_stagewise(itertools.chain([first], rest))
        elif self.batched:
            result = TrackingFrontend(self.cam, **self._params).run([first, *rest])
        else:
            online = OnlineFrontend(self.cam, **self._params)
            result = OnlineRecords(self._params["frame_cross_check"])
            for frame in itertools.chain([first], rest):
                pair = online.push(*frame)
                if pair is not None:
                    result.add(pair)
        db = db_adapter.replay_into_db(db_factory(), result, link_factory=link_factory, n_frames=total)
        if self.plot:                                                           # :226-228
            pairs = [result.pair(p) for p in range(total - 1)]
            self.matches_per_frame = [int(len(p["mask"])) for p in pairs]
            self.inliers_per_frame = [int(np.count_nonzero(p["mask"])) for p in pairs]
        if out_path is not None and hasattr(db, "serialize"):
            db.serialize(out_path)                                              # :249
        self.relative_Rts = np.asarray(result.relative_Rts(), dtype=np.float64)[:total - 1]
        if rel_path is not None:
            db_adapter.save_relative_Rts(self, rel_path)                        # :252-255
        return db

    def get_relative_Rts(self):                                                 # :258-259
        return self.relative_Rts
