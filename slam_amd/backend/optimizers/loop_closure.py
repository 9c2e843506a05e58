"""The geometric check of loop closure behind the reference's own method names and
signatures (LoopClosure.frame_data / compute_triangulation / consensus_pnp_match,
backend/optimizers/loop_closure.py:152-249), for a caller that keeps the reference's
`run_loop_closure` and only wants its per-candidate check to run on the GPU:

    from slam.backend.optimizers.loop_closure import LoopClosure as ReferenceLoopClosure
    from slam_amd.backend.optimizers.loop_closure import GpuConsensusCheck

    class LoopClosure(GpuConsensusCheck, ReferenceLoopClosure):
        pass

(`slam_amd.backend.optimizers.loop_closure.LoopClosure` is that class when the reference's
module can be imported, i.e. where gtsam, networkx and OpenCV are installed.)  Everything
else of LoopClosure -- the Mahalanobis gate, the pose graph, the optimiser
(loop_closure.py:51-149, 252-438) -- is GTSAM / networkx code and stays the reference's.

`consensus_pnp_match(kf_curr, kf_cand, db)` reads a keyframe's links and descriptors from the
TrackingDB the first time the keyframe is seen (frame_data, :152-164) and keeps them on the
device; a keyframe is uploaded once however many candidates it is checked against.
`consensus_pnp_matches(kf_curr, kf_cands, db)` checks all candidates of one keyframe (the
loop of :341-346, up to k_best of them) in one call.
"""
from __future__ import annotations

import numpy as np

from ...frontend.io.camera_model import CameraModel
from ..tracking.feature_cache import binary_descriptors
from .loop_verify import PairVerifier


def _pose3_factory():
    """gtsam.Pose3(gtsam.Rot3(R), gtsam.Point3(t)) as loop_closure.py:226-227 builds it, or None
    where gtsam is not installed (Delta_inv is then the 4 x 4 inverse pose as an array)."""
    try:
        import gtsam
    except ImportError:
        return None
    return lambda R, t: gtsam.Pose3(gtsam.Rot3(np.asarray(R)), gtsam.Point3(np.asarray(t)))


class GpuConsensusCheck:
    """Mix-in over the reference's LoopClosure (no constructor of its own: the reference's runs).
    It reads what LoopClosure.__init__ has set (loop_closure.py:32-40): `self.pnp` (the camera
    matrices and the 1.5 px threshold), `self.frame_cross_check`, `self.min_inliers_loop`.
    `ConsensusCheck` below is the same without the reference's class."""

    # ------------------------------------------------------------------ set-up
    def _camera(self):
        cam = getattr(self, "_cam", None)
        if cam is not None:
            return cam
        pnp = self.pnp                                                     # loop_closure.py:32
        return CameraModel(pnp.K, pnp.M_left, pnp.M_right)

    def _verifier_for(self, db) -> PairVerifier:
        v = getattr(self, "_verifier", None)
        pnp = getattr(self, "pnp", None)
        thresh = float(getattr(pnp, "pixel_thresh", getattr(self, "_pixel_thresh", 1.5)))   # PnP(cam, 1.5), :32
        key = (id(db), bool(self.frame_cross_check), int(self.min_inliers_loop), thresh)
        if v is None or self._verifier_key != key:
            v = PairVerifier(self._camera(), pixel_thresh=thresh, min_inliers_loop=int(self.min_inliers_loop),
                             frame_cross_check=bool(self.frame_cross_check), seed=getattr(self, "_seed", 0))
            self._verifier, self._verifier_key, self._kf_index = v, key, {}
        return v

    def forget_keyframes(self):
        """Drops the keyframes kept on the device (after the TrackingDB has changed)."""
        self._verifier = None

    def _index_of(self, db, kf_id: int) -> int:
        v = self._verifier_for(db)
        at = self._kf_index.get(kf_id)
        if at is None:
            des = db.features(kf_id)                                       # :157
            links = db.all_frame_links(kf_id)                              # :158
            rows = np.array([[ln.x_left, ln.x_right, ln.y] for ln in links], dtype=np.float64).reshape(-1, 3)
            des = np.zeros((0, 61), np.uint8) if des is None else binary_descriptors(des, "keyframe descriptors")
            at = self._kf_index[kf_id] = v.add_keyframes([(rows, des)])[0]
        return at

    # -------------------------------------------- the reference's three methods
    def frame_data(self, db, kf_id: int):
        """(kpL, kpR, descL) of a keyframe as loop_closure.py:152-164 rebuilds them from the
        links: left pixels (x_left, y) and right pixels (x_right, y) as (n, 2) float32 arrays
        (cv2.KeyPoint stores float32), and the left descriptors."""
        descL = db.features(kf_id)
        links = db.all_frame_links(kf_id)
        rows = np.array([[ln.x_left, ln.x_right, ln.y] for ln in links], dtype=np.float64).reshape(-1, 3)
        return (np.ascontiguousarray(rows[:, [0, 2]]).astype(np.float32),
                np.ascontiguousarray(rows[:, [1, 2]]).astype(np.float32), descL)

    def compute_triangulation(self, P_L, P_R, kpL, kpR):
        """(N, 3) float64 points in the left camera frame: cv2.triangulatePoints on float64
        pixels and the division by the fourth coordinate (loop_closure.py:237-249)."""
        from ... import kernels
        from ...types import keypoints_to_array
        kl = keypoints_to_array(kpL).astype(np.float64)
        kr = keypoints_to_array(kpR).astype(np.float64)
        if len(kl) != len(kr):
            raise ValueError("left and right keypoints differ in length")
        if len(kl) == 0:
            return np.zeros((0, 3))
        _, xyz = kernels.stereo_filter_triangulate(kl, kr, None, None, P_L, P_R, do_filter=False, fp64=True)
        return xyz.cpu().numpy()

    def consensus_pnp_match(self, kf_curr: int, kf_cand: int, db):
        """The reference's signature and return tuple (loop_closure.py:167-235):
        (True, Delta_inv, mask, (kf_curr, kf_cand), px) with Delta_inv a gtsam.Pose3 where gtsam
        is installed, or (False, None, None, (None, None), {})."""
        return self.consensus_pnp_matches(kf_curr, [kf_cand], db)[0]

    def consensus_pnp_matches(self, kf_curr: int, kf_cands, db) -> list:
        """consensus_pnp_match for every candidate of one keyframe, in one call to the GPU."""
        kf_cands = [int(c) for c in kf_cands]
        n = self._index_of(db, int(kf_curr))
        cands = [self._index_of(db, c) for c in kf_cands]
        # sampler key of a pair: its two frame ids, so a pair draws the same samples whenever
        # and in whatever company it is checked
        ids = [(int(kf_curr) << 24) ^ c for c in kf_cands]
        out = self._verifier_for(db).consensus_pnp_matches(n, cands, _pose3_factory(), pair_ids=ids)
        # the pair is reported by frame ids, as the reference does (:235)
        return [(ok, d, m, (kf_curr, c) if ok else pr, px) for (ok, d, m, pr, px), c in zip(out, kf_cands)]


class ConsensusCheck(GpuConsensusCheck):
    """The check on its own, for a caller without the reference's LoopClosure (no gtsam):
    defaults as loop_closure.py:23-32."""

    def __init__(self, cam, frame_cross_check=False, min_inliers_loop=60, pixel_thresh=1.5, seed=0):
        self._cam = cam
        self.frame_cross_check = frame_cross_check
        self.min_inliers_loop = min_inliers_loop
        self._pixel_thresh, self._seed = float(pixel_thresh), int(seed)


def __getattr__(name):
    # `LoopClosure`: the reference's class with the check above in front of its own, built on
    # first use, where the reference's module (gtsam, networkx, OpenCV) can be imported
    if name == "LoopClosure":
        try:
            from slam.backend.optimizers.loop_closure import LoopClosure as ReferenceLoopClosure
        except ImportError as e:
            raise ImportError("the reference's LoopClosure (slam.backend.optimizers.loop_closure: gtsam, "
                              "networkx, OpenCV) cannot be imported; ConsensusCheck stands alone") from e
        cls = type("LoopClosure", (GpuConsensusCheck, ReferenceLoopClosure), {"__module__": __name__})
        globals()["LoopClosure"] = cls
        return cls
    raise AttributeError(name)
