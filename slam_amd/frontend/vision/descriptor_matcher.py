"""Descriptor matching behind the reference's `DescriptorMatcher` interface
(frontend/vision/descriptor_matcher.py of the reference), executed by the
Hamming top-2 HIP kernel for uint8 (binary) descriptors and by the L2 top-2 kernel
for any other dtype - the reference's own norm selection (`_get_bf`, :23-28).

`match` returns what the reference's call sites consume: a sequence of
DMatch-like objects sorted by distance (stable, so equal distances stay in
ascending queryIdx order).  `match_arrays` returns the same result as device
tensors without building any Python object per match.
"""
from __future__ import annotations

import numpy as np
import torch

from ... import kernels
from ...types import DMatch, MatchList


class KnnMatches(list):
    """knn_match result: the list cv2.BFMatcher.knnMatch returns, plus the norm the
    distances were computed under ("hamming" or "l2"), which ratio_test reads."""
    norm = None


class DescriptorMatcher:
    def __init__(self, cross_check, matcher_type="bf", superglue_cfg=None):
        self.matcher_type = matcher_type.lower()
        self.cross_check = cross_check
        if self.matcher_type == "bf":
            pass
        elif self.matcher_type == "superglue":
            # The reference imports the external SuperPoint/SuperGlue package here;
            # that neural matcher is not part of this library.
            raise ImportError("matcher_type='superglue' needs the external "
                              "superpoint_superglue_deployment package; only 'bf' is provided")
        else:
            raise ValueError(f"Unsupported matcher_type: {matcher_type}")

    # ------------------------------------------------------------------ arrays
    @staticmethod
    def _is_binary(des) -> bool:
        """The reference's norm selection (descriptor_matcher.py:23-28): Hamming for
        uint8 descriptors, L2 for anything else."""
        return des.dtype in (np.uint8, torch.uint8)

    def match_arrays(self, desL, desR):
        """(queryIdx, trainIdx, distance) device tensors in sorted order; distance is
        int32 for binary descriptors, float32 for float descriptors."""
        if not self._is_binary(desL):
            if len(desL) == 0 or len(desR) == 0:
                z = torch.zeros(0, dtype=torch.int32, device="cuda")
                return z, z.clone(), torch.zeros(0, dtype=torch.float32, device="cuda")
            if self.cross_check:
                ti, di = kernels.l2_cross_check_match(desL, desR)
            else:
                idx2, dist2 = kernels.l2_knn2(desL, desR)
                ti, di = idx2[:, 0].contiguous(), dist2[:, 0].contiguous()
            order = kernels.sort_by_distance_f32(di)      # unmatched rows (-1) sort last
            n_valid = int((ti >= 0).sum().item())
            order = order[:n_valid].long()
            return order.int(), ti[order], di[order]
        if self.cross_check:
            ti, di = kernels.cross_check_match(desL, desR)
        else:
            idx2, dist2 = kernels.hamming_knn2(desL, desR)
            ti, di = idx2[:, 0].contiguous(), dist2[:, 0].contiguous()
        order = kernels.sort_by_distance(di)          # unmatched rows (-1) sort last
        n_valid = int((ti >= 0).sum().item())
        order = order[:n_valid].long()
        return order.int(), ti[order], di[order]

    def knn_arrays(self, desL, desR):
        """(idx2, dist2): (nq,2) device tensors, -1 where absent."""
        if not self._is_binary(desL):
            return kernels.l2_knn2(desL, desR)
        return kernels.hamming_knn2(desL, desR)

    # ------------------------------------------------- reference-shaped methods
    def match(self, desL, desR, kpsL=None, kpsR=None, shapeL=None, shapeR=None):
        qi, ti, di = self.match_arrays(desL, desR)
        return MatchList(qi.cpu().numpy(), ti.cpu().numpy(), di.cpu().numpy().astype(np.float32))

    def knn_match(self, desL, desR, k=2):
        """cv2.BFMatcher.knnMatch shape: one tuple of up to k DMatch per query."""
        if k not in (1, 2):
            raise ValueError("k must be 1 or 2")
        if self.cross_check and k != 1:
            raise ValueError("cross_check requires k == 1")  # cv2 asserts the same
        idx2, dist2 = self.knn_arrays(desL, desR)
        idx2, dist2 = idx2.cpu().numpy(), dist2.cpu().numpy()
        out = KnnMatches()
        out.norm = "hamming" if self._is_binary(desL) else "l2"
        for i in range(idx2.shape[0]):
            out.append(tuple(DMatch(i, idx2[i, j], 0, dist2[i, j])
                             for j in range(k) if idx2[i, j] >= 0))
        return out

    @staticmethod
    def ratio_test(raw, ratio=0.75):
        """Lowe ratio test on knn_match output -> (good, rejected) lists of the
        best match per query (the shape the commented call at
        backend/optimizers/loop_closure.py:193-194 expects)."""
        good, rejected = [], []
        if len(raw) == 0:
            return good, rejected
        d = np.full((len(raw), 2), -1.0, dtype=np.float64)
        for i, pair in enumerate(raw):
            for j, m in enumerate(pair[:2]):
                d[i, j] = float(m.distance)
        if getattr(raw, "norm", None) == "hamming":     # integer distances: the int32 kernel
            keep = kernels.ratio_test(d.astype(np.int32), ratio).cpu().numpy()
        else:
            # float32 distances (L2), or a plain list whose norm is not recorded: the
            # float kernel widens to fp64 before comparing, so Hamming distances (exact
            # in float32) get the same decisions from it as from the int32 kernel
            keep = kernels.ratio_test_f32(d.astype(np.float32), ratio).cpu().numpy()
        for i, pair in enumerate(raw):
            if len(pair) == 0:
                continue
            (good if keep[i] else rejected).append(pair[0])
        return good, rejected

    @staticmethod
    def extract_matched_pixels(kpsL, kpsR, matches):
        """Two (N,2) float32 arrays of matched pixel coordinates."""
        from ...types import keypoints_to_array, matches_to_arrays
        q, t = matches_to_arrays(matches)
        return (keypoints_to_array(kpsL)[q].astype(np.float32).reshape(-1, 2),
                keypoints_to_array(kpsR)[t].astype(np.float32).reshape(-1, 2))

    @staticmethod
    def extract_matched_keypoints(matches, kpsL, kpsR):
        return ([kpsL[m.queryIdx] for m in matches], [kpsR[m.trainIdx] for m in matches])
