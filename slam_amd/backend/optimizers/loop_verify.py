"""Loop-closure candidate verification for many keyframe pairs at once: the
arithmetic of LoopClosure.consensus_pnp_match
(backend/optimizers/loop_closure.py:167-235 of the reference) executed by
`slamhip_verify_pairs`.  The pose-graph / GTSAM side of loop closure stays in
the reference; this class produces what it consumes per verified pair: success
flag, relative pose, inlier mask and the matches.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import numpy as np
import torch

from ... import _lib, kernels
from ..._lib import lib, check, VerifyArgs
from ..tracking.feature_cache import binary_descriptors


class VerifyResult(dict):
    """Result of PairVerifier.verify.  The per-pair arrays (ok, Rt, Rt_ransac,
    n_matches, n_inliers, iterations, best_index, best_count, n_iter_required, out_off)
    are host arrays.  The per-match
    arrays (query_idx, train_idx, dist, mask; and xyz per keyframe row) stay on
    the device until asked for: `res["mask"]` copies the whole array once,
    `res.pair_matches(p)` copies one pair, `res.verified_matches()` gathers the
    rows of all verified pairs on the device and copies only those.  Almost all
    candidate pairs of an all-pairs screen fail, so the consumer of the verified
    pairs (loop_closure.py:226-235) never needs the rest."""

    _LAZY = {"query_idx": ("tm_query", None), "train_idx": ("tm_train", None),
             "dist": ("tm_dist", None), "mask": ("tm_mask", bool), "xyz": ("xyz", None)}

    def __init__(self, host: dict, device: dict, n_out: int, n_rows: int):
        super().__init__(host)
        self._dev, self._n_out, self._n_rows = device, n_out, n_rows

    def __missing__(self, key):
        if key not in self._LAZY:
            raise KeyError(key)
        name, cast = self._LAZY[key]
        a = self._dev[name].cpu().numpy()
        a = a.reshape(-1, 3)[:self._n_rows] if key == "xyz" else a[:self._n_out]
        if cast is not None:
            a = a.astype(cast)
        self[key] = a
        return a

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._LAZY

    def pair_matches(self, p: int) -> dict:
        """Matches of pair p: rows out_off[p] .. out_off[p] + n_matches[p] (all of the pair's
        rows without cross-check)."""
        lo = int(self["out_off"][p])
        hi = lo + int(self["n_matches"][p])
        out = {}
        for key, (name, cast) in self._LAZY.items():
            if key == "xyz":
                continue
            a = self._dev[name][lo:hi].cpu().numpy()
            out[key] = a.astype(cast) if cast is not None else a
        return out

    def verified_matches(self) -> dict:
        """{"pairs": indices of verified pairs, "off": (len+1,) offsets, and the
        per-match arrays of just those pairs, concatenated in that order}."""
        idx = np.flatnonzero(self["ok"])
        start = self["out_off"][idx].astype(np.int64)
        count = self["n_matches"][idx].astype(np.int64)
        off = np.concatenate([[0], np.cumsum(count)]).astype(np.int64)
        out = {"pairs": idx, "off": off}
        dev = self._dev["tm_query"].device
        # row list built on the host (verified pairs are few), one index_select per array
        rows_h = (np.concatenate([np.arange(s0, s0 + c, dtype=np.int64) for s0, c in zip(start, count)])
                  if len(idx) else np.zeros(0, np.int64))
        rows = torch.from_numpy(rows_h).to(dev)
        for key, (name, cast) in self._LAZY.items():
            if key == "xyz":
                continue
            a = self._dev[name].index_select(0, rows).cpu().numpy()
            out[key] = a.astype(cast) if cast is not None else a
        return out


class PairVerifier:
    """Defaults follow the reference: PnP pixel threshold 1.5 px and
    min_inliers_loop (loop_closure.py:24-32), p_success 0.99999 and
    n_iter_max 1500 (loop_closure.py:215-216)."""

    def __init__(self, cam, pixel_thresh: float = 1.5, min_inliers_loop: int = 60,
                 p_success: float = 0.99999, n_iter_max: int = 1500, refine_iters: int = 30,
                 seed: int = 0, frame_cross_check: bool = False):
        self.cam = cam
        # crossCheck of the matcher consensus_pnp_match builds (loop_closure.py:192: LoopClosure's
        # frame_cross_check, run_slam.py:35; default False)
        self.frame_cross_check = bool(frame_cross_check)
        self.pixel_thresh, self.min_inliers = float(pixel_thresh), int(min_inliers_loop)
        self.p_success, self.n_iter_max = float(p_success), int(n_iter_max)
        self.refine_iters, self.seed = int(refine_iters), int(seed)
        self._kf = None
        self._ws = None

    def set_keyframes(self, keyframes: Sequence[Tuple[np.ndarray, np.ndarray]]):
        """keyframes[i] = (links (n,3) float64 rows [x_left, x_right, y], des_left (n,D) uint8):
        what LoopClosure.frame_data reads back from the TrackingDB."""
        self._kf = None
        self.add_keyframes(keyframes)
        return self

    def add_keyframes(self, keyframes: Sequence[Tuple[np.ndarray, np.ndarray]]) -> range:
        """Appends keyframes to the ones already set (a caller that meets its keyframes one at a
        time, as the loop of run_loop_closure does, loop_closure.py:321-346); returns the indices
        they got.  The device arrays grow by doubling, so adding keyframes one by one uploads
        every row once (and copies it device to device a few times)."""
        for links, des in keyframes:
            if len(links) != len(des):
                raise ValueError("links and descriptors differ in length")
        if not len(keyframes):
            n_have = 0 if self._kf is None else len(self._kf["n"])
            if self._kf is None:
                self._kf = self._empty_store(61)
            return range(n_have, n_have)
        links = np.concatenate([np.asarray(k[0], dtype=np.float64).reshape(-1, 3) for k in keyframes])
        des = [binary_descriptors(k[1], "keyframe descriptors") for k in keyframes]
        D = next((d.shape[1] for d in des if d.ndim == 2 and d.shape[1] > 0), 61)
        des = np.concatenate([d.reshape(-1, D) for d in des])
        if self._kf is None:
            self._kf = self._empty_store(D)
        kf = self._kf
        if kf["D"] != D:
            if kf["off"][-1] != 0:
                raise ValueError("descriptor width differs from the keyframes already set")
            self._kf = kf = self._empty_store(D)
        # LoopClosure.frame_data rebuilds the keypoints as cv2.KeyPoint(x, y, 1)
        # (loop_closure.py:152-164), which stores float32: the fp64 triangulation sees the
        # link coordinates rounded to float32, and Link.y = (yL + yR) / 2 is in general
        # not a float32 value.
        kp_l32 = np.ascontiguousarray(links[:, [0, 2]]).astype(np.float32)
        kp_r32 = np.ascontiguousarray(links[:, [1, 2]]).astype(np.float32)
        first, at = len(kf["n"]), int(kf["off"][-1])
        n_new = [len(k[0]) for k in keyframes]
        rows = at + len(links)
        if rows > kf["cap"]:
            cap = max(2 * kf["cap"], rows, 1024)
            dev = kernels._require_gpu()
            for name, width, dt in (("kp_l", 2, torch.float64), ("kp_r", 2, torch.float64), ("des", D, torch.uint8)):
                grown = torch.empty((cap, width), dtype=dt, device=dev)
                if at:
                    grown[:at].copy_(kf[name][:at])
                kf[name] = grown
            kf["cap"] = cap
        if len(links):
            kf["kp_l"][at:rows].copy_(kernels.to_device(kp_l32.astype(np.float64), torch.float64))
            kf["kp_r"][at:rows].copy_(kernels.to_device(kp_r32.astype(np.float64), torch.float64))
            kf["des"][at:rows].copy_(kernels.to_device(des, torch.uint8))
        kf["n"] = np.concatenate([kf["n"], np.asarray(n_new, dtype=np.int64)])
        kf["off"] = np.concatenate([[0], np.cumsum(kf["n"])]).astype(np.int64)
        lo = 0
        for n in n_new:                  # left pixels as cv2.KeyPoint keeps them, per keyframe
            self._kf_host.append(kp_l32[lo:lo + n])
            lo += n
        return range(first, first + len(keyframes))

    def _empty_store(self, D: int) -> dict:
        dev = kernels._require_gpu()
        self._kf_host = []
        return {"n": np.zeros(0, np.int64), "off": np.zeros(1, np.int64), "cap": 0, "D": int(D),
                "kp_l": torch.empty((0, 2), dtype=torch.float64, device=dev),
                "kp_r": torch.empty((0, 2), dtype=torch.float64, device=dev),
                "des": torch.empty((0, int(D)), dtype=torch.uint8, device=dev)}

    def verify(self, pairs, pair_ids=None, samples=None) -> dict:
        """pairs: (P,2) int array of (candidate, current) keyframe indices.
        Returns a VerifyResult: ok (P,), Rt (P,3,4) candidate->current pose,
        Rt_ransac, n_matches, n_inliers, iterations, out_off (P+1,) on the host,
        and the flat per-match arrays query_idx / train_idx / dist / mask (rows
        out_off[p]:out_off[p+1] belong to pair p), copied from the device on
        first use.
        pair_ids (optional, (P,) int): each pair's index in the whole candidate
        list when `pairs` is one shard of it (e.g. whole[rank::world] with
        pair_ids = arange(len(whole))[rank::world]): the sampler is keyed by
        (seed, pair id), so the shards together give the results of the uncut list.
        samples (optional, (P, n_iter_max, 4) int): the idx4 sequence of every pair's RANSAC
        loop (what rng.choice(N, 4, replace=False) returns at pnp.py:126), as indices into the
        pair's sorted matches, instead of the device sampler: a CPU run of the reference's loop
        on the same sequence is then comparable count for count."""
        if self._kf is None:
            raise RuntimeError("call set_keyframes first")
        kf = self._kf
        pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
        P = len(pairs)
        cand, curr = pairs[:, 0], pairs[:, 1]
        qcnt, tcnt = kf["n"][cand], kf["n"][curr]
        obase = np.concatenate([[0], np.cumsum(qcnt)])
        O, N = int(obase[-1]), int(kf["off"][-1])
        if O >= 2 ** 31 or N >= 2 ** 31:
            raise ValueError("too many rows for 32-bit offsets; split the pair list")
        dev = kf["des"].device
        i32 = lambda a: kernels.to_device(np.ascontiguousarray(a, dtype=np.int32), torch.int32)
        d_qbase, d_qcnt = i32(kf["off"][cand]), i32(qcnt)
        d_tbase, d_tcnt, d_obase = i32(kf["off"][curr]), i32(tcnt), i32(obase[:-1])
        tobase = np.concatenate([[0], np.cumsum(tcnt)])
        if int(tobase[-1]) >= 2 ** 31:
            raise ValueError("too many rows for 32-bit offsets; split the pair list")
        d_tobase = i32(tobase[:-1])
        out = {
            "xyz": torch.empty(max(N, 1) * 3, dtype=torch.float32, device=dev),
            "tm_query": torch.empty(max(O, 1), dtype=torch.int32, device=dev),
            "tm_train": torch.empty(max(O, 1), dtype=torch.int32, device=dev),
            "tm_dist": torch.empty(max(O, 1), dtype=torch.int32, device=dev),
            "tm_mask": torch.zeros(max(O, 1), dtype=torch.uint8, device=dev),
            "pose_ransac": torch.zeros(max(P, 1) * 12, dtype=torch.float64, device=dev),
            "pose_refined": torch.zeros(max(P, 1) * 12, dtype=torch.float64, device=dev),
            "info": torch.zeros(max(P, 1) * 8, dtype=torch.int32, device=dev),
        }
        a = VerifyArgs()
        a.n_rows_total, a.n_pairs, a.n_out_total = N, P, O
        a.desc_bytes = a.desc_stride = int(kf["des"].shape[1])
        a.max_q = int(qcnt.max()) if P else 0
        a.max_t = int(tcnt.max()) if P else 0
        a.d_kp_left, a.d_kp_right, a.d_des = kf["kp_l"].data_ptr(), kf["kp_r"].data_ptr(), kf["des"].data_ptr()
        a.d_qbase, a.d_qcnt = d_qbase.data_ptr(), d_qcnt.data_ptr()
        a.d_tbase, a.d_tcnt, a.d_obase = d_tbase.data_ptr(), d_tcnt.data_ptr(), d_obase.data_ptr()
        for name, src, n in (("K", self.cam.K, 9), ("M_right", self.cam.M_right, 12),
                             ("P_left", self.cam.P_left, 12), ("P_right", self.cam.P_right, 12)):
            setattr(a, name, (C.c_double * n)(*np.asarray(src, dtype=np.float64).reshape(-1).tolist()))
        a.ransac_thresh, a.p_success = self.pixel_thresh, self.p_success
        a.n_iter_max, a.min_points, a.refine_iters = self.n_iter_max, self.min_inliers, self.refine_iters
        a.seed = self.seed
        a.cross_check = int(self.frame_cross_check)
        a.d_tobase, a.n_tout_total = d_tobase.data_ptr(), int(tobase[-1])
        d_ids = None
        if pair_ids is not None:
            ids = np.ascontiguousarray(pair_ids, dtype=np.int64).reshape(-1)
            if len(ids) != P:
                raise ValueError("pair_ids must have one entry per pair")
            d_ids = kernels.to_device(ids, torch.int64)
            a.d_pair_ids = d_ids.data_ptr()
        d_samples = None
        if samples is not None:
            smp = np.ascontiguousarray(samples, dtype=np.int32)
            if smp.shape != (P, self.n_iter_max, 4):
                raise ValueError("samples must have shape (n_pairs, n_iter_max, 4)")
            d_samples = kernels.to_device(smp, torch.int32)
            a.d_samples = d_samples.data_ptr()
        a.d_xyz = out["xyz"].data_ptr()
        a.d_tm_query, a.d_tm_train = out["tm_query"].data_ptr(), out["tm_train"].data_ptr()
        a.d_tm_dist, a.d_tm_mask = out["tm_dist"].data_ptr(), out["tm_mask"].data_ptr()
        a.d_pose_ransac, a.d_pose_refined = out["pose_ransac"].data_ptr(), out["pose_refined"].data_ptr()
        a.d_info = out["info"].data_ptr()
        need = lib.slamhip_verify_workspace_bytes(C.byref(a))
        if need == 0:
            check(_lib.EUNSUPPORTED, "slamhip_verify_workspace_bytes")
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        a.d_workspace, a.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
        check(lib.slamhip_verify_pairs(C.byref(a), torch.cuda.current_stream().cuda_stream),
              "slamhip_verify_pairs")
        self.last_device_outputs = out
        torch.cuda.synchronize()
        info = out["info"].cpu().numpy().reshape(-1, 8)[:P]
        n_matches, n_inl = info[:, 5], info[:, 6]
        ok = (n_matches >= self.min_inliers) & (n_inl >= self.min_inliers) & (info[:, 0] >= 0)
        host = {
            "ok": ok, "Rt": out["pose_refined"].cpu().numpy().reshape(-1, 3, 4)[:P],
            "Rt_ransac": out["pose_ransac"].cpu().numpy().reshape(-1, 3, 4)[:P],
            "n_matches": n_matches, "n_inliers": n_inl, "iterations": info[:, 2],
            "best_index": info[:, 0], "best_count": info[:, 1], "n_iter_required": info[:, 3],
            "out_off": obase,
        }
        return VerifyResult(host, out, O, N)

    def consensus_pnp_match(self, kf_curr: int, kf_cand: int, pose_factory=None, pair_id=None):
        """One pair, shaped like the reference's method (loop_closure.py:167-235):
        (True, Delta_inv, mask, (kf_curr, kf_cand), px) or (False, None, None, (None, None), {}).
        kf_curr / kf_cand index the list given to set_keyframes.  Delta_inv is
        `pose_factory(R, t).inverse()` when a factory is given (the reference builds
        gtsam.Pose3(gtsam.Rot3(R), gtsam.Point3(t)), loop_closure.py:226-228), else the
        inverse pose as a 4 x 4 array.  px holds the left pixels of inlier and outlier
        matches in both keyframes (float32), as loop_closure.py:230-233.  `pair_id`: the
        sampler key of the pair (default 0).  A caller with many candidates per keyframe
        should hand them to consensus_pnp_matches, or all pairs to verify(), at once."""
        return self.consensus_pnp_matches(kf_curr, [kf_cand], pose_factory,
                                          None if pair_id is None else [pair_id])[0]

    def consensus_pnp_matches(self, kf_curr: int, kf_cands, pose_factory=None, pair_ids=None) -> list:
        """The candidates of one keyframe (the `for i_kf in cand_kf_idxs` loop of
        loop_closure.py:341-346, whose iterations do not depend on one another) in one call: a
        list with consensus_pnp_match's tuple per candidate."""
        kf_cands = [int(c) for c in kf_cands]
        if not kf_cands:
            return []
        pairs = np.array([[c, kf_curr] for c in kf_cands], dtype=np.int64)
        res = self.verify(pairs, pair_ids=pair_ids)
        out = []
        for p, kf_cand in enumerate(kf_cands):
            if not bool(res["ok"][p]):
                out.append((False, None, None, (None, None), {}))
                continue
            m = res.pair_matches(p)
            mask = m["mask"]
            pts_curr = self._kf_host[kf_curr][m["train_idx"].astype(np.int64)]
            pts_cand = self._kf_host[kf_cand][m["query_idx"].astype(np.int64)]
            Rt = res["Rt"][p]
            R, t = Rt[:, :3], Rt[:, 3]
            if pose_factory is not None:
                delta_inv = pose_factory(R, t).inverse()
            else:
                delta_inv = np.eye(4)
                delta_inv[:3, :3] = R.T
                delta_inv[:3, 3] = -R.T @ t
            px = dict(inl_curr=pts_curr[mask], out_curr=pts_curr[~mask],
                      inl_cand=pts_cand[mask], out_cand=pts_cand[~mask])
            out.append((True, delta_inv, mask, (kf_curr, kf_cand), px))
        return out
