"""Batched front-end: the per-frame data flow of the reference's
TrackingPipeline.build() (backend/tracking/fill_database.py:85-255) with
features given, executed by `slamhip_frontend_batch` for a whole run of frames
at once.

A frame is (kp_left (n,2) float32, kp_right (m,2) float32, des_left (n,D) uint8,
des_right (m,D) uint8).  `run` returns device tensors; `FrontendResult.frame(f)`
/ `.pair(p)` give NumPy views shaped like what the reference builds per frame:
stereo inlier matches in sorted order, triangulated points, links, the inlier
descriptor subset, temporal matches (sorted), the RANSAC inlier mask and the
relative pose [R|t] (the rows of `relative_Rts`).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from ... import _lib
from ..._lib import lib, check, FrontendArgs
from ... import kernels
from .feature_cache import PackedFrames, binary_descriptors

Frame = Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]


# ---------------------------------------------------------------- host staging
# Page-locked staging buffers of `upload`, kept for the life of the process and
# grown on demand: allocating ~100 MB of page-locked memory costs several times the
# whole pass.  One event marks the last asynchronous copy out of them; the next
# upload waits for it before it overwrites them.  Not thread-safe: one uploader at a
# time.  (Filling them from several threads was measured slower on the GPU box's
# share of host cores: 5.6 ms with one thread, 9-12 ms with four or eight, for the
# 104 MB of 500 frames.)
_stage_bufs: dict = {}
_stage_event = None


def _staging(slot: int, dtype: torch.dtype, numel: int) -> torch.Tensor:
    buf = _stage_bufs.get(slot)
    if buf is None or buf.dtype != dtype or buf.numel() < numel:
        buf = torch.empty(max(int(numel * 1.25), 1), dtype=dtype, pin_memory=True)
        _stage_bufs[slot] = buf
    return buf[:numel]


def _staging_wait():
    if _stage_event is not None:
        _stage_event.synchronize()


def _staging_mark():
    global _stage_event
    if _stage_event is None:
        _stage_event = torch.cuda.Event()
    _stage_event.record(torch.cuda.current_stream())


@dataclass
class FrontendInputs:
    """Concatenated, device-resident inputs of a run of frames."""
    n_frames: int
    off_left: np.ndarray      # (F+1,) int32 host copy
    off_right: np.ndarray
    d_off_left: torch.Tensor
    d_off_right: torch.Tensor
    kp_left: torch.Tensor
    kp_right: torch.Tensor
    des_left: torch.Tensor
    des_right: torch.Tensor
    host: Optional[object] = None     # the PackedFrames the inputs came from, if they did

    @property
    def desc_bytes(self) -> int:
        return int(self.des_left.shape[1])


class FrontendResult:
    def __init__(self, inputs: FrontendInputs, W: int, t: dict):
        self.inputs, self.W, self.t = inputs, W, t
        self.frame_cross_check = False
        self._host = None
        self._pending = None

    # gathers of the inputs by the stereo match indices: two thirds of the result bytes, and
    # a host that still holds the inputs (PackedFrames) can redo them itself when asked
    _GATHERS = ("st_desc", "st_pix_left", "st_pix_right")

    def to_host(self, gathers: Optional[bool] = None) -> dict:
        """The result arrays as NumPy arrays.  They are copied through page-locked
        buffers (several times the rate of pageable copies) on the current stream, one
        synchronisation for all of them; the arrays keep their buffers alive.
        `gathers`: also copy the inlier descriptors and pixels (rows of the inputs gathered
        by the stereo matches); default: only when the inputs did not come from host
        arrays this object can gather from itself (`frame()` does either way)."""
        self.to_host_async(gathers)
        if self._pending is not None:
            ev, pinned = self._pending
            ev.synchronize()
            self._host = dict(self._host or {}, **{k: v.numpy() for k, v in pinned.items()})
            self._pending = None
        return self._host

    def to_host_async(self, gathers: Optional[bool] = None) -> None:
        """Starts the copies of to_host() on the current stream and returns; the next
        to_host() waits for them."""
        if gathers is None:
            gathers = self.inputs.host is None
        have = set(self._host or {})
        if self._pending is not None:
            have |= set(self._pending[1])
        want = [k for k in self.t if (gathers or k not in self._GATHERS) and k not in have]
        if not want:
            return
        if self._pending is not None:       # a second request before the first was collected
            ev, pinned = self._pending
            ev.synchronize()
            self._host = dict(self._host or {}, **{k: v.numpy() for k, v in pinned.items()})
        pinned = {k: torch.empty(self.t[k].shape, dtype=self.t[k].dtype, pin_memory=True) for k in want}
        for k, v in pinned.items():
            v.copy_(self.t[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._pending = (ev, pinned)

    # ------------------------------------------------------------ per-frame views
    def frame(self, f: int) -> dict:
        h = self.to_host()
        if "st_desc" not in h and self.inputs.host is None:
            h = self.to_host(gathers=True)      # no host copy of the inputs to gather from: fetch the rows
        b = int(self.inputs.off_left[f])
        m = int(h["st_count"][f])
        sl = slice(b, b + m)
        if "st_desc" in h:
            desc = h["st_desc"].reshape(-1, self.W)[sl].view(np.uint8)[:, :self.inputs.desc_bytes]
            pl, pr = h["st_pix_left"].reshape(-1, 2)[sl], h["st_pix_right"].reshape(-1, 2)[sl]
        else:           # the same rows taken from the host's own copy of the inputs
            src = self.inputs.host
            li = b + h["st_left"][sl].astype(np.int64)
            ri = int(self.inputs.off_right[f]) + h["st_right"][sl].astype(np.int64)
            desc, pl, pr = src.des_left[li], src.kp_left[li], src.kp_right[ri]
        return {
            "left_idx": h["st_left"][sl], "right_idx": h["st_right"][sl], "dist": h["st_dist"][sl],
            "xyz": h["st_xyz"].reshape(-1, 3)[sl], "pix_left": pl, "pix_right": pr,
            "inlier_des_left": desc,
            # Link(x_left, x_right, y) of TrackingDB.create_links (database.py:276)
            "links": np.column_stack([pl[:, 0].astype(np.float64), pr[:, 0].astype(np.float64),
                                      (pl[:, 1].astype(np.float64) + pr[:, 1].astype(np.float64)) / 2]),
        }

    def pair(self, p: int) -> dict:
        """Temporal stage of frames (p, p+1)."""
        h = self.to_host()
        b = int(self.inputs.off_left[p])
        info = h["ransac_info"].reshape(-1, 8)[p]
        n = int(info[5])
        sl = slice(b, b + n)
        return {
            "query_idx": h["tm_query"][sl], "train_idx": h["tm_train"][sl], "dist": h["tm_dist"][sl],
            "mask": h["tm_mask"][sl].astype(bool),
            "Rt_ransac": h["pose_ransac"].reshape(-1, 3, 4)[p],
            "Rt": h["pose_refined"].reshape(-1, 3, 4)[p],
            "best_index": int(info[0]), "best_count": int(info[1]), "iterations": int(info[2]),
            "n_iter_required": int(info[3]), "domain_error": bool(info[4]), "n_matches": n,
        }

    def relative_Rts(self) -> np.ndarray:
        """(F-1,3,4): what the reference pickles as `relative_Rts`
        (fill_database.py:252-255)."""
        return self.to_host()["pose_refined"].reshape(-1, 3, 4).copy()


class ChunkedResult:
    """Results of a sequence processed in chunks of consecutive pairs (TrackingFrontend.run_chunked):
    frame(f) / pair(p) / relative_Rts() as on FrontendResult, answered by the chunk that owns
    the frame or pair."""

    def __init__(self, n_frames: int, chunk_pairs: int, chunks):
        self.n_frames, self.chunk_pairs, self.chunks = n_frames, chunk_pairs, chunks

    def to_host(self):
        return [c.to_host() for c in self.chunks]

    def frame(self, f: int) -> dict:
        k = min(f // self.chunk_pairs, len(self.chunks) - 1)
        return self.chunks[k].frame(f - k * self.chunk_pairs)

    def pair(self, p: int) -> dict:
        k = min(p // self.chunk_pairs, len(self.chunks) - 1)
        return self.chunks[k].pair(p - k * self.chunk_pairs)

    def relative_Rts(self) -> np.ndarray:
        out = [c.to_host()["pose_refined"].reshape(-1, 3, 4)[:c.inputs.n_frames - 1] for c in self.chunks]
        return np.concatenate(out) if out else np.zeros((0, 3, 4))


class TrackingFrontend:
    """Parameters follow the reference's defaults: disparity_min / pixel_thresh
    (run_slam.py:31-32; pixel_thresh is shared by the stereo filter and the PnP
    inlier test, fill_database.py:81-82), p_success and n_iter_max (pnp.py:104-105),
    no cross-check in either matcher (run_slam.py:34-35)."""

    def __init__(self, cam, disparity_min: float = 1.2, pixel_thresh: float = 4.0,
                 p_success: float = 0.99999, n_iter_max: int = 2500, refine_iters: int = 30,
                 seed: int = 0, stereo_cross_check: bool = False, frame_cross_check: bool = False):
        self.cam = cam
        # crossCheck of the stereo / consecutive-frame matcher (fill_database.py:69-70,
        # run_slam.py:34-35)
        self.stereo_cross_check, self.frame_cross_check = bool(stereo_cross_check), bool(frame_cross_check)
        self.disparity_min, self.pixel_thresh = float(disparity_min), float(pixel_thresh)
        self.p_success, self.n_iter_max = float(p_success), int(n_iter_max)
        self.refine_iters, self.seed = int(refine_iters), int(seed)
        self._ws = None

    # ------------------------------------------------------------------ staging
    @staticmethod
    def upload(frames) -> FrontendInputs:
        """Frames -> device.  A list of per-frame tuples is concatenated into page-locked
        staging buffers first; a PackedFrames (already concatenated; page-locked after
        pin()) goes over as it is, four asynchronous copies."""
        if isinstance(frames, PackedFrames):
            if len(frames) < 1:
                raise ValueError("need at least one frame")
            dev = kernels.to_device
            d_off_l, d_off_r = dev(frames.off_left, torch.int32), dev(frames.off_right, torch.int32)
            kl, kr, dl, dr = (t.to(d_off_l.device, non_blocking=True) for t in frames.tensors())
            return FrontendInputs(len(frames), frames.off_left, frames.off_right, d_off_l, d_off_r,
                                  kl, kr, dl, dr, host=frames)
        if len(frames) < 1:
            raise ValueError("need at least one frame")
        nl = [len(f[0]) for f in frames]
        nr = [len(f[1]) for f in frames]
        for f in frames:
            if len(f[2]) != len(f[0]) or len(f[3]) != len(f[1]):
                raise ValueError("keypoints and descriptors differ in length")
        off_l = np.concatenate([[0], np.cumsum(nl)]).astype(np.int32)
        off_r = np.concatenate([[0], np.cumsum(nr)]).astype(np.int32)
        d = frames[0][2].shape[1]
        device = kernels.to_device(off_l, torch.int32).device

        def cat(i, np_dt, t_dt, w, total):
            # concatenated straight into a page-locked staging buffer, then one
            # asynchronous copy
            stage = _staging(i, t_dt, total * w).view(total, w)
            if total:
                np.concatenate([(binary_descriptors(f[i]) if np_dt == np.uint8 else np.asarray(f[i], dtype=np_dt))
                                .reshape(-1, w) for f in frames], out=stage.numpy())
            return stage.to(device, non_blocking=True)

        _staging_wait()
        dev = kernels.to_device
        inputs = FrontendInputs(len(frames), off_l, off_r, dev(off_l, torch.int32), dev(off_r, torch.int32),
                                cat(0, np.float32, torch.float32, 2, int(off_l[-1])),
                                cat(1, np.float32, torch.float32, 2, int(off_r[-1])),
                                cat(2, np.uint8, torch.uint8, d, int(off_l[-1])),
                                cat(3, np.uint8, torch.uint8, d, int(off_r[-1])))
        _staging_mark()
        return inputs

    # ------------------------------------------------------------------ chunks
    def run_chunked(self, packed: PackedFrames, chunk_pairs: int = 128, samples: Optional[np.ndarray] = None,
                    to_host: bool = True) -> ChunkedResult:
        """A sequence held in (page-locked) host memory in chunks of `chunk_pairs` consecutive
        pairs, three stages in flight: while the pass over chunk k runs on the current stream,
        the features of chunk k+1 go host-to-device on a second stream and the results of
        chunk k-1 device-to-host on a third.  Consecutive chunks share one frame (pair p needs
        frames p and p+1; its stereo stage is done twice).  Same results as one run() over
        the whole sequence: the device sampler keys pair p by its index in the sequence."""
        F = len(packed)
        if F < 1:
            raise ValueError("need at least one frame")
        chunk_pairs = max(1, int(chunk_pairs))
        starts = list(range(0, max(F - 1, 1), chunk_pairs))
        main = torch.cuda.current_stream()
        s_in, s_out = self._side_streams()
        s_in.wait_stream(main)
        s_out.wait_stream(main)

        def stage_in(k):
            a = starts[k]
            b = min(a + chunk_pairs, F - 1)
            with torch.cuda.stream(s_in):
                inp = self.upload(packed[a:b + 1])
                ev = torch.cuda.Event()
                ev.record(s_in)
            # allocated on the side stream, read by the pass on the main one: tell the caching
            # allocator, so that the memory is not handed out again while that pass is still
            # queued should the result be dropped early
            for t in (inp.d_off_left, inp.d_off_right, inp.kp_left, inp.kp_right, inp.des_left, inp.des_right):
                t.record_stream(main)
            return inp, ev

        results = []
        nxt = stage_in(0)
        for k, a in enumerate(starts):
            inp, ev = nxt
            if k + 1 < len(starts):
                nxt = stage_in(k + 1)
            main.wait_event(ev)
            smp = None if samples is None else samples[a:a + inp.n_frames - 1]
            res = self.run(inp, samples=smp, first_pair=a)
            results.append(res)
            if to_host:
                done = torch.cuda.Event()
                done.record(main)
                with torch.cuda.stream(s_out):
                    s_out.wait_event(done)
                    res.to_host_async()
        main.wait_stream(s_in)      # the inputs' memory may be reused only after their copies
        out = ChunkedResult(F, chunk_pairs, results)
        if to_host:
            out.to_host()
        return out

    def run_streamed(self, packed: PackedFrames, chunk_pairs: int = 64, samples: Optional[np.ndarray] = None
                     ) -> FrontendResult:
        """A sequence held in page-locked host memory, features up and results down while the
        passes run, as ONE result over the whole sequence (device arrays in `t`, the host copies
        already made).  All host-to-device copies are queued back to back on a second stream at
        the start, so that direction of the host link never waits for the host: the frames land
        in whole-sequence device arrays, `chunk_pairs` pairs' worth at a time with an event behind
        each piece; the passes follow the events on the current stream, each over the window
        [its first frame, its last frame] of those arrays, the stereo stage of a window's first
        frame being the one the window before has left in place (first_stereo_frame = 1: no
        frame is processed twice); a third stream copies every finished window's rows to the
        host.  Same results as run() over the whole sequence."""
        F = len(packed)
        if F < 2:
            return self.run(self.upload(packed), samples=samples)
        chunk_pairs = max(1, int(chunk_pairs))
        off_l, off_r = np.asarray(packed.off_left, np.int64), np.asarray(packed.off_right, np.int64)
        NL, NR = int(off_l[-1]), int(off_r[-1])
        kl_h, kr_h, dl_h, dr_h = packed.tensors()
        if dl_h.dim() != 2 or kl_h.dtype != torch.float32 or dl_h.dtype != torch.uint8:
            raise ValueError("keypoints must be float32 (n, 2) and descriptors uint8 (n, D)")
        D = int(dl_h.shape[1])
        W = 8 if D <= 32 else (16 if D <= 64 else 32)
        main = torch.cuda.current_stream()
        dev = main.device
        s_in, s_out = self._side_streams()
        s_in.wait_stream(main)
        s_out.wait_stream(main)
        starts = list(range(0, F - 1, chunk_pairs))
        windows = [(a, min(a + chunk_pairs, F - 1)) for a in starts]          # frames a .. b inclusive
        # offsets: the whole sequence's, then every window's own (counted from its first frame)
        offs = [off_l, off_r]
        for a, b in windows:
            offs += [off_l[a:b + 2] - off_l[a], off_r[a:b + 2] - off_r[a]]
        at = np.concatenate([[0], np.cumsum([len(o) for o in offs])])
        offs_h = torch.empty(int(at[-1]), dtype=torch.int32, pin_memory=True)
        offs_h.numpy()[:] = np.concatenate(offs)
        # device arrays of the whole sequence; allocated on the current stream, written by s_in
        d_in = {"kp_left": torch.empty((NL, 2), dtype=torch.float32, device=dev),
                "kp_right": torch.empty((NR, 2), dtype=torch.float32, device=dev),
                "des_left": torch.empty((NL, D), dtype=torch.uint8, device=dev),
                "des_right": torch.empty((NR, D), dtype=torch.uint8, device=dev)}
        d_offs = torch.empty(int(at[-1]), dtype=torch.int32, device=dev)
        rows = {"st_left": (torch.int32, 1), "st_right": (torch.int32, 1), "st_dist": (torch.int32, 1),
                "st_xyz": (torch.float32, 3), "st_pix_left": (torch.float32, 2), "st_pix_right": (torch.float32, 2),
                "st_desc": (torch.int32, W), "tm_query": (torch.int32, 1), "tm_train": (torch.int32, 1),
                "tm_dist": (torch.int32, 1), "tm_mask": (torch.uint8, 1)}
        t = {name: torch.empty(max(NL, 1) * w, dtype=dt, device=dev) for name, (dt, w) in rows.items()}
        t["st_count"] = torch.empty(F, dtype=torch.int32, device=dev)
        t["pose_ransac"] = torch.empty((F - 1) * 12, dtype=torch.float64, device=dev)
        t["pose_refined"] = torch.empty((F - 1) * 12, dtype=torch.float64, device=dev)
        t["ransac_info"] = torch.empty((F - 1) * 8, dtype=torch.int32, device=dev)
        d_samples = None
        if samples is not None:
            samples = np.ascontiguousarray(samples, dtype=np.int32)
            if samples.shape != (F - 1, self.n_iter_max, 4):
                raise ValueError("samples must have shape (n_frames-1, n_iter_max, 4)")
            d_samples = kernels.to_device(samples, torch.int32)
        for x in list(d_in.values()) + [d_offs]:
            x.record_stream(s_in)
        # ---- every copy of the inputs, back to back on s_in, an event behind each window's share
        events = []
        with torch.cuda.stream(s_in):
            d_offs.copy_(offs_h, non_blocking=True)
            for k, (a, b) in enumerate(windows):
                fa = a if k == 0 else a + 1                 # frame a came with the window before
                for name, src, off in (("kp_left", kl_h, off_l), ("kp_right", kr_h, off_r),
                                       ("des_left", dl_h, off_l), ("des_right", dr_h, off_r)):
                    r0, r1 = int(off[fa]), int(off[b + 1])
                    if r1 > r0:
                        d_in[name][r0:r1].copy_(src[r0:r1], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(s_in)
                events.append(ev)
        # ---- host arrays of the results (page-locked, through torch's caching host allocator)
        gathers = ("st_desc", "st_pix_left", "st_pix_right")
        host = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in t.items() if k not in gathers}
        for v in t.values():
            v.record_stream(s_out)
        # ---- the argument block: one for the whole sequence (kept with the result), one that
        # moves from window to window
        def block(n_frames, nl, nr, first_frame, row_l, row_r, off_at_l, off_at_r, first_stereo):
            a = self._new_args()
            a.n_frames, a.n_left_total, a.n_right_total = n_frames, nl, nr
            a.max_left, a.max_right = max_l, max_r
            a.desc_bytes, a.desc_stride = D, D
            a.d_kp_left = d_in["kp_left"].data_ptr() + 8 * row_l
            a.d_kp_right = d_in["kp_right"].data_ptr() + 8 * row_r
            a.d_des_left = d_in["des_left"].data_ptr() + D * row_l
            a.d_des_right = d_in["des_right"].data_ptr() + D * row_r
            a.d_off_left = d_offs.data_ptr() + 4 * off_at_l
            a.d_off_right = d_offs.data_ptr() + 4 * off_at_r
            a.first_pair, a.first_stereo_frame = first_frame, first_stereo
            for name, (dt, w) in rows.items():
                setattr(a, "d_" + name, t[name].data_ptr() + t[name].element_size() * w * row_l)
            a.d_st_count = t["st_count"].data_ptr() + 4 * first_frame
            a.d_pose_ransac = t["pose_ransac"].data_ptr() + 96 * first_frame
            a.d_pose_refined = t["pose_refined"].data_ptr() + 96 * first_frame
            a.d_ransac_info = t["ransac_info"].data_ptr() + 32 * first_frame
            if d_samples is not None:
                a.d_samples = d_samples.data_ptr() + 16 * self.n_iter_max * first_frame
            return a
        max_l, max_r = int(np.diff(off_l).max()), int(np.diff(off_r).max())
        blocks = []
        for k, (a, b) in enumerate(windows):
            blocks.append(block(b - a + 1, int(off_l[b + 1] - off_l[a]), int(off_r[b + 1] - off_r[a]), a,
                                int(off_l[a]), int(off_r[a]), int(at[2 + 2 * k]), int(at[3 + 2 * k]),
                                0 if k == 0 else 1))
        need = 0
        for blk in blocks:
            n = lib.slamhip_frontend_workspace_bytes(C.byref(blk))
            if n == 0:
                check(_lib.EUNSUPPORTED, "slamhip_frontend_workspace_bytes")
            need = max(need, n)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        # ---- passes on the current stream, results down on s_out
        for k, ((a, b), blk) in enumerate(zip(windows, blocks)):
            blk.d_workspace, blk.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
            main.wait_event(events[k])
            check(lib.slamhip_frontend_batch(C.byref(blk), main.cuda_stream), "slamhip_frontend_batch")
            done = torch.cuda.Event()
            done.record(main)
            fa = a if k == 0 else a + 1          # frames whose stereo rows this window has written
            with torch.cuda.stream(s_out):
                s_out.wait_event(done)
                for name, (dt, w) in rows.items():
                    if name in gathers:
                        continue
                    lo, hi = (a, b) if name.startswith("tm_") else (fa, b + 1)
                    r0, r1 = int(off_l[lo]) * w, int(off_l[hi]) * w
                    if r1 > r0:
                        host[name][r0:r1].copy_(t[name][r0:r1], non_blocking=True)
                host["st_count"][fa:b + 1].copy_(t["st_count"][fa:b + 1], non_blocking=True)
                for name, w in (("pose_ransac", 12), ("pose_refined", 12), ("ransac_info", 8)):
                    host[name][a * w:b * w].copy_(t[name][a * w:b * w], non_blocking=True)
        whole = block(F, NL, NR, 0, 0, 0, 0, int(at[1]), 0)
        whole.d_workspace, whole.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
        inputs = FrontendInputs(F, off_l.astype(np.int32), off_r.astype(np.int32), d_offs[:F + 1],
                                d_offs[int(at[1]):int(at[1]) + F + 1], d_in["kp_left"], d_in["kp_right"],
                                d_in["des_left"], d_in["des_right"], host=packed)
        res = FrontendResult(inputs, W, t)
        res.frame_cross_check = self.frame_cross_check
        res.args = whole
        res._keepalive = (d_samples, offs_h, blocks)
        main.wait_stream(s_in)
        main.wait_stream(s_out)
        s_out.synchronize()
        res._host = {k: v.numpy() for k, v in host.items()}
        res._pinned_host = host
        return res

    def _new_args(self) -> FrontendArgs:
        """An argument block with the camera and the parameters of this front-end filled in."""
        a = FrontendArgs()
        for name, src, n in (("K", self.cam.K, 9), ("M_right", self.cam.M_right, 12),
                             ("P_left", self.cam.P_left, 12), ("P_right", self.cam.P_right, 12)):
            vals = np.asarray(src, dtype=np.float64).reshape(-1)
            if vals.size != n:
                raise ValueError(f"camera.{name} has the wrong shape")
            setattr(a, name, (C.c_double * n)(*vals.tolist()))
        a.pixel_thresh, a.disparity_min = self.pixel_thresh, self.disparity_min
        a.ransac_thresh, a.p_success = self.pixel_thresh, self.p_success
        a.n_iter_max, a.refine_iters, a.seed = self.n_iter_max, self.refine_iters, self.seed
        a.stereo_cross_check, a.temporal_cross_check = int(self.stereo_cross_check), int(self.frame_cross_check)
        return a

    def _side_streams(self):
        if getattr(self, "_streams", None) is None:
            self._streams = (torch.cuda.Stream(), torch.cuda.Stream())
        return self._streams

    # ---------------------------------------------------------------------- run
    def run(self, frames_or_inputs, samples: Optional[np.ndarray] = None,
            first_pair: int = 0) -> FrontendResult:
        """`first_pair`: index of this batch's first pair in the whole sequence; the
        device sampler keys pair p's samples by (seed, first_pair + p), so batches and
        shards of a sequence draw the samples of the uncut run."""
        inp = frames_or_inputs if isinstance(frames_or_inputs, FrontendInputs) else self.upload(frames_or_inputs)
        if inp.kp_left.dtype != torch.float32 or inp.des_left.dtype != torch.uint8 or inp.des_left.dim() != 2:
            raise ValueError("keypoints must be float32 (n, 2) and descriptors uint8 (n, D)")
        F = inp.n_frames
        NL, NR = int(inp.off_left[-1]), int(inp.off_right[-1])
        D = inp.desc_bytes
        W = 8 if D <= 32 else (16 if D <= 64 else 32)
        dev = inp.kp_left.device
        i32 = lambda n: torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        f32 = lambda n: torch.empty(max(n, 1), dtype=torch.float32, device=dev)
        # The pair outputs share one allocation.  A call with at least one pair writes every
        # element of them (the mask rows of a frame beyond its matches included), so nothing is
        # filled beforehand; a call of one frame has no pair, and they read as zero.
        P = max(F - 1, 1)
        zsizes = (("pose_ransac", torch.float64, P * 12),
                  ("pose_refined", torch.float64, P * 12), ("ransac_info", torch.int32, P * 8),
                  ("tm_mask", torch.uint8, max(NL, 1)))
        offs, total = [], 0
        for _, dt, n in zsizes:
            offs.append(total)
            total = (total + n * torch.empty(0, dtype=dt).element_size() + 15) // 16 * 16
        zbuf = (torch.empty if F >= 2 and NL > 0 else torch.zeros)(total, dtype=torch.uint8, device=dev)
        t = {
            "st_count": i32(F), "st_left": i32(NL), "st_right": i32(NL), "st_dist": i32(NL),
            "st_xyz": f32(NL * 3), "st_pix_left": f32(NL * 2), "st_pix_right": f32(NL * 2),
            "tm_query": i32(NL), "tm_train": i32(NL), "tm_dist": i32(NL),
            # rows [off_left[f], +st_count[f]) of each frame are written, whole packed rows; the
            # rest is never read (64 bytes per feature: not worth a fill launch per call)
            "st_desc": i32(NL * W),
        }
        for (name, dt, n), o in zip(zsizes, offs):
            t[name] = zbuf[o:o + n * torch.empty(0, dtype=dt).element_size()].view(dt)
        a = FrontendArgs()
        a.n_frames, a.n_left_total, a.n_right_total = F, NL, NR
        a.max_left = int(np.diff(inp.off_left).max()) if F else 0
        a.max_right = int(np.diff(inp.off_right).max()) if F else 0
        # rows may lie in a wider buffer (e.g. 64-byte slots for 61-byte descriptors: aligned loads)
        if inp.des_left.stride(1) != 1 or inp.des_right.stride(1) != 1 or \
                (NR > 0 and inp.des_right.stride(0) != inp.des_left.stride(0)):
            raise ValueError("descriptor rows must be contiguous and both arrays must have one row stride")
        a.desc_bytes, a.desc_stride = D, int(inp.des_left.stride(0))
        a.d_kp_left, a.d_kp_right = inp.kp_left.data_ptr(), inp.kp_right.data_ptr()
        a.d_des_left, a.d_des_right = inp.des_left.data_ptr(), inp.des_right.data_ptr()
        a.d_off_left, a.d_off_right = inp.d_off_left.data_ptr(), inp.d_off_right.data_ptr()
        for name, src, n in (("K", self.cam.K, 9), ("M_right", self.cam.M_right, 12),
                             ("P_left", self.cam.P_left, 12), ("P_right", self.cam.P_right, 12)):
            vals = np.asarray(src, dtype=np.float64).reshape(-1)
            if vals.size != n:
                raise ValueError(f"camera.{name} has the wrong shape")
            setattr(a, name, (C.c_double * n)(*vals.tolist()))
        a.pixel_thresh, a.disparity_min = self.pixel_thresh, self.disparity_min
        a.ransac_thresh, a.p_success = self.pixel_thresh, self.p_success
        a.n_iter_max, a.refine_iters, a.seed = self.n_iter_max, self.refine_iters, self.seed
        a.first_pair = int(first_pair)
        a.stereo_cross_check, a.temporal_cross_check = int(self.stereo_cross_check), int(self.frame_cross_check)
        d_samples = None
        if samples is not None:
            samples = np.ascontiguousarray(samples, dtype=np.int32)
            if samples.shape != (max(F - 1, 0), self.n_iter_max, 4):
                raise ValueError("samples must have shape (n_frames-1, n_iter_max, 4)")
            d_samples = kernels.to_device(samples, torch.int32)
            a.d_samples = d_samples.data_ptr()
        for k, v in t.items():
            setattr(a, "d_" + k, v.data_ptr())
        need = lib.slamhip_frontend_workspace_bytes(C.byref(a))
        if need == 0:
            check(_lib.EUNSUPPORTED, "slamhip_frontend_workspace_bytes")
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        a.d_workspace, a.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
        check(lib.slamhip_frontend_batch(C.byref(a), torch.cuda.current_stream().cuda_stream),
              "slamhip_frontend_batch")
        res = FrontendResult(inp, W, t)
        res.frame_cross_check = self.frame_cross_check      # selects the hand-over shape (db_adapter)
        res._keepalive = (d_samples,)
        res.args = a            # the call's argument block (sharding packs hand-over records from it)
        return res
