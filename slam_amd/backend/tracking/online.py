"""Frame-at-a-time front-end: the calling pattern of the reference's own loop
(TrackingPipeline.build(), backend/tracking/fill_database.py:140-240), which gets one stereo
frame per iteration, matches it against the previous frame's stereo inliers and carries
the new frame's stereo stage over to the next iteration (:234-240).

`OnlineFrontend.push(frame)` uploads the new frame only, runs its stereo stage and the
temporal stage against the frame before it in ONE library call (`slamhip_frontend_batch`
with two frames and `first_stereo_frame = 1`), and returns the pair's result.  The previous
frame's stereo outputs stay where the call before left them: the frames of a sequence are
appended one behind the other in device arrays sized for many frames, and the library is
handed the window [previous frame, new frame] of every array.  When the arrays are full
the last frame's rows are moved to their front.  The device sampler is keyed by the pair's
number, so pushing a sequence frame by frame gives, bit for bit, the results of one batched
call over the whole sequence (`TrackingFrontend.run`).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from ..._lib import lib, check, FrontendArgs
from ... import _lib
from .feature_cache import binary_descriptors
from .frontend import FrontendInputs, FrontendResult, TrackingFrontend

# arrays with one entry (of `width` elements) per left feature / per right feature
_LEFT = (("kp_left", torch.float32, 2), ("des_left", torch.uint8, None),
         ("st_left", torch.int32, 1), ("st_right", torch.int32, 1), ("st_dist", torch.int32, 1),
         ("st_xyz", torch.float32, 3), ("st_pix_left", torch.float32, 2), ("st_pix_right", torch.float32, 2),
         ("st_desc", torch.int32, "W"), ("tm_query", torch.int32, 1), ("tm_train", torch.int32, 1),
         ("tm_dist", torch.int32, 1), ("tm_mask", torch.uint8, 1))
_RIGHT = (("kp_right", torch.float32, 2), ("des_right", torch.uint8, None))
_OUTPUTS = ("st_left", "st_right", "st_dist", "st_xyz", "st_pix_left", "st_pix_right", "st_desc",
            "tm_query", "tm_train", "tm_dist", "tm_mask")


class OnlinePair(FrontendResult):
    """Result of one push: a two-frame FrontendResult (frame(0) = the previous frame, frame(1) =
    the new one, pair(0) = the two), whose arrays are windows of the OnlineFrontend's device
    arrays -- valid until the push after the next one.  `pose()` is the cheap way to the
    relative pose alone; the windows themselves are only set up when something else is asked for."""

    def __init__(self, owner, window):
        self._owner, self._window = owner, window
        self._built = False
        self.frame_cross_check = owner.fe.frame_cross_check
        self._host = None
        self._pending = None

    def _build(self):
        if self._built:
            return
        b, (F, wl, wr, NL, NR, slot, off) = self._owner._buf, self._window
        if b is not self._window_buf:
            raise RuntimeError("the OnlineFrontend's arrays were reallocated since this push")
        t = {name: b[name][wl:wl + NL].reshape(-1) for name in _OUTPUTS}
        t["st_count"] = b["st_count"][slot:slot + F]
        for name in ("pose_ransac", "pose_refined", "ransac_info"):
            t[name] = b[name][slot]
        inp = FrontendInputs(F, off[:F + 1].copy(), off[3:3 + F + 1].copy(), b["off"][:3], b["off"][3:],
                             b["kp_left"][wl:wl + NL], b["kp_right"][wr:wr + NR],
                             b["des_left"][wl:wl + NL], b["des_right"][wr:wr + NR])
        FrontendResult.__init__(self, inp, b["W"], t)
        self.frame_cross_check = self._owner.fe.frame_cross_check
        self._built = True

    def __getattr__(self, name):          # inputs / W / t: set up on first use
        if name in ("inputs", "W", "t"):
            self._build()
            return self.__dict__[name]
        raise AttributeError(name)

    def pose(self):
        """(Rt (3,4) float64, info (8,) int32) of the pair: refined [R|t] previous -> new
        frame (a row of the reference's `relative_Rts`, fill_database.py:240) and the RANSAC
        record (best index, best count, iterations, n_iter_required, math-domain flag,
        matches, mask count, 0).  One small copy through a page-locked buffer."""
        o, slot = self._owner, self._window[5]
        o._pin_record.copy_(self._window_buf["record"][slot], non_blocking=True)
        o._stream.synchronize()
        rec = o._pin_record.numpy()
        return rec[:12].reshape(3, 4).copy(), rec[12:].view(np.int32).copy()


class OnlineRecords:
    """The records of a sequence pushed frame by frame, on the host: frame(f) / pair(p) /
    relative_Rts() / n_frames as on a FrontendResult, i.e. what db_adapter.replay_into_db,
    save_records and save_relative_Rts take (the hand-over to TrackingDB.add_frame,
    fill_database.py:231, and the `relative_Rts` file, :252-255).  `add(pair)` after every push
    that returned a pair."""

    def __init__(self, frame_cross_check: bool = False):
        self.frame_cross_check = bool(frame_cross_check)
        self._frames, self._pairs = [], []

    def add(self, pair: "OnlinePair") -> None:
        copy = lambda d: {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
        if not self._frames:
            self._frames.append(copy(pair.frame(0)))      # the sequence's first frame
        self._frames.append(copy(pair.frame(1)))
        self._pairs.append(copy(pair.pair(0)))

    @property
    def n_frames(self) -> int:
        return len(self._frames)

    def frame(self, f: int) -> dict:
        return self._frames[f]

    def pair(self, p: int) -> dict:
        return self._pairs[p]

    def relative_Rts(self) -> np.ndarray:
        return np.stack([p["Rt"] for p in self._pairs]) if self._pairs else np.zeros((0, 3, 4))


class OnlineFrontend:
    """`frames_capacity`: how many frames of the first frame's size the device arrays hold
    before the last frame's rows are moved to the front (a dozen small copies); parameters
    as TrackingFrontend."""

    _SLOTS = 4096      # per-frame / per-pair entries (counts, poses) before they wrap

    def __init__(self, cam, frames_capacity: int = 32, **params):
        self.fe = TrackingFrontend(cam, **params)
        self.frames_capacity = max(int(frames_capacity), 3)
        self.n_pushed = 0
        self._buf = None
        self._args = None

    # ------------------------------------------------------------------ storage
    def _allocate(self, cap_l: int, cap_r: int, D: int, dev):
        W = 8 if D <= 32 else (16 if D <= 64 else 32)
        widths = lambda w: D if w is None else (W if w == "W" else w)
        old = self._buf
        buf = {"cap_l": cap_l, "cap_r": cap_r, "D": D, "W": W}
        for name, dt, w in _LEFT:
            buf[name] = torch.empty((cap_l, widths(w)), dtype=dt, device=dev)
        for name, dt, w in _RIGHT:
            buf[name] = torch.empty((cap_r, widths(w)), dtype=dt, device=dev)
        buf["st_count"] = torch.zeros(self._SLOTS, dtype=torch.int32, device=dev)
        buf["pose_ransac"] = torch.zeros((self._SLOTS, 12), dtype=torch.float64, device=dev)
        # refined pose (12 float64) and RANSAC record (8 int32) of a pair side by side: one copy to the host
        buf["record"] = torch.zeros((self._SLOTS, 16), dtype=torch.float64, device=dev)
        buf["pose_refined"] = buf["record"][:, :12]
        buf["ransac_info"] = buf["record"][:, 12:].view(torch.int32)
        buf["off"] = torch.zeros(6, dtype=torch.int32, device=dev)
        # (address of row 0, bytes per row) of every array the library is handed a window of
        buf["where"] = {name: (buf[name].data_ptr(), buf[name].stride(0) * buf[name].element_size())
                        for name in [n for n, _, _ in _LEFT + _RIGHT] + ["st_count", "pose_ransac", "pose_refined",
                                                                          "ransac_info"]}
        self._buf = buf
        if old is None:
            self._pin_record = torch.zeros(16, dtype=torch.float64, pin_memory=True)
            self._pin_in = self._pin_np = self._packet = None
            self._off_event = self._ev = None
        return old

    def _move_previous_to_front(self, old=None):
        """Rows of the previous frame -> rows [0, n) of every array (of `old` arrays into the new
        ones after a reallocation)."""
        src = old if old is not None else self._buf
        ol, nl, orr, nr, j = self._ol, self._nl, self._or, self._nr, self._slot
        for name, _, _ in _LEFT:      # (source and destination may overlap: through a copy)
            self._buf[name][:nl].copy_(src[name][ol:ol + nl].clone())
        for name, _, _ in _RIGHT:
            self._buf[name][:nr].copy_(src[name][orr:orr + nr].clone())
        self._buf["st_count"][0:1].copy_(src["st_count"][j:j + 1].clone())
        self._ol = self._or = self._slot = 0

    # --------------------------------------------------------------------- push
    def push(self, kp_left, kp_right, des_left, des_right) -> Optional[OnlinePair]:
        """One new stereo frame.  Returns None for the first frame (the reference's frame 0
        has no pose either, fill_database.py:102-137), else the pair (previous, new)."""
        kp_left = np.ascontiguousarray(kp_left, dtype=np.float32).reshape(-1, 2)
        kp_right = np.ascontiguousarray(kp_right, dtype=np.float32).reshape(-1, 2)
        des_left, des_right = binary_descriptors(des_left, "des_left"), binary_descriptors(des_right, "des_right")
        if des_left.ndim != 2 or len(des_left) != len(kp_left) or len(des_right) != len(kp_right):
            raise ValueError("keypoints and descriptors differ in length")
        nl, nr, D = len(kp_left), len(kp_right), des_left.shape[1]
        if des_right.ndim != 2 or (nr and des_right.shape[1] != D):
            raise ValueError("left and right descriptors differ in width")
        first = self.n_pushed == 0
        stream = self._stream = torch.cuda.current_stream()
        dev = stream.device
        if first:
            self._ol = self._nl = self._or = self._nr = self._slot = 0
            if self._buf is None or self._buf["D"] != D or self._buf["cap_l"] < 2 * nl or self._buf["cap_r"] < 2 * nr:
                self._allocate(max(self.frames_capacity * nl, 1024), max(self.frames_capacity * nr, 1024), D, dev)
        elif D != self._buf["D"]:
            raise ValueError("descriptor width changed within a sequence")
        b = self._buf
        # where the new frame goes: right behind the previous one
        new_l, new_r = self._ol + self._nl, self._or + self._nr
        if not first and (self._nl + nl > b["cap_l"] or self._nr + nr > b["cap_r"]):
            old = self._allocate(max(b["cap_l"], 4 * (self._nl + nl)), max(b["cap_r"], 4 * (self._nr + nr)), D, dev)
            self._move_previous_to_front(old)
            b = self._buf
            new_l, new_r = self._nl, self._nr
        elif not first and (new_l + nl > b["cap_l"] or new_r + nr > b["cap_r"] or self._slot + 2 >= self._SLOTS):
            self._move_previous_to_front()
            new_l, new_r = self._nl, self._nr
        F = 1 if first else 2
        pl, pr = (0, 0) if first else (self._nl, self._nr)      # rows of the previous frame in the window
        off = np.array([0, pl, pl + nl, 0, pr, pr + nr] if not first else [0, nl, nl, 0, nr, nr], dtype=np.int32)
        # the new frame and the window's offsets go to the device as one packet: staged in a
        # page-locked buffer (free again once the copy of the push before has been made), one
        # asynchronous copy, one library launch that moves the pieces to their rows
        need = 24 + 8 * (nl + nr) + D * (nl + nr)
        if self._off_event is not None:
            self._off_event.synchronize()
        if self._pin_in is None or self._pin_in.numel() < need:
            self._pin_in = torch.empty(int(need * 1.5), dtype=torch.uint8, pin_memory=True)
            self._pin_np = self._pin_in.numpy()
        if self._packet is None or self._packet.numel() < need or self._packet.device != dev:
            self._packet = torch.empty(int(need * 1.5), dtype=torch.uint8, device=dev)
        stage, at = self._pin_np, 24
        stage[:24].view(np.int32)[:] = off
        for arr in (kp_left, kp_right, des_left, des_right):
            nb = arr.nbytes
            stage[at:at + nb] = arr.reshape(-1).view(np.uint8)
            at += nb
        self._packet[:need].copy_(self._pin_in[:need], non_blocking=True)
        if self._ev is None:
            self._ev = torch.cuda.Event()
        self._off_event = self._ev
        self._off_event.record(stream)
        where = b["where"]
        addr = lambda name, row: where[name][0] + row * where[name][1]      # device address of a row
        check(lib.slamhip_frontend_unpack_frame(self._packet.data_ptr(), nl, nr, D, D, addr("kp_left", new_l),
                                                addr("kp_right", new_r), addr("des_left", new_l),
                                                addr("des_right", new_r), b["off"].data_ptr(), stream.cuda_stream),
              "slamhip_frontend_unpack_frame")
        wl, wr = (new_l, new_r) if first else (self._ol, self._or)   # first row of the window
        NL, NR = pl + nl, pr + nr
        slot = self._slot                        # per-frame entry of the window's first frame
        a = self._arguments()
        a.n_frames, a.n_left_total, a.n_right_total = F, NL, NR
        a.max_left, a.max_right = max(pl, nl), max(pr, nr)
        a.desc_bytes, a.desc_stride = D, D
        a.d_kp_left, a.d_kp_right = addr("kp_left", wl), addr("kp_right", wr)
        a.d_des_left, a.d_des_right = addr("des_left", wl), addr("des_right", wr)
        d_off = b["off"].data_ptr()
        a.d_off_left, a.d_off_right = d_off, d_off + 12
        a.first_pair = max(self.n_pushed - 1, 0)
        a.first_stereo_frame = 0 if first else 1
        for name in _OUTPUTS:
            setattr(a, "d_" + name, addr(name, wl))
        for name in ("st_count", "pose_ransac", "pose_refined", "ransac_info"):
            setattr(a, "d_" + name, addr(name, slot))
        need = lib.slamhip_frontend_workspace_bytes(C.byref(a))
        if need == 0:
            check(_lib.EUNSUPPORTED, "slamhip_frontend_workspace_bytes")
        fe = self.fe
        if fe._ws is None or fe._ws.numel() < need or fe._ws.device != dev:
            fe._ws = torch.empty(int(need * 1.5), dtype=torch.uint8, device=dev)
        a.d_workspace, a.workspace_bytes = fe._ws.data_ptr(), fe._ws.numel()
        check(lib.slamhip_frontend_batch(C.byref(a), stream.cuda_stream), "slamhip_frontend_batch")
        res = OnlinePair(self, (F, wl, wr, NL, NR, slot, off))
        res._window_buf = b
        # the new frame is the next call's previous frame
        if not first:
            self._slot += 1
        self._ol, self._nl, self._or, self._nr = new_l, nl, new_r, nr
        self.n_pushed += 1
        return None if first else res

    def reset(self):
        """Forget the sequence: the next push is a first frame again."""
        self.n_pushed = 0

    def _arguments(self) -> FrontendArgs:
        if self._args is None:
            fe, a = self.fe, FrontendArgs()
            for name, src, n in (("K", fe.cam.K, 9), ("M_right", fe.cam.M_right, 12),
                                 ("P_left", fe.cam.P_left, 12), ("P_right", fe.cam.P_right, 12)):
                vals = np.asarray(src, dtype=np.float64).reshape(-1)
                if vals.size != n:
                    raise ValueError(f"camera.{name} has the wrong shape")
                setattr(a, name, (C.c_double * n)(*vals.tolist()))
            a.pixel_thresh, a.disparity_min = fe.pixel_thresh, fe.disparity_min
            a.ransac_thresh, a.p_success = fe.pixel_thresh, fe.p_success
            a.n_iter_max, a.refine_iters, a.seed = fe.n_iter_max, fe.refine_iters, fe.seed
            a.stereo_cross_check, a.temporal_cross_check = int(fe.stereo_cross_check), int(fe.frame_cross_check)
            self._args = a
        return self._args
