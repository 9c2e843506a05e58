"""Array-native entry points over the C ABI.

Inputs may be NumPy arrays (copied to the current ROCm device) or torch tensors
already on the device; outputs are torch tensors on the device.  torch is used
for device memory and the stream handle only; all arithmetic runs in
libslamhip.so.  Nothing here falls back to the CPU.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import lib, check

_workspaces = {}


def _require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("slam_amd needs a ROCm GPU (gfx950); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def to_device(x, dtype: torch.dtype) -> torch.Tensor:
    dev = _require_gpu()
    if isinstance(x, torch.Tensor):
        t = x.to(device=dev, dtype=dtype)
    else:
        t = torch.from_numpy(np.ascontiguousarray(x)).to(device=dev, dtype=dtype)
    return t.contiguous()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _workspace(nbytes: int) -> torch.Tensor:
    dev = _require_gpu()
    key = (dev.index, _stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
    return ws


def _binary(x, what):
    """Hamming descriptors are uint8 rows; anything else (float rows are matched under the L2
    norm, descriptor_matcher.py:23-28 of the reference) is refused instead of cast to bytes."""
    dt = x.dtype if isinstance(x, torch.Tensor) else np.asarray(x).dtype
    n = x.numel() if isinstance(x, torch.Tensor) else np.asarray(x).size
    if n and dt not in (torch.uint8, np.dtype(np.uint8)):
        raise TypeError(f"{what} descriptors must be uint8 for the Hamming matcher, got {dt} (l2_knn2 matches float rows)")
    return to_device(x, torch.uint8)


def _desc_pair(q, t):
    q = _binary(q, "query")
    t = _binary(t, "train")
    if q.dim() != 2 or t.dim() != 2:
        raise ValueError("descriptors must be 2-D (rows, bytes)")
    if t.shape[0] > 0 and q.shape[0] > 0 and q.shape[1] != t.shape[1]:
        raise ValueError("query and train descriptors differ in length")
    return q, t


def hamming_knn2(q, t, out=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Top-2 Hamming neighbours of every row of q among the rows of t.
    Returns (idx2, dist2): (nq, 2) int32 device tensors, -1 where absent.
    `out` = preallocated (idx2, dist2) to write into."""
    q, t = _desc_pair(q, t)
    nq, nt = q.shape[0], t.shape[0]
    d = q.shape[1] if nq else (t.shape[1] if nt else 1)
    if out is not None:
        idx2, dist2 = out
        if idx2.shape != (nq, 2) or dist2.shape != (nq, 2) or idx2.dtype != torch.int32 \
                or dist2.dtype != torch.int32 or not idx2.is_contiguous() or not dist2.is_contiguous() \
                or idx2.device != q.device or dist2.device != q.device:
            raise ValueError("out must be two contiguous (nq,2) int32 tensors on the descriptors' device")
    else:
        idx2 = torch.empty((nq, 2), dtype=torch.int32, device=q.device)
        dist2 = torch.empty((nq, 2), dtype=torch.int32, device=q.device)
    if nq == 0:
        return idx2, dist2
    if d < 1:
        raise ValueError("descriptor length must be >= 1 byte")
    need = lib.slamhip_hamming_knn2_workspace_bytes(nq, nt, 1, nq, nt, d)
    if need == 0:
        check(_lib.EUNSUPPORTED, "slamhip_hamming_knn2_workspace_bytes")
    ws = _workspace(need)
    check(lib.slamhip_hamming_knn2(_ptr(q), nq, _ptr(t), nt, d, d, _ptr(idx2), _ptr(dist2),
                                   _ptr(ws), ws.numel(), _stream()), "slamhip_hamming_knn2")
    return idx2, dist2


def hamming_knn2_ratio(q, t, ratio: float, out=None):
    """k=2 neighbours + ratio test in one call -> (idx2, dist2, keep)."""
    q, t = _desc_pair(q, t)
    nq, nt = q.shape[0], t.shape[0]
    d = q.shape[1] if nq else (t.shape[1] if nt else 1)
    if out is not None:
        idx2, dist2, keep = out
        for o, shape, dt in ((idx2, (nq, 2), torch.int32), (dist2, (nq, 2), torch.int32),
                             (keep, (nq,), torch.uint8)):
            if tuple(o.shape) != shape or o.dtype != dt or not o.is_contiguous() or o.device != q.device:
                raise ValueError("out must be contiguous device tensors on the descriptors' device: "
                                 "(nq,2) int32, (nq,2) int32, (nq,) uint8")
    else:
        idx2 = torch.empty((nq, 2), dtype=torch.int32, device=q.device)
        dist2 = torch.empty((nq, 2), dtype=torch.int32, device=q.device)
        keep = torch.empty(nq, dtype=torch.uint8, device=q.device)
    if nq == 0:
        return idx2, dist2, keep
    need = lib.slamhip_hamming_knn2_workspace_bytes(nq, nt, 1, nq, nt, d)
    if need == 0:
        check(_lib.EUNSUPPORTED, "slamhip_hamming_knn2_workspace_bytes")
    ws = _workspace(need)
    check(lib.slamhip_hamming_knn2_ratio(_ptr(q), nq, _ptr(t), nt, d, d, float(ratio), _ptr(idx2),
                                         _ptr(dist2), _ptr(keep), _ptr(ws), ws.numel(), _stream()),
          "slamhip_hamming_knn2_ratio")
    return idx2, dist2, keep


class BoundRatioMatch:
    """hamming_knn2_ratio with everything that does not change from call to call done once:
    device tensors checked, outputs and workspace allocated, the C arguments converted.
    Calling the object enqueues the matcher on the stream it was bound on and returns
    (idx2, dist2, keep), the same tensors every time.  For loops that match into the same
    buffers again and again (a tracker's per-frame call, a benchmark): the per-call host work
    is one foreign call instead of the argument checking of the array-native wrapper."""

    def __init__(self, q, t, ratio: float, out=None):
        import ctypes as C
        q, t = _desc_pair(q, t)
        nq, nt = q.shape[0], t.shape[0]
        if nq == 0 or nt == 0:
            raise ValueError("bind needs non-empty descriptor sets")
        d = q.shape[1]
        if out is None:
            out = (torch.empty((nq, 2), dtype=torch.int32, device=q.device),
                   torch.empty((nq, 2), dtype=torch.int32, device=q.device),
                   torch.empty(nq, dtype=torch.uint8, device=q.device))
        idx2, dist2, keep = out
        for o, shape, dt in ((idx2, (nq, 2), torch.int32), (dist2, (nq, 2), torch.int32), (keep, (nq,), torch.uint8)):
            if tuple(o.shape) != shape or o.dtype != dt or not o.is_contiguous() or o.device != q.device:
                raise ValueError("out must be contiguous device tensors on the descriptors' device: "
                                 "(nq,2) int32, (nq,2) int32, (nq,) uint8")
        need = lib.slamhip_hamming_knn2_workspace_bytes(nq, nt, 1, nq, nt, d)
        if need == 0:
            check(_lib.EUNSUPPORTED, "slamhip_hamming_knn2_workspace_bytes")
        ws = torch.empty(need, dtype=torch.uint8, device=q.device)      # its own: not shared with other calls
        self.out = out
        self._keep = (q, t, ws)
        self._fn = lib.slamhip_hamming_knn2_ratio
        self._args = (C.c_void_p(q.data_ptr()), C.c_int(nq), C.c_void_p(t.data_ptr()), C.c_int(nt), C.c_int(d),
                      C.c_int(d), C.c_double(float(ratio)), C.c_void_p(idx2.data_ptr()),
                      C.c_void_p(dist2.data_ptr()), C.c_void_p(keep.data_ptr()), C.c_void_p(ws.data_ptr()),
                      C.c_size_t(ws.numel()), C.c_void_p(_stream()))

    def __call__(self):
        rc = self._fn(*self._args)
        if rc != 0:
            check(rc, "slamhip_hamming_knn2_ratio")
        return self.out


def hamming_knn2_batched(q, t, q_off, t_off, max_nq: Optional[int] = None,
                         max_nt: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """n_pairs independent problems in one launch; offsets are (n_pairs+1,)
    int32 row offsets into the concatenated q / t arrays."""
    q, t = _desc_pair(q, t)
    if max_nq is None or max_nt is None:
        qo = np.asarray(q_off.cpu() if isinstance(q_off, torch.Tensor) else q_off)
        to = np.asarray(t_off.cpu() if isinstance(t_off, torch.Tensor) else t_off)
        max_nq = int(np.diff(qo).max()) if len(qo) > 1 else 0
        max_nt = int(np.diff(to).max()) if len(to) > 1 else 0
    q_off = to_device(q_off, torch.int32)
    t_off = to_device(t_off, torch.int32)
    n_pairs = q_off.numel() - 1
    if n_pairs < 1 or t_off.numel() != n_pairs + 1:
        raise ValueError("offset arrays must both have n_pairs+1 entries")
    nq, nt = q.shape[0], t.shape[0]
    d = q.shape[1] if nq else (t.shape[1] if nt else 1)
    idx2 = torch.empty((nq, 2), dtype=torch.int32, device=q.device)
    dist2 = torch.empty((nq, 2), dtype=torch.int32, device=q.device)
    if nq == 0:
        return idx2, dist2
    need = lib.slamhip_hamming_knn2_workspace_bytes(nq, nt, n_pairs, max_nq, max_nt, d)
    if need == 0:
        check(_lib.EUNSUPPORTED, "slamhip_hamming_knn2_workspace_bytes")
    ws = _workspace(need)
    check(lib.slamhip_hamming_knn2_batched(_ptr(q), nq, _ptr(t), nt, d, d, _ptr(q_off),
                                           _ptr(t_off), n_pairs, max_nq, max_nt, _ptr(idx2),
                                           _ptr(dist2), _ptr(ws), ws.numel(), _stream()),
          "slamhip_hamming_knn2_batched")
    return idx2, dist2


def l2_knn2(q, t) -> Tuple[torch.Tensor, torch.Tensor]:
    """Top-2 L2 neighbours of every row of q among the rows of t (float32 descriptors).
    Returns (idx2 (nq,2) int32, dist2 (nq,2) float32) device tensors, -1 where absent."""
    q = to_device(q, torch.float32)
    t = to_device(t, torch.float32)
    if q.dim() != 2 or t.dim() != 2:
        raise ValueError("descriptors must be 2-D (rows, floats)")
    nq, nt = q.shape[0], t.shape[0]
    if nq > 0 and nt > 0 and q.shape[1] != t.shape[1]:
        raise ValueError("query and train descriptors differ in length")
    d = q.shape[1] if nq else (t.shape[1] if nt else 1)
    idx2 = torch.empty((nq, 2), dtype=torch.int32, device=q.device)
    dist2 = torch.empty((nq, 2), dtype=torch.float32, device=q.device)
    if nq == 0:
        return idx2, dist2
    if d < 1:
        raise ValueError("descriptor length must be >= 1")
    need = lib.slamhip_l2_knn2_workspace_bytes(nq, nt, d)
    if need == 0:
        check(_lib.EUNSUPPORTED, "slamhip_l2_knn2_workspace_bytes")
    ws = _workspace(need)
    check(lib.slamhip_l2_knn2(_ptr(q), nq, _ptr(t), nt, d, d, _ptr(idx2), _ptr(dist2), _ptr(ws),
                              ws.numel(), _stream()), "slamhip_l2_knn2")
    return idx2, dist2


def l2_cross_check_match(q, t) -> Tuple[torch.Tensor, torch.Tensor]:
    """crossCheck=True matching under L2: (train index (nq,) int32, distance (nq,) float32),
    -1 where a query row has no match."""
    q = to_device(q, torch.float32)
    t = to_device(t, torch.float32)
    nq, nt = q.shape[0], t.shape[0]
    ridx, rdist = l2_knn2(t, q)
    mi = torch.empty(nq, dtype=torch.int32, device=q.device)
    md = torch.empty(nq, dtype=torch.float32, device=q.device)
    check(lib.slamhip_l2_cross_check(_ptr(ridx), _ptr(rdist), nt, nq, _ptr(mi), _ptr(md), _stream()),
          "slamhip_l2_cross_check")
    return mi, md


def ratio_test_f32(dist2, ratio: float) -> torch.Tensor:
    """Lowe ratio test on a float (nq,2) k=2 result: uint8 keep flags."""
    dist2 = to_device(dist2, torch.float32).contiguous()
    n = dist2.shape[0]
    keep = torch.empty(n, dtype=torch.uint8, device=dist2.device)
    check(lib.slamhip_ratio_test_f32(_ptr(dist2), n, float(ratio), _ptr(keep), _stream()),
          "slamhip_ratio_test_f32")
    return keep


def sort_by_distance_f32(dist: torch.Tensor) -> torch.Tensor:
    """Stable order of float distances (unmatched rows, -1.0, last): (n,) int32."""
    dist = to_device(dist, torch.float32).contiguous()
    n = dist.shape[0]
    order = torch.empty(n, dtype=torch.int32, device=dist.device)
    check(lib.slamhip_sort_by_distance_f32(_ptr(dist), 1, n, _ptr(order), _stream()),
          "slamhip_sort_by_distance_f32")
    return order


def ratio_test(dist2: torch.Tensor, ratio: float, out=None) -> torch.Tensor:
    dist2 = to_device(dist2, torch.int32)
    nq = dist2.shape[0]
    keep = out if out is not None else torch.empty(nq, dtype=torch.uint8, device=dist2.device)
    if keep.numel() != nq or keep.dtype != torch.uint8:
        raise ValueError("out must be a (nq,) uint8 device tensor")
    check(lib.slamhip_ratio_test(_ptr(dist2), nq, float(ratio), _ptr(keep), _stream()),
          "slamhip_ratio_test")
    return keep


def cross_check_match(q, t) -> Tuple[torch.Tensor, torch.Tensor]:
    """cv2 crossCheck=True result per query row: (train index or -1, distance)."""
    q, t = _desc_pair(q, t)
    nq, nt = q.shape[0], t.shape[0]
    t2q_idx, t2q_dist = hamming_knn2(t, q)
    mi = torch.empty(nq, dtype=torch.int32, device=q.device)
    md = torch.empty(nq, dtype=torch.int32, device=q.device)
    check(lib.slamhip_cross_check(_ptr(t2q_idx), _ptr(t2q_dist), nt, nq, _ptr(mi), _ptr(md),
                                  _stream()), "slamhip_cross_check")
    return mi, md


def sort_by_distance(dist: torch.Tensor, stride: int = 1, off=None,
                     max_n: Optional[int] = None) -> torch.Tensor:
    """Stable order by (distance, index).  `dist` is a flat int32 device tensor
    read with element stride `stride`; returns int32 order of length n."""
    dist = to_device(dist, torch.int32).reshape(-1)
    n = dist.numel() // stride
    order = torch.empty(n, dtype=torch.int32, device=dist.device)
    if off is None:
        check(lib.slamhip_sort_by_distance(_ptr(dist), stride, n, None, 1, n, _ptr(order),
                                           _stream()), "slamhip_sort_by_distance")
    else:
        if max_n is None:
            o = np.asarray(off.cpu() if isinstance(off, torch.Tensor) else off)
            max_n = int(np.diff(o).max()) if len(o) > 1 else 0
        off = to_device(off, torch.int32)
        check(lib.slamhip_sort_by_distance(_ptr(dist), stride, n, _ptr(off), off.numel() - 1,
                                           max_n, _ptr(order), _stream()),
              "slamhip_sort_by_distance")
    return order


def stereo_filter_triangulate(kp_left, kp_right, qidx, tidx, P_left, P_right,
                              pixel_thresh: float = 0.0, disparity_min: float = 0.0,
                              do_filter: bool = True, fp64: bool = False):
    """Row/disparity filter + DLT triangulation of matches (qidx[k], tidx[k]).
    fp64=False: float32 keypoints -> float32 points (the tracking call site);
    fp64=True : float64 keypoints -> float64 points (the loop-closure call site).
    Returns (keep uint8 or None, xyz (m,3))."""
    kdt = torch.float64 if fp64 else torch.float32
    kl = to_device(kp_left, kdt)
    kr = to_device(kp_right, kdt)
    qi = None if qidx is None else to_device(qidx, torch.int32)
    ti = None if tidx is None else to_device(tidx, torch.int32)
    m = qi.numel() if qi is not None else (ti.numel() if ti is not None else kl.shape[0])
    if (qi is None) != (ti is None):
        raise ValueError("qidx and tidx must both be given or both be None")
    if qi is None and kl.shape[0] != kr.shape[0]:
        raise ValueError("identity pairing needs equally long keypoint arrays")
    keep = torch.empty(m, dtype=torch.uint8, device=kl.device) if do_filter else None
    xyz = torch.empty((m, 3), dtype=kdt, device=kl.device)
    pl, _a = _lib.host_doubles(P_left, 12)
    pr, _b = _lib.host_doubles(P_right, 12)
    check(lib.slamhip_stereo_filter_triangulate(
        _ptr(kl), int(kl.shape[0]), _ptr(kr), int(kr.shape[0]), 1 if fp64 else 0, _ptr(qi), _ptr(ti), m, pl, pr,
        float(pixel_thresh), float(disparity_min), _ptr(keep), _ptr(xyz), 1 if fp64 else 0,
        _stream()), "slamhip_stereo_filter_triangulate")
    return keep, xyz


def _points(points):
    if isinstance(points, torch.Tensor):
        fp64 = points.dtype == torch.float64
    else:
        fp64 = np.asarray(points).dtype == np.float64
    return to_device(points, torch.float64 if fp64 else torch.float32), fp64


def ransac_score(points, meas_l, meas_r, Rts, valid, K, M_right, thresh: float) -> torch.Tensor:
    """Inlier count of each of the h hypotheses `Rts` (h,3,4) fp64."""
    pts, fp64 = _points(points)
    ml = to_device(meas_l, torch.float32)
    mr = to_device(meas_r, torch.float32)
    rts = to_device(Rts, torch.float64)
    va = None if valid is None else to_device(valid, torch.uint8)
    n, h = pts.shape[0], rts.shape[0]
    if ml.shape[0] != n or mr.shape[0] != n:
        raise ValueError("points and measurements differ in length")
    counts = torch.empty(h, dtype=torch.int32, device=pts.device)
    kp, _a = _lib.host_doubles(K, 9)
    mp, _b = _lib.host_doubles(M_right, 12)
    check(lib.slamhip_ransac_score(_ptr(pts), 1 if fp64 else 0, _ptr(ml), _ptr(mr), n, _ptr(rts),
                                   _ptr(va), h, kp, mp, float(thresh), _ptr(counts), _stream()),
          "slamhip_ransac_score")
    return counts


def ransac_mask(points, meas_l, meas_r, Rt, K, M_right, thresh: float) -> torch.Tensor:
    pts, fp64 = _points(points)
    ml = to_device(meas_l, torch.float32)
    mr = to_device(meas_r, torch.float32)
    rt = to_device(Rt, torch.float64)
    n = pts.shape[0]
    mask = torch.empty(n, dtype=torch.uint8, device=pts.device)
    kp, _a = _lib.host_doubles(K, 9)
    mp, _b = _lib.host_doubles(M_right, 12)
    check(lib.slamhip_ransac_mask(_ptr(pts), 1 if fp64 else 0, _ptr(ml), _ptr(mr), n, _ptr(rt),
                                  kp, mp, float(thresh), _ptr(mask), _stream()),
          "slamhip_ransac_mask")
    return mask


def ransac_replay(counts: torch.Tensor, n_points: int, p_success: float = 0.99999,
                  n_iter_max: int = 2500, early_inliers: Optional[int] = None) -> torch.Tensor:
    """Device-side adaptive-stop replay; returns 8 int32 (see slamhip.h)."""
    counts = to_device(counts, torch.int32)
    res = torch.empty(8, dtype=torch.int32, device=counts.device)
    check(lib.slamhip_ransac_replay(_ptr(counts), counts.numel(), int(n_points),
                                    float(p_success), int(n_iter_max),
                                    -1 if early_inliers is None else int(early_inliers),
                                    _ptr(res), _stream()), "slamhip_ransac_replay")
    return res


def project(points, P) -> torch.Tensor:
    """(n,2) fp64 pixels of `points` under the 3x4 matrix P."""
    pts, fp64 = _points(points)
    n = pts.shape[0]
    out = torch.empty((n, 2), dtype=torch.float64, device=pts.device)
    pp, _a = _lib.host_doubles(P, 12)
    check(lib.slamhip_project(_ptr(pts), 1 if fp64 else 0, n, pp, _ptr(out), _stream()),
          "slamhip_project")
    return out


def supporters_pair(meas_l, proj_l, meas_r, proj_r, thresh: float) -> torch.Tensor:
    ml = to_device(meas_l, torch.float64)
    pl = to_device(proj_l, torch.float64)
    mr = to_device(meas_r, torch.float64)
    pr = to_device(proj_r, torch.float64)
    n = ml.shape[0]
    if not (pl.shape[0] == mr.shape[0] == pr.shape[0] == n):
        raise ValueError("all four pixel arrays must have the same length")
    mask = torch.empty(n, dtype=torch.uint8, device=ml.device)
    check(lib.slamhip_supporters_pair(_ptr(ml), _ptr(pl), _ptr(mr), _ptr(pr), n, float(thresh),
                                      _ptr(mask), _stream()), "slamhip_supporters_pair")
    return mask


def compute_err(meas, proj) -> torch.Tensor:
    """Per-row Euclidean distance between two (n,2) pixel arrays, fp64 device tensor (n,)."""
    m = to_device(meas, torch.float64)
    p = to_device(proj, torch.float64)
    if m.dim() != 2 or m.shape[1] != 2 or m.shape != p.shape:
        raise ValueError("compute_err takes two (n, 2) arrays of the same length")
    err = torch.empty((m.shape[0],), dtype=torch.float64, device=m.device)
    check(lib.slamhip_compute_err(_ptr(m), _ptr(p), m.shape[0], _ptr(err), _stream()), "slamhip_compute_err")
    return err


def p3p_solve_indexed(points, pixels, idx4, K):
    """Minimal pose per row of idx4 (h,4) into points (n,3) / pixels (n,2).
    Returns (Rts (h,3,4) fp64, valid (h,) uint8) on the device."""
    pts, fp64 = _points(points)
    pix = to_device(pixels, torch.float32)
    ix = None if idx4 is None else to_device(idx4, torch.int32)
    h = ix.shape[0] if ix is not None else pts.shape[0] // 4
    Rts = torch.empty((h, 3, 4), dtype=torch.float64, device=pts.device)
    valid = torch.empty(h, dtype=torch.uint8, device=pts.device)
    kp, _a = _lib.host_doubles(K, 9)
    if pix.shape[0] != pts.shape[0]:
        raise ValueError("points and pixels differ in length")
    check(lib.slamhip_p3p_solve(_ptr(pts), 1 if fp64 else 0, _ptr(pix), int(pts.shape[0]), _ptr(ix), h, kp,
                                _ptr(Rts), _ptr(valid), _stream()), "slamhip_p3p_solve")
    return Rts, valid


def p3p_solve(obj4, img4, K):
    """Minimal pose for each of h gathered samples: obj4 (h,4,3), img4 (h,4,2)."""
    o = np.ascontiguousarray(obj4)
    i = np.ascontiguousarray(img4, dtype=np.float32)
    return p3p_solve_indexed(o.reshape(-1, 3), i.reshape(-1, 2), None, K)


def refine_pose_from(points, pixels, mask, Rt_init, K, max_iters: int = 50):
    """LM refinement of one or several starting poses on one data set.
    Rt_init (3,4) or (s,3,4).  Returns (Rts (s,3,4), info (s,2) = cost, iters)."""
    pts, fp64 = _points(points)
    pix = to_device(pixels, torch.float32)
    mk = None if mask is None else to_device(mask, torch.uint8)
    init = to_device(Rt_init, torch.float64).reshape(-1, 3, 4)
    s = init.shape[0]
    out = torch.empty_like(init)
    info = torch.empty((s, 2), dtype=torch.float64, device=pts.device)
    kp, _a = _lib.host_doubles(K, 9)
    check(lib.slamhip_refine_pose(_ptr(pts), 1 if fp64 else 0, _ptr(pix), _ptr(mk), pts.shape[0],
                                  None, s, kp, _ptr(init), int(max_iters), _ptr(out), _ptr(info),
                                  _stream()), "slamhip_refine_pose")
    return out, info


def refine_pose(obj, img, K, n_starts: int = 16):
    """Least-squares pose over all correspondences with no start given: minimal
    solutions of `n_starts` spread 4-samples seed the LM; the lowest final cost
    wins.  Returns a (3,4) float64 NumPy array or None."""
    obj = np.ascontiguousarray(obj)
    img = np.ascontiguousarray(img, dtype=np.float32)
    n = obj.shape[0]
    if n < 4:
        return None
    base = np.arange(n_starts)[:, None] * max(1, n // n_starts)
    idx4 = (base + np.array([0, n // 4, n // 2, (3 * n) // 4])[None, :]) % n
    idx4 = idx4[[len(set(r)) == 4 for r in idx4.tolist()]]
    if len(idx4) == 0:
        return None
    Rts, valid = p3p_solve_indexed(obj, img, idx4.astype(np.int32), K)
    keep = valid.bool()
    if not bool(keep.any().item()):
        return None
    out, info = refine_pose_from(obj, img, None, Rts[keep], K)
    best = int(torch.argmin(info[:, 0]).item())
    return out[best].cpu().numpy()
