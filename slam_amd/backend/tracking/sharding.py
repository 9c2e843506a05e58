"""Frame-range sharding of the front-end over the GPUs of one node and the
gather of the per-frame records.

Consecutive-frame pairs are independent (the pose of pair (k-1,k) needs only the
features of frames k-1 and k; fill_database.py:140-240 of the reference), so
rank r processes a contiguous range of pairs and re-computes the stereo stage
of one halo frame.  There is no collective on the data path.  At the end every
rank contributes ONE variable-length byte record - poses, links, inlier
descriptors, temporal matches, masks: everything TrackingDB.add_frame and the
bundle-adjustment back-end consume (run_slam.py:88-89) - to one all_gather
(RCCL over xGMI with the nccl backend; gloo on CPU for tests).  The record is
assembled on the device (one row-compaction kernel), so with nccl the
results never take a host hop before the collective.

Record layout (little endian): int64 header [n_frames, n_pairs, first_frame,
first_pair, desc_bytes], int64 stereo counts, int64 temporal counts, then the
fields below, each flat over all owned frames / pairs.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# (name, numpy dtype, columns) in record order
FRAME_FIELDS = (("left_idx", np.int32, 1), ("right_idx", np.int32, 1), ("dist", np.int32, 1),
                ("xyz", np.float32, 3), ("pix_left", np.float32, 2), ("pix_right", np.float32, 2),
                ("inlier_des_left", np.uint8, -1))          # -1: desc_bytes columns
PAIR_ROW_FIELDS = (("query_idx", np.int32, 1), ("train_idx", np.int32, 1), ("dist", np.int32, 1),
                   ("mask", np.uint8, 1))
PAIR_FIXED_FIELDS = (("Rt", np.float64, 12), ("Rt_ransac", np.float64, 12), ("info", np.int32, 8))


def pair_ranges(n_frames: int, world: int) -> List[Tuple[int, int]]:
    """Pairs p = (p, p+1), p in [0, n_frames-1), split into `world` contiguous
    ranges [lo, hi) of near-equal size (some may be empty)."""
    n_pairs = max(n_frames - 1, 0)
    base, extra = divmod(n_pairs, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def frames_for_rank(n_frames: int, world: int, rank: int):
    """((first_frame, last_frame_exclusive) to load, pair range owned, frame range
    whose stereo records are owned).  Frame f is owned by the rank owning pair f;
    the final frame by the rank owning the last pair.  With a single frame (no
    pairs) rank 0 owns it."""
    ranges = pair_ranges(n_frames, world)
    lo, hi = ranges[rank]
    if n_frames == 1:
        return (0, 1, (0, 0), (0, 1)) if rank == 0 else (0, 0, (0, 0), (0, 0))
    if hi == lo:
        return lo, lo, (lo, hi), (lo, lo)
    last_owner = max(r for r, (a, b) in enumerate(ranges) if b > a)
    own_hi = hi + 1 if rank == last_owner else hi
    return lo, hi + 1, (lo, hi), (lo, own_hi)


def pack_result(res, frame_range: Tuple[int, int], pair_range: Tuple[int, int], first_frame: int,
                first_pair: int) -> torch.Tensor:
    """Serialise the owned records of a FrontendResult into one uint8 tensor on
    the result's device (one kernel: slamhip_frontend_pack_record).  Ranges are local
    to the result's batch; first_frame / first_pair are the global indices of the
    first owned frame / pair.  The owned frames' and pairs' row counts are read back
    first (one small copy): they size the record."""
    import ctypes as C
    from ..._lib import lib, check
    t = res.t
    dev = t["st_count"].device
    flo, fhi = frame_range
    plo, phi = pair_range
    nf, npairs = fhi - flo, phi - plo
    counts = torch.cat([t["st_count"][flo:fhi], t["ransac_info"].reshape(-1, 8)[plo:phi, 5]]).cpu().numpy()
    offs = np.zeros(nf + npairs + 2, np.int64)
    np.cumsum(counts[:nf], out=offs[1:nf + 1])
    np.cumsum(counts[nf:], out=offs[nf + 2:])
    ns, nt = int(offs[nf]), int(offs[-1])
    d_offs = torch.from_numpy(offs).to(dev)
    nbytes = lib.slamhip_frontend_record_bytes(nf, npairs, ns, nt, res.inputs.desc_bytes)
    blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib.slamhip_frontend_pack_record(C.byref(res.args), flo, fhi, plo, phi, first_frame, first_pair,
                                           d_offs.data_ptr(), d_offs.data_ptr() + 8 * (nf + 1), ns, nt,
                                           blob.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream),
          "slamhip_frontend_pack_record")
    return blob


def pack_records(frames: Sequence[dict], pairs: Sequence[dict], first_frame: int, first_pair: int,
                 desc_bytes: int) -> np.ndarray:
    """Same record from per-frame / per-pair dicts (host); used by tests and by
    callers that hold results in that form."""
    head = np.array([len(frames), len(pairs), first_frame, first_pair, desc_bytes], dtype=np.int64)
    st_cnt = np.array([len(f["left_idx"]) for f in frames], dtype=np.int64)
    tm_cnt = np.array([len(p["query_idx"]) for p in pairs], dtype=np.int64)
    parts = [head.view(np.uint8), st_cnt.view(np.uint8), tm_cnt.view(np.uint8)]
    for name, dt, w in FRAME_FIELDS:
        cols = desc_bytes if w < 0 else w
        arrs = [np.asarray(f[name], dtype=dt).reshape(-1, cols) for f in frames]
        cat = np.concatenate(arrs) if arrs else np.zeros((0, cols), dt)
        parts.append(np.ascontiguousarray(cat).reshape(-1).view(np.uint8))
    for name, dt, w in PAIR_ROW_FIELDS:
        arrs = [np.asarray(p[name]).astype(dt).reshape(-1) for p in pairs]
        cat = np.concatenate(arrs) if arrs else np.zeros(0, dt)
        parts.append(np.ascontiguousarray(cat).view(np.uint8))
    for name, dt, w in PAIR_FIXED_FIELDS:
        if name == "info":
            rows = [np.array([p["best_index"], p["best_count"], p["iterations"],
                              p.get("n_iter_required", 0), int(p.get("domain_error", False)),
                              len(p["query_idx"]), int(np.sum(p["mask"])), 0], dtype=dt) for p in pairs]
        else:
            rows = [np.asarray(p[name], dtype=dt).reshape(-1) for p in pairs]
        cat = np.concatenate(rows) if rows else np.zeros(0, dt)
        parts.append(np.ascontiguousarray(cat).view(np.uint8))
    return np.concatenate(parts)


class SequenceResult:
    """A whole sequence reassembled from per-rank records.  frame(f) / pair(p) give views shaped
    like what the reference builds per frame; they index the owning rank's record where it lies
    (no copy of the large per-row fields is made to merge the records).  `f` holds the flat
    per-sequence arrays, concatenated on first use."""

    def __init__(self, n_frames: int, desc_bytes: int, parts):
        """parts: parsed records (first_frame, first_pair, desc_bytes, st_cnt, tm_cnt, fields),
        those with frames sorted by first_frame and covering [0, n_frames) without gaps."""
        self.n_frames, self.desc_bytes = n_frames, desc_bytes
        self.frame_cross_check = False        # set by run_sharded; selects the hand-over shape (db_adapter)
        self._parts = list(parts)
        self._by_frame = [p for p in parts if len(p[3])]
        self._by_pair = sorted((p for p in parts if len(p[4])), key=lambda p: p[1])
        self._frame_first = np.array([p[0] for p in self._by_frame], dtype=np.int64)
        self._pair_first = np.array([p[1] for p in self._by_pair], dtype=np.int64)
        self._frame_off = [np.concatenate([[0], np.cumsum(p[3])]).astype(np.int64) for p in self._by_frame]
        self._pair_off = [np.concatenate([[0], np.cumsum(p[4])]).astype(np.int64) for p in self._by_pair]
        z = np.zeros(0, np.int64)
        self.st_cnt = np.concatenate([p[3] for p in self._by_frame]) if self._by_frame else z
        self.tm_cnt = np.concatenate([p[4] for p in self._by_pair]) if self._by_pair else z
        self.st_off = np.concatenate([[0], np.cumsum(self.st_cnt)]).astype(np.int64)
        self.tm_off = np.concatenate([[0], np.cumsum(self.tm_cnt)]).astype(np.int64)
        self._Rt = (np.concatenate([p[5]["Rt"] for p in self._by_pair]).reshape(-1, 3, 4) if self._by_pair
                    else np.zeros((0, 3, 4)))
        self._flat = None

    def _locate(self, firsts, i):
        k = int(np.searchsorted(firsts, i, side="right")) - 1
        return k, i - int(firsts[k])

    def frame(self, f: int) -> dict:
        k, j = self._locate(self._frame_first, f)
        fields, off = self._by_frame[k][5], self._frame_off[k]
        sl = slice(int(off[j]), int(off[j + 1]))
        out = {name: fields[name][sl] for name, _, _ in FRAME_FIELDS}
        pl, pr = out["pix_left"], out["pix_right"]
        out["links"] = np.column_stack([pl[:, 0].astype(np.float64), pr[:, 0].astype(np.float64),
                                        (pl[:, 1].astype(np.float64) + pr[:, 1].astype(np.float64)) / 2])
        return out

    def pair(self, p: int) -> dict:
        k, j = self._locate(self._pair_first, p)
        fields, off = self._by_pair[k][5], self._pair_off[k]
        sl = slice(int(off[j]), int(off[j + 1]))
        info = fields["info"][j]
        return {"query_idx": fields["tm_query_idx"][sl], "train_idx": fields["tm_train_idx"][sl],
                "dist": fields["tm_dist"][sl], "mask": fields["tm_mask"][sl].astype(bool),
                "Rt": fields["Rt"][j].reshape(3, 4), "Rt_ransac": fields["Rt_ransac"][j].reshape(3, 4),
                "best_index": int(info[0]), "best_count": int(info[1]), "iterations": int(info[2]),
                "n_iter_required": int(info[3]), "domain_error": bool(info[4]),
                "n_matches": int(off[j + 1] - off[j])}

    @property
    def f(self) -> Dict[str, np.ndarray]:
        """Flat per-sequence arrays (every field over all frames / pairs), built on first use."""
        if self._flat is None:
            D = self.desc_bytes
            out = {}
            for name, dt, w in FRAME_FIELDS:
                cols = D if w < 0 else w
                arrs = [p[5][name] for p in self._by_frame]
                out[name] = np.concatenate(arrs) if arrs else np.zeros((0, cols) if cols > 1 or w < 0 else 0, dt)
            for name, dt, w in PAIR_ROW_FIELDS:
                arrs = [p[5]["tm_" + name] for p in self._by_pair]
                out["tm_" + name] = np.concatenate(arrs) if arrs else np.zeros(0, dt)
            for name, dt, w in PAIR_FIXED_FIELDS:
                arrs = [p[5][name] for p in self._by_pair]
                out[name] = np.concatenate(arrs) if arrs else np.zeros((0, w), dt)
            out["Rt"] = out["Rt"].reshape(-1, 3, 4)
            out["Rt_ransac"] = out["Rt_ransac"].reshape(-1, 3, 4)
            self._flat = out
        return self._flat

    @property
    def nbytes(self) -> int:
        """Bytes of all fields of all records."""
        return int(sum(a.nbytes for p in self._parts for a in p[5].values()))

    @property
    def relative_Rts(self) -> np.ndarray:
        """(n_frames-1,3,4): the array the reference pickles as `relative_Rts`."""
        return self._Rt

    @property
    def frames(self):
        return [self.frame(f) for f in range(self.n_frames)]

    @property
    def pairs(self):
        return [self.pair(p) for p in range(max(self.n_frames - 1, 0))]


def _parse(blob: np.ndarray):
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    pos = 0

    def take(dt, count):
        nonlocal pos
        nbytes = np.dtype(dt).itemsize * count
        if pos + nbytes > blob.size:
            raise ValueError("truncated record")
        out = blob[pos:pos + nbytes].view(dt)
        pos += nbytes
        return out

    nf, npairs, first_frame, first_pair, D = (int(v) for v in take(np.int64, 5))
    st_cnt = take(np.int64, nf).copy()
    tm_cnt = take(np.int64, npairs).copy()
    ns, nt = int(st_cnt.sum()), int(tm_cnt.sum())
    fields = {}
    for name, dt, w in FRAME_FIELDS:
        cols = D if w < 0 else w
        a = take(dt, ns * cols)
        fields[name] = a.reshape(ns, cols) if cols > 1 or w < 0 else a
    for name, dt, w in PAIR_ROW_FIELDS:
        fields["tm_" + name] = take(dt, nt)
    for name, dt, w in PAIR_FIXED_FIELDS:
        fields[name] = take(dt, npairs * w).reshape(npairs, w)
    return first_frame, first_pair, D, st_cnt, tm_cnt, fields


def merge_records(blobs: Sequence[np.ndarray], n_frames: int) -> SequenceResult:
    """Reassemble the whole sequence from the per-rank records (any order)."""
    parsed = [_parse(b) for b in blobs if b is not None and len(b) > 0]
    parsed = [p for p in parsed if len(p[3]) or len(p[4])]
    by_frame = sorted((p for p in parsed if len(p[3])), key=lambda p: p[0])
    by_pair = sorted((p for p in parsed if len(p[4])), key=lambda p: p[1])
    nf = sum(len(p[3]) for p in by_frame)
    npairs = sum(len(p[4]) for p in by_pair)
    exp_f = 0
    for p in by_frame:
        if p[0] != exp_f:
            raise RuntimeError(f"frame records missing or duplicated at frame {exp_f}")
        exp_f += len(p[3])
    exp_p = 0
    for p in by_pair:
        if p[1] != exp_p:
            raise RuntimeError(f"pair records missing or duplicated at pair {exp_p}")
        exp_p += len(p[4])
    if nf != n_frames or npairs != max(n_frames - 1, 0):
        raise RuntimeError(f"gathered {nf} frames / {npairs} pairs, expected {n_frames} / {n_frames - 1}")
    D = parsed[0][2] if parsed else 0
    return SequenceResult(n_frames, D, by_frame + [p for p in by_pair if not len(p[3])])


def records_to_host(blobs: Sequence[torch.Tensor], reuse_buffer: bool = False) -> List[np.ndarray]:
    """Device records -> NumPy arrays through one page-locked buffer (several times the rate of
    pageable copies), one synchronisation for all of them.  The arrays are views of that buffer
    and keep it alive.  `reuse_buffer`: take the module's buffer, kept and grown on demand
    (allocating page-locked memory costs more than the copy), instead of a new one -- the arrays
    of the previous call with this option are overwritten then."""
    global _host_stage
    sizes = [int(b.numel()) for b in blobs]
    offs = np.concatenate([[0], np.cumsum([(n + 63) // 64 * 64 for n in sizes])]).astype(np.int64)
    total = max(int(offs[-1]), 1)
    pin = torch.cuda.is_available()
    if not reuse_buffer:
        stage = torch.empty(total, dtype=torch.uint8, pin_memory=pin)
    else:
        if _host_stage is None or _host_stage.numel() < total:
            _host_stage = torch.empty(int(total * 1.25), dtype=torch.uint8, pin_memory=pin)
        stage = _host_stage
    out = []
    for b, o, n in zip(blobs, offs[:-1], sizes):
        dst = stage[int(o):int(o) + n]
        dst.copy_(b.reshape(-1), non_blocking=True)
        out.append(dst.numpy())
    if torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()
    return out


_host_stage = None


def all_gather_records(blob, group=None) -> List[np.ndarray]:
    """One all_gather of variable-length byte records (padded to the longest).
    `blob`: uint8 torch tensor (device tensor for nccl, CPU for gloo) or NumPy
    array.  Returns every rank's record as NumPy arrays."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    if not isinstance(blob, torch.Tensor):
        blob = torch.from_numpy(np.ascontiguousarray(blob))
    blob = blob.to(device)
    n = torch.tensor([blob.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    buf[:blob.numel()] = blob
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return [o[:s].cpu().numpy() for o, s in zip(out, sizes)]


def gather_records(blob: torch.Tensor, dst: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """Gather of the variable-length records onto rank `dst` only (the host of
    the back-end): sizes by a tiny all_gather, payload by one gather of buffers
    padded to the longest record.  Returns the list of records (uint8 tensors on
    the collective's device, trimmed) on `dst`, None elsewhere."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    blob = blob.to(device)
    n = torch.tensor([blob.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    buf[:blob.numel()] = blob
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst, group=group)
    if rank != dst:
        return None
    return [o[:s] for o, s in zip(out, sizes)]


def run_sharded(frontend, frames: Sequence, rank: int, world: int, group=None,
                samples: Optional[np.ndarray] = None, dst: int = 0,
                force_collective: bool = False, reuse_host_buffer: bool = False) -> Optional[SequenceResult]:
    """Process this rank's share of `frames` on its GPU and gather the records
    on rank `dst`, which returns the merged result (other ranks return None).
    With one rank the record is parsed directly; `force_collective` sends it
    through the gather all the same (needs an initialised process group).
    `reuse_host_buffer`: see records_to_host (the previous result's arrays are overwritten)."""
    n_frames = len(frames)
    lo, hi, (plo, phi), (flo, fhi) = frames_for_rank(n_frames, world, rank)
    desc_bytes = int(np.asarray(frames[0][2]).shape[1])
    if hi > lo:
        s = None if samples is None else samples[plo:phi]
        res = frontend.run(list(frames[lo:hi]), samples=s, first_pair=plo)
        blob = pack_result(res, (flo - lo, fhi - lo), (plo - lo, phi - lo), flo, plo)
    else:
        blob = torch.from_numpy(pack_records([], [], 0, 0, desc_bytes))
    if world == 1 and not force_collective:
        merged = merge_records(records_to_host([blob], reuse_host_buffer), n_frames)
    else:
        got = gather_records(blob, dst, group)
        if got is None:
            return None
        merged = merge_records(records_to_host(got, reuse_host_buffer), n_frames)
    merged.frame_cross_check = bool(getattr(frontend, "frame_cross_check", False))   # for the hand-over
    return merged
