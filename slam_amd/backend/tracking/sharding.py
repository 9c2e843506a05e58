"""Frame-range sharding of the front-end over the GPUs of one node and the
gather of the per-frame records.

Consecutive-frame pairs are independent (the pose of pair (k-1,k) needs only the
features of frames k-1 and k; fill_database.py:140-240 of the reference), so
rank r processes a contiguous range of pairs and re-computes the stereo stage
of one halo frame.  There is no collective on the data path.  At the end every
rank contributes one variable-length byte record (poses, links, inlier
descriptors, temporal matches, masks - everything TrackingDB.add_frame and the
bundle-adjustment back-end consume, run_slam.py:88-89) to ONE all_gather
(RCCL over xGMI with the nccl backend; gloo on CPU for tests).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

FRAME_FIELDS = (("left_idx", np.int32, 1), ("right_idx", np.int32, 1), ("dist", np.int32, 1),
                ("xyz", np.float32, 3), ("pix_left", np.float32, 2), ("pix_right", np.float32, 2))
PAIR_FIELDS = (("query_idx", np.int32, 1), ("train_idx", np.int32, 1), ("dist", np.int32, 1),
               ("mask", np.uint8, 1))


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


def frames_for_rank(n_frames: int, world: int, rank: int) -> Tuple[int, int, Tuple[int, int], Tuple[int, int]]:
    """(first_frame, last_frame_exclusive) this rank must load, the pair range it
    owns, and the frame range whose stereo records it owns.  Frame f is owned by
    the rank owning pair f; the final frame by the rank owning the last pair.
    With a single frame (no pairs) rank 0 owns it."""
    lo, hi = pair_ranges(n_frames, world)[rank]
    if n_frames == 1:
        return (0, 1, (0, 0), (0, 1)) if rank == 0 else (0, 0, (0, 0), (0, 0))
    if hi == lo:
        return lo, lo, (lo, hi), (lo, lo)
    last_owner = max(r for r, (a, b) in enumerate(pair_ranges(n_frames, world)) if b > a)
    own_hi = hi + 1 if rank == last_owner else hi
    return lo, hi + 1, (lo, hi), (lo, own_hi)


def pack_records(frames: Sequence[dict], pairs: Sequence[dict], first_frame: int, first_pair: int,
                 desc_bytes: int) -> np.ndarray:
    """Serialise owned per-frame / per-pair records into one uint8 array."""
    head = [len(frames), len(pairs), first_frame, first_pair, desc_bytes]
    head += [len(f["left_idx"]) for f in frames]
    head += [len(p["query_idx"]) for p in pairs]
    parts = [np.asarray(head, dtype=np.int64).view(np.uint8)]
    for f in frames:
        for name, dt, w in FRAME_FIELDS:
            parts.append(np.ascontiguousarray(f[name], dtype=dt).reshape(-1).view(np.uint8))
        parts.append(np.ascontiguousarray(f["inlier_des_left"], dtype=np.uint8).reshape(-1))
    for p in pairs:
        for name, dt, w in PAIR_FIELDS:
            parts.append(np.ascontiguousarray(p[name], dtype=dt).reshape(-1).view(np.uint8))
        parts.append(np.ascontiguousarray(p["Rt"], dtype=np.float64).reshape(-1).view(np.uint8))
        parts.append(np.ascontiguousarray(p["Rt_ransac"], dtype=np.float64).reshape(-1).view(np.uint8))
        parts.append(np.asarray([p["best_index"], p["best_count"], p["iterations"]],
                                dtype=np.int64).view(np.uint8))
    return np.concatenate(parts)


def unpack_records(blob: np.ndarray):
    """Inverse of pack_records -> (first_frame, frames, first_pair, pairs)."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    pos = 0

    def take(dt, count):
        nonlocal pos
        nbytes = np.dtype(dt).itemsize * count
        out = blob[pos:pos + nbytes].copy().view(dt)
        pos += nbytes
        return out

    nf, npairs, first_frame, first_pair, desc_bytes = (int(v) for v in take(np.int64, 5))
    fcounts = take(np.int64, nf)
    pcounts = take(np.int64, npairs)
    frames, pairs = [], []
    for m in fcounts:
        m = int(m)
        f = {}
        for name, dt, w in FRAME_FIELDS:
            a = take(dt, m * w)
            f[name] = a.reshape(m, w) if w > 1 else a
        f["inlier_des_left"] = take(np.uint8, m * desc_bytes).reshape(m, desc_bytes)
        frames.append(f)
    for n in pcounts:
        n = int(n)
        p = {}
        for name, dt, w in PAIR_FIELDS:
            p[name] = take(dt, n)
        p["mask"] = p["mask"].astype(bool)
        p["Rt"] = take(np.float64, 12).reshape(3, 4)
        p["Rt_ransac"] = take(np.float64, 12).reshape(3, 4)
        bi, bc, it = (int(v) for v in take(np.int64, 3))
        p.update(best_index=bi, best_count=bc, iterations=it, n_matches=n)
        pairs.append(p)
    return first_frame, frames, first_pair, pairs


def all_gather_records(blob: np.ndarray, group=None, device=None) -> List[np.ndarray]:
    """One all_gather of variable-length byte records (padded to the longest).
    Needs an initialised torch.distributed process group; `device` = 'cuda' for
    the nccl (RCCL) backend, 'cpu' for gloo."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    n = torch.tensor([blob.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    if blob.size:
        buf[:blob.size] = torch.from_numpy(blob).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return [o[:s].cpu().numpy() for o, s in zip(out, sizes)]


def merge_records(blobs: Sequence[np.ndarray], n_frames: int) -> Dict[str, list]:
    """Reassemble the whole sequence: {'frames': [n_frames dicts], 'pairs':
    [n_frames-1 dicts], 'relative_Rts': (n_frames-1,3,4)}."""
    frames: List = [None] * n_frames
    pairs: List = [None] * max(n_frames - 1, 0)
    for blob in blobs:
        if blob.size == 0:
            continue
        ff, fr, fp, pr = unpack_records(blob)
        for i, f in enumerate(fr):
            frames[ff + i] = f
        for i, p in enumerate(pr):
            pairs[fp + i] = p
    missing = [i for i, f in enumerate(frames) if f is None] + [i for i, p in enumerate(pairs) if p is None]
    if missing:
        raise RuntimeError(f"records missing after gather: {missing[:8]}")
    rel = np.stack([p["Rt"] for p in pairs]) if pairs else np.zeros((0, 3, 4))
    return {"frames": frames, "pairs": pairs, "relative_Rts": rel}


def run_sharded(frontend, frames: Sequence, rank: int, world: int, group=None, samples=None):
    """Process this rank's share of `frames` on its GPU and gather everything.
    Every rank returns the merged result (all_gather)."""
    n_frames = len(frames)
    lo, hi, (plo, phi), (flo, fhi) = frames_for_rank(n_frames, world, rank)
    desc_bytes = int(np.asarray(frames[0][2]).shape[1])
    own_frames, own_pairs = [], []
    if hi > lo:
        s = None if samples is None else samples[plo:phi]
        res = frontend.run(list(frames[lo:hi]), samples=s)
        own_frames = [res.frame(f - lo) for f in range(flo, fhi)]
        own_pairs = [res.pair(p - lo) for p in range(plo, phi)]
    blob = pack_records(own_frames, own_pairs, flo, plo, desc_bytes)
    if world == 1:
        return merge_records([blob], n_frames)
    return merge_records(all_gather_records(blob, group), n_frames)
