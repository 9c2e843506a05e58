"""CPU tests of the multi-GPU path: range partitioning, record (de)serialisation
and the world_size-2 gather over gloo, fed by the CPU restatement of the
pipeline instead of the GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from slam_amd.backend.tracking import sharding


def test_pair_ranges_cover_everything_once():
    for n_frames in (1, 2, 3, 7, 500, 2761):
        for world in (1, 2, 4, 8):
            ranges = sharding.pair_ranges(n_frames, world)
            covered = [p for lo, hi in ranges for p in range(lo, hi)]
            assert covered == list(range(max(n_frames - 1, 0)))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
            owned = []
            for r in range(world):
                lo, hi, (plo, phi), (flo, fhi) = sharding.frames_for_rank(n_frames, world, r)
                owned += list(range(flo, fhi))
                if phi > plo:
                    assert lo == plo and hi == phi + 1      # one halo frame
            assert sorted(owned) == list(range(n_frames)), (n_frames, world, owned)


def _fake_records(rng, n_frames, desc_bytes=61):
    frames, pairs = [], []
    for f in range(n_frames):
        m = int(rng.integers(0, 40))
        frames.append({"left_idx": rng.integers(0, 99, m).astype(np.int32),
                       "right_idx": rng.integers(0, 99, m).astype(np.int32),
                       "dist": rng.integers(0, 488, m).astype(np.int32),
                       "xyz": rng.normal(size=(m, 3)).astype(np.float32),
                       "pix_left": rng.normal(size=(m, 2)).astype(np.float32),
                       "pix_right": rng.normal(size=(m, 2)).astype(np.float32),
                       "inlier_des_left": rng.integers(0, 256, (m, desc_bytes), dtype=np.uint8)})
    for p in range(n_frames - 1):
        n = len(frames[p]["left_idx"])
        pairs.append({"query_idx": rng.permutation(n).astype(np.int32),
                      "train_idx": rng.integers(0, 50, n).astype(np.int32),
                      "dist": rng.integers(0, 488, n).astype(np.int32),
                      "mask": rng.random(n) < 0.6, "Rt": rng.normal(size=(3, 4)),
                      "Rt_ransac": rng.normal(size=(3, 4)), "best_index": int(rng.integers(0, 99)),
                      "best_count": int(rng.integers(0, 99)), "iterations": int(rng.integers(1, 99)),
                      "n_matches": n})
    return frames, pairs


def _same(a, b):
    assert a.keys() >= {"left_idx"} or a.keys() >= {"query_idx"}
    for k, v in b.items():
        if k == "n_matches" and k not in a:
            continue
        if isinstance(v, np.ndarray):
            np.testing.assert_array_equal(a[k], v, err_msg=k)
        else:
            assert a[k] == v, k


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(0)
    frames, pairs = _fake_records(rng, 6)
    a = sharding.pack_records(frames[0:3], pairs[0:3], 0, 0, 61)
    b = sharding.pack_records(frames[3:6], pairs[3:5], 3, 3, 61)
    empty = sharding.pack_records([], [], 0, 0, 61)
    merged = sharding.merge_records([b, empty, a], 6)       # any order, empty ranks allowed
    for got, ref in zip(merged.frames, frames):
        _same(got, ref)
    for got, ref in zip(merged.pairs, pairs):
        _same(got, ref)
    assert merged.relative_Rts.shape == (5, 3, 4)
    with pytest.raises(RuntimeError):
        sharding.merge_records([a], 6)                       # a rank's record is missing
    with pytest.raises(RuntimeError):
        sharding.merge_records([a, a, b], 6)                 # duplicated


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)                 # same data on both ranks
        frames, pairs = _fake_records(rng, n_frames)
        lo, hi, (plo, phi), (flo, fhi) = sharding.frames_for_rank(n_frames, world, rank)
        blob = sharding.pack_records(frames[flo:fhi], pairs[plo:phi], flo, plo, 61)
        merged = sharding.merge_records(sharding.all_gather_records(blob), n_frames)
        ok = len(merged.frames) == n_frames and len(merged.pairs) == n_frames - 1
        for a, b in zip(merged.frames, frames):
            _same(a, b)
        for a, b in zip(merged.pairs, pairs):
            _same(a, b)
        ok = ok and np.array_equal(merged.relative_Rts, np.stack([p["Rt"] for p in pairs]))
        # gather onto rank 0 only: same records, nothing returned elsewhere
        got = sharding.gather_records(torch.from_numpy(blob), dst=0)
        if rank == 0:
            m2 = sharding.merge_records([g.numpy() for g in got], n_frames)
            ok = ok and np.array_equal(m2.relative_Rts, merged.relative_Rts)
        else:
            ok = ok and got is None
        q.put((rank, bool(ok)))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [2, 9])
def test_gather_world_size_2_gloo(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)], results
