"""GPU parity test of the batched front-end against the frame-at-a-time CPU
restatement of fill_database.py's data flow."""
import os

import numpy as np
import pytest
import torch

from oracle import pipeline_oracle as po
from oracle import slam_oracle as so

pytestmark = pytest.mark.gpu


def _sequence(n_frames, n_features, seed):
    from slam_amd import CameraModel
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    seq = StereoSequence(cam, n_frames=n_frames, n_features=n_features, seed=seed)
    frames = []
    for k in range(n_frames):
        f = seq.frame(k)
        frames.append((f.kp_left, f.kp_right, f.des_left, f.des_right))
    return cam, seq, frames


def test_frontend_matches_frame_at_a_time_oracle():
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    n_iter = 512
    cam, seq, frames = _sequence(6, 700, seed=11)
    # ragged: drop features of one frame, empty right image in another
    frames[2] = (frames[2][0][:300], frames[2][1][:450], frames[2][2][:300], frames[2][3][:450])
    rng = np.random.default_rng(5)
    stereo_o = [po.stereo_stage(cam, f) for f in frames]
    samples = np.zeros((len(frames) - 1, n_iter, 4), np.int32)
    for p in range(len(frames) - 1):
        N = len(stereo_o[p]["left_idx"])
        if N >= 4:
            samples[p] = np.stack([rng.choice(N, 4, replace=False) for _ in range(n_iter)])
    fe = TrackingFrontend(cam, n_iter_max=n_iter)
    res = fe.run(frames, samples=samples)
    stereo_ref, pairs_ref = po.run(cam, frames, samples=samples, n_iter_max=n_iter)

    for f in range(len(frames)):
        got, ref = res.frame(f), stereo_ref[f]
        np.testing.assert_array_equal(got["left_idx"], ref["left_idx"], err_msg=f"frame {f}")
        np.testing.assert_array_equal(got["right_idx"], ref["right_idx"])
        np.testing.assert_array_equal(got["dist"], ref["dist"])
        np.testing.assert_array_equal(got["inlier_des_left"], ref["inlier_des_left"])
        np.testing.assert_array_equal(got["pix_left"], ref["pix_left"])
        np.testing.assert_array_equal(got["pix_right"], ref["pix_right"])
        rel = np.abs(got["xyz"] - ref["xyz"]) / np.abs(ref["xyz"]).max(axis=1, keepdims=True)
        assert rel.max() < 1e-4
        assert len(got["left_idx"]) > 100

    same_best = 0
    for p in range(len(frames) - 1):
        got, ref = res.pair(p), pairs_ref[p]
        assert got["n_matches"] == ref["n_matches"]
        np.testing.assert_array_equal(got["query_idx"], ref["query_idx"], err_msg=f"pair {p}")
        np.testing.assert_array_equal(got["train_idx"], ref["train_idx"])
        np.testing.assert_array_equal(got["dist"], ref["dist"])
        # the device mask is exactly the reference formula applied to the device's own winner
        X, pL, pR = ref["X_corr"], ref["pix_L"], ref["pix_R"]
        eL, eR = so.hypothesis_errors(cam.K, cam.M_right, got["Rt_ransac"], X, pL, pR)
        refmask = (eL <= 4.0) & (eR <= 4.0)
        bad = np.flatnonzero(refmask != got["mask"])
        assert all(min(abs(eL[i] - 4.0), abs(eR[i] - 4.0)) < 1e-6 for i in bad)
        assert got["best_count"] == got["mask"].sum()
        # same samples through the CPU minimal solver: same winner (root-finder rounding aside)
        if got["best_index"] == ref["best_index"]:
            same_best += 1
            assert np.abs(got["Rt_ransac"] - ref["Rt_ransac"]).max() < 1e-6
            assert (got["mask"] != ref["mask"]).sum() <= 1
            assert got["iterations"] == ref["iterations"]
            # refinement = least-squares minimum over the device's own inlier set
            gold, _ = so.refine_pose(X, pL, cam.K, got["Rt_ransac"], got["mask"])
            assert np.abs(got["Rt"] - gold).max() < 1e-7, np.abs(got["Rt"] - gold).max()
        truth = seq.relative_pose(p + 1)
        if p in (1, 2):      # pairs touching the truncated frame: few true matches, no accuracy claim
            continue
        # triangulated landmarks carry metres of depth noise at range, so the pose is
        # only centimetre-accurate against ground truth
        assert np.abs(got["Rt"][:, :3] - truth[:, :3]).max() < 3e-3
        assert np.abs(got["Rt"][:, 3] - truth[:, 3]).max() < 0.1, (p, got["Rt"][:, 3], truth[:, 3])
        assert got["mask"].mean() > 0.25
    assert same_best >= len(frames) - 2
    assert res.relative_Rts().shape == (len(frames) - 1, 3, 4)


def test_frontend_device_sampler_and_edge_cases():
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    cam, seq, frames = _sequence(4, 500, seed=3)
    fe = TrackingFrontend(cam, seed=7)
    a = fe.run(frames)
    b = fe.run(frames)
    for p in range(3):
        pa, pb = a.pair(p), b.pair(p)
        np.testing.assert_array_equal(pa["mask"], pb["mask"])          # seeded sampler is deterministic
        np.testing.assert_array_equal(pa["Rt"], pb["Rt"])
        assert np.abs(pa["Rt"] - seq.relative_pose(p + 1)).max() < 0.1
        assert pa["iterations"] >= 1 and pa["best_count"] == pa["mask"].sum()
    # a frame with no right features: no stereo inliers, its pairs have no matches
    empty = (frames[1][0], np.zeros((0, 2), np.float32), frames[1][2], np.zeros((0, 61), np.uint8))
    r = fe.run([frames[0], empty, frames[2]])
    assert len(r.frame(1)["left_idx"]) == 0
    assert r.pair(0)["n_matches"] == 0 and r.pair(1)["n_matches"] == 0
    assert r.pair(0)["best_index"] == -1 and not r.pair(0)["mask"].any()
    np.testing.assert_array_equal(r.pair(0)["Rt"], np.hstack([np.eye(3), np.zeros((3, 1))]))
    # single frame: stereo stage only
    r1 = fe.run(frames[:1])
    assert len(r1.frame(0)["left_idx"]) > 50
    # no right features at all; frames of one to five features (fewer rows than one 16-byte load
    # of the in-place reader per array is the library's business, not the caller's)
    none_right = [(f[0], np.zeros((0, 2), np.float32), f[2], np.zeros((0, 61), np.uint8)) for f in frames[:3]]
    r = fe.run(none_right)
    assert not r.to_host()["st_count"][:3].any() and r.pair(1)["n_matches"] == 0
    tiny = [tuple(a[:k] for a in frames[i]) for i, k in enumerate((1, 2, 3, 5))]
    r = fe.run(tiny)
    h = r.to_host()
    assert (h["st_count"][:4] <= np.array([1, 2, 3, 5])).all()
    for p in range(3):
        assert r.pair(p)["n_matches"] <= h["st_count"][p] and r.pair(p)["best_index"] == -1   # < 4 points: not attempted


def _shard_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share the one GPU
    try:
        from slam_amd.backend.tracking import sharding
        from slam_amd.backend.tracking.frontend import TrackingFrontend
        torch.cuda.set_device(0)
        cam, seq, frames = _sequence(7, 400, seed=21)
        n_iter = 256
        rng = np.random.default_rng(9)
        samples = rng.integers(0, 50, (len(frames) - 1, n_iter, 4)).astype(np.int32)
        samples[..., 1] += 50; samples[..., 2] += 100; samples[..., 3] += 150   # distinct rows
        fe = TrackingFrontend(cam, n_iter_max=n_iter)
        merged = sharding.run_sharded(fe, frames, rank, world, samples=samples)
        if rank != 0:
            q.put((rank, merged is None))
            return
        whole = sharding.run_sharded(fe, frames, 0, 1, samples=samples)
        ok = True
        direct = fe.run(frames, samples=samples)          # no packing at all
        for f, (a, b) in enumerate(zip(merged.frames, whole.frames)):
            c = direct.frame(f)
            for k in b:
                ok = ok and np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k])
        for p, (a, b) in enumerate(zip(merged.pairs, whole.pairs)):
            c = direct.pair(p)
            for k in ("query_idx", "train_idx", "dist", "mask", "Rt", "Rt_ransac"):
                ok = ok and np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k])
            ok = ok and a["best_index"] == b["best_index"] == c["best_index"]
            ok = ok and a["iterations"] == b["iterations"] == c["iterations"]
        ok = ok and merged.relative_Rts.shape == (6, 3, 4)
        ok = ok and np.array_equal(merged.relative_Rts, direct.relative_Rts())
        q.put((rank, bool(ok)))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_sharded_run_two_ranks_equals_single_rank():
    """Frame-range sharding + gather (gloo here, RCCL on a multi-GPU node) gives
    exactly the single-process result."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)], results


@pytest.mark.parametrize("mode", [[], ["--online"]])
def test_run_sequence_tool_from_feature_cache(tmp_path, mode):
    """tools/run_sequence.py end to end on a synthesised cache, batched and one frame at a time
    (--online): the pickled poses and the saved records equal what the front-end returns for the
    same frames."""
    import pickle
    import subprocess
    import sys
    from slam_amd import CameraModel
    from slam_amd.backend.tracking import db_adapter, feature_cache
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    feats, out = str(tmp_path / "features"), str(tmp_path / "out")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "run_sequence.py"), "--features", feats,
                        "--out", out, "--synthesize", "6"] + mode, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "5 relative poses" in r.stdout
    with open(os.path.join(out, "relative_Rts"), "rb") as fh:
        rel = pickle.load(fh)
    assert rel.shape == (5, 3, 4) and rel.dtype == np.float64
    cache = feature_cache.FeatureCache(feats)
    res = TrackingFrontend(CameraModel.kitti_like(), seed=0).run(cache[0:6])
    np.testing.assert_array_equal(rel, res.relative_Rts())
    back = db_adapter.SavedRecords(os.path.join(out, "frontend_records.npz"))
    for f in range(6):
        np.testing.assert_array_equal(back.frame(f)["pix_left"], res.frame(f)["pix_left"])
    for p in range(5):
        np.testing.assert_array_equal(back.pair(p)["mask"], res.pair(p)["mask"])


@pytest.mark.parametrize("stereo_cc,frame_cc", [(True, False), (False, True), (True, True)])
def test_batched_cross_check_equals_frame_at_a_time(stereo_cc, frame_cc):
    """crossCheck in the stereo and / or the consecutive-frame matcher
    (fill_database.py:69-70): match lists, their order, the inlier sets and
    everything gathered from them equal the frame-at-a-time restatement, including a
    frame with few features, one with an empty right image and duplicated rows that
    force ties."""
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    n_iter = 256
    cam, seq, frames = _sequence(6, 500, seed=23)
    frames[2] = (frames[2][0][:200], frames[2][1][:320], frames[2][2][:200], frames[2][3][:320])
    frames[4] = (frames[4][0], frames[4][1][:0], frames[4][2], frames[4][3][:0])
    f1 = [a.copy() for a in frames[1]]
    f1[2][10:20] = f1[2][30:40]          # equal left rows: several queries want the same train row
    f1[3][50:55] = f1[3][60:65]          # equal right rows: ties between train rows
    frames[1] = tuple(f1)
    stereo_o = [po.stereo_stage(cam, f, cross_check=stereo_cc) for f in frames]
    rng = np.random.default_rng(9)
    samples = np.zeros((len(frames) - 1, n_iter, 4), np.int32)
    pairs_n = []
    for p in range(len(frames) - 1):
        t = po.temporal_stage(cam, stereo_o[p], stereo_o[p + 1], n_iter_max=1, cross_check=frame_cc,
                              samples=np.zeros((1, 4), np.int32) + np.arange(4))
        pairs_n.append(t["n_matches"])
        if t["n_matches"] >= 4:
            samples[p] = np.stack([rng.choice(t["n_matches"], 4, replace=False) for _ in range(n_iter)])
    fe = TrackingFrontend(cam, n_iter_max=n_iter, stereo_cross_check=stereo_cc, frame_cross_check=frame_cc)
    res = fe.run(frames, samples=samples)
    plain = TrackingFrontend(cam, n_iter_max=n_iter).run(frames, samples=samples)
    fewer = 0
    for f in range(len(frames)):
        got, ref = res.frame(f), stereo_o[f]
        for k in ("left_idx", "right_idx", "dist", "inlier_des_left", "pix_left", "pix_right"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"frame {f} {k}")
        fewer += len(plain.frame(f)["left_idx"]) - len(got["left_idx"])
    assert (fewer > 0) == stereo_cc            # the option really removes matches
    assert len(res.frame(4)["left_idx"]) == 0
    for p in range(len(frames) - 1):
        got = res.pair(p)
        ref = po.temporal_stage(cam, stereo_o[p], stereo_o[p + 1], n_iter_max=1, cross_check=frame_cc,
                                samples=np.zeros((1, 4), np.int32) + np.arange(4))
        assert got["n_matches"] == ref["n_matches"] == pairs_n[p], p
        for k in ("query_idx", "train_idx", "dist"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"pair {p} {k}")
        if frame_cc:
            assert len(set(got["train_idx"].tolist())) == len(got["train_idx"])   # one-to-one
        if got["n_matches"] >= 4 and got["best_index"] >= 0:
            X = stereo_o[p]["xyz"][ref["query_idx"]]
            eL, eR = so.hypothesis_errors(cam.K, cam.M_right, got["Rt_ransac"], X, ref["pix_L"], ref["pix_R"])
            refmask = (eL <= 4.0) & (eR <= 4.0)
            bad = np.flatnonzero(refmask != got["mask"])
            assert all(min(abs(eL[i] - 4.0), abs(eR[i] - 4.0)) < 1e-6 for i in bad)
    assert np.abs(res.pair(0)["Rt"][:, 3] - seq.relative_pose(1)[:, 3]).max() < 0.1


# tune knob 4: 1 = rounds, 0 = auto (single pass here), 2 = three rounds, then the rest of the cap in one launch
@pytest.mark.parametrize("scheme", [1, 0, 2])
@pytest.mark.parametrize("seed,n_iter,clutter,thresh", [(31, 1100, 0.72, 4.0), (5, 257, 0.5, 2.0),
                                                        (77, 2500, 0.8, 1.5)])
def test_ransac_rounds_equal_stand_alone_kernels(seed, n_iter, clutter, thresh, scheme):
    """The batched RANSAC in its launch schemes -- rounds of 256 hypotheses (late
    rounds on the narrow grid, hypotheses beyond the current requirement not
    generated), the single pass a call with few pairs takes, and the rounds with
    everything after the third merged into one launch (long calls) -- against the
    same samples pushed through the stand-alone kernels one pair at a time:
    P3P -> consensus counts -> sequential replay.  Same device arithmetic, so
    winner, count, iteration trace, pose and mask must agree exactly, for pairs
    that stop in different rounds and an iteration cap that is not a multiple of
    the round size."""
    from slam_amd import CameraModel, kernels, _lib
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    if seed == 5:
        # a right camera that is not exactly the left one shifted along x: the counting
        # kernels then project through both matrices instead of sharing y and w
        MR = cam.M_right.copy()
        a = 1e-4
        MR[:, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        cam = CameraModel(cam.K, cam.M_left, MR)
    _lib.lib.slamhip_tune(4, scheme)
    try:
        _rounds_case(cam, kernels, TrackingFrontend, StereoSequence, seed, n_iter, clutter, thresh)
    finally:
        _lib.lib.slamhip_tune(4, 0)


def _rounds_case(cam, kernels, TrackingFrontend, StereoSequence, seed, n_iter, clutter, thresh):
    seq = StereoSequence(cam, n_frames=7, n_features=900, seed=seed, clutter=clutter, bit_noise=0.05)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(7))]
    frames[3] = tuple(a[:350] for a in frames[3])          # a sparse frame: its pairs behave differently
    counts = TrackingFrontend(cam, pixel_thresh=thresh, n_iter_max=1).run(frames).to_host()["st_count"]
    rng = np.random.default_rng(seed + 46)
    samples = np.zeros((6, n_iter, 4), np.int32)
    for p in range(6):
        samples[p] = np.stack([rng.choice(int(counts[p]), 4, replace=False) for _ in range(n_iter)])
    res = TrackingFrontend(cam, pixel_thresh=thresh, n_iter_max=n_iter).run(frames, samples=samples)
    iters = []
    for p in range(6):
        got = res.pair(p)
        a, b = res.frame(p), res.frame(p + 1)
        X = a["xyz"][got["query_idx"]]
        pL, pR = b["pix_left"][got["train_idx"]], b["pix_right"][got["train_idx"]]
        Rts, valid = kernels.p3p_solve_indexed(X, pL, samples[p], cam.K)
        c = kernels.ransac_score(X, pL, pR, Rts, valid, cam.K, cam.M_right, thresh)
        rep = kernels.ransac_replay(c, len(X), 0.99999, n_iter).cpu().numpy()
        assert (got["best_index"], got["best_count"], got["iterations"], got["n_iter_required"]) == \
               (rep[0], rep[1], rep[2], rep[3]), (p, rep[:4])
        np.testing.assert_array_equal(got["Rt_ransac"], Rts.cpu().numpy()[rep[0]])
        mask = kernels.ransac_mask(X, pL, pR, got["Rt_ransac"], cam.K, cam.M_right, thresh).cpu().numpy()
        np.testing.assert_array_equal(got["mask"], mask.astype(bool))
        iters.append(int(got["iterations"]))
    if n_iter == 1100:   # this case must actually reach the late rounds and stop in different ones
        assert max(iters) > 768 and min(iters) < max(iters), iters


def test_frontend_result_builds_a_consistent_track_db():
    """The batched result, replayed through the hand-over adapter into the
    restatement of the reference's TrackingDB.add_frame (pinned to the reference's
    own DB by tests/golden/ref_tracking_db.npz): every invariant add_frame asserts
    holds, and the tracks it builds follow real landmarks across frames."""
    from oracle.tracking_db_oracle import TrackDBOracle
    from slam_amd.backend.tracking import db_adapter
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    cam, seq, frames = _sequence(6, 600, seed=17)
    res = TrackingFrontend(cam, seed=2).run(frames)
    db = db_adapter.replay_into_db(TrackDBOracle(), res)
    s = db.summary()
    assert int(s["last_frame"]) == 5
    assert (s["track_len"] >= 2).all() and s["track_len"].sum() == len(s["link_keys"])
    assert len(s["track_keys"]) > 200 and s["track_len"].max() == 6      # landmarks seen in all six frames
    # links of a frame: one per RANSAC-inlier match into or out of it
    for f in range(6):
        n_on_f = int((s["link_keys"][:, 0] == f).sum())
        ids = np.asarray(db.track_ids[f])
        assert n_on_f == int((ids >= 0).sum()) == len(set(ids[ids >= 0].tolist()))
        assert len(db.all_frame_links(f)) == len(res.frame(f)["left_idx"])


def test_batched_ransac_flags_the_all_inlier_case():
    """The same noise-free frame three times: every temporal match is right and some
    hypothesis has every correspondence as an inlier, where the reference's loop
    raises `math domain error`
    (pnp.py:150-153, pinned by ref_ransac_loop.npz case `all_inliers`).  The batched
    path cannot raise per pair: it sets the pair's domain_error flag, stops at that
    hypothesis and still returns its pose and an all-true mask."""
    from slam_amd import CameraModel
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    seq = StereoSequence(cam, n_frames=3, n_features=300, seed=9, clutter=0.0, pixel_sigma=0.0, bit_noise=0.0)
    f = seq.frame(0)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right)] * 3
    res = TrackingFrontend(cam, pixel_thresh=4.0, seed=1).run(frames)
    flagged = 0
    for p in range(2):
        pr = res.pair(p)
        if pr["domain_error"]:
            flagged += 1
            assert pr["best_count"] == pr["n_matches"] and pr["mask"].all()
            assert pr["iterations"] == pr["best_index"] + 1          # stopped right there
            assert np.abs(pr["Rt"] - np.hstack([np.eye(3), np.zeros((3, 1))])).max() < 1e-4
        else:
            assert pr["best_count"] < pr["n_matches"]
    assert flagged == 2


def test_front_end_call_captures_into_a_hip_graph():
    """The library allocates nothing and launches everything on the caller's stream, so a
    whole front-end call can be recorded into a HIP graph and replayed: replaying it on
    new input values in the same buffers gives what an ordinary call gives."""
    from slam_amd import CameraModel
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    seq = StereoSequence(cam, n_frames=4, n_features=400, seed=21, clutter=0.5)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(4))]
    # same feature counts in both pairs, so the recorded launch shapes fit both
    n = min(len(f[0]) for f in frames)
    m = min(len(f[1]) for f in frames)
    frames = [(f[0][:n], f[1][:m], f[2][:n], f[3][:m]) for f in frames]
    fe = TrackingFrontend(cam, seed=4)
    inp = fe.upload(frames[:2])
    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        fe.run(inp)                       # sizes the workspace outside the capture
    torch.cuda.current_stream().wait_stream(warm)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = fe.run(inp)
    other = fe.upload(frames[2:])
    for name in ("kp_left", "kp_right", "des_left", "des_right"):
        getattr(inp, name).copy_(getattr(other, name))
    graph.replay()
    torch.cuda.synchronize()
    expect = TrackingFrontend(cam, seed=4).run(frames[2:])
    torch.cuda.synchronize()
    for f in range(2):              # the valid part of every per-frame and per-pair array
        a, b = captured.frame(f), expect.frame(f)
        for k in b:
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"frame {f} {k}")
    a, b = captured.pair(0), expect.pair(0)
    for k in b:
        np.testing.assert_array_equal(a[k], b[k], err_msg=f"pair {k}")


@pytest.mark.parametrize("cross_check", [False, True])
def test_front_end_against_the_references_own_frame_loop(golden, cross_check):
    """tests/golden/ref_build_flow.npz: what the reference's TrackingPipeline.build()
    handed to TrackingDB.add_frame and to its pose refinement when it ran, as it is, on
    these frames (its OpenCV leaves on the CPU restatement, its RANSAC sampler seeded;
    oracle/make_golden_flow.py), once with the default matchers and once with both
    cross-checked (densified hand-over, fill_database.py:209-223).  The batched HIP
    front-end on the same frames, with the same sample sequence: identical links, inlier
    descriptors and temporal matches; the same RANSAC inliers and relative poses up to the
    minimal solver's rounding; and then the same TrackingDB state through the adapter."""
    from oracle.tracking_db_oracle import TrackDBOracle
    from slam_amd import CameraModel
    from slam_amd.backend.tracking import db_adapter
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    g = golden("ref_build_flow.npz")
    px = "cc_" if cross_check else ""
    F, seed0 = int(g["n_frames"]), int(g[px + "seed0"])
    cam = CameraModel.from_projections(g["P0"], g["P1"])
    frames = [(g[f"kp_left_{k}"], g[f"kp_right_{k}"], g[f"des_left_{k}"], g[f"des_right_{k}"]) for k in range(F)]
    n_iter = 2500
    samples = np.zeros((F - 1, n_iter, 4), np.int32)
    for p in range(F - 1):
        rng = np.random.default_rng(seed0 + p)          # the generator the reference's loop drew from
        N = int((g[f"{px}tm_train_{p + 1}"] >= 0).sum())
        for i in range(200):                            # far more than the loop consumed
            samples[p, i] = rng.choice(N, 4, replace=False)
    fe = TrackingFrontend(cam, n_iter_max=n_iter, stereo_cross_check=cross_check, frame_cross_check=cross_check)
    res = fe.run(frames, samples=samples)
    for k in range(F):
        got = res.frame(k)
        np.testing.assert_array_equal(got["links"], g[f"{px}links_{k}"], err_msg=f"links of frame {k}")
        np.testing.assert_array_equal(got["inlier_des_left"], g[f"{px}inlier_des_{k}"])
    masks_equal = True
    for p in range(F - 1):
        got = res.pair(p)
        if cross_check:
            m, mask = db_adapter.dense_matches(got, len(g[f"{px}links_{p}"]))
            q, tr = m.query_idx, m.train_idx
        else:
            q, tr, mask = got["query_idx"], got["train_idx"], got["mask"]
        np.testing.assert_array_equal(q, g[f"{px}tm_query_{p + 1}"], err_msg=f"pair {p}")
        np.testing.assert_array_equal(tr, g[f"{px}tm_train_{p + 1}"])
        ref_mask = g[f"{px}tm_mask_{p + 1}"]
        assert got["iterations"] < 200
        assert (mask != ref_mask).sum() <= 1, f"pair {p}: {(mask != ref_mask).sum()} points differ"
        if (mask == ref_mask).all():
            assert np.abs(got["Rt"] - g[px + "relative_Rts"][p]).max() < 1e-6
        else:
            masks_equal = False
    if not masks_equal:
        pytest.skip("a borderline point fell on the other side of the threshold: no database comparison")
    # then the hand-over fills the track database exactly as the reference's build() did
    s = db_adapter.replay_into_db(TrackDBOracle(), res).summary()
    for k in ("track_ids", "track_ids_off", "last_track", "last_frame", "link_keys", "link_xyz",
              "track_keys", "track_len", "leftover_count"):
        np.testing.assert_array_equal(s[k], g[px + "db_" + k], err_msg=k)


@pytest.mark.parametrize("nbytes", [32, 100])
def test_front_end_with_other_descriptor_widths(nbytes):
    """32-byte (ORB) and 100-byte descriptors through the whole batched flow (8- and
    32-word packed rows: the second hand-scheduled scan and the compiler-scheduled
    one): every index array equals the frame-at-a-time restatement."""
    from slam_amd import CameraModel
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    seq = StereoSequence(cam, n_frames=4, n_features=500, seed=31, nbytes=nbytes)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(4))]
    assert frames[0][2].shape[1] == nbytes
    n_iter = 256
    rng = np.random.default_rng(9)
    stereo_o = [po.stereo_stage(cam, f) for f in frames]
    samples = np.stack([np.stack([rng.choice(len(stereo_o[p]["left_idx"]), 4, replace=False)
                                  for _ in range(n_iter)]) for p in range(3)]).astype(np.int32)
    res = TrackingFrontend(cam, n_iter_max=n_iter).run(frames, samples=samples)
    stereo_ref, pairs_ref = po.run(cam, frames, samples=samples, n_iter_max=n_iter)
    for f in range(4):
        got, ref = res.frame(f), stereo_ref[f]
        for k in ("left_idx", "right_idx", "dist", "inlier_des_left", "pix_left", "pix_right"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"frame {f} {k}")
        assert len(got["left_idx"]) > 100
    for p in range(3):
        got, ref = res.pair(p), pairs_ref[p]
        for k in ("query_idx", "train_idx", "dist"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"pair {p} {k}")
        assert (got["mask"] != ref["mask"]).sum() <= 1 or got["best_index"] != ref["best_index"]
        assert got["mask"].mean() > 0.25


def test_packed_frames_and_chunked_run_equal_the_plain_run():
    """The same sequence handed over as per-frame tuples, as one page-locked PackedFrames, and
    in overlapped chunks (one pass, chunks of 3 and of 1 pair): identical stereo and temporal
    results, poses included (the device sampler is keyed by the pair's index in the sequence);
    without a device copy of the gathered descriptors / pixels, frame() takes the same rows
    from the host's own arrays."""
    from slam_amd.backend.tracking.feature_cache import PackedFrames
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    cam, seq, frames = _sequence(8, 500, seed=21)
    frames[3] = (frames[3][0][:200], frames[3][1][:310], frames[3][2][:200], frames[3][3][:310])
    fe = TrackingFrontend(cam, n_iter_max=300, seed=4)
    ref = fe.run(frames)
    packed = PackedFrames.from_frames(frames).pin()
    assert packed.is_pinned
    one = fe.run(packed)
    assert "st_desc" not in one.to_host() and "st_desc" in one.to_host(gathers=True)
    light = fe.run(packed)                       # frame() from the host arrays only
    for chunked in (fe.run_chunked(packed, chunk_pairs=3), fe.run_chunked(packed, chunk_pairs=1),
                    fe.run_chunked(packed, chunk_pairs=100)):
        for f in range(len(frames)):
            a, b, c, d = ref.frame(f), one.frame(f), chunked.frame(f), light.frame(f)
            for k in a:
                np.testing.assert_array_equal(a[k], b[k], err_msg=f"{k} frame {f}")
                np.testing.assert_array_equal(a[k], c[k], err_msg=f"{k} frame {f} (chunked)")
                np.testing.assert_array_equal(a[k], d[k], err_msg=f"{k} frame {f} (host gather)")
        for p in range(len(frames) - 1):
            a, b, c = ref.pair(p), one.pair(p), chunked.pair(p)
            for k in a:
                np.testing.assert_array_equal(a[k], b[k], err_msg=f"{k} pair {p}")
                np.testing.assert_array_equal(a[k], c[k], err_msg=f"{k} pair {p} (chunked)")
        np.testing.assert_array_equal(chunked.relative_Rts(), ref.relative_Rts())
    # the streamed run: one result over the whole sequence, every window's first frame taken
    # from the window before (no frame's stereo stage runs twice)
    for chunk in (1, 3, 4, 100):
        streamed = fe.run_streamed(packed, chunk_pairs=chunk)
        for f in range(len(frames)):
            a, c = ref.frame(f), streamed.frame(f)
            for k in a:
                np.testing.assert_array_equal(a[k], c[k], err_msg=f"{k} frame {f} (streamed, {chunk})")
        for p in range(len(frames) - 1):
            a, c = ref.pair(p), streamed.pair(p)
            for k in a:
                np.testing.assert_array_equal(a[k], c[k], err_msg=f"{k} pair {p} (streamed, {chunk})")
        np.testing.assert_array_equal(streamed.relative_Rts(), ref.relative_Rts())
        np.testing.assert_array_equal(streamed.t["st_count"].cpu().numpy(), ref.to_host()["st_count"])
    # ... and with a given sample sequence per pair
    st = ref.to_host()["st_count"]
    rng = np.random.default_rng(5)
    samples = np.stack([rng.integers(0, max(int(st[p]), 1), (300, 4)) for p in range(len(frames) - 1)]).astype(np.int32)
    with_samples = fe.run(packed, samples=samples)
    streamed = fe.run_streamed(packed, chunk_pairs=3, samples=samples)
    for p in range(len(frames) - 1):
        a, c = with_samples.pair(p), streamed.pair(p)
        for k in a:
            np.testing.assert_array_equal(a[k], c[k], err_msg=f"{k} pair {p} (streamed, samples)")


@pytest.mark.parametrize("nbytes", [61, 32])
def test_device_record_equals_the_host_record_byte_for_byte(nbytes):
    """slamhip_frontend_pack_record (one kernel on the device) writes the same bytes as
    sharding.pack_records builds on the host from the per-frame / per-pair views: whole batch,
    an inner range of frames and pairs, and an empty range; merge_records reads it back."""
    from slam_amd.backend.tracking import sharding
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    cam, seq, frames = _sequence(7, 450, seed=13)
    if nbytes != 61:
        frames = [(a, b, c[:, :nbytes].copy(), d[:, :nbytes].copy()) for a, b, c, d in frames]
    frames[2] = (frames[2][0][:130], frames[2][1][:200], frames[2][2][:130], frames[2][3][:200])
    res = TrackingFrontend(cam, n_iter_max=200, seed=2).run(frames)
    for (flo, fhi), (plo, phi) in (((0, 7), (0, 6)), ((1, 4), (1, 3)), ((3, 3), (2, 2)), ((6, 7), (5, 5))):
        blob = sharding.pack_result(res, (flo, fhi), (plo, phi), 100 + flo, 50 + plo).cpu().numpy()
        fr = [res.frame(f) for f in range(flo, fhi)]
        pr = []
        for p in range(plo, phi):
            d = res.pair(p)
            d["n_iter_required"], d["domain_error"] = d["n_iter_required"], d["domain_error"]
            pr.append(d)
        want = sharding.pack_records(fr, pr, 100 + flo, 50 + plo, nbytes)
        assert blob.shape == want.shape, (blob.shape, want.shape)
        np.testing.assert_array_equal(blob, want)
    whole = sharding.merge_records([sharding.pack_result(res, (0, 7), (0, 6), 0, 0).cpu().numpy()], 7)
    for f in range(7):
        a, c = whole.frame(f), res.frame(f)
        for k in c:
            np.testing.assert_array_equal(a[k], c[k], err_msg=k)


@pytest.mark.parametrize("nbytes", [61, 32, 16])
@pytest.mark.parametrize("cross", [False, True])
def test_frontend_reads_descriptor_rows_in_place(nbytes, cross):
    """The stereo matcher and the inlier-descriptor gather read the caller's rows where
    they lie (61-byte rows: nothing aligned, a tail to mask in every row, the last row
    ending at the array's last byte) -- every output, the packed inlier descriptors
    included, equals the run that packs the rows first (tune knob 9 = 0)."""
    from slam_amd import CameraModel, _lib
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    F = 5
    seq = StereoSequence(cam, n_frames=F, n_features=(300, 900), seed=nbytes + cross, clutter=0.5, nbytes=nbytes)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
    fe = TrackingFrontend(cam, n_iter_max=300, seed=3, stereo_cross_check=cross, frame_cross_check=cross)
    inp = fe.upload(frames)

    def snapshot():
        res = fe.run(inp)
        rows = [res.frame(f) for f in range(F)] + [res.pair(p) for p in range(F - 1)]
        host = res.to_host()
        return [{k: np.array(v, copy=True) for k, v in r.items() if isinstance(v, np.ndarray)} for r in rows], host

    try:
        _lib.lib.slamhip_tune(9, 1)
        a, ha = snapshot()
        _lib.lib.slamhip_tune(9, 0)
        fe._ws = None
        b, hb = snapshot()
    finally:
        _lib.lib.slamhip_tune(9, 1)
    assert sum(len(r["xyz"]) for r in a[:F]) > 200
    for x, y in zip(a, b):
        assert x.keys() == y.keys()
        for k in x:
            np.testing.assert_array_equal(x[k], y[k], err_msg=k)
    # the same rows lying in wider slots (row stride 80, whatever fills the gaps)
    import torch
    dl, dr = inp.des_left, inp.des_right
    wide_l = torch.full((dl.shape[0], 80), 0xA5, dtype=torch.uint8, device=dl.device)
    wide_r = torch.full((dr.shape[0], 80), 0x5A, dtype=torch.uint8, device=dr.device)
    wide_l[:, :nbytes], wide_r[:, :nbytes] = dl, dr
    inp.des_left, inp.des_right = wide_l[:, :nbytes], wide_r[:, :nbytes]
    fe._ws = None
    c, _ = snapshot()
    inp.des_left, inp.des_right = dl, dr
    for x, y in zip(a, c):
        for k in x:
            np.testing.assert_array_equal(x[k], y[k], err_msg="strided " + k)
    # the packed inlier descriptor rows the temporal matcher reads, zero padding included
    W = 8 if nbytes <= 32 else 16
    da, db = ha["st_desc"].reshape(-1, W), hb["st_desc"].reshape(-1, W)
    for f in range(F):
        o, m = int(inp.off_left[f]), int(ha["st_count"][f])
        np.testing.assert_array_equal(da[o:o + m], db[o:o + m])
        assert not da[o:o + m].view(np.uint8)[:, nbytes:].any()


@pytest.mark.parametrize("cross", [False, True])
def test_fused_sort_launches_equal_the_separate_ones(cross):
    """Where a frame's matches fit a block's LDS the front-end sorts and compacts (stereo) or
    sorts and gathers (temporal) in one launch; forcing a sort algorithm (tune knob 5) takes
    the separate sort launch instead.  Every output must agree, for ragged frames, with and
    without cross-check (unmatched rows sort last and end the gather)."""
    from slam_amd import CameraModel, _lib
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    F = 6
    seq = StereoSequence(cam, n_frames=F, n_features=(100, 2600), seed=17 + cross, clutter=0.6)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
    frames[2] = tuple(a[:3] for a in frames[2])        # a frame of three features
    fe = TrackingFrontend(cam, n_iter_max=300, seed=5, stereo_cross_check=cross, frame_cross_check=cross)
    inp = fe.upload(frames)
    got = []
    try:
        for algo in (0, 2, 1):
            _lib.lib.slamhip_tune(5, algo)
            res = fe.run(inp)
            rows = [res.frame(f) for f in range(F)] + [res.pair(p) for p in range(F - 1)]
            got.append([{k: np.array(v, copy=True) for k, v in r.items() if isinstance(v, np.ndarray)} for r in rows])
    finally:
        _lib.lib.slamhip_tune(5, 0)
    assert sum(len(r["xyz"]) for r in got[0][:F]) > 500
    for other in got[1:]:
        for x, y in zip(got[0], other):
            assert x.keys() == y.keys()
            for k in x:
                np.testing.assert_array_equal(x[k], y[k], err_msg=k)


def test_any_number_of_rounds_before_the_merged_tail_gives_the_same_results():
    """A long front-end call runs 1, 2 (the default) or 3 rounds of 256 hypotheses over all pairs
    before the rest of the iteration cap goes through the merged launch (tune knob 13): the same
    samples are replayed in the same order, so every output is identical -- with pairs that stop
    in the first round, in the tail, and at the cap."""
    from slam_amd import CameraModel, _lib
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    keys = ("ransac_info", "pose_ransac", "pose_refined", "tm_mask")
    try:
        for kw, n_iter in ((dict(clutter=0.63, landmark_density=0.58), 2500), (dict(clutter=0.6, landmark_density=1.7), 1100),
                           (dict(clutter=0.5), 700)):
            seq = StereoSequence(cam, n_frames=70, n_features=(300, 1000), seed=17, **kw)
            frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(70))]
            ref = None
            for knob in (3, 2, 1):
                assert _lib.lib.slamhip_tune(13, knob) == 0
                h = TrackingFrontend(cam, n_iter_max=n_iter, seed=9).run(frames).to_host()   # its own workspace
                got = {k: h[k].copy() for k in keys}
                ref = ref or got
                for k in keys:
                    np.testing.assert_array_equal(got[k], ref[k], err_msg=f"{kw}, {knob} rounds: {k}")
            it = ref["ransac_info"].reshape(-1, 8)[:69, 2]
            if n_iter == 1100:
                assert it.max() > 768, it.max()          # pairs that run into the tail and to the cap
            if n_iter == 2500:
                assert it.min() < 256 < it.max(), (it.min(), it.max())   # first round and beyond
    finally:
        _lib.lib.slamhip_tune(13, 2)


@pytest.mark.parametrize("cross", [False, True])
@pytest.mark.parametrize("F,nfeat,nbytes", [(2, (1200, 1600), 61), (3, (100, 2600), 61), (2, (2100, 4000), 32),
                                            (3, (40, 700), 100)])
def test_calls_of_a_few_frames_run_fused_launches_with_the_same_results(cross, F, nfeat, nbytes):
    """A front-end call of up to three frames (the frame-at-a-time use of the reference's own loop,
    fill_database.py:140) folds the matcher's train splits in the launch behind the matcher, runs
    sort + filter + triangulation + descriptor gather as one launch, the refinement behind the RANSAC
    replay, and deals the consensus count in shorter spans (tune knobs 14 / 15).  Every output must
    equal that of the launches a long call uses: ragged frames, a frame of three features, both
    matcher read modes, with and without cross-check."""
    from slam_amd import CameraModel, _lib
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    seq = StereoSequence(cam, n_frames=F, n_features=nfeat, seed=23 + cross + F, clutter=0.6)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
    if nbytes != 61:
        widen = lambda d: (np.concatenate([d, d[:, :nbytes - 61] ^ 0x5A], axis=1) if nbytes > 61 else d[:, :nbytes]).copy()
        frames = [(kl, kr, widen(dl), widen(dr)) for kl, kr, dl, dr in frames]
    if F == 3:
        frames[1] = tuple(a[:3] for a in frames[1])        # a frame of three features
    fe = TrackingFrontend(cam, n_iter_max=700, seed=5, stereo_cross_check=cross, frame_cross_check=cross)
    inp = fe.upload(frames)

    def run():
        res = fe.run(inp)
        rows = [res.frame(f) for f in range(F)] + [res.pair(p) for p in range(F - 1)]
        return [{k: np.array(v, copy=True) for k, v in r.items()} for r in rows]

    try:
        _lib.lib.slamhip_tune(14, 0)
        want = run()
        _lib.lib.slamhip_tune(14, 1)
        got = [run()]
        for span in (32, 64, 200):
            _lib.lib.slamhip_tune(15, span)
            got.append(run())
        _lib.lib.slamhip_tune(15, 0)
        _lib.lib.slamhip_tune(9, 0)          # the matcher reads packed rows
        got.append(run())
    finally:
        _lib.lib.slamhip_tune(14, 1)
        _lib.lib.slamhip_tune(15, 0)
        _lib.lib.slamhip_tune(9, 1)
    assert sum(len(r["xyz"]) for r in want[:F]) > 100
    for other in got:
        for x, y in zip(want, other):
            assert x.keys() == y.keys()
            for k in x:
                np.testing.assert_array_equal(x[k], y[k], err_msg=k)


@pytest.mark.parametrize("cross", [False, True])
def test_pushing_frames_one_at_a_time_equals_the_batched_call(cross):
    """OnlineFrontend.push is the reference's own calling pattern (one stereo frame per iteration,
    the previous frame's stereo stage carried over, fill_database.py:140-240): each push uploads
    the new frame only and runs one library call with first_stereo_frame = 1.  Frame by frame it
    must give, bit for bit, what one batched call over the whole sequence gives -- through the
    moves of the last frame to the front of the device arrays (a capacity of three frames), a
    frame without right features, a frame of three features and a growing frame size."""
    from slam_amd import CameraModel
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.backend.tracking.online import OnlineFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    F = 11
    seq = StereoSequence(cam, n_frames=F, n_features=(300, 1500), seed=41 + cross, clutter=0.6)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
    frames[4] = (frames[4][0], np.zeros((0, 2), np.float32), frames[4][2], np.zeros((0, 61), np.uint8))
    frames[7] = tuple(a[:3] for a in frames[7])
    big = StereoSequence(cam, n_frames=F, n_features=4000, seed=43, clutter=0.6).frame(9)
    frames[9] = (big.kp_left, big.kp_right, big.des_left, big.des_right)      # outgrows the arrays
    kw = dict(n_iter_max=600, seed=11, stereo_cross_check=cross, frame_cross_check=cross)
    want = TrackingFrontend(cam, **kw).run(frames)
    online = OnlineFrontend(cam, frames_capacity=3, **kw)
    assert online.push(*frames[0]) is None
    for k in range(1, F):
        got = online.push(*frames[k])
        Rt, info = got.pose()
        for (x, y) in ((got.frame(0), want.frame(k - 1)), (got.frame(1), want.frame(k)), (got.pair(0), want.pair(k - 1))):
            assert x.keys() == y.keys()
            for key in x:
                np.testing.assert_array_equal(x[key], y[key], err_msg=f"frame {k}: {key}")
        np.testing.assert_array_equal(Rt, want.pair(k - 1)["Rt"])
        assert info[5] == want.pair(k - 1)["n_matches"]
    assert sum(want.pair(p)["mask"].sum() for p in range(F - 1)) > 100
    # the library refuses a first stereo frame outside the call's frames
    import ctypes as C
    from slam_amd import _lib
    args = want.args
    for bad in (-1, F):
        args.first_stereo_frame = bad
        assert _lib.lib.slamhip_frontend_batch(C.byref(args), torch.cuda.current_stream().cuda_stream) == _lib.EINVAL
    args.first_stereo_frame = 0
    # a second sequence through the same object
    online.reset()
    assert online.push(*frames[1]) is None
    np.testing.assert_array_equal(online.push(*frames[2]).pair(0)["query_idx"], want.pair(1)["query_idx"])


@pytest.mark.gpu
@pytest.mark.parametrize("n_left,n_right,nbytes,stride", [(1500, 1400, 61, 61), (7, 0, 32, 40), (0, 5, 61, 64),
                                                           (0, 0, 61, 61), (4097, 3, 1, 1)])
def test_one_packet_per_frame_reaches_the_rows_it_names(n_left, n_right, nbytes, stride):
    """slamhip_frontend_unpack_frame: the packet's five pieces land at the addresses given and
    nothing next to them changes."""
    from slam_amd import _lib
    import ctypes as C
    rng = np.random.default_rng(n_left + 3 * n_right)
    off = rng.integers(0, 1 << 20, 6).astype(np.int32)
    kp_l = rng.random((n_left, 2), dtype=np.float32)
    kp_r = rng.random((n_right, 2), dtype=np.float32)
    de_l = rng.integers(0, 256, (n_left, nbytes), dtype=np.uint8)
    de_r = rng.integers(0, 256, (n_right, nbytes), dtype=np.uint8)
    packet = np.concatenate([a.reshape(-1).view(np.uint8) for a in (off, kp_l, kp_r, de_l, de_r)])
    d_packet = torch.from_numpy(packet).cuda()
    row_l, row_r = 3, 5                                  # where the frame starts in the caller's arrays
    fill = 0xA5
    d_kp_l = torch.full((row_l + n_left + 2, 2), float("nan"), device="cuda")
    d_kp_r = torch.full((row_r + n_right + 2, 2), float("nan"), device="cuda")
    d_de_l = torch.full((row_l + n_left + 2, stride), fill, dtype=torch.uint8, device="cuda")
    d_de_r = torch.full((row_r + n_right + 2, stride), fill, dtype=torch.uint8, device="cuda")
    d_off = torch.full((8,), -1, dtype=torch.int32, device="cuda")
    for with_off in (True, False):
        rc = _lib.lib.slamhip_frontend_unpack_frame(
            d_packet.data_ptr(), n_left, n_right, nbytes, stride, d_kp_l[row_l:].data_ptr(), d_kp_r[row_r:].data_ptr(),
            d_de_l[row_l:].data_ptr(), d_de_r[row_r:].data_ptr(), d_off.data_ptr() if with_off else None,
            torch.cuda.current_stream().cuda_stream)
        assert rc == _lib.OK
        torch.cuda.synchronize()
        for got, want, row, n in ((d_kp_l, kp_l, row_l, n_left), (d_kp_r, kp_r, row_r, n_right)):
            g = got.cpu().numpy()
            assert np.array_equal(g[row:row + n], want)
            assert np.isnan(g[:row]).all() and np.isnan(g[row + n:]).all()
        for got, want, row, n in ((d_de_l, de_l, row_l, n_left), (d_de_r, de_r, row_r, n_right)):
            g = got.cpu().numpy()
            assert np.array_equal(g[row:row + n, :nbytes], want)
            assert (g[:row] == fill).all() and (g[row + n:] == fill).all() and (g[:, nbytes:] == fill).all()
        assert np.array_equal(d_off.cpu().numpy(), np.concatenate([off, [-1, -1]]))
    assert _lib.lib.slamhip_frontend_unpack_frame(d_packet.data_ptr(), 1, 0, nbytes, nbytes - 1, d_kp_l.data_ptr(),
                                                  d_kp_r.data_ptr(), d_de_l.data_ptr(), d_de_r.data_ptr(), None,
                                                  None) == _lib.EINVAL
