"""GPU test of loop-closure pair verification against the CPU restatement of
LoopClosure.consensus_pnp_match."""
import numpy as np
import pytest
import torch

from oracle import pipeline_oracle as po
from oracle import slam_oracle as so

from _verify_helpers import _assert_same_consensus, _cpu_loop_on_device_poses, _idx4_table

pytestmark = pytest.mark.gpu


def test_verify_pairs_against_oracle():
    from slam_amd import CameraModel
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    seq = StereoSequence(cam, n_frames=120, n_features=900, seed=5, clutter=0.3)
    kf_ids = [0, 2, 4, 60, 62, 118]
    keyframes = []
    for k in kf_ids:
        f = seq.frame(k)
        st = po.stereo_stage(cam, (f.kp_left, f.kp_right, f.des_left, f.des_right))
        pl, pr = st["pix_left"].astype(np.float64), st["pix_right"].astype(np.float64)
        links = np.column_stack([pl[:, 0], pr[:, 0], (pl[:, 1] + pr[:, 1]) / 2])   # Link(x_l, x_r, y)
        keyframes.append((links, st["inlier_des_left"]))
    pairs = np.array([(i, n) for n in range(len(kf_ids)) for i in range(n)], dtype=np.int64)
    ver = PairVerifier(cam, seed=3).set_keyframes(keyframes)
    res = ver.verify(pairs)
    off = np.concatenate([[0], np.cumsum([len(k[0]) for k in keyframes])])
    n_ok = 0
    for p, (i, n) in enumerate(pairs):
        ref = po.consensus_pnp_match(cam, keyframes[i], keyframes[n], rng=np.random.default_rng(p))
        sl = slice(int(res["out_off"][p]), int(res["out_off"][p + 1]))
        np.testing.assert_array_equal(res["query_idx"][sl], ref["query_idx"], err_msg=f"pair {p}")
        np.testing.assert_array_equal(res["train_idx"][sl], ref["train_idx"])
        np.testing.assert_array_equal(res["dist"][sl], ref["dist"])
        xyz = res["xyz"][off[i]:off[i + 1]]
        obj32 = ref["obj"].astype(np.float32)
        assert np.abs(xyz - obj32).max() <= 2e-6 * np.abs(obj32).max()
        assert bool(res["ok"][p]) == bool(ref["ok"]), (p, i, n, res["n_inliers"][p], ref.get("best_count"))
        if not res["ok"][p]:
            continue
        n_ok += 1
        # mask = reference formula on the device's own winner; pose = LS minimum on that mask
        eL, eR = so.hypothesis_errors(cam.K, cam.M_right, res["Rt_ransac"][p], ref["P3D"], ref["ptsL"], ref["ptsR"])
        refmask = (eL <= 1.5) & (eR <= 1.5)
        bad = np.flatnonzero(refmask != res["mask"][sl])
        assert all(min(abs(eL[k] - 1.5), abs(eR[k] - 1.5)) < 1e-6 for k in bad)
        assert res["n_inliers"][p] == res["mask"][sl].sum()
        gold, _ = so.refine_pose(ref["P3D"], ref["ptsL"], cam.K, res["Rt_ransac"][p], res["mask"][sl])
        assert np.abs(res["Rt"][p] - gold).max() < 1e-7
        # both estimate the same motion (different random samples, same consensus set up to a few points)
        assert np.abs(res["Rt"][p] - ref["Rt"]).max() < 0.1
        A = np.vstack([seq.poses[kf_ids[i]], [0, 0, 0, 1]])
        B = np.vstack([seq.poses[kf_ids[n]], [0, 0, 0, 1]])
        truth = (B @ np.linalg.inv(A))[:3]
        assert np.abs(res["Rt"][p][:, 3] - truth[:, 3]).max() < 0.25
    assert 3 <= n_ok < len(pairs)          # near keyframes verify, far ones do not


def test_verify_pairs_with_a_given_sample_sequence_count_for_count():
    """d_samples on the pair-verification path: every pair runs the idx4 sequence it is given, so
    the CPU loop on the same sequence is comparable iteration for iteration.  With the device's
    own minimal solutions as hypotheses the CPU loop must reproduce winner, iteration count,
    consensus count and mask exactly; with its own minimal solver (another root finder) the same
    winner in nearly every pair, and then the same iteration count and the same mask to one point."""
    from slam_amd import CameraModel
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    seq = StereoSequence(cam, n_frames=120, n_features=900, seed=5, clutter=0.3)
    kf_ids = [0, 2, 4, 60, 62, 118]
    keyframes = []
    for k in kf_ids:
        f = seq.frame(k)
        st = po.stereo_stage(cam, (f.kp_left, f.kp_right, f.des_left, f.des_right))
        pl, pr = st["pix_left"].astype(np.float64), st["pix_right"].astype(np.float64)
        keyframes.append((np.column_stack([pl[:, 0], pr[:, 0], (pl[:, 1] + pr[:, 1]) / 2]), st["inlier_des_left"]))
    keyframes.append((keyframes[0][0][:30], keyframes[0][1][:30]))         # below the first gate: never sampled
    pairs = np.array([(i, n) for n in range(len(keyframes)) for i in range(len(keyframes)) if i != n], dtype=np.int64)
    n_iter = 300
    counts = [len(keyframes[i][0]) for i, _ in pairs]                      # one match per candidate row
    samples = _idx4_table(np.random.default_rng(9), counts, n_iter)
    ver = PairVerifier(cam, seed=3, n_iter_max=n_iter).set_keyframes(keyframes)
    res = ver.verify(pairs, samples=samples)
    again = ver.verify(pairs, samples=samples, pair_ids=np.arange(len(pairs))[::-1].copy())
    for k in ("best_index", "iterations", "n_inliers", "Rt"):             # the table, not the sampler keys, decides
        np.testing.assert_array_equal(np.array(again[k]), np.array(res[k]))
    n_ok = same_best = attempted = 0
    for p, (i, n) in enumerate(pairs):
        own, ref = _cpu_loop_on_device_poses(cam, keyframes[i], keyframes[n], samples[p], n_iter_max=n_iter)
        sl = slice(int(res["out_off"][p]), int(res["out_off"][p + 1]))
        if "P3D" not in ref:
            assert res["iterations"][p] == 0 and not res["ok"][p]
            continue
        attempted += 1
        _assert_same_consensus(cam, res, p, sl, ref)
        assert bool(res["ok"][p]) == bool(ref["ok"])
        if res["ok"][p]:
            n_ok += 1
            assert np.abs(res["Rt"][p] - ref["Rt"]).max() < 1e-7
        if own["best_index"] == ref["best_index"]:
            same_best += 1
            assert own["iterations"] == res["iterations"][p]
            assert (own["mask"] != res["mask"][sl]).sum() <= 1
    assert n_ok >= 6 and attempted >= 30 and same_best >= attempted - 3
    with pytest.raises(ValueError):
        ver.verify(pairs, samples=samples[:, :10])


def test_verify_reproduces_the_references_consensus_check_on_its_own_samples(golden):
    """tests/golden/ref_consensus_flow.npz again, now with the reference run's own idx4 sequence:
    the fixture records the seed its numpy.random.default_rng() calls were given, and the
    reference draws rng.choice(N, 4, replace=False) once per iteration (pnp.py:126).  Fed that
    sequence the device must end with the reference's mask (to one point: the fixture's
    hypotheses came from the CPU minimal solver) and its relative pose."""
    from slam_amd import CameraModel
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    g = golden("ref_consensus_flow.npz")
    cam = CameraModel.from_projections(g["P0"], g["P1"])
    ids = sorted(int(k.split("_")[1]) for k in g.files if k.startswith("links_"))
    keyframes = [(g[f"links_{k}"], g[f"des_{k}"]) for k in ids]
    seed = int(g["seed0"])
    checked = 0
    for i, (curr, cand, gate, cc) in enumerate(g["cases"]):
        calls = int(g[f"rng_calls_{i}"])
        if calls == 0:
            continue                                    # stopped at the first gate: no generator made
        rng = np.random.default_rng(seed)
        seed += calls
        N = len(g[f"mask_{i}"])
        ver = PairVerifier(cam, pixel_thresh=1.5, min_inliers_loop=int(gate), seed=i,
                           frame_cross_check=bool(cc)).set_keyframes(keyframes)
        if N == 0:                                      # second gate: the reference returned no mask
            res = ver.verify(np.array([[ids.index(int(cand)), ids.index(int(curr))]], dtype=np.int64))
            N = int(res["n_matches"][0])
        samples = _idx4_table(rng, [N], ver.n_iter_max)
        res = ver.verify(np.array([[ids.index(int(cand)), ids.index(int(curr))]], dtype=np.int64), samples=samples)
        assert bool(res["ok"][0]) == bool(g[f"ok_{i}"]), f"case {i}"
        if not res["ok"][0]:
            continue
        mask, ref_mask = res.pair_matches(0)["mask"], g[f"mask_{i}"]
        assert len(mask) == len(ref_mask) and (mask != ref_mask).sum() <= 1, (i, (mask != ref_mask).sum())
        T = np.eye(4)
        T[:3] = res["Rt"][0]
        D = np.linalg.inv(T) - g[f"delta_inv_{i}"]
        assert np.abs(D).max() < (1e-6 if (mask == ref_mask).all() else 1e-3), (i, np.abs(D).max())
        checked += 1
    assert checked >= 2


def test_verify_pairs_edge_cases():
    from slam_amd import CameraModel
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    cam = CameraModel.kitti_like()
    rng = np.random.default_rng(0)
    def kf(n):
        x = rng.uniform(100, 1000, n); y = rng.uniform(10, 300, n)
        return (np.column_stack([x, x - rng.uniform(2, 50, n), y]), rng.integers(0, 256, (n, 61), dtype=np.uint8))
    ver = PairVerifier(cam).set_keyframes([kf(10), kf(0), kf(80)])
    res = ver.verify([(0, 2), (1, 2), (2, 1), (2, 0)])
    assert not res["ok"].any()
    assert res["n_matches"].tolist() == [10, 0, 0, 80]
    assert res["iterations"][0] == 0          # fewer than min_inliers_loop matches: not attempted
    assert ver.verify(np.zeros((0, 2), int))["ok"].shape == (0,)


def test_lazy_match_arrays_agree():
    """pair_matches(p) and verified_matches() return the same rows as the full
    per-match arrays."""
    from slam_amd import CameraModel
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    seq = StereoSequence(cam, n_frames=12, n_features=500, seed=3, clutter=0.4)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in (0, 1, 2, 9))]
    st = TrackingFrontend(cam, n_iter_max=1).run(frames)
    kfs = [(st.frame(k)["links"], st.frame(k)["inlier_des_left"]) for k in range(4)]
    ver = PairVerifier(cam, min_inliers_loop=40, seed=1).set_keyframes(kfs)
    res = ver.verify([(0, 1), (0, 3), (1, 2), (2, 3)])
    assert res["ok"][0] and res["ok"][2]
    vm = res.verified_matches()
    assert vm["pairs"].tolist() == np.flatnonzero(res["ok"]).tolist()
    for k, p in enumerate(vm["pairs"]):
        one = res.pair_matches(int(p))
        sl = slice(int(res["out_off"][p]), int(res["out_off"][p + 1]))
        vs = slice(int(vm["off"][k]), int(vm["off"][k + 1]))
        for key in ("query_idx", "train_idx", "dist", "mask"):
            assert key in res
            np.testing.assert_array_equal(one[key], res[key][sl])
            np.testing.assert_array_equal(vm[key][vs], res[key][sl])
        assert one["mask"].dtype == bool and one["mask"].sum() == res["n_inliers"][p]


def test_verify_against_the_references_own_consensus_check(golden):
    """tests/golden/ref_consensus_flow.npz: outcomes of the reference's own
    LoopClosure.consensus_pnp_match (run as it is, OpenCV leaves on the CPU restatement,
    oracle/make_golden_flow.py) for a verified pair, a pair stopped at the first
    `min_inliers_loop` gate and one stopped at the second.  The device draws other random
    samples and the adaptive stop keeps the first hypothesis that is good enough, so the
    consensus sets overlap without being equal (1.5 px threshold on 0.4 px noise); the
    decision, the gates and the relative pose must agree."""
    from slam_amd import CameraModel
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    g = golden("ref_consensus_flow.npz")
    cam = CameraModel.from_projections(g["P0"], g["P1"])
    ids = sorted(int(k.split("_")[1]) for k in g.files if k.startswith("links_"))
    keyframes = [(g[f"links_{k}"], g[f"des_{k}"]) for k in ids]
    for i, (curr, cand, gate, cc) in enumerate(g["cases"]):
        ver = PairVerifier(cam, pixel_thresh=1.5, min_inliers_loop=int(gate), seed=i,
                           frame_cross_check=bool(cc)).set_keyframes(keyframes)
        res = ver.verify(np.array([[ids.index(int(cand)), ids.index(int(curr))]], dtype=np.int64))
        assert bool(res["ok"][0]) == bool(g[f"ok_{i}"]), f"case {i}"
        if int(g[f"rng_calls_{i}"]) == 0:
            assert res["iterations"][0] == 0              # first gate: RANSAC never ran
        if not res["ok"][0]:
            continue
        mask = res.pair_matches(0)["mask"]
        ref_mask = g[f"mask_{i}"]
        assert len(mask) == len(ref_mask)                 # the same matches (one per candidate feature unless cross-checked)
        assert mask.sum() >= gate and ref_mask.sum() >= gate
        assert (mask & ref_mask).sum() > 0.5 * ref_mask.sum()
        T = np.eye(4)
        T[:3] = res["Rt"][0]
        D = np.linalg.inv(T) - g[f"delta_inv_{i}"]
        assert np.abs(D[:3, :3]).max() < 5e-3 and np.abs(D[:3, 3]).max() < 0.15, D


def test_single_pair_method_has_the_references_return_shape(golden):
    """PairVerifier.consensus_pnp_match(kf_curr, kf_cand): the tuple of
    loop_closure.py:207/235, consistent with verify() on the same pair; the pixel sets are
    the reference's own where the match lists are (every row of px comes from the same
    keypoints; which rows are inliers depends on the samples drawn)."""
    from slam_amd import CameraModel
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    g = golden("ref_consensus_flow.npz")
    cam = CameraModel.from_projections(g["P0"], g["P1"])
    ids = sorted(int(k.split("_")[1]) for k in g.files if k.startswith("links_"))
    keyframes = [(g[f"links_{k}"], g[f"des_{k}"]) for k in ids]
    seen = set()
    for i, (curr, cand, gate, cc) in enumerate(g["cases"]):
        ver = PairVerifier(cam, pixel_thresh=1.5, min_inliers_loop=int(gate), seed=i,
                           frame_cross_check=bool(cc)).set_keyframes(keyframes)
        ci, ni = ids.index(int(cand)), ids.index(int(curr))
        out = ver.consensus_pnp_match(ni, ci)
        assert len(out) == 5 and out[0] == bool(g[f"ok_{i}"])
        seen.add(out[0])
        if not out[0]:
            assert out[1:] == (None, None, (None, None), {})
            continue
        ok, delta_inv, mask, pair, px = out
        assert pair == (ni, ci) and mask.dtype == bool
        res = ver.verify(np.array([[ci, ni]], dtype=np.int64))          # same seed, same pair: same result
        np.testing.assert_array_equal(mask, res.pair_matches(0)["mask"])
        T = np.eye(4)
        T[:3] = res["Rt"][0]
        np.testing.assert_allclose(delta_inv @ T, np.eye(4), atol=1e-12)
        assert np.abs(delta_inv - g[f"delta_inv_{i}"])[:3, :3].max() < 5e-3
        assert set(px) == {"inl_curr", "out_curr", "inl_cand", "out_cand"}
        for k in px:
            assert px[k].dtype == np.float32 and px[k].shape[1] == 2
        assert len(px["inl_curr"]) == len(px["inl_cand"]) == mask.sum()
        assert len(px["out_curr"]) == len(px["out_cand"]) == (~mask).sum()
        # the union of inlier and outlier pixels is the reference's, whatever the split
        ref_curr = {tuple(r) for r in g[f"inl_curr_{i}"]}
        ours_curr = {tuple(r) for r in np.vstack([px["inl_curr"], px["out_curr"]])}
        assert ref_curr <= ours_curr
        ref_cand = {tuple(r) for r in g[f"inl_cand_{i}"]}
        assert ref_cand <= {tuple(r) for r in np.vstack([px["inl_cand"], px["out_cand"]])}

        class Pose:                                                      # what gtsam.Pose3 offers here
            def __init__(self, R, t):
                self.R, self.t = R, t

            def inverse(self):
                return ("inverse of", self.R.shape, self.t.shape)
        assert ver.consensus_pnp_match(ni, ci, pose_factory=Pose)[1] == ("inverse of", (3, 3), (3,))
    assert seen == {True, False}


def test_verify_pairs_with_cross_checked_matcher():
    """frame_cross_check=True (the matcher of loop_closure.py:192 built with crossCheck):
    fewer matches per pair, the same ones as the CPU restatement, in the same order; the
    decision agrees and the pose is the least-squares minimum on the device's own mask."""
    from slam_amd import CameraModel
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    seq = StereoSequence(cam, n_frames=70, n_features=700, seed=8, clutter=0.4)
    kf_ids = [0, 2, 3, 60, 61]
    keyframes = []
    for k in kf_ids:
        f = seq.frame(k)
        st = po.stereo_stage(cam, (f.kp_left, f.kp_right, f.des_left, f.des_right))
        pl, pr = st["pix_left"].astype(np.float64), st["pix_right"].astype(np.float64)
        keyframes.append((np.column_stack([pl[:, 0], pr[:, 0], (pl[:, 1] + pr[:, 1]) / 2]), st["inlier_des_left"]))
    keyframes[2] = (keyframes[2][0][:150], keyframes[2][1][:150])        # pairs of unequal size
    pairs = np.array([(i, n) for n in range(len(kf_ids)) for i in range(n)], dtype=np.int64)
    res = PairVerifier(cam, seed=5, frame_cross_check=True).set_keyframes(keyframes).verify(pairs)
    plain = PairVerifier(cam, seed=5).set_keyframes(keyframes).verify(pairs)
    n_ok = 0
    for p, (i, n) in enumerate(pairs):
        ref = po.consensus_pnp_match(cam, keyframes[i], keyframes[n], rng=np.random.default_rng(p), cross_check=True)
        got = res.pair_matches(p)
        assert res["n_matches"][p] == len(ref["query_idx"]) <= len(keyframes[i][0])
        np.testing.assert_array_equal(got["query_idx"], ref["query_idx"], err_msg=f"pair {p}")
        np.testing.assert_array_equal(got["train_idx"], ref["train_idx"])
        np.testing.assert_array_equal(got["dist"], ref["dist"])
        assert res["n_matches"][p] <= plain["n_matches"][p]
        assert bool(res["ok"][p]) == bool(ref["ok"]), (p, res["n_inliers"][p], ref.get("best_count"))
        if res["ok"][p]:
            n_ok += 1
            gold, _ = so.refine_pose(ref["P3D"], ref["ptsL"], cam.K, res["Rt_ransac"][p], got["mask"])
            assert np.abs(res["Rt"][p] - gold).max() < 1e-7
    assert (res["n_matches"] < plain["n_matches"]).any() and 2 <= n_ok < len(pairs)
    vm = res.verified_matches()
    assert vm["off"][-1] == res["n_matches"][res["ok"]].sum() == len(vm["mask"])


def test_pair_list_longer_than_one_chunk_with_a_short_tail():
    """More than 32768 pairs run as a full chunk plus a short one through the same
    workspace; the short chunk must not outgrow it (it runs with the full chunk's
    split count) and must give the same matches as the same pairs in the full chunk."""
    from slam_amd import CameraModel
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    seq = StereoSequence(cam, n_frames=4, n_features=900, seed=8, clutter=0.3)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(4))]
    st = TrackingFrontend(cam, n_iter_max=1).run(frames)
    kfs = [(st.frame(k)["links"], st.frame(k)["inlier_des_left"]) for k in range(4)]
    assert min(len(k[0]) for k in kfs) > 300
    base = np.array([(i, n) for n in range(4) for i in range(n)], dtype=np.int64)       # 6 pairs
    n_pairs = 32768 + 100
    pairs = base[np.arange(n_pairs) % len(base)]
    ver = PairVerifier(cam, n_iter_max=256, seed=2).set_keyframes(kfs)
    res = ver.verify(pairs)
    assert (res["n_matches"] == np.array([len(kfs[i][0]) for i in pairs[:, 0]])).all()
    tail = np.arange(32768, n_pairs)
    for p in tail[:: 7]:
        twin = int(p % len(base))                 # the same keyframe pair inside the first chunk
        a, b = res.pair_matches(int(p)), res.pair_matches(twin)
        for key in ("query_idx", "train_idx", "dist"):
            np.testing.assert_array_equal(a[key], b[key])
        assert a["mask"].sum() == res["n_inliers"][p]
    # consecutive frames verify in the tail as they do in the first chunk
    # (every copy of a pair draws its own samples, so a borderline pair may differ between copies)
    kind = np.arange(n_pairs) % len(base)
    sure = [k for k in range(len(base)) if res["ok"][:32768][kind[:32768] == k].mean() > 0.999]
    assert sure, "no keyframe pair verifies reliably"
    assert all(res["ok"][32768:][kind[32768:] == k].all() for k in sure)
    # sampler keys across the chunk boundary: the list's own numbering changes nothing, and keys
    # that give every copy of a keyframe pair the number of its first copy make the copies equal
    first = {k: np.array(res[k]).copy() for k in ("ok", "n_inliers", "iterations", "Rt_ransac")}
    keyed = ver.verify(pairs, pair_ids=np.arange(n_pairs))
    for k in first:
        np.testing.assert_array_equal(np.array(keyed[k]), first[k], err_msg=k)
    same = ver.verify(pairs, pair_ids=kind)
    for k in first:
        np.testing.assert_array_equal(np.array(same[k]), first[k][:len(base)][kind], err_msg=k)


def test_keyframe_points_use_float32_keypoints_like_cv2_keypoint(golden):
    """LoopClosure.frame_data builds cv2.KeyPoint(x, y, 1) from the stored links, so the
    fp64 triangulation sees Link.y = (yL + yR) / 2 rounded to float32.  The reference's own
    run holds the resulting float32 points of the consensus set (refine_obj): the device's
    points of the same features are the same float32 values."""
    from slam_amd import CameraModel
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    g = golden("ref_consensus_flow.npz")
    cam = CameraModel.from_projections(g["P0"], g["P1"])
    ids = sorted(int(k.split("_")[1]) for k in g.files if k.startswith("links_"))
    keyframes = [(g[f"links_{k}"], g[f"des_{k}"]) for k in ids]
    off = np.concatenate([[0], np.cumsum([len(k[0]) for k in keyframes])])
    y = np.concatenate([k[0][:, 2] for k in keyframes])
    assert (y.astype(np.float32).astype(np.float64) != y).mean() > 0.2     # the rounding matters here
    checked = 0
    for i, (curr, cand, gate, cc) in enumerate(g["cases"]):
        if not bool(g[f"ok_{i}"]):
            continue
        ver = PairVerifier(cam, pixel_thresh=1.5, min_inliers_loop=int(gate), seed=i,
                           frame_cross_check=bool(cc)).set_keyframes(keyframes)
        c = ids.index(int(cand))
        res = ver.verify(np.array([[c, ids.index(int(curr))]], dtype=np.int64))
        m = res.pair_matches(0)
        ref_mask = g[f"mask_{i}"]
        got = res["xyz"][off[c]:off[c + 1]][m["query_idx"]][ref_mask]
        ref = g[f"refine_obj_{i}"]
        assert got.shape == ref.shape and got.dtype == np.float32 and ref.dtype == np.float32
        assert (got == ref).mean() > 0.99 and np.abs(got - ref).max() <= 2.5e-7 * np.abs(ref).max()
        checked += 1
    assert checked >= 2


def test_counting_schemes_give_the_same_consensus():
    """The consensus count of pair verification dismisses most (hypothesis, point) combinations
    of unrelated keyframes by the left image alone and takes each hypothesis' own sample points
    from the launch that solved it (tune knob 2 = 1, the default here; 3: the front-end too);
    knob 2 = 2 screens every combination and 0 evaluates all of them in fp64.  Same counts, so
    the same winners, iteration numbers, masks and poses -- on related and unrelated pairs,
    odd and tiny match counts, pairs of one chunk and of many."""
    from slam_amd import CameraModel, _lib
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    seq = StereoSequence(cam, n_frames=200, n_features=(40, 1400), seed=12, clutter=0.5)
    ids = [0, 1, 2, 3, 50, 51, 52, 120, 121, 199]
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in ids)]
    st = TrackingFrontend(cam, n_iter_max=1).run(frames)
    keyframes = [(st.frame(k)["links"], st.frame(k)["inlier_des_left"]) for k in range(len(ids))]
    keyframes.append((keyframes[0][0][:5], keyframes[0][1][:5]))            # fewer rows than a sample needs spare
    keyframes.append((keyframes[1][0][:129], keyframes[1][1][:129]))        # one point past a chunk of 128
    pairs = np.array([(i, n) for n in range(len(keyframes)) for i in range(len(keyframes)) if i != n], dtype=np.int64)
    keys = ("ok", "n_matches", "n_inliers", "iterations", "Rt", "Rt_ransac", "mask")
    fe_seq = StereoSequence(cam, n_frames=70, n_features=(300, 900), seed=3, clutter=0.6)
    fe_frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (fe_seq.frame(k) for k in range(70))]
    fe_keys = ("ransac_info", "pose_ransac", "pose_refined", "tm_mask")
    ref = fe_ref = None
    try:
        for knob in (2, 1, 3, 0):
            assert _lib.lib.slamhip_tune(2, knob) == 0
            for min_inl in (10, 30):
                ver = PairVerifier(cam, seed=4, min_inliers_loop=min_inl).set_keyframes(keyframes)
                res = ver.verify(pairs)
                got = {k: np.array(res[k]).copy() for k in keys}
                if knob == 2:
                    ref = ref or {}
                    ref[min_inl] = got
                    assert 4 <= got["ok"].sum() < len(pairs) // 2
                for k in keys:
                    np.testing.assert_array_equal(got[k], ref[min_inl][k], err_msg=f"knob {knob}: {k}")
            for F in (70, 9):        # rounds with the merged tail; the single pass of short calls
                h = TrackingFrontend(cam, n_iter_max=700, seed=8).run(fe_frames[:F]).to_host()
                got = {k: h[k].copy() for k in fe_keys}
                if knob == 2:
                    fe_ref = fe_ref or {}
                    fe_ref[F] = got
                for k in fe_keys:
                    np.testing.assert_array_equal(got[k], fe_ref[F][k], err_msg=f"front-end, knob {knob}: {k}")
    finally:
        _lib.lib.slamhip_tune(2, 1)


def test_counting_schemes_agree_on_points_at_the_threshold():
    """A keyframe pair made so that a quarter of the hypotheses (samples of exact points) is the
    true motion to rounding and the other points sit at the threshold distance from their
    projections, times 1 +- {1e-7 ... 5e-3}: across the band the packed-fp32 screen leaves
    undecided and on both sides of it.  The sparse counting scheme (tune knob 2 = 1 / 3), the
    plain screen (2) and fp64 alone (0) must count the same."""
    from slam_amd import CameraModel, _lib
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    cam = CameraModel.kitti_like()
    fx, fy, cx, cy = cam.K[0, 0], cam.K[1, 1], cam.K[0, 2], cam.K[1, 2]
    base = -cam.M_right[0, 3]
    rng = np.random.default_rng(77)
    n = 700
    Z = rng.uniform(6.0, 60.0, n)
    X = np.column_stack([rng.uniform(-0.8, 0.8, n) * Z, rng.uniform(-0.2, 0.2, n) * Z, Z])

    def links_of(P):
        u, v = fx * P[:, 0] / P[:, 2] + cx, fy * P[:, 1] / P[:, 2] + cy
        return np.column_stack([u, u - fx * base / P[:, 2], v])

    ang = 0.03
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.1, -0.02, -0.9])
    la, lb = links_of(X), links_of(X @ R.T + t)
    # float32 pixels are what the verifier keeps; the points it triangulates come from la
    la, lb = la.astype(np.float32).astype(np.float64), lb.astype(np.float32).astype(np.float64)
    thr = 1.5
    moved = rng.random(n) < 0.3
    factor = 1.0 + rng.choice([1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 2e-3, 3e-3, 5e-3], n) * rng.choice([-1.0, 1.0], n)
    phi = rng.uniform(0, 2 * np.pi, n)
    which = rng.integers(0, 3, n)          # move: the left x and the row / the right x alone / the left x alone
    d = thr * factor
    lb[moved & (which == 0), 0] += (d * np.cos(phi))[moved & (which == 0)]
    lb[moved & (which == 0), 2] += (d * np.sin(phi))[moved & (which == 0)]
    lb[moved & (which == 1), 1] += d[moved & (which == 1)]
    lb[moved & (which == 2), 0] -= d[moved & (which == 2)]
    des = rng.integers(0, 256, (n, 61), dtype=np.uint8)
    keyframes = [(la, des), (lb, des.copy())]
    ref = None
    try:
        for knob in (0, 2, 1, 3):
            assert _lib.lib.slamhip_tune(2, knob) == 0
            for seed in (1, 2):
                res = PairVerifier(cam, seed=seed, min_inliers_loop=20, n_iter_max=600).set_keyframes(keyframes).verify([(0, 1), (1, 0)])
                got = {k: np.array(res[k]).copy() for k in ("ok", "n_inliers", "iterations", "Rt_ransac", "mask")}
                if knob == 0:
                    ref = ref or {}
                    ref[seed] = got
                    assert got["ok"].all() and 0.6 * n < got["n_inliers"].min() and got["n_inliers"].max() < 0.95 * n
                for k in got:
                    np.testing.assert_array_equal(got[k], ref[seed][k], err_msg=f"knob {knob} seed {seed}: {k}")
    finally:
        _lib.lib.slamhip_tune(2, 1)


def test_a_pair_list_dealt_over_ranks_gives_the_results_of_the_whole_list():
    """Pairs dealt round-robin over 3 or 8 ranks, each shard verified with the pairs' indices in
    the whole list as sampler keys: pair for pair the flags, counts, iteration numbers, poses and
    masks of the one call over the whole list (and not without the keys)."""
    from slam_amd import CameraModel
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    from slam_amd.backend.tracking.frontend import TrackingFrontend
    from slam_amd.utils.synth import StereoSequence
    cam = CameraModel.kitti_like()
    KF = 24
    seq = StereoSequence(cam, n_frames=KF * 5, n_features=(500, 1100), seed=19, clutter=0.5)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(5 * k) for k in range(KF))]
    st = TrackingFrontend(cam, n_iter_max=1).run(frames)
    keyframes = [(st.frame(k)["links"], st.frame(k)["inlier_des_left"]) for k in range(KF)]
    every = np.array([(i, n) for n in range(KF) for i in range(n)], dtype=np.int64)
    ver = PairVerifier(cam, seed=6, min_inliers_loop=30).set_keyframes(keyframes)
    whole = ver.verify(every)
    keys = ("ok", "n_matches", "n_inliers", "iterations", "Rt", "Rt_ransac")
    ref = {k: np.array(whole[k]).copy() for k in keys}
    ref_mask, ref_off = np.array(whole["mask"]).copy(), np.array(whole["out_off"]).copy()
    assert 5 < ref["ok"].sum() < len(every) // 2
    np.testing.assert_array_equal(np.array(ver.verify(every, pair_ids=np.arange(len(every)))["Rt_ransac"]), ref["Rt_ransac"])
    for world in (3, 8):
        for rank in range(world):
            ids = np.arange(len(every))[rank::world]
            part = ver.verify(every[rank::world], pair_ids=ids)
            for k in keys:
                np.testing.assert_array_equal(np.array(part[k]), ref[k][ids], err_msg=f"world {world} rank {rank}: {k}")
            off, mask = np.array(part["out_off"]), np.array(part["mask"])
            for j, p in enumerate(ids):
                np.testing.assert_array_equal(mask[off[j]:off[j + 1]], ref_mask[ref_off[p]:ref_off[p + 1]])
    unkeyed = ver.verify(every[1::3])
    assert not np.array_equal(np.array(unkeyed["Rt_ransac"]), ref["Rt_ransac"][1::3])
    with pytest.raises(ValueError):
        ver.verify(every[:4], pair_ids=[0, 1])


def test_pairs_with_more_matches_than_a_sample_index_field_holds():
    """The sparse counting scheme keeps a hypothesis' sample indices in 16 bits each; a call with
    more than 65535 rows per keyframe counts the plain way, with the same results as with the
    scheme switched off."""
    from slam_amd import CameraModel, _lib
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    cam = CameraModel.kitti_like()
    rng = np.random.default_rng(5)
    n = 66000

    def kf():
        x = rng.uniform(50, 1200, n); y = rng.uniform(5, 365, n)
        return (np.column_stack([x, x - rng.uniform(2, 60, n), y]), rng.integers(0, 256, (n, 32), dtype=np.uint8))
    ver = PairVerifier(cam, seed=1, n_iter_max=300).set_keyframes([kf(), kf()])
    got = {}
    try:
        for knob in (1, 2):
            _lib.lib.slamhip_tune(2, knob)
            res = ver.verify([(0, 1)])
            got[knob] = {k: np.array(res[k]).copy() for k in ("ok", "n_matches", "n_inliers", "iterations", "Rt_ransac")}
    finally:
        _lib.lib.slamhip_tune(2, 1)
    assert got[1]["n_matches"][0] == n and 1 <= got[1]["iterations"][0] <= 300
    for k in got[1]:
        np.testing.assert_array_equal(got[1][k], got[2][k], err_msg=k)


class _LinkRow:
    def __init__(self, row):
        self.x_left, self.x_right, self.y = (float(v) for v in row)


class _KeyframeDB:
    """What consensus_pnp_match reads of a TrackingDB (loop_closure.py:157-158)."""

    def __init__(self, g, ids):
        self._links = {k: [_LinkRow(r) for r in g[f"links_{k}"]] for k in ids}
        self._des = {k: g[f"des_{k}"] for k in ids}
        self.reads = []

    def features(self, kf):
        self.reads.append(kf)
        return self._des[kf]

    def all_frame_links(self, kf):
        return self._links[kf]


@pytest.mark.gpu
def test_consensus_check_with_the_references_signature_reads_the_tracking_db(golden):
    """ConsensusCheck.consensus_pnp_match(kf_curr, kf_cand, db) (the signature of
    loop_closure.py:167) on the keyframes of the reference's own run: keyframes are read from
    the DB and uploaded the first time they are met, one by one; the results are those of
    PairVerifier over the whole keyframe list with the same sampler keys, bit for bit, the
    decisions are the reference's, and the candidates of a keyframe checked in one call give
    what they give one by one."""
    from slam_amd import CameraModel
    from slam_amd.backend.optimizers.loop_closure import ConsensusCheck, GpuConsensusCheck
    from slam_amd.backend.optimizers.loop_verify import PairVerifier
    g = golden("ref_consensus_flow.npz")
    cam = CameraModel.from_projections(g["P0"], g["P1"])
    ids = sorted(int(k.split("_")[1]) for k in g.files if k.startswith("links_"))
    keyframes = [(g[f"links_{k}"], g[f"des_{k}"]) for k in ids]
    seen = set()
    for i, (curr, cand, gate, cc) in enumerate(g["cases"]):
        curr, cand = int(curr), int(cand)
        db = _KeyframeDB(g, ids)
        chk = ConsensusCheck(cam, frame_cross_check=bool(cc), min_inliers_loop=int(gate), seed=i)
        out = chk.consensus_pnp_match(curr, cand, db)
        assert len(out) == 5 and out[0] == bool(g[f"ok_{i}"])
        seen.add(out[0])
        again = chk.consensus_pnp_match(curr, cand, db)
        assert sorted(db.reads) == sorted({curr, cand})                   # each keyframe read once
        whole = PairVerifier(cam, pixel_thresh=1.5, min_inliers_loop=int(gate), seed=i,
                             frame_cross_check=bool(cc)).set_keyframes(keyframes)
        res = whole.verify(np.array([[ids.index(cand), ids.index(curr)]]), pair_ids=[(curr << 24) ^ cand])
        assert bool(res["ok"][0]) == out[0]
        if not out[0]:
            assert out[1:] == (None, None, (None, None), {}) and again == out
            continue
        ok, delta_inv, mask, pair, px = out
        assert pair == (curr, cand)                                        # frame ids, as :235
        np.testing.assert_array_equal(mask, res.pair_matches(0)["mask"])
        np.testing.assert_array_equal(mask, again[2])
        T = np.eye(4)
        T[:3] = res["Rt"][0]
        np.testing.assert_allclose(delta_inv @ T, np.eye(4), atol=1e-12)
        np.testing.assert_array_equal(delta_inv, again[1])
        assert np.abs(delta_inv - g[f"delta_inv_{i}"])[:3, :3].max() < 5e-3
        assert len(px["inl_curr"]) == len(px["inl_cand"]) == mask.sum()
        assert {tuple(r) for r in g[f"inl_curr_{i}"]} <= {tuple(r) for r in np.vstack([px["inl_curr"], px["out_curr"]])}
    assert seen == {True, False}

    # all candidates of one keyframe in one call = one by one; keyframes met in any order
    db = _KeyframeDB(g, ids)
    chk = ConsensusCheck(cam, min_inliers_loop=60)
    curr, cands = ids[-1], ids[:-1]
    together = chk.consensus_pnp_matches(curr, cands, db)
    alone = ConsensusCheck(cam, min_inliers_loop=60)
    for c, t in zip(reversed(cands), reversed(together)):
        o = alone.consensus_pnp_match(curr, c, db)
        assert o[0] == t[0] and o[3] == t[3]
        if o[0]:
            np.testing.assert_array_equal(o[2], t[2])
            np.testing.assert_array_equal(o[1], t[1])
    assert chk.consensus_pnp_matches(curr, [], db) == []

    # frame_data / compute_triangulation, the two helpers of the method (:152-164, :237-249)
    kpL, kpR, desL = chk.frame_data(db, ids[0])
    links = g[f"links_{ids[0]}"]
    assert kpL.dtype == np.float32 and kpL.shape == (len(links), 2) and desL is db._des[ids[0]]
    np.testing.assert_array_equal(kpL, links[:, [0, 2]].astype(np.float32))
    np.testing.assert_array_equal(kpR, links[:, [1, 2]].astype(np.float32))
    xyz = chk.compute_triangulation(cam.P_left, cam.P_right, kpL, kpR)
    assert xyz.dtype == np.float64 and xyz.shape == (len(links), 3)
    ref = so.triangulate_cv(cam.P_left, cam.P_right, kpL.astype(np.float64), kpR.astype(np.float64), out_fp64=True)
    np.testing.assert_allclose(xyz, ref, rtol=1e-9, atol=1e-9)

    # as a mix-in: the attributes LoopClosure.__init__ sets are what it reads (:32-40)
    class RefShaped:
        def __init__(self, cam, frame_cross_check, min_inliers_loop=60):
            from slam_amd.frontend.geometry.pnp import PnP
            self.pnp = PnP(cam, 1.5)
            self.frame_cross_check, self.min_inliers_loop = frame_cross_check, min_inliers_loop

    class Mixed(GpuConsensusCheck, RefShaped):
        pass
    mixed = Mixed(cam, False, 60).consensus_pnp_matches(curr, cands, db)
    for a, b in zip(mixed, together):
        assert a[0] == b[0] and a[3] == b[3]
        if a[0]:
            np.testing.assert_array_equal(a[2], b[2])
