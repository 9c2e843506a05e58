"""GPU parity tests of PnP-RANSAC consensus scoring: inlier counts identical to
the reference arithmetic, except for points within 1e-6 px of the threshold."""
import numpy as np
import pytest
import torch

from oracle import slam_oracle as so

pytestmark = pytest.mark.gpu

BAND = 1e-6  # px; stated tolerance band around the reprojection threshold


def _assert_masks(mask, eL, eR, thr, what):
    ref = (eL <= thr) & (eR <= thr)
    diff = np.flatnonzero(mask.astype(bool) != ref)
    for p in diff:
        near = min(abs(eL[p] - thr), abs(eR[p] - thr))
        assert near <= BAND, f"{what}: point {p} differs, errors {eL[p]}, {eR[p]}"
    return len(diff)


def test_scores_and_masks_match_reference_golden(golden):
    from slam_amd import kernels
    g = golden("ref_ransac_score.npz")
    K, MR, thr = g["K"], g["M_right"], float(g["pixel_thresh"])
    counts = kernels.ransac_score(g["X"], g["meas_L"], g["meas_R"], g["Rts"], None, K, MR, thr)
    counts = counts.cpu().numpy()
    n_diff = 0
    for h in range(g["Rts"].shape[0]):
        mask = kernels.ransac_mask(g["X"], g["meas_L"], g["meas_R"], g["Rts"][h], K, MR, thr)
        mask = mask.cpu().numpy()
        n_diff += _assert_masks(mask, g["err_L"][h], g["err_R"][h], thr, f"hyp {h}")
        assert counts[h] == mask.sum()
    np.testing.assert_array_equal(counts, g["counts"]) if n_diff == 0 else None
    assert n_diff == 0  # nothing in this fixture sits inside the band


def test_project_and_supporters_match_reference_golden(golden):
    from slam_amd import kernels
    g = golden("ref_ransac_score.npz")
    K, thr = g["K"], float(g["pixel_thresh"])
    for h in (0, 5):
        P = K @ g["Rts"][h]
        pix = kernels.project(g["X"], P).cpu().numpy()
        ref = g["proj_L"][h]
        ok = np.isfinite(ref).all(axis=1)
        np.testing.assert_allclose(pix[ok], ref[ok], rtol=1e-13, atol=0)
        m = kernels.supporters_pair(g["meas_L"], g["proj_L"][h], g["meas_R"], g["proj_R"][h], thr)
        np.testing.assert_array_equal(m.cpu().numpy(), g["masks"][h])


def test_compute_err_equals_the_references_bit_for_bit(golden):
    """compute_err (utils/graph_utils.py:7-11) on the projections the reference computed:
    the errors it computed from them, every bit (no fused multiply-add in dx*dx + dy*dy);
    and the supporters mask from those same errors."""
    from slam_amd import kernels
    from slam_amd.utils.graph_utils import compute_err
    g = golden("ref_ransac_score.npz")
    thr = float(g["pixel_thresh"])
    n_checked = 0
    for h in range(len(g["Rts"])):
        for meas, proj, ref in ((g["meas_L"], g["proj_L"][h], g["err_L"][h]), (g["meas_R"], g["proj_R"][h], g["err_R"][h])):
            err = compute_err(meas.astype(np.float64), proj)
            assert isinstance(err, np.ndarray) and err.dtype == np.float64
            ok = np.isfinite(ref)
            np.testing.assert_array_equal(err[ok], ref[ok])
            assert (np.isnan(err) == np.isnan(ref)).all()
            n_checked += int(ok.sum())
        m = kernels.supporters_pair(g["meas_L"], g["proj_L"][h], g["meas_R"], g["proj_R"][h], thr).cpu().numpy()
        np.testing.assert_array_equal(m.astype(bool), (g["err_L"][h] <= thr) & (g["err_R"][h] <= thr))
    assert n_checked > 5000
    dev = compute_err(torch.from_numpy(g["meas_L"].astype(np.float64)).cuda(), torch.from_numpy(g["proj_L"][0]).cuda())
    assert dev.is_cuda and dev.shape == (len(g["meas_L"]),)
    # random pairs against NumPy's own norm (separately rounded squares, correctly rounded root)
    rng = np.random.default_rng(0)
    a, b = rng.normal(0, 300, (20000, 2)), rng.normal(0, 300, (20000, 2))
    np.testing.assert_array_equal(compute_err(a, b), np.linalg.norm(a - b, axis=1))


def test_scores_large_random_scene_with_invalid_hypotheses():
    from slam_amd import kernels, CameraModel
    from slam_amd.utils.synth import perturbed_hypotheses
    cam = CameraModel.kitti_like()
    rng = np.random.default_rng(8)
    n, H = 1111, 2500
    X = np.column_stack([rng.uniform(-25, 25, n), rng.uniform(-3, 2, n), rng.uniform(4, 120, n)]).astype(np.float32)
    ang = 0.01
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    Rt = np.hstack([R, np.array([[0.02], [-0.01], [-0.8]])])
    Rt_R = (so.homogenous(cam.M_right) @ so.homogenous(Rt))[:3]
    mL = (so.project(X, cam.K @ Rt) + rng.normal(0, 0.7, (n, 2))).astype(np.float32)
    mR = (so.project(X, cam.K @ Rt_R) + rng.normal(0, 0.7, (n, 2))).astype(np.float32)
    bad = rng.random(n) < 0.35
    mL[bad] += rng.normal(0, 30, (int(bad.sum()), 2)).astype(np.float32)
    Rts, valid = perturbed_hypotheses(rng, Rt, H)
    for thr in (4.0, 1.5):
        counts = kernels.ransac_score(X, mL, mR, Rts, valid, cam.K, cam.M_right, thr).cpu().numpy()
        ref = so.score_hypotheses(cam.K, cam.M_right, Rts, valid, X, mL, mR, thr)
        bad_h = np.flatnonzero(counts != ref)
        for h in bad_h:  # any difference must be explained by points inside the band
            eL, eR = so.hypothesis_errors(cam.K, cam.M_right, Rts[h], X, mL, mR)
            in_band = (np.abs(eL - thr) <= BAND) | (np.abs(eR - thr) <= BAND)
            assert abs(int(counts[h]) - int(ref[h])) <= in_band.sum()
        assert len(bad_h) <= 2
        assert (counts[valid == 0] == -1).all()
    # fp64 landmarks (loop-closure call site) take the same path
    c64 = kernels.ransac_score(X.astype(np.float64), mL, mR, Rts[:64], None, cam.K, cam.M_right, 4.0)
    c32 = kernels.ransac_score(X, mL, mR, Rts[:64], None, cam.K, cam.M_right, 4.0)
    np.testing.assert_array_equal(c64.cpu().numpy(), c32.cpu().numpy())


def test_general_right_camera_matrix():
    """M_right with a rotation: the full 4x4 product must be honoured."""
    from slam_amd import kernels, CameraModel
    rng = np.random.default_rng(2)
    cam = CameraModel.kitti_like()
    a = 0.02
    Rr = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    MR = np.hstack([Rr, np.array([[-0.54], [0.01], [0.002]])])
    n = 300
    X = np.column_stack([rng.uniform(-10, 10, n), rng.uniform(-3, 2, n), rng.uniform(4, 60, n)]).astype(np.float32)
    Rt = np.hstack([np.eye(3), np.array([[0.0], [0.0], [-0.5]])])
    mL = so.project(X, cam.K @ Rt).astype(np.float32)
    mR = so.project(X, cam.K @ (so.homogenous(MR) @ so.homogenous(Rt))[:3]).astype(np.float32)
    Rts = np.stack([Rt, np.hstack([np.eye(3), np.array([[0.3], [0.0], [-0.5]])])])
    counts = kernels.ransac_score(X, mL, mR, Rts, None, cam.K, MR, 1.0).cpu().numpy()
    ref = so.score_hypotheses(cam.K, MR, Rts, None, X, mL, mR, 1.0)
    np.testing.assert_array_equal(counts, ref)
    assert counts[0] == n


def test_device_replay_matches_oracle(golden):
    from slam_amd import kernels
    g = golden("oracle_ransac_replay.npz")
    for i in range(4):
        res = kernels.ransac_replay(g[f"counts{i}"], int(g[f"N{i}"])).cpu().numpy()
        np.testing.assert_array_equal(res[:4], g[f"res{i}"], err_msg=f"trace {i}")
        assert res[4] == 0
    res = kernels.ransac_replay(np.array([10, 50, 3], np.int32), 50).cpu().numpy()
    assert res[0] == 1 and res[1] == 50 and res[4] == 1   # where the reference raises ValueError
    res = kernels.ransac_replay(np.array([5, 9, 7, 30, 2], np.int32), 60, early_inliers=9).cpu().numpy()
    assert tuple(res[:3]) == (1, 9, 2)


def test_pnp_facade_with_injected_hypotheses():
    from slam_amd import PnP, CameraModel
    from slam_amd.utils.synth import perturbed_hypotheses
    cam = CameraModel.kitti_like()
    rng = np.random.default_rng(21)
    n = 550
    X = np.column_stack([rng.uniform(-25, 25, n), rng.uniform(-3, 2, n), rng.uniform(4, 120, n)]).astype(np.float32)
    Rt = np.hstack([np.eye(3), np.array([[0.01], [0.0], [-0.8]])])
    Rt_R = (so.homogenous(cam.M_right) @ so.homogenous(Rt))[:3]
    mL = (so.project(X, cam.K @ Rt) + rng.normal(0, 0.5, (n, 2))).astype(np.float32)
    mR = (so.project(X, cam.K @ Rt_R) + rng.normal(0, 0.5, (n, 2))).astype(np.float32)
    bad = rng.random(n) < 0.35
    mL[bad] += rng.normal(0, 30, (int(bad.sum()), 2)).astype(np.float32)
    Rts, valid = perturbed_hypotheses(rng, Rt, 2500, good_fraction=0.1)
    pnp = PnP(cam, pixel_thresh=4.0)
    for kwargs in ({}, {"n_iter_max": 1500}, {"early_inliers": 200}, {"block": 7}):
        mask = pnp.ransac_pnp_pair1(X, mL, mR, hypotheses=(Rts, valid), **kwargs)
        okw = {k: v for k, v in kwargs.items() if k != "block"}
        counts = so.score_hypotheses(cam.K, cam.M_right, Rts, valid, X, mL, mR, 4.0)
        best, best_count, it, req = so.ransac_replay(counts, n, 0.99999, okw.get("n_iter_max", 2500),
                                                      okw.get("early_inliers"))
        ref = so.hypothesis_mask(cam.K, cam.M_right, Rts[best], X, mL, mR, 4.0)
        np.testing.assert_array_equal(mask, ref)
        assert mask.dtype == bool and mask.sum() == best_count
    # reference-shaped helpers
    P = cam.K @ Rt
    np.testing.assert_allclose(PnP.project(X, P), so.project(X, P), rtol=1e-13)
    s = pnp.supporters_pair(mL, so.project(X, P), mR, so.project(X, cam.K @ Rt_R))
    np.testing.assert_array_equal(s, so.supporters_pair(mL, so.project(X, P), mR,
                                                        so.project(X, cam.K @ Rt_R), 4.0))


def _screen_camera(kind):
    """Camera pairs that take the different screening kernels: a rectified pair
    (right camera = left shifted along x) shares y and w between the cameras; a skewed
    K still does; a rotated right camera or a K with entries below the diagonal do not."""
    from slam_amd import CameraModel
    cam = CameraModel.kitti_like()
    K, MR = cam.K.copy(), cam.M_right.copy()
    if kind == "skewed_k":
        K[0, 1] = 0.37
    elif kind == "rotated_right":
        a = 2e-3
        MR[:, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        MR[1, 3] = 0.004
    elif kind == "lower_k":
        K[1, 0] = 0.02
        K[2, 0] = 1e-6
    elif kind == "wide_baseline":
        MR[0, 3] = -350.0           # c = fx * tx far above every other entry of P
    return CameraModel(K, cam.M_left, MR)


@pytest.mark.parametrize("kind", ["rectified", "skewed_k", "rotated_right", "lower_k", "wide_baseline"])
def test_packed_fp32_screening_never_changes_a_count(kind):
    """The counting kernels screen in packed fp32 and hand every undecided point
    to the fp64 path; with the screen switched off (tune knob 2) the same calls must
    return identical counts.  The scene is built to stress the bound: measurements
    placed exactly on, and a few ulps either side of, the threshold circle of the
    true pose; near (2 m) and far (400 m) points; points behind the camera and at
    w ~ 0; hypotheses from exact to wild, including NaN, infinity, a translation
    whose projection overflows fp64, and finite poses with |t| from 1e-200 to 1e200."""
    from slam_amd import kernels, _lib
    from slam_amd.utils.synth import perturbed_hypotheses
    cam = _screen_camera(kind)
    rng = np.random.default_rng(21)
    n, H, thr = 3001, 6000, 2.0
    X = np.column_stack([rng.uniform(-40, 40, n), rng.uniform(-5, 5, n), rng.uniform(2, 400, n)])
    X[:40, 2] = rng.uniform(-5, 0.5, 40)            # behind / at the camera plane
    X[40:44] *= 1e3                                 # farther than the screen accepts
    X = X.astype(np.float32)
    Rt = np.hstack([np.eye(3), np.array([[0.03], [-0.01], [-0.8]])])
    Rr = so.homogenous(cam.M_right) @ so.homogenous(Rt)
    pL = so.project(X.astype(np.float64), cam.K @ Rt)
    pR = so.project(X.astype(np.float64), cam.K @ Rr[:3])
    phi = rng.uniform(0, 2 * np.pi, n)
    rad = np.where(rng.random(n) < 0.5, thr, rng.uniform(0, 2 * thr, n))   # half exactly on the circle
    mL = (pL + (rad * np.array([np.cos(phi), np.sin(phi)])).T).astype(np.float32)
    mR = (pR + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    for k in range(0, 600, 3):                      # nudge some by single float32 ulps
        mL[k, 0] = np.nextafter(mL[k, 0], np.float32(np.inf if k % 2 else -np.inf))
    mL[700, 0] = 3e5                                # a pixel coordinate beyond the screen's range
    mR[900, 1] = np.nan
    mL[1100, 1] = np.inf
    Rts, valid = perturbed_hypotheses(rng, Rt, H, good_fraction=0.5)
    Rts[0] = Rt
    Rts[1, 2, 3] = 0.0                              # t_z = 0: w ~ Z
    Rts[2] = np.nan
    Rts[3] *= 1e6                                   # overflows fp32 squares
    Rts[4, 0, 3] = -1.7e308                         # fx * t_x overflows fp64
    Rts[5, 0, 3] = np.inf
    # finite poses of absurd size, as a minimal sample with a repeated pixel produces them: the
    # screen sees the matrices scaled by a power of two, so these are decided in fp32 as well
    for k, size in zip(range(6, 11), (1e7, 1e16, 1e31, 1e200, 1e-200)):
        Rts[k] = Rt
        Rts[k, :, 3] = np.array([0.3, -0.1, 1.0]) * size
    Rts[11] = Rt * 1e-300                           # every entry tiny: scaled up instead
    Rts[12] = Rt
    Rts[12, :, 3] *= 1e5                            # the true direction, far out
    valid[:13] = 1
    try:
        assert _lib.lib.slamhip_tune(2, 0) == 0
        ref = kernels.ransac_score(X, mL, mR, Rts, valid, cam.K, cam.M_right, thr).cpu().numpy()
        assert _lib.lib.slamhip_tune(2, 1) == 0
        got = kernels.ransac_score(X, mL, mR, Rts, valid, cam.K, cam.M_right, thr).cpu().numpy()
    finally:
        _lib.lib.slamhip_tune(2, 1)
    np.testing.assert_array_equal(got, ref)
    assert ref[0] > 0 and (ref[valid == 0] == -1).all() and ref[2] == 0
    # a threshold nothing can meet: the reference's `error <= thr` is false everywhere
    for bad_thr in (-1.0, float("nan")):
        none = kernels.ransac_score(X, mL, mR, Rts[:64], valid[:64], cam.K, cam.M_right, bad_thr).cpu().numpy()
        assert (none[valid[:64] == 1] == 0).all()
        assert not kernels.ransac_mask(X, mL, mR, Rts[0], cam.K, cam.M_right, bad_thr).cpu().numpy().any()
    # and one beyond the screen's range is still counted (in fp64)
    wide = kernels.ransac_score(X, mL, mR, Rts[:64], valid[:64], cam.K, cam.M_right, 5e3).cpu().numpy()
    assert wide[0] >= ref[0]
    # and against the reference arithmetic, outside the stated band
    eL, eR = so.hypothesis_errors(cam.K, cam.M_right, Rts[0], X, mL, mR)
    sure = (np.abs(eL - thr) > BAND) & (np.abs(eR - thr) > BAND)
    assert abs(int(((eL <= thr) & (eR <= thr))[sure].sum()) - int(ref[0])) <= int((~sure).sum())


def test_facade_loop_equals_reference_loop(golden):
    """ref_ransac_loop.npz is the reference's own ransac_pnp_pair1 loop fed a
    prepared pose list through its solve_pnp.  The façade fed the same list returns
    the same mask (None / ValueError in the same cases), and the device replay
    stops after the same number of iterations."""
    from slam_amd import kernels, CameraModel, PnP
    g = golden("ref_ransac_loop.npz")
    cam = CameraModel.kitti_like()
    np.testing.assert_array_equal(cam.K, g["K"])
    thr = float(g["pixel_thresh"])
    for name in g["names"]:
        k = str(name)
        X, mL, mR = g[k + "_X"], g[k + "_mL"], g[k + "_mR"]
        Rts, valid = g[k + "_Rts"], g[k + "_valid"]
        p_success, n_iter_max, early = g[k + "_kw"]
        early = None if early < 0 else int(early)
        pnp = PnP(cam, thr)
        raised = str(g[k + "_raised"])
        if raised:
            with pytest.raises(ValueError, match=raised):
                pnp.ransac_pnp_pair1(X, mL, mR, float(p_success), int(n_iter_max), early,
                                     hypotheses=(Rts, valid))
            continue
        for block in (256, 7):       # the replay carries its state across scoring blocks of any size
            mask = pnp.ransac_pnp_pair1(X, mL, mR, float(p_success), int(n_iter_max), early,
                                        hypotheses=(Rts, valid), block=block)
            if not bool(g[k + "_has_mask"]):
                assert mask is None, k
            else:
                assert mask.dtype == bool
                np.testing.assert_array_equal(mask, g[k + "_mask"], err_msg=f"{k} block {block}")
        counts = kernels.ransac_score(X, mL, mR, Rts, valid, cam.K, cam.M_right, thr)
        rep = kernels.ransac_replay(counts, len(X), float(p_success), int(n_iter_max), early).cpu().numpy()
        assert rep[2] == int(g[k + "_calls"]), (k, rep)
        assert rep[1] == int(g[k + "_mask"].sum())
