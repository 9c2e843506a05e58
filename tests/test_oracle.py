"""CPU tests: the oracle against the reference-pinned golden vectors and
against its own literal (sequential) restatements."""
import numpy as np
import pytest

from oracle import slam_oracle as so


def test_camera_model_matches_reference(golden):
    g = golden("ref_camera.npz")
    cam = so.CameraModel.from_projection_rows(g["P0"], g["P1"])
    for name in ("K", "M_left", "M_right", "P_left", "P_right"):
        np.testing.assert_array_equal(getattr(cam, name), g[name])


def test_stereo_filter_matches_reference(golden):
    g = golden("ref_stereo_filter.npz")
    keep = so.stereo_filter(g["kp_left"], g["kp_right"], g["qidx"], g["tidx"],
                            float(g["pixel_thresh"]), float(g["disparity_min"]))
    np.testing.assert_array_equal(keep.astype(np.uint8), g["keep"])
    # the crafted boundary rows: |dv| == thr keeps, one ulp above rejects
    assert g["keep"][0] == 1 and g["keep"][1] == 0 and g["keep"][5] == 1 and g["keep"][6] == 0


def test_dlt_own_matches_reference(golden):
    g = golden("ref_dlt_own.npz")
    cam = so.CameraModel.from_projection_rows(golden("ref_camera.npz")["P0"],
                                              golden("ref_camera.npz")["P1"])
    A = so.dlt_matrices_own(cam.P_left, cam.P_right, g["kp_left"], g["kp_right"])
    np.testing.assert_array_equal(A, g["A"])
    X = so.triangulate_own(cam.P_left, cam.P_right, g["kp_left"], g["kp_right"])
    np.testing.assert_allclose(X, g["X"], rtol=1e-12, atol=0)
    # the cv2-row-order system has the same null space
    Xcv = so.triangulate_cv(cam.P_left, cam.P_right, g["kp_left"].astype(np.float64),
                            g["kp_right"].astype(np.float64), out_fp64=True)
    np.testing.assert_allclose(Xcv, g["X"], rtol=1e-9, atol=0)


def test_ransac_scoring_matches_reference(golden):
    g = golden("ref_ransac_score.npz")
    K, MR, thr = g["K"], g["M_right"], float(g["pixel_thresh"])
    for h in range(g["Rts"].shape[0]):
        with np.errstate(all="ignore"):
            eL, eR = so.hypothesis_errors(K, MR, g["Rts"][h], g["X"], g["meas_L"], g["meas_R"])
        np.testing.assert_array_equal(eL, g["err_L"][h])
        np.testing.assert_array_equal(eR, g["err_R"][h])
        m = so.hypothesis_mask(K, MR, g["Rts"][h], g["X"], g["meas_L"], g["meas_R"], thr)
        np.testing.assert_array_equal(m.astype(np.uint8), g["masks"][h])
    counts = so.score_hypotheses(K, MR, g["Rts"], None, g["X"], g["meas_L"], g["meas_R"], thr)
    np.testing.assert_array_equal(counts, g["counts"])


def test_hamming_oracle_vectorised_equals_sequential():
    rng = np.random.default_rng(3)
    for nq, nt, d in [(40, 33, 61), (25, 70, 32), (9, 1, 61), (9, 2, 61), (5, 0, 61)]:
        q = rng.integers(0, 256, (nq, d), dtype=np.uint8)
        t = rng.integers(0, 4, (nt, d), dtype=np.uint8)  # few distinct values -> many ties
        q[:, :] &= 3
        a = so.hamming_knn2(q, t)
        b = so.hamming_knn2_sequential(q, t)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])


def test_hamming_oracle_golden(golden, oracle_c):
    g = golden("oracle_hamming.npz")
    for name in ("akaze", "orb", "nt1", "nt2"):
        idx2, dist2 = so.hamming_knn2(g[f"{name}_q"], g[f"{name}_t"])
        np.testing.assert_array_equal(idx2, g[f"{name}_idx2"])
        np.testing.assert_array_equal(dist2, g[f"{name}_dist2"])
        ci, cd = oracle_c(g[f"{name}_q"], g[f"{name}_t"])
        np.testing.assert_array_equal(ci, g[f"{name}_idx2"])
        np.testing.assert_array_equal(cd, g[f"{name}_dist2"])
    # duplicates of train row 3 live at rows 5 and nt-1: ties go to the lowest index
    assert tuple(g["akaze_idx2"][0]) == (3, 5) and tuple(g["akaze_dist2"][0]) == (0, 0)


def test_c_oracle_equals_numpy_oracle_midsize(oracle_c):
    from slam_amd.utils.synth import matcher_workload
    q, t = matcher_workload(700, 900, 61, seed=5)
    a = so.hamming_knn2(q, t)
    b = oracle_c(q, t, threads=4)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_cross_check_and_sort_semantics():
    rng = np.random.default_rng(11)
    q = rng.integers(0, 256, (60, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (45, 32), dtype=np.uint8)
    qi, ti, di = so.bf_match(q, t, cross_check=False)
    assert len(qi) == 60 and (np.diff(di) >= 0).all()
    same = np.flatnonzero(np.diff(di) == 0)
    assert (qi[same + 1] > qi[same]).all()          # stable: ascending queryIdx inside ties
    qi2, ti2, di2 = so.bf_match(q, t, cross_check=True)
    assert len(qi2) <= 45 and len(set(ti2.tolist())) == len(ti2)
    d = so.hamming_distance_matrix(q, t)
    for a, b, c in zip(qi2, ti2, di2):
        assert d[a, b] == c and d[:, b].min() == c  # the train row's nearest query is a


def test_ransac_replay_golden(golden):
    g = golden("oracle_ransac_replay.npz")
    for i in range(4):
        res = so.ransac_replay(g[f"counts{i}"], int(g[f"N{i}"]))
        np.testing.assert_array_equal(np.array(res), g[f"res{i}"])
    np.testing.assert_array_equal(so.sample_indices(550, 64), g["idx4_seed12345_N550"])


def test_ransac_replay_domain_error_like_reference():
    with pytest.raises(ValueError):
        so.ransac_replay([10, 50], 50)  # all points inliers -> math.log(0)


def test_ransac_loop_matches_reference_loop(golden):
    """ref_ransac_loop.npz: the reference's own ransac_pnp_pair1 loop run with its
    solve_pnp replaced by a prepared pose list.  The oracle's scoring + replay of
    the same list must give the same mask, the same number of iterations, and
    raise where the reference raised."""
    g = golden("ref_ransac_loop.npz")
    K, MR, thr = g["K"], g["M_right"], float(g["pixel_thresh"])
    for name in g["names"]:
        k = str(name)
        X, mL, mR = g[k + "_X"], g[k + "_mL"], g[k + "_mR"]
        Rts, valid = g[k + "_Rts"], g[k + "_valid"]
        p_success, n_iter_max, early = g[k + "_kw"]
        early = None if early < 0 else int(early)
        counts = so.score_hypotheses(K, MR, Rts, valid, X, mL, mR, thr)
        raised = str(g[k + "_raised"])
        if raised:
            with pytest.raises(ValueError, match=raised):
                so.ransac_replay(counts, len(X), float(p_success), int(n_iter_max), early)
            continue
        best, best_count, iters, _ = so.ransac_replay(counts, len(X), float(p_success), int(n_iter_max), early)
        assert iters == int(g[k + "_calls"]), k
        if not bool(g[k + "_has_mask"]):
            assert best == -1
            continue
        mask = so.hypothesis_mask(K, MR, Rts[best], X, mL, mR, thr)
        np.testing.assert_array_equal(mask, g[k + "_mask"], err_msg=k)
        assert best_count == int(g[k + "_mask"].sum())


def test_l2_restatement_is_the_l2_distance():
    """The float32 L2 restatement (index-order float32 sum, sqrt) against float64
    arithmetic: same distances to float32 accuracy, same neighbours wherever the float64
    distances are not nearly tied; an exact duplicate wins with the lowest index."""
    rng = np.random.default_rng(0)
    q = rng.normal(size=(50, 96)).astype(np.float32)
    t = rng.normal(size=(80, 96)).astype(np.float32)
    t[7] = t[2]
    q[0] = t[2]
    d32 = so.l2_distance_matrix(q, t)
    d64 = np.sqrt(((q[:, None, :].astype(np.float64) - t[None, :, :].astype(np.float64)) ** 2).sum(-1))
    assert d32.dtype == np.float32 and np.abs(d32 - d64).max() <= 2e-6 * d64.max()
    idx2, dist2 = so.l2_knn2(q, t)
    assert idx2[0, 0] == 2 and idx2[0, 1] == 7 and dist2[0, 0] == 0.0
    order = np.argsort(d64, axis=1, kind="stable")
    gap = np.take_along_axis(d64, order[:, 2:3], 1)[:, 0] - np.take_along_axis(d64, order[:, 1:2], 1)[:, 0]
    clear = (gap > 1e-4) & (np.take_along_axis(d64, order[:, 1:2], 1)[:, 0]
                            - np.take_along_axis(d64, order[:, 0:1], 1)[:, 0] > 1e-4)
    assert clear.sum() > 30
    np.testing.assert_array_equal(idx2[clear], order[clear, :2])
    qi, ti, di = so.bf_match_l2(q, t)
    assert (np.diff(di) >= 0).all() and sorted(qi.tolist()) == list(range(50))
