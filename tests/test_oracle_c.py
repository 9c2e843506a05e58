"""The C/OpenMP restatement of the front-end (oracle/frontend_oracle.c, the CPU baseline that
bench.py times) against the NumPy restatement and against the reference's own RANSAC loop."""
import numpy as np
import pytest

from oracle import frontend_c as fc
from oracle import pipeline_oracle as po
from oracle import slam_oracle as so


def _cam():
    from slam_amd import CameraModel
    return CameraModel.kitti_like()


def test_p3p_equals_the_numpy_solver():
    cam = _cam()
    rng = np.random.default_rng(3)
    from scipy.spatial.transform import Rotation
    solved = 0
    for _ in range(200):
        X = np.column_stack([rng.uniform(-10, 10, 4), rng.uniform(-3, 3, 4), rng.uniform(5, 40, 4)])
        R = Rotation.from_rotvec(rng.normal(size=3) * 0.05).as_matrix()
        t = rng.normal(size=3) * 0.5
        Xc = X @ R.T + t
        p = Xc @ cam.K.T
        img = p[:, :2] / p[:, 2:3] + rng.normal(size=(4, 2)) * 0.3
        a = so.p3p_solve(X, img, cam.K)
        b = fc.p3p_solve(X, img, cam.K)
        assert (a is None) == (b is None)
        if a is not None:
            solved += 1
            np.testing.assert_allclose(b, a, rtol=0, atol=1e-7)
    assert solved > 150


def test_ransac_loop_matches_reference_loop(golden):
    """tests/golden/ref_ransac_loop.npz is the reference's own ransac_pnp_pair1 run with a prepared
    pose list where solve_pnp stands: same mask, same number of iterations, and the math.log(0)
    of an all-inlier hypothesis where the reference raised."""
    g = golden("ref_ransac_loop.npz")
    K, MR, thr = g["K"], g["M_right"], float(g["pixel_thresh"])
    for name in g["names"]:
        k = str(name)
        p_success, n_iter_max, early = g[k + "_kw"]
        kw = dict(p_success=float(p_success), n_iter_max=int(n_iter_max), early_inliers=None if early < 0 else int(early),
                  hyps=g[k + "_Rts"], hyp_valid=g[k + "_valid"])
        if str(g[k + "_raised"]):
            with pytest.raises(ValueError):
                fc.ransac_loop(K, MR, g[k + "_X"], g[k + "_mL"], g[k + "_mR"], thr, **kw)
            continue
        r = fc.ransac_loop(K, MR, g[k + "_X"], g[k + "_mL"], g[k + "_mR"], thr, **kw)
        assert r["iterations"] == int(g[k + "_calls"]), k
        if not bool(g[k + "_has_mask"]):
            assert r["best_index"] == -1
            continue
        np.testing.assert_array_equal(r["mask"], g[k + "_mask"], err_msg=k)


def test_frontend_equals_the_numpy_restatement():
    """Whole sequence: stereo matches, triangulated points, temporal matches, RANSAC masks and
    iteration counts identical to oracle/pipeline_oracle.py with the same idx4 samples; refined
    poses within 1e-6."""
    from slam_amd.utils.synth import StereoSequence
    cam = _cam()
    F = 5
    seq = StereoSequence(cam, n_frames=F, n_features=300, seed=11, clutter=0.5)
    frames = []
    for k in range(F):
        f = seq.frame(k)
        frames.append((f.kp_left, f.kp_right, f.des_left, f.des_right))
    rng = np.random.default_rng(5)
    n_iter_max = 400
    stereo, _ = po.run(cam, frames, n_iter_max=1, samples=[np.zeros((1, 4), np.int64)] * (F - 1), refine=False)
    samples = []
    for p in range(F - 1):
        n = len(stereo[p]["left_idx"])        # matches of pair p = stereo inliers of frame p
        samples.append(np.stack([rng.choice(n, 4, replace=False) for _ in range(n_iter_max)]))
    stereo, pairs = po.run(cam, frames, samples=samples, n_iter_max=n_iter_max)
    c = fc.frontend_run(cam, frames, samples=samples, n_iter_max=n_iter_max)
    off = c["offL"]
    for f in range(F):
        n = c["st_count"][f]
        assert n == len(stereo[f]["left_idx"]) and n > 50
        s = slice(off[f], off[f] + n)
        np.testing.assert_array_equal(c["st_left"][s], stereo[f]["left_idx"])
        np.testing.assert_array_equal(c["st_right"][s], stereo[f]["right_idx"])
        np.testing.assert_array_equal(c["st_dist"][s], stereo[f]["dist"])
        np.testing.assert_allclose(c["st_xyz"][s], stereo[f]["xyz"], rtol=2e-6, atol=0)
    for p in range(F - 1):
        n = c["tm_count"][p]
        assert n == pairs[p]["n_matches"]
        s = slice(off[p], off[p] + n)
        np.testing.assert_array_equal(c["tm_q"][s], pairs[p]["query_idx"])
        np.testing.assert_array_equal(c["tm_t"][s], pairs[p]["train_idx"])
        assert c["info"][p, 2] == pairs[p]["iterations"], p
        assert c["info"][p, 0] == pairs[p]["best_index"], p
        np.testing.assert_array_equal(c["mask"][s].astype(bool), pairs[p]["mask"])
        np.testing.assert_allclose(c["Rt"][p], pairs[p]["Rt"], rtol=0, atol=1e-6)
