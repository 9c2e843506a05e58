"""GPU parity tests: stereo filter (exact) and DLT triangulation (1e-6 relative
in fp64, 1e-4 in fp32 are the stated bars; the asserts are tighter)."""
import numpy as np
import pytest
import torch

from oracle import slam_oracle as so

pytestmark = pytest.mark.gpu


def _cam(golden):
    g = golden("ref_camera.npz")
    from slam_amd import CameraModel
    cam = CameraModel.from_projections(g["P0"], g["P1"])
    for name in ("K", "M_left", "M_right", "P_left", "P_right"):
        np.testing.assert_array_equal(getattr(cam, name), g[name])
    return cam


def test_filter_decisions_match_reference(golden):
    from slam_amd import kernels
    g = golden("ref_stereo_filter.npz")
    cam = _cam(golden)
    keep, xyz = kernels.stereo_filter_triangulate(g["kp_left"], g["kp_right"], g["qidx"], g["tidx"],
                                                  cam.P_left, cam.P_right, float(g["pixel_thresh"]),
                                                  float(g["disparity_min"]))
    keep = keep.cpu().numpy()
    np.testing.assert_array_equal(keep, g["keep"])
    assert (xyz.cpu().numpy()[keep == 0] == 0).all()


def test_triangulation_fp32_and_fp64_vs_oracle(golden):
    from slam_amd import kernels
    g = golden("oracle_dlt_cv.npz")
    _, x32 = kernels.stereo_filter_triangulate(g["kp_left"], g["kp_right"], None, None,
                                               g["P_left"], g["P_right"], do_filter=False)
    x32 = x32.cpu().numpy()
    assert x32.dtype == np.float32
    rel32 = np.abs(x32 - g["X32"]) / np.abs(g["X32"]).max(axis=1, keepdims=True)
    assert rel32.max() < 1e-4, rel32.max()          # stated bar
    assert rel32.max() < 2e-6, rel32.max()          # what it actually achieves (fp32 ulps)
    _, x64 = kernels.stereo_filter_triangulate(g["kp_left"].astype(np.float64),
                                               g["kp_right"].astype(np.float64), None, None,
                                               g["P_left"], g["P_right"], do_filter=False, fp64=True)
    x64 = x64.cpu().numpy()
    rel64 = np.abs(x64 - g["X64"]) / np.abs(g["X64"]).max(axis=1, keepdims=True)
    assert rel64.max() < 1e-6, rel64.max()          # stated bar
    assert rel64.max() < 1e-10, rel64.max()


def test_triangulation_vs_reference_own_dlt(golden):
    from slam_amd import kernels
    g = golden("ref_dlt_own.npz")
    cam = _cam(golden)
    _, x64 = kernels.stereo_filter_triangulate(g["kp_left"].astype(np.float64),
                                               g["kp_right"].astype(np.float64), None, None,
                                               cam.P_left, cam.P_right, do_filter=False, fp64=True)
    rel = np.abs(x64.cpu().numpy() - g["X"]) / np.abs(g["X"]).max(axis=1, keepdims=True)
    assert rel.max() < 1e-9, rel.max()


def test_triangulation_hard_geometry():
    """Far points (sub-pixel disparity), wide range of depths, large batch,
    exact projections (rank-deficient A: smallest singular value ~ 0)."""
    from slam_amd import kernels, CameraModel
    cam = CameraModel.kitti_like()
    rng = np.random.default_rng(5)
    n = 20000
    X = np.column_stack([rng.uniform(-40, 40, n), rng.uniform(-5, 5, n), 10 ** rng.uniform(0.3, 3.0, n)])
    Xh = np.column_stack([X, np.ones(n)])
    pl, pr = Xh @ cam.P_left.T, Xh @ cam.P_right.T
    uvl = pl[:, :2] / pl[:, 2:3]
    uvr = pr[:, :2] / pr[:, 2:3]
    _, x64 = kernels.stereo_filter_triangulate(uvl, uvr, None, None, cam.P_left, cam.P_right,
                                               do_filter=False, fp64=True)
    ref = so.triangulate_cv(cam.P_left, cam.P_right, uvl, uvr, out_fp64=True)
    x64 = x64.cpu().numpy()
    # noise-free: both recover the point; compare to truth scaled by depth conditioning
    cond = (X[:, 2] ** 2) / (707.0912 * 0.5372)          # dZ/d(disparity)
    assert (np.abs(x64 - X).max(axis=1) < 1e-9 * np.maximum(cond, 1.0) * 1e3).all()
    assert np.isfinite(x64).all() and np.isfinite(ref).all()
    noisy_l = (uvl + rng.normal(0, 0.5, uvl.shape)).astype(np.float32)
    noisy_r = (uvr + rng.normal(0, 0.5, uvr.shape)).astype(np.float32)
    near = X[:, 2] < 120
    _, y64 = kernels.stereo_filter_triangulate(noisy_l[near].astype(np.float64),
                                               noisy_r[near].astype(np.float64), None, None,
                                               cam.P_left, cam.P_right, do_filter=False, fp64=True)
    ref = so.triangulate_cv(cam.P_left, cam.P_right, noisy_l[near].astype(np.float64),
                            noisy_r[near].astype(np.float64), out_fp64=True)
    rel = np.abs(y64.cpu().numpy() - ref) / np.abs(ref).max(axis=1, keepdims=True)
    assert rel.max() < 1e-6, rel.max()
    assert np.median(rel) < 1e-12


def test_triangulation_facade(golden):
    from slam_amd import Triangulation, DMatch, KeyPoint
    g = golden("ref_stereo_filter.npz")
    cam = _cam(golden)
    tri = Triangulation(cam, disparity_min=1.2, pixel_thresh=4.0)
    kpl = [KeyPoint(*p) for p in g["kp_left"]]
    kpr = [KeyPoint(*p) for p in g["kp_right"]]
    matches = [DMatch(q, t, 0, 1.0) for q, t in zip(g["qidx"], g["tidx"])]
    inl, outl = tri.create_inlier_outlier_matches(matches, kpl, kpr)
    assert [m.queryIdx for m in inl] == g["qidx"][g["keep"] == 1].tolist()
    assert [m.queryIdx for m in outl] == g["qidx"][g["keep"] == 0].tolist()
    X = tri.triangulate_opencv(kpl, kpr, inl)
    qi = np.array([m.queryIdx for m in inl]); ti = np.array([m.trainIdx for m in inl])
    ref = so.triangulate_cv(cam.P_left, cam.P_right, g["kp_left"][qi], g["kp_right"][ti])
    assert X.dtype == np.float32 and X.shape == ref.shape
    rel = np.abs(X - ref) / np.abs(ref).max(axis=1, keepdims=True)
    assert rel.max() < 1e-4
    X2 = tri.triangulate_matches_batched(kpl, kpr, inl)
    ref2 = so.triangulate_own(cam.P_left, cam.P_right, g["kp_left"][qi], g["kp_right"][ti])
    rel2 = np.abs(X2 - ref2) / np.abs(ref2).max(axis=1, keepdims=True)
    assert rel2.max() < 1e-6
    assert tri.create_inlier_outlier_matches([], kpl, kpr) == ([], [])


def test_matches_naming_missing_keypoints_are_rejected(golden):
    """Index -1 (the matcher's "no neighbour") or a row past the end: keep 0, xyz 0,
    nothing read; the other matches are unaffected."""
    from slam_amd import kernels, CameraModel
    g = golden("ref_stereo_filter.npz")
    cam = CameraModel.kitti_like()
    q, t = g["qidx"].copy(), g["tidx"].copy()
    ref_keep, ref_xyz = kernels.stereo_filter_triangulate(g["kp_left"], g["kp_right"], q, t, cam.P_left,
                                                          cam.P_right, 4.0, 1.2)
    bad = [1, 7, 20]
    t[bad[0]] = -1
    q[bad[1]] = len(g["kp_left"])
    t[bad[2]] = 2 ** 31 - 1
    keep, xyz = kernels.stereo_filter_triangulate(g["kp_left"], g["kp_right"], q, t, cam.P_left, cam.P_right,
                                                  4.0, 1.2)
    keep, xyz, ref_keep, ref_xyz = (a.cpu().numpy() for a in (keep, xyz, ref_keep, ref_xyz))
    assert not keep[bad].any() and not xyz[bad].any()
    ok = np.ones(len(q), bool); ok[bad] = False
    np.testing.assert_array_equal(keep[ok], ref_keep[ok])
    np.testing.assert_array_equal(xyz[ok], ref_xyz[ok])
