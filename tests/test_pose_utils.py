"""slam_amd/utils/utils.py against the reference's own pose-chain helpers (utils/utils.py:17-141,
run as they are: tests/golden/ref_pose_utils.npz from oracle/make_golden_pose_utils.py)."""
import numpy as np

from slam_amd.utils import utils as pu


def test_pose_chain_helpers_equal_the_references(golden):
    g = golden("ref_pose_utils.npz")
    rel, gt = g["rel"], g["gt"]
    abs_w2c = pu.chain_rel_to_abs_w2c(rel)
    tol = dict(rtol=0, atol=1e-12)
    np.testing.assert_allclose(abs_w2c, g["chain_rel_to_abs_w2c"], **tol)
    np.testing.assert_allclose(pu.relative_chain_to_absolute(rel), g["relative_chain_to_absolute"], **tol)
    np.testing.assert_allclose(pu.centers_from_Rts(abs_w2c), g["centers_from_Rts"], **tol)
    np.testing.assert_allclose(pu.rel_position_err(abs_w2c, gt), g["rel_position_err"], **tol)
    e = pu.compute_position_errs(abs_w2c, gt)
    for k in ("ex", "ey", "ez", "epos"):
        np.testing.assert_allclose(e[k], g[k], **tol)
    np.testing.assert_array_equal(pu.to_hom(abs_w2c), g["to_hom"])
    np.testing.assert_allclose(pu.invert_T(pu.to_hom(abs_w2c)), g["invert_T"], **tol)
    np.testing.assert_allclose(pu.relatives_from_abs_w2c_endpoints(abs_w2c, g["starts"], g["ends"]), g["endpoints"], **tol)
    np.testing.assert_allclose(pu.relatives_from_absolutes_w2c_adjacent(abs_w2c), g["adjacent"], **tol)
    np.testing.assert_allclose(pu.relatives_from_absolutes_w2c_adjacent(abs_w2c, g["kf"]), g["adjacent_kf"], **tol)
    np.testing.assert_allclose(pu.trajectory_rmse(abs_w2c, gt), np.sqrt(np.mean(g["epos"] ** 2)), rtol=1e-12)
    # chaining the adjacent relatives back gives the chain again
    adj = pu.invert_T(pu.relatives_from_absolutes_w2c_adjacent(abs_w2c))[:, :3]
    np.testing.assert_allclose(pu.chain_rel_to_abs_w2c(adj), abs_w2c, rtol=0, atol=1e-10)


def test_rotation_error_is_the_rotation_angle():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(0)
    ang = np.array([0.0, 1e-9, 1e-3, 0.5, 3.0, np.pi - 1e-6])
    axis = rng.normal(size=(len(ang), 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    base = Rotation.from_rotvec(rng.normal(size=(len(ang), 3))).as_matrix()
    est = np.zeros((len(ang), 3, 4))
    gt = np.zeros((len(ang), 3, 4))
    gt[:, :, :3] = base
    est[:, :, :3] = base @ Rotation.from_rotvec(axis * ang[:, None]).as_matrix()
    np.testing.assert_allclose(pu.rot_err_deg(est, gt), np.degrees(ang), rtol=1e-9, atol=1e-10)
    assert pu.chain_rel_to_abs_w2c(np.zeros((0, 3, 4))).shape == (1, 3, 4)
