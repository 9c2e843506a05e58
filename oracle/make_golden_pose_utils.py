"""Generates tests/golden/ref_pose_utils.npz: outputs of the reference's own pose-chain helpers
(utils/utils.py:17-141, pure NumPy bodies) on a seeded random pose chain.  Run in the build
container only:  python oracle/make_golden_pose_utils.py
The reference module imports cv2 / gtsam at the top; they are absent and stand as empty
placeholders for the import only (oracle/make_golden.py::import_reference); rot_err_deg calls
cv2.Rodrigues and is therefore not recorded."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.make_golden import import_reference, OUT  # noqa: E402


def main():
    import_reference()
    import matplotlib
    matplotlib.use("Agg")
    from slam.utils import utils as ru  # type: ignore
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(42)
    n = 40
    rel = np.zeros((n, 3, 4))
    rel[:, :, :3] = Rotation.from_rotvec(rng.normal(size=(n, 3)) * 0.05).as_matrix()
    rel[:, :, 3] = rng.normal(size=(n, 3)) * [0.05, 0.02, 0.8]
    abs_w2c = ru.chain_rel_to_abs_w2c(rel)
    gt = ru.chain_rel_to_abs_w2c(rel + rng.normal(size=rel.shape) * 1e-3 * (np.arange(12).reshape(3, 4) % 4 == 3))
    errs = ru.compute_position_errs(abs_w2c, gt)
    kf = np.array([0, 3, 9, 17, 30, 40])
    starts, ends = np.array([1, 5, 12]), np.array([0, 2, 40])
    out = dict(
        rel=rel, gt=gt, kf=kf, starts=starts, ends=ends,
        chain_rel_to_abs_w2c=abs_w2c,
        relative_chain_to_absolute=ru.relative_chain_to_absolute(rel),
        centers_from_Rts=ru.centers_from_Rts(abs_w2c),
        rel_position_err=ru.rel_position_err(abs_w2c, gt),
        ex=errs["ex"], ey=errs["ey"], ez=errs["ez"], epos=errs["epos"],
        to_hom=ru.to_hom(abs_w2c), invert_T=ru.invert_T(ru.to_hom(abs_w2c)),
        endpoints=ru.relatives_from_abs_w2c_endpoints(abs_w2c, starts, ends),
        adjacent=ru.relatives_from_absolutes_w2c_adjacent(abs_w2c),
        adjacent_kf=ru.relatives_from_absolutes_w2c_adjacent(abs_w2c, kf),
    )
    np.savez_compressed(os.path.join(OUT, "ref_pose_utils.npz"), **out)
    print("wrote ref_pose_utils.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
