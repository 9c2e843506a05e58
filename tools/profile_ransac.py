"""Times the two ingredients of the pair RANSAC separately at the front-end's
shape (for rocprofv3): P3P on pairs*256 samples and consensus scoring of
pairs*256 hypotheses against n points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel, kernels
from slam_amd.utils.synth import perturbed_hypotheses

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 499
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
h = pairs * 256
rng = np.random.default_rng(0)
cam = CameraModel.kitti_like()
X = np.column_stack([rng.uniform(-20, 20, n), rng.uniform(-3, 2, n), rng.uniform(5, 80, n)])
Rt = np.hstack([np.eye(3), np.array([[0.01], [0.0], [-0.8]])])
Xh = np.column_stack([X, np.ones(n)])
pl = Xh @ (cam.K @ Rt).T
pl = pl[:, :2] / pl[:, 2:3] + rng.normal(0, 0.4, (n, 2))
Rr = cam.M_right[:, :3] @ Rt[:, :3]
tr = cam.M_right[:, :3] @ Rt[:, 3] + cam.M_right[:, 3]
pr = Xh @ (cam.K @ np.hstack([Rr, tr[:, None]])).T
pr = pr[:, :2] / pr[:, 2:3] + rng.normal(0, 0.4, (n, 2))
bad = rng.random(n) < 0.3
pl[bad] += rng.uniform(-40, 40, (bad.sum(), 2))
idx4 = np.stack([rng.permutation(n)[:4] for _ in range(4096)])
idx4 = np.tile(idx4, (h // 4096 + 1, 1))[:h].astype(np.int32)
Xd = torch.from_numpy(X.astype(np.float32)).cuda()
pld = torch.from_numpy(pl.astype(np.float32)).cuda()
prd = torch.from_numpy(pr.astype(np.float32)).cuda()
ixd = torch.from_numpy(idx4).cuda()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


from slam_amd import _lib
t_p3p, (Rts, valid) = timed(lambda: kernels.p3p_solve_indexed(Xd, pld, ixd, cam.K))
t_sc, counts = timed(lambda: kernels.ransac_score(Xd, pld, prd, Rts, valid, cam.K, cam.M_right, 2.0))
_lib.lib.slamhip_tune(2, 0)          # fp64 only
t_64, counts64 = timed(lambda: kernels.ransac_score(Xd, pld, prd, Rts, valid, cam.K, cam.M_right, 2.0))
_lib.lib.slamhip_tune(2, 1)
print(f"h={h} n={n}: p3p {t_p3p*1e6:.0f} us ({int(valid.sum())} valid), score {t_sc*1e6:.0f} us "
      f"(fp64 only: {t_64*1e6:.0f} us, same counts: {bool((counts == counts64).all())}), "
      f"best count {int(counts.max())}")
