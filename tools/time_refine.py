"""Duration of the pose-refinement launch against the number of problems (the same data set and
start for every problem: what one problem costs alone and what many cost together)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel, kernels

cam = CameraModel.kitti_like()
rng = np.random.default_rng(0)
n = 600
Z = rng.uniform(6, 60, n)
X = np.column_stack([rng.uniform(-0.8, 0.8, n) * Z, rng.uniform(-0.2, 0.2, n) * Z, Z]).astype(np.float32)
a = 0.01
R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
t = np.array([0.02, -0.01, -0.8])
Xc = X @ R.T + t
pix = (np.column_stack([cam.K[0, 0] * Xc[:, 0] / Xc[:, 2] + cam.K[0, 2], cam.K[1, 1] * Xc[:, 1] / Xc[:, 2] + cam.K[1, 2]])
       + rng.normal(0, 0.5, (n, 2))).astype(np.float32)
mask = (rng.random(n) < 0.65).astype(np.uint8)
start = np.hstack([R, (t + np.array([0.03, 0.01, -0.05]))[:, None]])       # a RANSAC-grade start
for P in (1, 64, 128, 256, 499, 512, 1024, 2048):
    init = np.repeat(start[None], P, axis=0)
    args = (kernels.to_device(X, torch.float32), kernels.to_device(pix, torch.float32), kernels.to_device(mask, torch.uint8),
            kernels.to_device(init, torch.float64), cam.K)
    for _ in range(5):
        out, info = kernels.refine_pose_from(*args)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        out, info = kernels.refine_pose_from(*args)
    ev1.record()
    torch.cuda.synchronize()
    print(f"{P:5d} problems: {ev0.elapsed_time(ev1) / 20 * 1e3:7.1f} us per launch, {info[0, 1].item():.0f} iterations, "
          f"cost {info[0, 0].item():.3f}", flush=True)
print("one problem against the iteration cap (0: one pass over the points and no solve):")
args1 = (kernels.to_device(X, torch.float32), kernels.to_device(pix, torch.float32), kernels.to_device(mask, torch.uint8),
         kernels.to_device(start[None], torch.float64), cam.K)
for cap in (0, 1, 2, 3, 50):
    for _ in range(5):
        kernels.refine_pose_from(*args1, max_iters=cap)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(50):
        out, info = kernels.refine_pose_from(*args1, max_iters=cap)
    ev1.record()
    torch.cuda.synchronize()
    print(f"cap {cap:2d}: {ev0.elapsed_time(ev1) / 50 * 1e3:6.1f} us, {info[0, 1].item():.0f} iterations", flush=True)
