"""Where the host-inclusive front-end pass spends its time: concatenation into page-locked
buffers + H2D, the device pass, D2H of every result array."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

F = int(sys.argv[1]) if len(sys.argv) > 1 else 500
cam = CameraModel.kitti_like()
seq = StereoSequence(cam, n_frames=F, n_features=1500, seed=100, clutter=0.6)
frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
fe = TrackingFrontend(cam)
for _ in range(2):
    fe.run(fe.upload(frames)).to_host()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    inp = fe.upload(frames); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res = fe.run(inp); torch.cuda.synchronize(); t3 = time.perf_counter()
    h = res.to_host(); t4 = time.perf_counter()
    print(f"upload (host part) {1e3*(t1-t0):.2f} ms, H2D drain {1e3*(t2-t1):.2f} ms, pass {1e3*(t3-t2):.2f} ms, "
          f"to_host {1e3*(t4-t3):.2f} ms, total {1e3*(t4-t0):.2f} ms -> {F/(t4-t0):.0f} frames/s", flush=True)
