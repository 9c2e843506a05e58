"""Runs the batched front-end a few times on synthetic frames (for rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

F = int(sys.argv[1]) if len(sys.argv) > 1 else 500
nfeat = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
if len(sys.argv) > 3:                      # feature count per image drawn from [nfeat, argv[3]]
    nfeat = (nfeat, int(sys.argv[3]))
cam = CameraModel.kitti_like()
seq = StereoSequence(cam, n_frames=F, n_features=nfeat, seed=100, clutter=0.6)
frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
fe = TrackingFrontend(cam)
inp = fe.upload(frames)
for _ in range(3):
    fe.run(inp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    res = fe.run(inp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"{F} frames x {nfeat} features: {dt*1e3:.3f} ms per pass, {F/dt:.0f} frames/s")
