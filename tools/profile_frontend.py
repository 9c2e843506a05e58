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
# SLAMHIP_PROFILE_SHAPE=readme: the reference's published KITTI-05 statistics (~565 temporal
# matches per frame, ~65 % of them PnP inliers) instead of 41 % inliers
shape = dict(clutter=0.63, landmark_density=0.58) if os.environ.get("SLAMHIP_PROFILE_SHAPE") == "readme" else dict(clutter=0.6)
seq = StereoSequence(cam, n_frames=F, n_features=nfeat, seed=100, **shape)
frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
fe = TrackingFrontend(cam)
if os.environ.get("SLAMHIP_PROFILE_VARIANT"):          # matcher kernel variant forced (tune knob 0)
    from slam_amd import _lib
    _lib.lib.slamhip_tune(0, int(os.environ["SLAMHIP_PROFILE_VARIANT"]))
for kv in os.environ.get("SLAMHIP_PROFILE_TUNE", "").split():     # knob=value pairs for slamhip_tune
    from slam_amd import _lib
    k, v = kv.split("=")
    assert _lib.lib.slamhip_tune(int(k), int(v)) == 0
inp = fe.upload(frames)
for _ in range(3):
    fe.run(inp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    res = fe.run(inp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
info = res.to_host()["ransac_info"].reshape(-1, 8)
print(f"{F} frames x {nfeat} features: {dt*1e3:.3f} ms per pass, {F/dt:.0f} frames/s; "
      f"{info[:, 5].mean():.0f} temporal matches, {(info[:, 1] / np.maximum(info[:, 5], 1)).mean():.2f} inliers, "
      f"{info[:, 2].mean():.0f} RANSAC iterations on average")
