"""Host-side cost of OnlineFrontend.push(frame).pose(): wall time per frame and, with an
argument, the cProfile table of the loop."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel
from slam_amd.backend.tracking.online import OnlineFrontend
from slam_amd.utils.synth import StereoSequence

cam = CameraModel.kitti_like()
n = 40
seq = StereoSequence(cam, n_frames=n, n_features=1500, seed=3, clutter=0.63, landmark_density=0.58)
frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(n))]
on = OnlineFrontend(cam)
for f in frames[:5]:
    r = on.push(*f)
r.pose()

def loop(sync_pose=True):
    on.reset()
    on.push(*frames[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in frames[1:]:
        r = on.push(*f)
        if sync_pose:
            r.pose()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n - 1) * 1e6

for _ in range(3):
    print(f"push + pose: {loop():.1f} us per frame; push only (one sync at the end): {loop(False):.1f}")
if len(sys.argv) > 1:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        loop()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
