"""The online (frame-at-a-time) use: one front-end call per new frame, two frames per
call.  Prints the time per call; run under rocprofv3 --kernel-trace --stats for the
per-kernel durations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slam_amd import CameraModel
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

nfeat = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
# further arguments: knob=value pairs for slamhip_tune (e.g. 14=0: the launches of a long call)
from slam_amd import _lib
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    assert _lib.lib.slamhip_tune(int(k), int(v)) == 0
cam = CameraModel.kitti_like()
seq = StereoSequence(cam, n_frames=2, n_features=nfeat, seed=3, clutter=0.6)
frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(2))]
fe = TrackingFrontend(cam)
inp = fe.upload(frames)
for _ in range(5):
    fe.run(inp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    fe.run(inp)
torch.cuda.synchronize()
print(f"{sys.argv[2:]} one pair of {nfeat}-feature frames: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per call")
