"""Phases of the fused launches of a one-pair call (library built by tools/build_stamps.sh, picked
through SLAMHIP_LIB_PATH): microseconds between the phase clocks of each launch's first workgroup."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SLAMHIP_LIB_PATH", os.path.join(ROOT, "slam_amd", "libslamhip_stamps.so"))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
from slam_amd import CameraModel, _lib
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

nfeat = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
cam = CameraModel.kitti_like()
seq = StereoSequence(cam, n_frames=2, n_features=nfeat, seed=3, clutter=0.6)
frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(2))]
fe = TrackingFrontend(cam)
inp = fe.upload(frames)
names = {(0, 1): "stereo tail: start -> splits folded, keys in LDS", (1, 2): "  sort", (2, 3): "  filter + compaction",
         (3, 4): "  triangulation", (4, 5): "  descriptor gather",
         (8, 9): "temporal tail: start -> splits folded", (9, 10): "  sort", (10, 11): "  gather",
         (16, 17): "finish: start -> replay done", (17, 18): "  winner's mask", (18, 19): "  refinement"}
acc = {k: [] for k in names}
out = (C.c_ulonglong * 64)()
for rep in range(30):
    fe.run(inp)
    torch.cuda.synchronize()
    assert _lib.lib.slamhip_debug_stamps(out) == 0
    if rep >= 5:
        for (a, b) in names:
            acc[(a, b)].append((out[b] - out[a]) / 100.0)
for k, label in names.items():
    print(f"{label:55s} {np.median(acc[k]):6.2f} us")
