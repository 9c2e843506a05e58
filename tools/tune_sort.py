"""Times the batched front-end with the sort algorithm forced (slamhip_tune knob 5:
0 = the library's choice, 1 = rank counting, 2 = bitonic network)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel, _lib, kernels
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

cam = CameraModel.kitti_like()
for F, nfeat in ((60, (2000, 8000)), (500, 1500), (64, 1500), (16, 1500), (8, 1500), (2, 1500), (2, 8000)):
    seq = StereoSequence(cam, n_frames=F, n_features=nfeat, seed=100, clutter=0.6)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
    fe = TrackingFrontend(cam)
    inp = fe.upload(frames)
    for mode in (0, 1, 2, 0):
        _lib.lib.slamhip_tune(5, mode)
        for _ in range(3):
            fe.run(inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fe.run(inp)
        torch.cuda.synchronize()
        print(f"F={F} nfeat={nfeat} sort={mode}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
    _lib.lib.slamhip_tune(5, 0)
