"""Times the batched front-end and the loop verification with the matrix-core matcher's blocks dealt
to the XCDs pair by pair (slamhip_tune knob 11 = 1) against the plain grid (0):
    python tools/tune_xcd.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel, _lib
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

cam = CameraModel.kitti_like()
for F, nfeat in ((500, 1500), (250, 3000), (100, (2000, 8000)), (66, 1500), (17, 1500)):
    seq = StereoSequence(cam, n_frames=F, n_features=nfeat, seed=100, clutter=0.6)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
    fe = TrackingFrontend(cam)
    inp = fe.upload(frames)
    for rnd in range(3):
        for v in (1, 0):
            _lib.lib.slamhip_tune(11, v)
            for _ in range(3):
                fe.run(inp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fe.run(inp)
            torch.cuda.synchronize()
            print(f"F={F} nfeat={nfeat} xcd_dealing={v}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
    _lib.lib.slamhip_tune(11, 1)
