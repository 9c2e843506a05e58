"""Times the batched front-end under each pair-RANSAC launch scheme (slamhip_tune knob 4:
0 auto, 1 rounds of 256 hypotheses, 2 rounds with the merged tail):
    python tools/tune_ransac_scheme.py [schemes...]   (default: 0 1 2)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel, _lib
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

schemes = [int(v) for v in sys.argv[1:]] or [0, 1, 2]
cam = CameraModel.kitti_like()
for F, nfeat, clutter in ((500, 1500, 0.6), (500, 1500, 0.85), (100, (2000, 8000), 0.6), (66, 1500, 0.6)):
    seq = StereoSequence(cam, n_frames=F, n_features=nfeat, seed=100, clutter=clutter)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
    fe = TrackingFrontend(cam)
    inp = fe.upload(frames)
    for rnd in range(2):
        for v in schemes:
            _lib.lib.slamhip_tune(4, v)
            fe._ws = None
            for _ in range(3):
                fe.run(inp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fe.run(inp)
            torch.cuda.synchronize()
            print(f"F={F} nfeat={nfeat} clutter={clutter} scheme={v}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
    _lib.lib.slamhip_tune(4, 0)
