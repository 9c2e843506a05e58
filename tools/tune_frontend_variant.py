"""Times the batched front-end with the matcher kernel variant forced (slamhip_tune knob 0):
    python tools/tune_frontend_variant.py [variants...]   (default: -1 10 18 9)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel, _lib
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

variants = [int(v) for v in sys.argv[1:]] or [-1, 10, 18, 9]
cam = CameraModel.kitti_like()
for F, nfeat in ((500, 1500), (60, (2000, 8000)), (8, 1500), (2, 1500)):
    seq = StereoSequence(cam, n_frames=F, n_features=nfeat, seed=100, clutter=0.6)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
    fe = TrackingFrontend(cam)
    inp = fe.upload(frames)
    for rnd in range(2):
        for v in variants:
            _lib.lib.slamhip_tune(0, v)
            fe._ws = None
            for _ in range(3):
                fe.run(inp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fe.run(inp)
            torch.cuda.synchronize()
            print(f"F={F} nfeat={nfeat} variant={v}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
    _lib.lib.slamhip_tune(0, -1)
