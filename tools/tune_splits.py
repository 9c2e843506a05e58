"""Times the batched front-end with the matcher's number of train splits forced
(slamhip_tune knob 1; 0 = the plan's own choice)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slam_amd import CameraModel, _lib
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

cam = CameraModel.kitti_like()
for F, nfeat in ((60, (2000, 8000)), (500, 1500), (64, 1500), (8, 1500)):
    seq = StereoSequence(cam, n_frames=F, n_features=nfeat, seed=100, clutter=0.6)
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
    fe = TrackingFrontend(cam)
    inp = fe.upload(frames)
    for splits in (0, 1, 2, 4, 8, 16):
        _lib.lib.slamhip_tune(1, splits)
        for _ in range(2):
            fe.run(inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            fe.run(inp)
        torch.cuda.synchronize()
        print(f"F={F} nfeat={nfeat} splits={splits}: {(time.perf_counter() - t0) / 8 * 1e3:.3f} ms", flush=True)
    _lib.lib.slamhip_tune(1, 0)
