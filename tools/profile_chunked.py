"""Host-inclusive pass (page-locked host arrays -> result arrays on the host) by scheme, and the
host-side time line of one chunked run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel
from slam_amd.backend.tracking.feature_cache import PackedFrames
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

F = int(sys.argv[1]) if len(sys.argv) > 1 else 500
cam = CameraModel.kitti_like()
seq = StereoSequence(cam, n_frames=F, n_features=1500, seed=100, clutter=0.6)
frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
packed = PackedFrames.from_frames(frames).pin()
fe = TrackingFrontend(cam)


def timed(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print(f"one pass: {timed(lambda: fe.run(fe.upload(packed)).to_host()) * 1e3:.2f} ms")
for chunk in (64, 100, 128, 167, 256):
    print(f"chunks of {chunk}: {timed(lambda: fe.run_chunked(packed, chunk_pairs=chunk)) * 1e3:.2f} ms", flush=True)
if hasattr(fe, "run_streamed"):
    for chunk in (32, 64, 100, 128, 167, 256):
        print(f"streamed, chunks of {chunk}: {timed(lambda: fe.run_streamed(packed, chunk_pairs=chunk)) * 1e3:.2f} ms",
              flush=True)
