"""One chunked host-inclusive run after warm-up, for a copy + kernel time line under
rocprofv3 --kernel-trace --memory-copy-trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slam_amd import CameraModel
from slam_amd.backend.tracking.feature_cache import PackedFrames
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

F, chunk = 500, int(sys.argv[1]) if len(sys.argv) > 1 else 128
mode = sys.argv[2] if len(sys.argv) > 2 else "chunked"
cam = CameraModel.kitti_like()
seq = StereoSequence(cam, n_frames=F, n_features=1500, seed=100, clutter=0.6)
frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
packed = PackedFrames.from_frames(frames).pin()
fe = TrackingFrontend(cam)
fn = (lambda: fe.run_streamed(packed, chunk_pairs=chunk)) if mode == "streamed" else \
     (lambda: fe.run_chunked(packed, chunk_pairs=chunk))
for _ in range(4):
    fn()
torch.cuda.synchronize()
time.sleep(0.05)          # a visible gap in the trace before the run that is looked at
t0 = time.perf_counter()
fn()
torch.cuda.synchronize()
print(f"{mode} run: {(time.perf_counter() - t0) * 1e3:.2f} ms")
