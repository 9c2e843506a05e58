"""Runs the all-pairs keyframe verification once warm (for rocprofv3) and prints
host-side timings of its parts:  python tools/profile_loop_verify.py [keyframes] [matcher variant]
(the variant is slamhip_tune knob 0; default: the library's choice)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel
from slam_amd.backend.optimizers.loop_verify import PairVerifier
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

KF = int(sys.argv[1]) if len(sys.argv) > 1 else 270
if len(sys.argv) > 2:
    from slam_amd import _lib
    _lib.lib.slamhip_tune(0, int(sys.argv[2]))
for kv in os.environ.get("SLAMHIP_PROFILE_TUNE", "").split():     # knob=value pairs for slamhip_tune
    from slam_amd import _lib
    k, v = kv.split("=")
    assert _lib.lib.slamhip_tune(int(k), int(v)) == 0
cam = CameraModel.kitti_like()
seq = StereoSequence(cam, n_frames=KF * 17, n_features=1500, seed=7, clutter=0.6)
frames = []
for k in range(KF):
    f = seq.frame(k * 17)
    frames.append((f.kp_left, f.kp_right, f.des_left, f.des_right))
st = TrackingFrontend(cam, n_iter_max=1).run(frames)
keyframes = [(st.frame(k)["links"], st.frame(k)["inlier_des_left"]) for k in range(KF)]
pairs = np.array([(i, n) for n in range(KF) for i in range(n)], dtype=np.int64)
ver = PairVerifier(cam, seed=0).set_keyframes(keyframes)
ver.verify(pairs[:64])
torch.cuda.synchronize()
t0 = time.perf_counter()
res = ver.verify(pairs)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
t0 = time.perf_counter()
vm = res.verified_matches()
torch.cuda.synchronize()
dt_vm = time.perf_counter() - t0
t0 = time.perf_counter()
res2 = ver.verify(pairs)
torch.cuda.synchronize()
dt2 = time.perf_counter() - t0
print(f"verified_matches: {dt_vm*1e3:.1f} ms ({int(vm['off'][-1])} rows); second full verify: {dt2*1e3:.1f} ms")
print(f"{len(pairs)} pairs of {KF} keyframes: {dt*1e3:.1f} ms, {len(pairs)/dt:.0f} pairs/s, "
      f"{int(res['ok'].sum())} verified, mean iterations {res['iterations'].mean():.0f}")
