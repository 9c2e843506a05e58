"""Consensus counting of pair verification with and without the sparse scheme (slamhip_tune knob 2:
2 = never, 3 = always, 1 = the library's choice) on the all-pairs keyframe verification and on
the batched front-end; results must be identical.   python tools/tune_count_sparse.py [keyframes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel, _lib
from slam_amd.backend.optimizers.loop_verify import PairVerifier
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

KF = int(sys.argv[1]) if len(sys.argv) > 1 else 270
cam = CameraModel.kitti_like()
seq = StereoSequence(cam, n_frames=KF * 17, n_features=1500, seed=7, clutter=0.6)
frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k * 17) for k in range(KF))]
st = TrackingFrontend(cam, n_iter_max=1).run(frames)
keyframes = [(st.frame(k)["links"], st.frame(k)["inlier_des_left"]) for k in range(KF)]
pairs = np.array([(i, n) for n in range(KF) for i in range(n)], dtype=np.int64)
ver = PairVerifier(cam, seed=0).set_keyframes(keyframes)
keys = ("ok", "n_matches", "n_inliers", "iterations", "Rt", "Rt_ransac")
ref = None
for knob in (2, 3, 1, 2, 3):
    _lib.lib.slamhip_tune(2, knob)
    ver.verify(pairs[:64])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        res = ver.verify(pairs)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    got = {k: np.array(res[k]).copy() for k in keys}
    if ref is None:
        ref = got
    same = all(np.array_equal(ref[k], got[k]) for k in keys)
    print(f"verify {len(pairs)} pairs, knob 2 = {knob}: {best*1e3:.2f} ms, same results: {same}", flush=True)

seq = StereoSequence(cam, n_frames=500, n_features=1500, seed=100, clutter=0.6)
frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(500))]
fe = TrackingFrontend(cam)
inp = fe.upload(frames)
ref = None
for knob in (2, 3, 2, 3):
    _lib.lib.slamhip_tune(2, knob)
    for _ in range(3):
        fe.run(inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        res = fe.run(inp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    h = res.to_host()
    got = {k: h[k].copy() for k in ("ransac_info", "pose_ransac", "pose_refined", "tm_mask")}
    if ref is None:
        ref = got
    same = all(np.array_equal(ref[k], got[k]) for k in got)
    print(f"front-end 500 x 1500, knob 2 = {knob}: {dt*1e3:.3f} ms per pass, same results: {same}", flush=True)
_lib.lib.slamhip_tune(2, 1)
