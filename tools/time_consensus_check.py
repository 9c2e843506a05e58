"""Latency of the loop-closure check called the way run_loop_closure calls it
(loop_closure.py:341-346): one keyframe against its k_best candidates, keyframes of 600
stereo features read from a TrackingDB-shaped object.  Prints the time per
consensus_pnp_match call and per consensus_pnp_matches call of three candidates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel
from slam_amd.backend.optimizers.loop_closure import ConsensusCheck
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

cam = CameraModel.kitti_like()
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 40
step = 2                                    # frames between keyframes: neighbours overlap, so checks succeed
seq = StereoSequence(cam, n_frames=n_kf * step, n_features=1500, seed=7, clutter=0.6)
frames = [seq.frame(k * step) for k in range(n_kf)]
st = TrackingFrontend(cam, n_iter_max=1).run([(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in frames])
keyframes = [(st.frame(k)["links"], st.frame(k)["inlier_des_left"]) for k in range(n_kf)]


class Link:
    __slots__ = ("x_left", "x_right", "y")

    def __init__(self, r):
        self.x_left, self.x_right, self.y = r


class DB:
    def __init__(self):
        self.links = [[Link(r) for r in np.asarray(k[0]).tolist()] for k in keyframes]

    def features(self, kf):
        return keyframes[kf][1]

    def all_frame_links(self, kf):
        return self.links[kf]


db, chk = DB(), ConsensusCheck(cam)
for kf in range(n_kf):
    chk._index_of(db, kf)                   # keyframes on the device
chk.consensus_pnp_match(5, 0, db)
torch.cuda.synchronize()
t0 = time.perf_counter()
n_ok = 0
for n in range(6, n_kf):
    n_ok += chk.consensus_pnp_match(n, n - 6, db)[0]
dt1 = (time.perf_counter() - t0) / (n_kf - 6)
t0 = time.perf_counter()
for n in range(8, n_kf):
    out = chk.consensus_pnp_matches(n, [n - 8, n - 7, n - 6], db)
dt3 = (time.perf_counter() - t0) / (n_kf - 8)
print(f"consensus_pnp_match: {dt1 * 1e3:.3f} ms per call ({n_ok} verified of {n_kf - 6}); "
      f"three candidates in one call: {dt3 * 1e3:.3f} ms")
