"""One-off soak of the loop-closure pair verification for run-to-run determinism:
random keyframe sets (3..40 keyframes of 40..900 links), all pairs, both matcher modes,
random gates; verify() twice and once more with every combination of hypothesis and point
screened (tune knob 2 = 2) instead of the sparse counting scheme, every output identical.
python tools/soak_verify.py [cases [seed]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel, _lib
from slam_amd.backend.optimizers.loop_verify import PairVerifier
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
cam = CameraModel.kitti_like()
bad = 0
for case in range(cases):
    KF = int(rng.integers(3, 41))
    step = int(rng.choice([1, 3, 17]))
    lo = int(rng.integers(100, 1500))
    seq = StereoSequence(cam, n_frames=KF * step, n_features=(lo, lo + 600), seed=int(rng.integers(1 << 30)), clutter=0.6)
    frames = []
    for k in range(KF):
        f = seq.frame(k * step)
        frames.append((f.kp_left, f.kp_right, f.des_left, f.des_right))
    st = TrackingFrontend(cam, n_iter_max=1).run(frames)
    keyframes = [(st.frame(k)["links"], st.frame(k)["inlier_des_left"]) for k in range(KF)]
    pairs = np.array([(i, n) for n in range(KF) for i in range(n)], dtype=np.int64)
    ver = PairVerifier(cam, seed=case, min_inliers_loop=int(rng.choice([20, 60, 200])),
                       n_iter_max=int(rng.choice([100, 1500])),
                       frame_cross_check=bool(rng.integers(2))).set_keyframes(keyframes)
    a, b = ver.verify(pairs), ver.verify(pairs)
    _lib.lib.slamhip_tune(2, 2)
    c = ver.verify(pairs)
    _lib.lib.slamhip_tune(2, 1)
    ok = True
    for other in (b, c):
        for k in ("ok", "Rt", "Rt_ransac", "n_matches", "n_inliers", "iterations", "out_off"):
            ok = ok and np.array_equal(a[k], other[k], equal_nan=True)
        for p in np.flatnonzero(a["ok"])[:20]:
            ma, mb = a.pair_matches(int(p)), other.pair_matches(int(p))
            ok = ok and all(np.array_equal(ma[k], mb[k]) for k in ma)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: KF={KF} step={step}", flush=True)
    if case % 5 == 0:
        print(f"case {case} ok so far (KF={KF}, {len(pairs)} pairs, {int(a['ok'].sum())} verified), mismatches {bad}", flush=True)
print(f"done: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
