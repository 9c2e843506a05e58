"""Soak of pair verification against the CPU restatement of the reference's loop on GIVEN idx4
sequences (slamhip_verify_args.d_samples): random keyframe sets (3..8 keyframes, 40..900 links),
all ordered pairs, both matcher modes, random gates and iteration caps.  Per pair the CPU loop
(oracle/pipeline_oracle.consensus_pnp_match) runs the same sequence with the device's minimal
solutions as hypotheses: winner, iteration count, requirement, consensus count and mask must be
the device's exactly (a point within 1e-6 px of the threshold may differ).
python tools/soak_verify_samples.py [cases [seed]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from _verify_helpers import _assert_same_consensus, _cpu_loop_on_device_poses, _idx4_table
from slam_amd import CameraModel
from slam_amd.backend.optimizers.loop_verify import PairVerifier
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
cam = CameraModel.kitti_like()
bad = pairs_done = verified = 0
for case in range(cases):
    KF = int(rng.integers(3, 9))
    step = int(rng.choice([1, 2, 17]))
    lo = int(rng.integers(100, 1300))
    seq = StereoSequence(cam, n_frames=KF * step, n_features=(lo, lo + 500), seed=int(rng.integers(1 << 30)),
                         clutter=float(rng.choice([0.3, 0.6])), landmark_density=float(rng.choice([0.6, 1.0])))
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k * step) for k in range(KF))]
    st = TrackingFrontend(cam, n_iter_max=1).run(frames)
    keyframes = [(st.frame(k)["links"], st.frame(k)["inlier_des_left"]) for k in range(KF)]
    pairs = np.array([(i, n) for n in range(KF) for i in range(KF) if i != n], dtype=np.int64)
    cc = bool(rng.integers(2))
    n_iter = int(rng.choice([40, 257, 600]))
    gate = int(rng.choice([20, 60, 200]))
    thr = float(rng.choice([1.5, 4.0]))
    ver = PairVerifier(cam, pixel_thresh=thr, seed=case, min_inliers_loop=gate, n_iter_max=n_iter,
                       frame_cross_check=cc).set_keyframes(keyframes)
    counts = ver.verify(pairs)["n_matches"]                     # with cross-check the match count is an output
    table = _idx4_table(rng, counts, n_iter)
    res = ver.verify(pairs, samples=table)
    for p, (i, n) in enumerate(pairs):
        _, ref = _cpu_loop_on_device_poses(cam, keyframes[i], keyframes[n], table[p], n_iter_max=n_iter,
                                           pixel_thresh=thr, min_inliers_loop=gate, cross_check=cc)
        sl = slice(int(res["out_off"][p]), int(res["out_off"][p]) + int(res["n_matches"][p]))
        try:
            if "P3D" not in ref:
                assert res["iterations"][p] == 0 and not res["ok"][p]
            else:
                _assert_same_consensus(cam, res, p, sl, ref, thr)
                assert bool(res["ok"][p]) == bool(ref["ok"])
                if res["ok"][p]:
                    verified += 1
                    assert np.abs(res["Rt"][p] - ref["Rt"]).max() < 1e-6
        except AssertionError as e:
            bad += 1
            print(f"MISMATCH case {case} pair {p} ({i},{n}) cc={cc} n_iter={n_iter} gate={gate}: {e!r}"[:400], flush=True)
        pairs_done += 1
    print(f"case {case}: KF={KF} step={step} cc={cc} n_iter={n_iter} gate={gate} thr={thr}: {pairs_done} pairs so far, "
          f"{verified} verified, mismatches {bad}", flush=True)
print(f"done: {cases} cases, {pairs_done} pairs, {verified} verified, {bad} mismatches")
sys.exit(1 if bad else 0)
