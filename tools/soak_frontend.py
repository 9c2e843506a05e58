"""One-off soak of the batched front-end for run-to-run determinism and agreement of its
launch schemes: random sequences (2..40 frames, every fourth 64..90; 60..2500 features per image, both
cross-check flags, random iteration caps), each run twice and once more with each of the
other RANSAC launch schemes forced (rounds; rounds with the merged tail) and once with the
descriptor rows packed in front of the stereo matcher, once with the sorts as launches of
their own, once with pair verification's consensus counting scheme and once with the consensus
counted in fp64 alone, once with the fused launches of short calls switched off; every fifth case is
a call of two or three frames (those launches), and sequences of up to twelve frames are also pushed
one frame at a time through OnlineFrontend; every valid output row must be identical bit for bit.
python tools/soak_frontend.py [cases [seed]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel, _lib
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.backend.tracking.online import OnlineFrontend
from slam_amd.utils.synth import StereoSequence

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
cam = CameraModel.kitti_like()


def snapshot(res, F):
    out = []
    for f in range(F):
        fr = res.frame(f)
        out.append({k: np.array(v, copy=True) for k, v in fr.items()})
    for p in range(F - 1):
        pr = res.pair(p)
        out.append({k: np.array(v, copy=True) for k, v in pr.items() if isinstance(v, np.ndarray)})
    return out


def same(a, b):
    if len(a) != len(b):
        return False
    for x, y in zip(a, b):
        if x.keys() != y.keys():
            return False
        for k in x:
            if x[k].shape != y[k].shape or not np.array_equal(x[k], y[k], equal_nan=True):
                return k
    return True


bad = 0
for case in range(cases):
    # every fourth case is long enough for the launch schemes of long calls (merged tail,
    # blocks dealt to the XCDs by pair: 64 pairs or more)
    F = int(rng.integers(2, 41)) if case % 4 else int(rng.integers(64, 91))
    if case % 5 == 1:
        F = int(rng.integers(2, 4))
    lo = int(rng.integers(60, 1200))
    nfeat = (lo, lo + int(rng.integers(1, 1300)))
    seq = StereoSequence(cam, n_frames=F, n_features=nfeat, seed=int(rng.integers(1 << 30)), clutter=float(rng.uniform(0.2, 0.8)))
    frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]
    kw = dict(n_iter_max=int(rng.choice([1, 100, 700, 2500])), seed=case,
              stereo_cross_check=bool(rng.integers(2)), frame_cross_check=bool(rng.integers(2)))
    fe = TrackingFrontend(cam, **kw)
    inp = fe.upload(frames)
    _lib.lib.slamhip_tune(4, 0)
    a = snapshot(fe.run(inp), F)
    b = snapshot(fe.run(inp), F)
    _lib.lib.slamhip_tune(4, 1)
    c = snapshot(fe.run(inp), F)
    _lib.lib.slamhip_tune(4, 2)
    d = snapshot(fe.run(inp), F)
    _lib.lib.slamhip_tune(4, 0)
    _lib.lib.slamhip_tune(9, 0)          # descriptor rows packed first instead of read in place
    fe._ws = None
    e = snapshot(fe.run(inp), F)
    _lib.lib.slamhip_tune(9, 1)
    fe._ws = None
    _lib.lib.slamhip_tune(5, 2)          # the sorts as launches of their own (bitonic network)
    g = snapshot(fe.run(inp), F)
    _lib.lib.slamhip_tune(5, 0)
    _lib.lib.slamhip_tune(2, 3)          # the consensus counting scheme of pair verification
    h = snapshot(fe.run(inp), F)
    _lib.lib.slamhip_tune(2, 0)          # consensus counting in fp64 alone
    i = snapshot(fe.run(inp), F)
    _lib.lib.slamhip_tune(2, 1)
    _lib.lib.slamhip_tune(14, 0)         # a short call through the launches of a long one
    j = snapshot(fe.run(inp), F)
    _lib.lib.slamhip_tune(14, 1)
    _lib.lib.slamhip_tune(16, 0)         # the first round's consensus count as one launch
    k16 = snapshot(fe.run(inp), F)
    _lib.lib.slamhip_tune(16, 1)
    r10 = same(a, k16)
    r8 = same(a, j)
    r9 = True
    if F <= 12:                          # the same frames pushed one at a time
        on = OnlineFrontend(cam, frames_capacity=int(rng.integers(3, 6)), **kw)
        on.push(*frames[0])
        for k in range(1, F):
            got = on.push(*frames[k])
            w = [{kk: np.array(v, copy=True) for kk, v in got.frame(1).items()},
                 {kk: np.array(v, copy=True) for kk, v in got.pair(0).items() if isinstance(v, np.ndarray)}]
            rr = same(w, [a[k], a[F + k - 1]])
            if rr is not True:
                r9 = (k, rr)
                break
    r1, r2, r3, r4, r5, r6, r7 = same(a, b), same(a, c), same(a, d), same(a, e), same(a, g), same(a, h), same(a, i)
    if not all(r is True for r in (r1, r2, r3, r4, r5, r6, r7, r8, r9, r10)):
        bad += 1
        print(f"MISMATCH case {case}: F={F} nfeat={nfeat} rerun={r1} rounds={r2} tail={r3} packed={r4} "
              f"sort={r5} sparse count={r6} fp64 count={r7} unfused={r8} pushed={r9} one count launch={r10}", flush=True)
    if case % 5 == 0:
        print(f"case {case} ok so far (F={F}, features {nfeat}), mismatches {bad}", flush=True)
print(f"done: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
