"""Does a device-resident front-end pass gain from running as K sub-sequences on K streams at
once (each with its halo frame and its own workspace), so that one part's launch tails overlap
with another's kernels?   python tools/tune_streams.py [frames [features [max_features]]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import CameraModel
from slam_amd.backend.tracking import sharding
from slam_amd.backend.tracking.frontend import TrackingFrontend
from slam_amd.utils.synth import StereoSequence

F = int(sys.argv[1]) if len(sys.argv) > 1 else 500
nfeat = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
if len(sys.argv) > 3:
    nfeat = (nfeat, int(sys.argv[3]))
cam = CameraModel.kitti_like()
seq = StereoSequence(cam, n_frames=F, n_features=nfeat, seed=100, clutter=0.6)
frames = [(f.kp_left, f.kp_right, f.des_left, f.des_right) for f in (seq.frame(k) for k in range(F))]


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for parts in (1, 2, 3, 4):
    fes, inps, streams, firsts = [], [], [], []
    for r in range(parts):
        lo, hi, (plo, phi), _ = sharding.frames_for_rank(F, parts, r)
        fe = TrackingFrontend(cam)
        fes.append(fe)
        inps.append(fe.upload(frames[lo:hi]))
        firsts.append(plo)
        streams.append(torch.cuda.Stream())
    main = torch.cuda.current_stream()

    def once():
        for fe, inp, st, p0 in zip(fes, inps, streams, firsts):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                fe.run(inp, first_pair=p0)
        for st in streams:
            main.wait_stream(st)

    dt = timed(once)
    print(f"{F} frames x {nfeat} features as {parts} concurrent part(s): {dt*1e3:.3f} ms per pass", flush=True)
