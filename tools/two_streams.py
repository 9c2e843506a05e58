"""The 8192 x 8192 matcher step issued on one, two and three streams in turn (a bound call, outputs and
workspace per stream): microseconds per step and whether every stream's outputs equal the first one's.
python tools/two_streams.py"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import kernels
from slam_amd.utils.synth import matcher_workload
q_np, t_np = matcher_workload(8192, 8192, 61, seed=0)
q = kernels.to_device(q_np, torch.uint8); t = kernels.to_device(t_np, torch.uint8)
def run(n_streams, K=4000):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    steps = []
    for s in streams:
        with torch.cuda.stream(s):
            steps.append(kernels.BoundRatioMatch(q, t, 0.8))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < 1.0:
        for i in range(100): steps[i % n_streams]()
        torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(K): steps[i % n_streams]()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K)
    ref = steps[0].out
    same = all(all(torch.equal(a, b) for a, b in zip(ref, st.out)) for st in steps[1:])
    return best * 1e6, same
for n in (1, 2, 3, 1, 2):
    us, same = run(n)
    print(f"{n} stream(s): {us:.2f} us per step = {8192*8192/us*1e6:.3e} compares/s, outputs equal: {same}")
