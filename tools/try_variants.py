"""Times the 8192 x 8192 x 61 B k=2 + ratio call with matcher variants forced through tune knob 0
(8 + G: four waves per block; 16 + G: eight) and checks every variant against the default's result."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slam_amd import _lib, kernels
from slam_amd.utils.synth import matcher_workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
qn, tn = matcher_workload(n, n, 61, seed=0)
q, t = kernels.to_device(qn, torch.uint8), kernels.to_device(tn, torch.uint8)
ref = None
for variant in (-1, 9, 10, 12, 18, -1):
    assert _lib.lib.slamhip_tune(0, variant) == 0
    call = kernels.BoundRatioMatch(q, t, 0.8)
    for _ in range(200):
        out = call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        out = call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2000
    got = [o.clone() for o in out]
    if ref is None:
        ref = got
    same = all(torch.equal(a, b) for a, b in zip(ref, got))
    print(f"variant {variant:3d}: {dt * 1e6:7.2f} us per call, equal to default: {same}", flush=True)
_lib.lib.slamhip_tune(0, -1)
