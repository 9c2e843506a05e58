import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_amd import _lib, kernels
from slam_amd.utils.synth import matcher_workload
q_np, t_np = matcher_workload(8192, 8192, 61, seed=0)
q = kernels.to_device(q_np, torch.uint8); t = kernels.to_device(t_np, torch.uint8)
out = (torch.empty((8192, 2), dtype=torch.int32, device='cuda'), torch.empty((8192, 2), dtype=torch.int32, device='cuda'), torch.empty(8192, dtype=torch.uint8, device='cuda'))
for threads in (256, 128):
    for splits in (0, 16, 32, 64):
        _lib.lib.slamhip_tune(3, threads); _lib.lib.slamhip_tune(1, splits)
        for _ in range(50): kernels.hamming_knn2_ratio(q, t, 0.8, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(500): kernels.hamming_knn2_ratio(q, t, 0.8, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
        print(f"threads {threads} splits {splits:2d}: {dt*1e6:.1f} us/call  {8192*8192/dt:.3e} compares/s")
