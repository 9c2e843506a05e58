#!/usr/bin/env python3
"""A few dozen matcher calls of one variant, for profiler runs:
    python tools/run_matcher_once.py VARIANT [n] [desc_bytes] [splits] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, kernels          # noqa: E402
from slam_amd.utils.synth import matcher_workload   # noqa: E402

variant = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
D = int(sys.argv[3]) if len(sys.argv) > 3 else 61
splits = int(sys.argv[4]) if len(sys.argv) > 4 else 0
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
qn, tn = matcher_workload(n, n, D, seed=3)
q = kernels.to_device(qn, torch.uint8)
t = kernels.to_device(tn, torch.uint8)
out = (torch.empty((n, 2), dtype=torch.int32, device=q.device),
       torch.empty((n, 2), dtype=torch.int32, device=q.device),
       torch.empty(n, dtype=torch.uint8, device=q.device))
_lib.lib.slamhip_tune(0, variant)
_lib.lib.slamhip_tune(1, splits)
for _ in range(reps):
    kernels.hamming_knn2_ratio(q, t, 0.8, out=out)
torch.cuda.synchronize()
