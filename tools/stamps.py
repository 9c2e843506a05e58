#!/usr/bin/env python3
"""Prints the shader-clock stamps a diagnostics build of the matrix-core matcher
(SLAMHIP_ABLATE=8, library named by SLAMHIP_LIB_PATH) leaves in idx2."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, kernels
from slam_amd.utils.synth import matcher_workload
n = 8192
qn, tn = matcher_workload(n, n, 61, seed=3)
q = kernels.to_device(qn, torch.uint8); t = kernels.to_device(tn, torch.uint8)
_lib.lib.slamhip_tune(0, 10); _lib.lib.slamhip_tune(1, 1)      # one split: idx2 is written by the kernel itself
for _ in range(5):
    i2, d2 = kernels.hamming_knn2(q, t)
torch.cuda.synchronize()
st = d2.cpu().numpy().reshape(-1)[:64 * 40].reshape(64, 40)
np.set_printoptions(linewidth=250)
cal = d2.cpu().numpy().reshape(-1)[64 * 40:64 * 40 + 2]
print("whole block 0: realtime ticks (100 MHz)", cal[0], "memtime ticks", cal[1], "-> memtime MHz", cal[1] / max(cal[0], 1) * 100)
for b in (0, 1, 5, 17, 31):
    v = st[b]; v = v[v >= 0]
    print("block", b, "stamps:", v[:3], "stage deltas (tile0, tile1, store, barrier):")
    d = np.diff(v)
    print(d[2:2 + 4 * 6].reshape(-1, 4), " total", v[-1])
