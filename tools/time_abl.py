#!/usr/bin/env python3
"""Kernel time of matcher variant 10 (splits 16) at 8192^2 for the library named by
SLAMHIP_LIB_PATH (timing experiments with parts of the kernel compiled out)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import _lib, kernels
from slam_amd.utils.synth import matcher_workload
n = 8192
qn, tn = matcher_workload(n, n, 61, seed=3)
q = kernels.to_device(qn, torch.uint8); t = kernels.to_device(tn, torch.uint8)
out = (torch.empty((n, 2), dtype=torch.int32, device=q.device), torch.empty((n, 2), dtype=torch.int32, device=q.device),
       torch.empty(n, dtype=torch.uint8, device=q.device))
for v, s in ((10, 16), (10, 8), (9, 16)):
    _lib.lib.slamhip_tune(0, v); _lib.lib.slamhip_tune(1, s)
    for _ in range(50):
        kernels.hamming_knn2_ratio(q, t, 0.8, out=out)
    torch.cuda.synchronize()
    _lib.check(_lib.lib.slamhip_profile_begin(200), "b")
    for _ in range(200):
        kernels.hamming_knn2_ratio(q, t, 0.8, out=out)
    torch.cuda.synchronize()
    ms = C.c_double(0.0)
    k = _lib.lib.slamhip_profile_end(C.byref(ms))
    print(os.environ.get("SLAMHIP_LIB_PATH", "default"), "variant", v, "splits", s, f"kernel {ms.value / k * 1e3:.1f} us", flush=True)
